#!/usr/bin/env python
"""Time of the ensemble gradient (grad.EnsembleGradient: one device sweep for the whole batch) at cfg3's shape,
beside the forward step and round 3's one-star-per-call gradient:  python tools/grad_timing.py [S] [K]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from starry_process_amd.engine import get_engine, make_stars  # noqa: E402
from starry_process_amd.grad import EnsembleGradient, hyper_gradient  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402
from starry_process_amd.upstream_device import ylm_moments_device  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sts = [synthetic_star(s, K) for s in range(S)]
t, flux, p = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), np.array([s["p"] for s in sts])
eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
for _ in range(3):
    total, g = eg()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for k in range(n):
    total, g = eg(r=20.0 + 0.01 * k)
torch.cuda.synchronize()
ms_grad = 1e3 * (time.perf_counter() - t0) / n
# the device sweep alone (tables given)
e = eg._e
mu, Sig = ylm_moments_device(e)
e.set_moments_dev(mu, Sig)
tab, mv = e.kernel_table(eg._rta1, 300)
for _ in range(3):
    e.lnlike_grad_marginal(eg._t, eg._flux, eg._stars, tab, mv, workspace=eg._ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    e.lnlike_grad_marginal(eg._t, eg._flux, eg._stars, tab, mv, workspace=eg._ws)
torch.cuda.synchronize()
ms_sweep = 1e3 * (time.perf_counter() - t0) / n
# the forward step
ws = e.workspace(S, K, 1)
fl3 = eg._flux[:, None, :].contiguous()
for _ in range(5):
    e.lnlike_ensemble(eg._t, fl3, eg._stars, tab=tab, meanvar=mv, workspace=ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    e.set_moments_dev(mu, Sig)
    tab, mv = e.kernel_table(eg._rta1, 300)
    e.lnlike_ensemble(eg._t, fl3, eg._stars, tab=tab, meanvar=mv, workspace=ws)
torch.cuda.synchronize()
ms_fwd = 1e3 * (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
l1, g1 = hyper_gradient(t[0], flux[0], 1e-6, p=float(p[0]))
torch.cuda.synchronize()
ms_one = 1e3 * (time.perf_counter() - t0)
fl = S * (K ** 3 / 3.0)
print("S = %d, K = %d" % (S, K))
print("forward step (moments -> table -> lnL of %d stars), one at a time:   %.3f ms" % (S, ms_fwd))
print("ensemble gradient, device sweep alone (C, C^-1, adjoints):           %.3f ms  = %.2f x forward; "
      "%.1f TFLOP/s of the 3.5 K^3/3 flops of factor + triangular inverse + L^-T L^-1" %
      (ms_sweep, ms_sweep / ms_fwd, 3.5 * fl / (ms_sweep * 1e-3) / 1e12))
print("ensemble gradient, whole call (moments with their exact tangents, 3 + 2 table evaluations on three more streams, "
      "sweep): %.3f ms = %.2f x forward" % (ms_grad, ms_grad / ms_fwd))
eg_fd = EnsembleGradient(t, flux, ferr=1e-3, p=p, exact=False)
for _ in range(3):
    eg_fd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    total_fd, g_fd = eg_fd(r=20.0 + 0.01 * k)
torch.cuda.synchronize()
ms_fd = 1e3 * (time.perf_counter() - t0) / n
print("   ... with round 4's central differences of the table in r, a, b (exact=False: nine table evaluations): %.3f ms; "
      "largest relative difference of the two gradients %.1e" % (ms_fd, max(abs(g[k] - g_fd[k]) / abs(g_fd[k]) for k in g)))
print("round 3: hyper_gradient, ONE star per call:                          %.3f ms  (x %d stars = %.0f ms)" % (ms_one, S, ms_one * S))
print("gradient:", {k: float("%.6g" % v) for k, v in g.items()}, " lnL = %.6f" % total)
