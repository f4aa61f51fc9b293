#!/usr/bin/env python
"""Only the device sweep of the ensemble gradient (sp_lnlike_grad_marginal), N times: for rocprofv3 --kernel-trace.
python tools/grad_sweep_only.py [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from starry_process_amd.grad import EnsembleGradient  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402
from starry_process_amd.upstream_device import ylm_moments_device  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
S, K = 64, 1000
sts = [synthetic_star(s, K) for s in range(S)]
t, flux, p = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), np.array([s["p"] for s in sts])
eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
e = eg._e
mu, Sig = ylm_moments_device(e)
e.set_moments_dev(mu, Sig)
tab, mv = e.kernel_table(eg._rta1, 300)
for _ in range(2):
    e.lnlike_grad_marginal(eg._t, eg._flux, eg._stars, tab, mv, workspace=eg._ws)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    e.lnlike_grad_marginal(eg._t, eg._flux, eg._stars, tab, mv, workspace=eg._ws)
torch.cuda.synchronize()
print("sweep ms", 1e3 * (time.perf_counter() - t0) / N)
