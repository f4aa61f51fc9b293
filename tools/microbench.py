#!/usr/bin/env python
"""Time the Cholesky phases in isolation with events on the launch stream.
python tools/microbench.py [S] [K]"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import get_engine, make_stars
from starry_process_amd.synthetic import synthetic_star
from starry_process_amd._lib import check

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
e = get_engine(15, 2, 0)
mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
sts = [synthetic_star(s, K) for s in range(S)]
t_d = e.f64(np.array([s["t"] for s in sts])); f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
stars_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
ws = e.workspace(S, K, 1)
e.lnlike_ensemble(t_d, f_d, stars_d, tab=tab, meanvar=mv, workspace=ws)
torch.cuda.synchronize()
st = e._stream()
def timeit(phase, j, reps=20):
    for _ in range(3):
        check(e._L.sp_debug_cholesky_phase(e._h, S, K, 1, e._p(ws), phase, j, st))
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        check(e._L.sp_debug_cholesky_phase(e._h, S, K, 1, e._p(ws), phase, j, st))
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
names = ["diag", "solve", "update"]
nsteps = (K + 63) // 64
tot = [0, 0, 0]
for j in range(nsteps):
    row = []
    for ph in range(3):
        if ph == 2 and (j + 1) * 64 >= K:
            row.append(0.0); continue
        us = timeit(ph, j); row.append(us); tot[ph] += us
    n = ((K + 1 + 63) // 64) * 64 - 64 * (j + 1)
    print("step %2d  n=%4d  diag %6.1f us  solve %6.1f us  update %7.1f us" % (j, n, *row))
print("totals: diag %.1f  solve %.1f  update %.1f  sum %.1f us" % (*tot, sum(tot)))
# in-kernel timestamps of the diagonal-block kernel (stderr)
check(e._L.sp_debug_cholesky_phase(e._h, S, K, 1, e._p(ws), 3, 0, st))
torch.cuda.synchronize()
