#!/usr/bin/env python
"""Time the pieces of the recursive factorisation alone (events on the launch stream):
the top-level strip solve, the top-level symmetric update, the four base blocks.
python tools/strip_bench.py [S] [K]"""
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
nb = (K + 63) // 64
bm = ((nb // 2 + 3) // 4) * 4
n2 = (nb - bm) * 64
us = timeit(6, 0)
fl = S * n2 * (bm * 64.0) ** 2
print("strip solve  %4d rows x %4d cols: %7.1f us  %5.1f TFLOP/s algorithmic (%.2f of 78.6)" % (n2, bm * 64, us, fl / us * 1e-6, fl / us * 1e-6 / 78.6))
us = timeit(7, 0)
fl = S * n2 * (n2 + 1.0) * bm * 64
print("sym. update  n = %4d, k = %4d:   %7.1f us  %5.1f TFLOP/s algorithmic (%.2f of 78.6)" % (n2, bm * 64, us, fl / us * 1e-6, fl / us * 1e-6 / 78.6))
base = int(os.environ.get("SP_REC_BASE", "8"))
for j in range((nb + base - 1) // base):
    print("base block %d (%d panels): %7.1f us" % (j, base, timeit(8, j)))
