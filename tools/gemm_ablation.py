#!/usr/bin/env python
"""Time the rank-256 trailing update (first super-panel, 64 stars, K = 1000) alone,
optionally with an ablation (env SP_GEMM_ABL = 1..4, see sp_gemm.hip).
python tools/gemm_ablation.py"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import get_engine, make_stars
from starry_process_amd.synthetic import synthetic_star
from starry_process_amd._lib import check

S, K = 64, 1000
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
us = timeit(4, 0)
n, kd = 1024 - 256, 256
flops = S * n * (n + 1) * kd          # algorithmic (lower triangle)
tiles = S * (n // 64) * (n // 64 + 1) // 2
print("ABL=%s  rank-256 update n=%d: %.1f us  algorithmic %.1f TF  executed %.1f TF" % (
    os.environ.get("SP_GEMM_ABL", "0"), n, us, flops / us * 1e-6, tiles * 2 * 64 * 64 * kd / us * 1e-6))
if os.environ.get("SP_PEAK"):
    for kk in (1000, 1002, 1004, 1006):      # 8, 4, 2, 1 accumulators per wave
        for occ in (1, 4):
            check(e._L.sp_debug_cholesky_phase(e._h, S, kk, 1, e._p(ws), 5, occ, st))
