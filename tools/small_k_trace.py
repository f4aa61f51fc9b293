"""In-kernel timeline of the small-K kernel's phases (first 64 workgroups): needs the variant library
    bash tools/ab_build.sh trace -DSP_SMALL_TRACE ;  SP_LIB_VARIANT=trace python tools/small_k_trace.py [K] [S]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from starry_process_amd import _lib  # noqa: E402
from starry_process_amd.engine import Engine, make_stars  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
e = Engine(15, 2, 0)
mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
base = [synthetic_star(s, K) for s in range(64)]
sts = [base[s % 64] for s in range(S)]
t_d = e.f64(np.array([s["t"] for s in sts]))
f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
s_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
plan = e.plan_data(t_d, f_d, s_d, covpts=300)
for _ in range(3):
    e.lnlike_ensemble_planned(plan, None, None, s_d, tab, mv)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(64 * 16, dtype=np.int64)
assert L.sp_debug_small_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
tr = buf.reshape(64, 16).astype(float)
# stamps: 0 start, 1 prologue done, 2 tile (0, 0) assembled and in LDS, 3 pivot block 0 factored, 8 rows' first half
# solved, 9 tile (1, 0) assembled and X in registers, 10 X / rows in LDS, 11 rows' second half updated, 12 tile (1, 1)
# assembled and updated,
# 4 tile (1, 1) in LDS, 5 pivot block 1 factored, 6 rows solved, 7 reduced
order = [0, 1, 2, 3, 8, 9, 10, 11, 12, 4, 5, 6, 7] if K > 64 else [0, 1, 2, 3, 8, 6, 7]
names = {1: "prologue (loads, m, coefficients)", 2: "assembly", 3: "pivot block 0 (diag_block)", 8: "rows: first half solve",
         9: "tile (1, 0) assembled + X = T10 L00^-T", 10: "barriers + X to LDS", 11: "rows: second half update", 12: "tile (1, 1) assembled, -= X X^T", 4: "T11 to LDS",
         5: "pivot block 1 (diag_block)", 6: "rows: solve / barriers", 7: "reduction"}
print("K %d, S %d: phases of workgroups 0 .. 63, us (mean / min / max); whole %.1f us" % (K, S, np.mean(tr[:, 7] - tr[:, 0]) * 0.01))
for a, b in zip(order[:-1], order[1:]):
    d = (tr[:, b] - tr[:, a]) * 0.01
    print("  %-36s %6.2f  %6.2f  %6.2f" % (names[b], d.mean(), d.min(), d.max()))

# diag_block's own stamps (shader clock cycles) of workgroup 0's first pivot block: per block column kb the owner
# wavefront's {start, leaf begins, leaf done, stores done, -}, then the next owner's {after the rows below, after its
# part of the update}
dg = np.zeros(64, dtype=np.int64)
if hasattr(L, "sp_debug_small_diag") and L.sp_debug_small_diag(dg.ctypes.data_as(ctypes.c_void_p)) == 0:
    d = dg.reshape(8, 8)[:4]
    t0 = d[0, 0]
    print("diag_block of workgroup 0, cycles from its start (block column: leaf start, leaf done, stored | rows below done, update done):")
    for kb in range(4):
        print("  kb %d: %6d %6d %6d | %6d %6d" % (kb, d[kb, 1] - t0, d[kb, 2] - t0, d[kb, 3] - t0, d[kb, 5] - t0, d[kb, 6] - t0))
