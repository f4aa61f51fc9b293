#!/usr/bin/env python
"""Timeline of the round-3 panel kernel (sp_panel.hip; K = 1000, 64 stars, one step at a time) from
in-kernel wall-clock stamps of star 0's work items.  Needs the variant library:
    bash tools/ab_build.sh trace -DSP_PANEL_TRACE && SP_LIB_VARIANT=trace python tools/panel2_trace.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from starry_process_amd._lib import check
from _common import engine, setup, run

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
e = engine()
a = setup(e, S, K)
run(e, a, reps=5)
f = e._L.sp_debug_panel2_trace
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p]
check(f(None))
run(e, a, reps=1)
buf = np.zeros(64 * 3 * 16, dtype=np.int64)
check(f(buf.ctypes.data_as(ctypes.c_void_p)))
t = buf.reshape(64, 3, 16).astype(np.float64)
t0 = t[t > 0].min()
us = lambda v: (v - t0) / 100.0
d = lambda p, a, b: (us(p[b]) - us(p[a])) if p[a] and p[b] else float("nan")
print("star 0, us.  D item: start | load->diag_block | diag_block | image + L out | tail (partial rows, publish)")
print("T items (first = next pivot row tile, last = bottom row tile): start | C tile / lazy | product | wait | solve | store | eager")
for j in range(64):
    if not t[j].any():
        continue
    D, F, L = t[j]
    line = "j %2d" % j
    if D[0]:
        line += " | D%s @%7.1f: load %4.1f diag %5.1f img %4.1f tail %4.1f = %5.1f" % (
            "(j+1, tail)" if F[0] and D[0] > F[0] + 1 else "", us(D[0]), d(D, 0, 8), d(D, 8, 9), d(D, 9, 10), d(D, 10, 2), d(D, 0, 2))
    for name, p in (("first", F), ("last", L)):
        if p[0]:
            line += " | %s @%7.1f: C %4.1f prod %5.1f wait %5.1f solve %4.1f store %4.1f eager %4.1f end @%7.1f" % (
                name, us(p[0]), d(p, 0, 1), d(p, 1, 2), d(p, 2, 3), d(p, 3, 4), d(p, 4, 5), d(p, 5, 6),
                us(p[6] if p[6] else p[5]))
    print(line)

# every star's tail block: how the 64 chains of a launch sit on the CUs
cb = np.zeros(16 * 64 * 4 + 16 * 512 + 16 * 1024 * 2, dtype=np.int64)
check(e._L.sp_debug_panel2_chain(cb.ctypes.data_as(ctypes.c_void_p)))
cu = cb[16 * 64 * 4:16 * 64 * 4 + 16 * 512].view(np.int32).reshape(16, 1024)
wg = cb[16 * 64 * 4 + 16 * 512:].reshape(16, 1024, 2)
cb = cb[:16 * 64 * 4].reshape(16, 64, 4)
if len(sys.argv) > 3:
    # where the workgroups of launch j ran: XCD-local dispatch index -> CU (key within the XCD), per XCD
    j = int(sys.argv[3])
    for x in range(2):
        ks = cu[j][x::8]
        ks = ks[ks > 0] - 1
        cus = {k: i for i, k in enumerate(sorted(set(ks)))}
        print("launch %d, XCD slot %d: %d workgroups on %d CUs; CU index by dispatch order:" % (j, x, len(ks), len(cus)))
        print(" ".join("%2d" % cus[k] for k in ks))
print("per launch: first items start (min..max) | blocks start | blocks end | block duration min / median / max | CUs used, most chains on one CU")
for j in range(16):
    c = cb[j][cb[j][:, 2] > 0]
    if not len(c):
        continue
    f0, b0, b1 = [us(c[:, k].astype(np.float64)) for k in range(3)]
    dur = b1 - b0
    keys, cnt = np.unique(c[:, 3], return_counts=True)
    print("j %2d | first %7.1f..%7.1f | block start %7.1f..%7.1f | end %7.1f..%7.1f | %5.1f / %5.1f / %5.1f | %d CUs, max %d" % (
        j, f0.min(), f0.max(), b0.min(), b0.max(), b1.min(), b1.max(), dur.min(), np.median(dur), dur.max(), len(keys), cnt.max()))

print("workgroups of a launch (first 1024): end - launch start, by XCD-local dispatch index k (XCD slot 0): per CU = k mod 32")
for j in range(16):
    w = wg[j]
    ok = (w[:, 1] > 0) & (w[:, 0] > 0)
    if not ok.any():
        continue
    t00 = w[ok, 0].min()
    dur = np.where(ok, (w[:, 1] - t00) / 100.0, np.nan)
    loc = dur[0::8]
    loc = loc[:96] if len(loc) >= 96 else loc
    n = len(loc)
    print("j %2d: %4d workgroups, ends: median %5.1f, 90%% %5.1f, max %5.1f us | XCD 0 by round: %s" % (
        j, ok.sum(), np.nanmedian(dur), np.nanpercentile(dur[ok], 90), np.nanmax(dur),
        " / ".join("%.0f-%.0f" % (np.nanmin(loc[r * 32:(r + 1) * 32]), np.nanmax(loc[r * 32:(r + 1) * 32]))
                   for r in range((n + 31) // 32) if np.isfinite(loc[r * 32:(r + 1) * 32]).any())))
    if len(sys.argv) > 3 and int(sys.argv[3]) == j:
        for r in range((n + 31) // 32):
            print("   round %d: " % r + " ".join("%3.0f" % v for v in loc[r * 32:(r + 1) * 32]))
