#!/usr/bin/env python
"""Timeline of the trailing update's workgroups at cfg3's shape (library built with -DSP_MM_STAMPS,
SP_LIB_VARIANT=mmstamps): begin / end (100 MHz wall clock) and the cycles before the loop, in it, after it."""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("SP_LIB_VARIANT", "mmstamps")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from starry_process_amd import _lib  # noqa: E402

bench.bench_shape(torch, dist, ydeg=15, Kc=1000, S=64, tspan=4.0, tau=None, u=(0.0, 0.0), conditional=False, F=1,
                  steps=2, device=0)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(8 * 4096, dtype=np.int64)
L.sp_debug_mm_stamps.restype = ctypes.c_int
L.sp_debug_mm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.sp_debug_mm_stamps(buf.ctypes.data, buf.size) == 0
rows = buf.reshape(-1, 8)
rows = rows[rows[:, 3] > 0]
t0 = rows[:, 2].min()
print("workgroups (products)", len(rows), " span %.1f us" % ((rows[:, 3].max() - t0) / 100.0))
dur = (rows[:, 3] - rows[:, 2]) / 100.0
for lazy in (0, 1):
    m = rows[:, 1] == lazy
    if m.any():
        print("lazy=%d: %d workgroups, duration median %.1f (min %.1f max %.1f) us; cycles before loop %d, loop %d, store %d" %
              (lazy, m.sum(), np.median(dur[m]), dur[m].min(), dur[m].max(), np.median(rows[m, 4]), np.median(rows[m, 5]), np.median(rows[m, 6])))
st = (rows[:, 2] - t0) / 100.0
en = (rows[:, 3] - t0) / 100.0
print("starts: ", np.round(np.percentile(st, [0, 10, 25, 33, 50, 66, 75, 90, 100]), 1))
print("ends:   ", np.round(np.percentile(en, [0, 10, 25, 33, 50, 66, 75, 90, 100]), 1))
percu = defaultdict(list)
for r in rows:
    percu[int(r[7])].append(r)
n = [len(v) for v in percu.values()]
print("CUs", len(percu), "workgroups per CU: min %d max %d" % (min(n), max(n)))
last = sorted(((max(x[3] for x in v) - t0) / 100.0 for v in percu.values()))
print("last end per CU: min %.1f median %.1f max %.1f us" % (last[0], last[len(last) // 2], last[-1]))
busy = [sum((x[3] - x[2]) for x in v) / 100.0 for v in percu.values()]
print("sum of workgroup durations per CU: min %.1f median %.1f max %.1f us" % (min(busy), np.median(busy), max(busy)))
