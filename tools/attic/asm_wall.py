#!/usr/bin/env python
"""Timeline of the assembly kernel's workgroups (library built with -DSP_ASM_STAMPS, SP_LIB_VARIANT=stamps):
when each started and ended (100 MHz wall clock), the cycles of its phases.  python tools/asm_wall.py [tiles-per-wg]"""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    os.environ["SP_ASM_TILES"] = sys.argv[1]
os.environ.setdefault("SP_LIB_VARIANT", "stamps")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from starry_process_amd import _lib  # noqa: E402

bench.bench_shape(torch, dist, ydeg=15, Kc=1000, S=64, tspan=4.0, tau=None, u=(0.0, 0.0), conditional=False, F=1,
                  steps=2, device=0)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(8 * 8192, dtype=np.int64)
L.sp_debug_asm_stamps.restype = ctypes.c_int
L.sp_debug_asm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.sp_debug_asm_stamps(buf.ctypes.data, buf.size) == 0
rows = buf.reshape(-1, 8)
rows = rows[rows[:, 3] > 0]
t0 = rows[:, 2].min()
print("workgroups", len(rows), " kernel span %.1f us" % ((rows[:, 3].max() - t0) / 100.0))
dur = (rows[:, 3] - rows[:, 2]) / 100.0
print("duration us: min %.1f median %.1f max %.1f;  start: median %.1f max %.1f" %
      (dur.min(), np.median(dur), dur.max(), np.median(rows[:, 2] - t0) / 100.0, (rows[:, 2].max() - t0) / 100.0))
for x in sorted(set(rows[:, 0])):
    r = rows[rows[:, 0] == x]
    d = (r[:, 3] - r[:, 2]) / 100.0
    print("chunk %2d: duration median %.1f max %.1f us, ends at %.1f..%.1f;  cycles prologue %d eval %d sums+stores %d rest %d" %
          (x, np.median(d), d.max(), (r[:, 3].min() - t0) / 100.0, (r[:, 3].max() - t0) / 100.0,
           np.median(r[:, 4]), np.median(r[:, 5]), np.median(r[:, 6]), np.median(r[:, 7])))
