#!/usr/bin/env python
"""Timeline of the one-launch-per-panel kernel (K = 1000, 64 stars, one step at a time) from in-kernel
wall-clock stamps: needs the variant library  bash tools/ab_build.sh trace -DSP_PANEL_TRACE
SP_LIB_VARIANT=trace python tools/panel_trace.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from starry_process_amd._lib import check
from chain_check import engine, setup, run

e = engine(0, 1)
a = setup(e, 64, 1000)
run(e, a, reps=5)
check(e._L.sp_debug_panel_trace(None))
run(e, a, reps=1)
buf = np.zeros(64 * 4 * 16, dtype=np.int64)
check(e._L.sp_debug_panel_trace(buf.ctypes.data_as(ctypes.c_void_p)))
t = buf.reshape(64, 4, 16).astype(np.float64)
t0 = t[t > 0].min()
us = lambda v: (v - t0) / 100.0
names = ["start", "product done", "image in LDS", "solved", "eager done", "diag start", "diag end", "end"]
print("pivot workgroup of star 0 (us from the first stamp of the step); then strip 3's start / product / image / solve")
prev_end = None
for l in range(64):
    if not t[l, 0].any():
        continue
    p = t[l, 0]
    line = "launch %2d: start %7.1f" % (l, us(p[0]))
    if prev_end is not None:
        line += " (gap %4.1f)" % (us(p[0]) - prev_end)
    line += " | product +%5.1f image +%4.1f solve +%4.1f eager +%4.1f | diag %4.1f | tail %4.1f | total %5.1f" % (
        us(p[1]) - us(p[0]), us(p[2]) - us(p[1]), us(p[3]) - us(p[2]), us(p[4]) - us(p[3]),
        us(p[6]) - us(p[5]), us(p[7]) - us(p[6]), us(p[7]) - us(p[0]))
    if p[8] and p[9]:
        line += " | forming the tile: %4.1f (at +%4.1f)" % (us(p[9]) - us(p[8]), us(p[8]) - us(p[0]))
    o = t[l, 1]
    if o[0]:
        line += " || strip 3: start %+5.1f product +%5.1f image +%4.1f solve +%4.1f" % (
            us(o[0]) - us(p[0]), us(o[1]) - us(o[0]), us(o[2]) - us(o[1]), us(o[3]) - us(o[2]))
    prev_end = us(p[7])
    print(line)
