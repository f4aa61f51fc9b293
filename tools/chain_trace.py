#!/usr/bin/env python
"""Timeline of the dataflow panel chain (sp_set_chol_mode 3) for star 0 of a K = 1000, 64-star step:
in-kernel wall-clock stamps of every strip (sp_debug_chain_trace), microseconds from the first one.
python tools/chain_trace.py [S] [K]"""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd._lib import check
from chain_check import engine, setup, run

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
e = engine(3)
a = setup(e, S, K)
run(e, a, reps=5)
Kp = -(-(K + 4) // 64) * 64
ntile = Kp // 64
nl = 8
buf = torch.zeros(nl * ntile * 128, dtype=torch.int64, device="cuda")
check(e._L.sp_debug_chain_trace(e._h, buf.data_ptr()))
run(e, a, reps=1)
check(e._L.sp_debug_chain_trace(e._h, None))
tr = buf.cpu().numpy().reshape(nl, ntile, 16, 8).astype(np.float64)
for l in range(nl):
    t = tr[l]
    if not t.any():
        continue
    t0 = t[t > 0].min()
    us = lambda v: (v - t0) / 100.0 if v > 0 else float("nan")
    print("launch %d (100 MHz wall clock; us from the first stamp)" % l)
    for s in range(ntile):
        if not t[s].any():
            continue
        line = "strip %2d start %6.1f |" % (s, us(t[s, 15, 0]))
        for j in range(15):
            if t[s, j].any():
                line += " j%d: w%5.1f p%5.1f d%5.1f s%5.1f e%5.1f |" % (
                    j, us(t[s, j, 1]) - us(t[s, j, 0]) if t[s, j, 1] else 0.0,
                    us(t[s, j, 2]) - us(t[s, j, 1]) if t[s, j, 2] else 0.0,
                    us(t[s, j, 3]) - (us(t[s, j, 2]) if t[s, j, 2] else us(t[s, j, 0])),
                    us(t[s, j, 4]) - us(t[s, j, 3]), us(t[s, j, 5]))
        if t[s, 15, 1]:
            line += " diag %6.1f -> %6.1f pub %6.1f" % (us(t[s, 15, 1]), us(t[s, 15, 2]), us(t[s, 15, 3]))
        print(line)
