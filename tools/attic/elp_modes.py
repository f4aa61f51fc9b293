#!/usr/bin/env python
"""calibrate.EnsembleLogProb: likelihood streams and where the upstream runs (its own stream / inline), alternating on
one box:  python tools/elp_modes.py [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from starry_process_amd.calibrate import EnsembleLogProb
from starry_process_amd.synthetic import synthetic_star
S, K = 64, 1000
sts = [synthetic_star(s, K) for s in range(S)]
t = np.array([s["t"] for s in sts]); flux = np.array([s["flux"] for s in sts]); p = np.array([s["p"] for s in sts])
samples = np.array([[20.0 + 0.01 * i, 0.4, 0.27, 0.1, 10.0] for i in range(240)])
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
modes = [("3 streams + upstream stream", 3, True), ("4 streams, upstream inline", 4, False),
         ("3 streams, upstream inline", 3, False), ("4 streams + upstream stream", 4, True)]
lps = [EnsembleLogProb(t, flux, ferr=1e-3, p=p, depth=d, upstream_stream=u) for _, d, u in modes]
res = [[] for _ in modes]
ref = None
for r in range(rounds + 1):
    for k, lp in enumerate(lps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = lp(samples)
        dt = (time.perf_counter() - t0) / len(samples)
        if r:
            res[k].append(1e3 * dt)
        ref = v if ref is None else ref
        assert np.max(np.abs(v - ref)) < 1e-9 * np.max(np.abs(ref))
for (name, _, _), x in zip(modes, res):
    print("%-30s %.4f ms per sample (%.4f .. %.4f)" % (name, sum(x) / len(x), min(x), max(x)))
