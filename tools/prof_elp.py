import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from starry_process_amd.calibrate import EnsembleLogProb
from starry_process_amd.synthetic import synthetic_star
S, K = 64, 1000
sts = [synthetic_star(s, K) for s in range(S)]
t = np.array([s["t"] for s in sts]); flux = np.array([s["flux"] for s in sts]); p = np.array([s["p"] for s in sts])
samples = np.array([[20.0 + 0.01 * i, 0.4, 0.27, 0.1, 10.0] for i in range(120)])
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lp = EnsembleLogProb(t, flux, ferr=1e-3, p=p, depth=depth)
lp(samples[:6]); torch.cuda.synchronize()
t0 = time.perf_counter(); vals = lp(samples); dt = (time.perf_counter() - t0) / len(samples)
print("depth", depth, "ms per sample", 1e3 * dt)
if len(sys.argv) > 2:
    sys.exit(0)
pr = cProfile.Profile(); pr.enable(); lp(samples); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
