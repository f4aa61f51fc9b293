#!/usr/bin/env python
"""Where the HOST time of one EnsembleGradient call goes (cProfile):  python tools/grad_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from starry_process_amd.grad import EnsembleGradient  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402

S, K = 64, 1000
sts = [synthetic_star(s, K) for s in range(S)]
t, flux, p = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), np.array([s["p"] for s in sts])
eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
for _ in range(3):
    eg()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(20):
    eg(r=20.0 + 0.01 * k)
print("ms per call", 1e3 * (time.perf_counter() - t0) / 20)
pr = cProfile.Profile()
pr.enable()
for k in range(20):
    eg(r=20.0 + 0.01 * k)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
