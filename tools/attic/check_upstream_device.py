#!/usr/bin/env python
"""Device upstream (quadrature of rotations) against the reference fixtures, per degree, and timed."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import golden
from starry_process_amd.engine import get_engine
from starry_process_amd.upstream_device import ylm_moments_device
from starry_process_amd import upstream

for L, names in ((5, ["default"]), (15, ["default", "hilat", "spread"]), (20, ["default"])):
    e = get_engine(L, 2, 0)
    g = golden("moments_L%d" % L)
    for name in names:
        r, dr, a, b, c, n = g[name + "_hyper"]
        dr = None if np.isnan(dr) else dr
        mu, S = ylm_moments_device(e, r=r, dr=dr, a=a, b=b, c=c, n=n)
        mu, S = mu.cpu().numpy(), S.cpu().numpy()
        mr, Sr = g[name + "_mean_ylm"], g[name + "_cov_ylm"]
        d = np.abs(S - Sr)
        prof = " ".join("l%d:%.0e" % (l, d[l * l:(l + 1) ** 2].max()) for l in sorted({x for x in (1, 4, 8, 12, L) if x <= L}))
        print("L=%2d %-8s mean %.2e | cov/max %.2e | %s | sym %.1e" % (
            L, name, np.abs(mu - mr).max() / np.abs(mr).max(), d.max() / np.abs(Sr).max(), prof, np.abs(S - S.T).max()))
e = get_engine(15, 2, 0)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(10):
        mu, S = ylm_moments_device(e, r=20.0 + 0.1 * i, a=0.4, b=0.27)
    torch.cuda.synchronize()
    print("device upstream: %.2f ms per call" % ((time.perf_counter() - t0) / 10 * 1e3))
t0 = time.perf_counter()
for i in range(5):
    upstream.ylm_moments(r=20.0 + 0.1 * i, a=0.4, b=0.27, ydeg=15)
print("host upstream: %.2f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3))
