#!/usr/bin/env python
"""End-to-end cost of one sampler step through calibrate.get_log_prob_ensemble: 64 stars,
K = 1000, own period each, hyperparameters changing every call -- with the reference's
moment algorithm on the host and with the device upstream; and the vectorised
form (EnsembleLogProb: 60 samples per call, 1 or 3 in flight)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.calibrate import get_log_prob_ensemble
from starry_process_amd.synthetic import synthetic_star

S, K = 64, 1000
sts = [synthetic_star(s, K) for s in range(S)]
t = np.array([s["t"] for s in sts]); flux = np.array([s["flux"] for s in sts]); p = np.array([s["p"] for s in sts])
out = {}
for how in ("reference", "device"):
    f = get_log_prob_ensemble(t, flux, ferr=1e-3, p=p, upstream=how)
    f(20.0, 0.4, 0.27, 0.1, 10.0)
    torch.cuda.synchronize()
    nrep = 5 if how == "reference" else 30
    t0 = time.perf_counter()
    for i in range(nrep):
        v = f(20.0 + 0.01 * i, 0.4, 0.27, 0.1, 10.0)
    dt = (time.perf_counter() - t0) / nrep
    out[how] = {"ms_per_call": 1e3 * dt, "stars_per_s": S / dt, "value": v}
# many samples per call, data resident, samples in flight (calibrate.EnsembleLogProb)
from starry_process_amd.calibrate import EnsembleLogProb
samples = np.array([[20.0 + 0.01 * i, 0.4, 0.27, 0.1, 10.0] for i in range(60)])
for depth in (1, 3):
    lp = EnsembleLogProb(t, flux, ferr=1e-3, p=p, depth=depth)
    lp(samples[:6])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vals = lp(samples)
    dt = (time.perf_counter() - t0) / len(samples)
    out["EnsembleLogProb_depth%d" % depth] = {"ms_per_sample": 1e3 * dt, "stars_per_s": S / dt, "value": float(vals[-1])}
print(json.dumps(out))
