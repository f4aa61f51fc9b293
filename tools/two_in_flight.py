#!/usr/bin/env python
"""Experiment: consecutive (independent) likelihood steps alternate between NF streams, each with its
own handle, workspace and outputs, so that the latency-bound phases of one step can overlap the
throughput-bound phases of another.  python tools/two_in_flight.py [NF] [steps]"""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
from starry_process_amd.engine import Engine, make_stars
from starry_process_amd.synthetic import synthetic_star

NF = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
S, K, L = 64, 1000, 15
mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
sts = [synthetic_star(s, K) for s in range(S)]
ctx = []
for k in range(NF):
    e = Engine(L, 2, 0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        c = dict(e=e, st=st, t=e.f64(np.array([s["t"] for s in sts])),
                 f=e.f64(np.array([s["flux"] for s in sts])[:, None, :]),
                 stars=e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6)),
                 mu=e.f64(mom["default_mean_ylm"]), Sig=e.f64(mom["default_cov_ylm"]),
                 rta1=e.f64(e.rTA1L([0.0, 0.0])), ws=e.workspace(S, K, 1), out=e.empty(S),
                 status=torch.zeros(S, dtype=torch.int32, device=e.device))
        e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    ctx.append(c)
torch.cuda.synchronize()

def step(c):
    with torch.cuda.stream(c["st"]):
        e = c["e"]
        e.set_moments_dev(c["mu"], c["Sig"])
        tab, mv = e.kernel_table(c["rta1"], 300)
        e.lnlike_ensemble(c["t"], c["f"], c["stars"], covpts=300, tab=tab, meanvar=mv, normalized=True,
                          out=c["out"], status=c["status"], workspace=c["ws"])

for i in range(20):
    step(ctx[i % NF])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(ctx[i % NF])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"in_flight": NF, "ms_per_step": 1e3 * dt / steps, "evals_per_s": S * steps / dt,
                  "lnl0": float(ctx[0]["out"][0].item())}))
