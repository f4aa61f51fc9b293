#!/usr/bin/env python
"""cfg3's step replayed from a captured graph per slot against the eager launches: python tools/graph_probe.py [F] [steps]
(host enqueue per step: ~0.1 ms eager -- 23 launches --, one graph launch captured)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from starry_process_amd.engine import engine_slots, make_stars  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S, Kc, ydeg = 64, 1000, 15
mom = np.load(os.path.join(bench.ROOT, "tests", "golden", "moments_L%d.npz" % ydeg))
mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]
sts = [synthetic_star(s, Kc, 4.0) for s in range(S)]
pairs = engine_slots(ydeg, bench.UDEG, 0, F)
e0 = pairs[0][0]
stars_h = make_stars(S, period=[s["p"] for s in sts], inc_deg=[s["i"] for s in sts], tau=0.0, data_var=1e-6)
inputs = dict(t_d=e0.f64(np.array([s["t"] for s in sts])), f_d=e0.f64(np.array([s["flux"] for s in sts])[:, None, :]),
              stars_d=e0.stars_to_device(stars_h), mu_d=e0.f64(mu), Sig_d=e0.f64(Sig), rta1_d=e0.f64(e0.rTA1L([0.0, 0.0])),
              temporal=None)
slots = []
for ek, stream in pairs:
    ek.set_moments(mu, Sig)
    sl = bench.Slot(torch, ek, stream, S, Kc, 1, False, dist, bench.COVPTS, False)
    sl.bind(**inputs)
    slots.append(sl)
for _ in range(3):
    for sl in slots:
        sl.run()
torch.cuda.synchronize()
ref = slots[0].out.clone()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3


def prewarm(fn):
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 0.3:
        fn(i)
        i += 1
    torch.cuda.synchronize()


eager = lambda i: slots[i % F].run()
prewarm(eager)
res = {"eager": [timed(eager, steps) for _ in range(5)]}
graphs = []
for sl in slots:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=sl.stream):
        sl.step()
    graphs.append(g)
torch.cuda.synchronize()


def replay(i):
    with torch.cuda.stream(slots[i % F].stream):
        graphs[i % F].replay()


prewarm(replay)
res["graph"] = [timed(replay, steps) for _ in range(5)]
ok = bool(torch.equal(slots[0].out, ref))
for k, v in res.items():
    print(k, "ms per step (wall, host enqueue):", [(round(a, 4), round(b, 4)) for a, b in v],
          " evals/s best %.0f" % (S / min(a for a, _ in v) * 1e3))
print("graph output identical:", ok)
