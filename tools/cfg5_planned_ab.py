#!/usr/bin/env python3
"""cfg5's shape (ydeg 20, K 3000, Matern-3/2, u = [0.4, 0.2], 32 stars) planned against unplanned, alternating on one
box: python tools/cfg5_planned_ab.py [rounds] [steps]      (environment: SP_PLAN_TEMPORAL_LAZY, SP_SUPER, ...)"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %r)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, torch.distributed as dist
import bench
out = {}
for F in (4, 1):
    r = bench.bench_shape(torch, dist, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0, u=(0.4, 0.2), conditional=False, F=F,
                          steps=int(sys.argv[2]) if F > 1 else 8, device=0, planned=sys.argv[1] == "1")
    out[F] = (r["evals_per_s"], r["whole_step_frac"], r["finite"])
print(json.dumps(out))
""" % ROOT


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = sys.argv[2] if len(sys.argv) > 2 else "24"
    cfgs = [("planned", "1", {}), ("planned-all-assembled", "1", {"SP_PLAN_TEMPORAL_LAZY": "0"}), ("unplanned", "0", {})]
    res = {c[0]: [] for c in cfgs}
    for _ in range(rounds):
        for label, pl, env in cfgs:
            out = subprocess.run([sys.executable, "-c", CHILD, pl, steps], env=dict(os.environ, **env),
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600).stdout.decode()
            res[label].append(json.loads(out.strip().splitlines()[-1]))
    for label, _, _ in cfgs:
        v = res[label]
        f4 = [x["4"][0] for x in v]
        f1 = [x["1"][0] for x in v]
        print("%-22s four in flight %6.0f evals/s (%.0f .. %.0f), whole step %.3f of peak;  one at a time %6.0f  finite %s"
              % (label, sum(f4) / len(f4), min(f4), max(f4), sum(x["4"][1] for x in v) / len(v), sum(f1) / len(f1),
                 all(x["4"][2] and x["1"][2] for x in v)))


if __name__ == "__main__":
    main()
