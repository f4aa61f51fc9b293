#!/usr/bin/env python3
"""cfg5's shape (ydeg 20, K 3000, Matern-3/2, u = [0.4, 0.2], 32 stars, planned), alternating runs of environment
variants on one box:  python tools/cfg5_env_ab.py [--rounds 2] [--steps 24] [--flight 4,1] label[:ENV=v,ENV=v] ..."""
import argparse
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
for F in [int(x) for x in sys.argv[2].split(",")]:
    r = bench.bench_shape(torch, dist, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0, u=(0.4, 0.2), conditional=False, F=F,
                          steps=int(sys.argv[1]) if F > 1 else 8, device=0, planned=True)
    out[F] = (r["evals_per_s"], r["whole_step_frac"], r["finite"])
print(json.dumps(out))
""" % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", default="24")
    ap.add_argument("--flight", default="4,1")
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    cfgs = []
    for c in a.configs:
        parts = c.split(":")
        cfgs.append((parts[0], dict(kv.split("=", 1) for kv in parts[1].split(",") if kv) if len(parts) > 1 else {}))
    res = {c[0]: [] for c in cfgs}
    for _ in range(a.rounds):
        for label, env in cfgs:
            out = subprocess.run([sys.executable, "-c", CHILD, a.steps, a.flight], env=dict(os.environ, **env),
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900).stdout.decode()
            try:
                res[label].append(json.loads(out.strip().splitlines()[-1]))
            except Exception as exc:
                print("%s: run failed (%r)" % (label, exc))
            sys.stdout.flush()
    for label, _ in cfgs:
        v = res[label]
        if not v:
            continue
        line = "%-16s" % label
        for F in a.flight.split(","):
            x = [r[F][0] for r in v]
            line += "  F=%s %6.0f evals/s (%.0f .. %.0f) %.3f of peak" % (F, sum(x) / len(x), min(x), max(x),
                                                                          sum(r[F][1] for r in v) / len(v))
        print(line + "  finite %s" % all(r[F][2] for r in v for F in a.flight.split(",")))


if __name__ == "__main__":
    main()
