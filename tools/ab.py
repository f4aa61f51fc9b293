#!/usr/bin/env python3
"""Alternating A/B runs of bench.py on ONE box (the only comparison this project trusts: box to box the in-flight
numbers move by +-5 %).

    python tools/ab.py [--rounds 3] [--steps 240] label[:ENV=v,ENV=v][:--flag,--flag] ...

e.g.  python tools/ab.py base  nofuse:SP_PLAN_FUSE0=0  unplanned::--unplanned  var:SP_LIB_VARIANT=x

Every configuration is run once per round, in the given order; per configuration: evaluations/s with the steps in
flight over `steps` timed steps, and the one-step-at-a-time figure of the same run (mean, min .. max over the rounds).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--in-flight", type=int, default=4)
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    cfgs = []
    for c in a.configs:
        parts = c.split(":")
        env = dict(kv.split("=", 1) for kv in parts[1].split(",") if kv) if len(parts) > 1 else {}
        flags = [f for f in parts[2].split(",") if f] if len(parts) > 2 else []
        cfgs.append((parts[0], env, flags))
    res = {c[0]: [] for c in cfgs}
    for r in range(a.rounds):
        for label, env, flags in cfgs:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "10",
                   "--in-flight", str(a.in_flight), "--no-cpu", "--no-extras"] + flags
            out = subprocess.run(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=600).stdout.decode()
            try:
                d = json.loads(out.strip().splitlines()[-1])
                res[label].append((d["value"], d["one_step_at_a_time"]["ms_per_step"], d["parity_ok"]))
            except Exception as exc:
                print("%s: run failed (%r)" % (label, exc))
            sys.stdout.flush()
    for label, _, _ in cfgs:
        v = res[label]
        if not v:
            continue
        fl = [x[0] for x in v]
        one = [x[1] for x in v]
        print("%-14s in flight %8.0f evals/s (%.0f .. %.0f)   one at a time %.4f ms (%.4f .. %.4f)   parity %s   n=%d"
              % (label, sum(fl) / len(fl), min(fl), max(fl), sum(one) / len(one), min(one), max(one),
                 all(x[2] for x in v), len(v)))


if __name__ == "__main__":
    main()
