#!/usr/bin/env python
"""cfg3's conditional branch (64 stars, K 1000, i per star) through bench.bench_shape: python tools/cond_sweep.py [F] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
r = bench.bench_shape(torch, dist, ydeg=15, Kc=1000, S=64, tspan=4.0, tau=None, u=(0.0, 0.0), conditional=True, F=F,
                      steps=steps, device=0)
print(json.dumps({k: r[k] for k in ("steps_in_flight", "evals_per_s", "ms_per_step", "whole_step_frac", "finite")}))
