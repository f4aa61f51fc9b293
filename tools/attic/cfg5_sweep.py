#!/usr/bin/env python
"""cfg5's shape (ydeg 20, K 3000, Matern-3/2, u = [0.4, 0.2], 32 stars per GPU) under different settings of the
factorisation: python tools/cfg5_sweep.py [F] [steps]   (environment: SP_SUPER, SP_LAZY_COV, ...)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
r = bench.bench_shape(torch, dist, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0, u=(0.4, 0.2), conditional=False, F=F,
                      steps=steps, device=0)
print(json.dumps({k: r[k] for k in ("steps_in_flight", "evals_per_s", "ms_per_step", "whole_step_frac", "finite")}))
