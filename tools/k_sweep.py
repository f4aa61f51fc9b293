"""bench.py's K sweep alone (the reference's benchmark protocol, joss/figures/speed.py:22-37): python tools/k_sweep.py [F] [K ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Ks = tuple(int(x) for x in sys.argv[2:]) or (64, 128, 256, 512, 1000, 2048, 4096)
res = bench.bench_k_sweep(torch, None, F, 0, Ks)
for k, v in res.items():
    print(k, json.dumps(v))
