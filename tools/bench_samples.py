"""bench.py's cfg2_batched_samples / ensemble8 lines alone (tools: quick look without the whole bench)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for Sd in (1, 8):
    print(json.dumps(bench.bench_samples(torch, 15, 1000, Sd, F, int(sys.argv[2]) if len(sys.argv) > 2 else 120, 0)))
