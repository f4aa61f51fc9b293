"""BASELINE cfg5's per-GPU shape (ydeg 20, K 3000, 32 stars, Matern-3/2, u = [0.4, 0.2]) through bench.bench_shape, planned:
python tools/cfg5_run.py [F] [steps]      (for kernel traces / statistics: tools/kstats_cmd.sh cfg5 tools/cfg5_run.py 1 6)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
r = bench.bench_shape(torch, None, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0, u=(0.4, 0.2), conditional=False, F=F,
                      steps=steps, device=0, planned=True)
print(json.dumps(r))
