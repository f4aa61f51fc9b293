#!/bin/bash
# per-launch timeline of ONE step at cfg5's shape, one step at a time: bash tools/cfg5_trace.sh   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/cfg5_trace; rm -rf $O
cat > /tmp/cfg5_run.py <<PY
import json, os, sys
sys.path.insert(0, "$ROOT")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, torch.distributed as dist
import bench
r = bench.bench_shape(torch, dist, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0, u=(0.4, 0.2), conditional=False, F=1, steps=4, device=0, planned=True)
print(json.dumps({k: r[k] for k in ("evals_per_s", "ms_per_step", "whole_step_frac")}))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 /tmp/cfg5_run.py > $O.log 2>&1
tail -1 $O.log
python3 tools/trace_step.py $(ls $O/*/*kernel_trace.csv | head -1) > gpurun_out/cfg5_trace.txt
rm -f $O/*/*kernel_trace.csv
tail -80 gpurun_out/cfg5_trace.txt
