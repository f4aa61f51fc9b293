#!/bin/bash
# Kernel stats of the in-flight regime (the driver's command under the kernel trace).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/stats_inflight
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $O/run.log 2>&1 || exit 1
python3 tools/kstats.py $O/run 485 | tee $O/kstats.txt
grep -o '"value": [0-9.]*' $O/run.log | head -1
