#!/bin/bash
# Kernel stats of the in-flight regime (3 steps in flight), old (SP_PANEL2=0) and new panel kernel.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/stats_inflight
rm -rf $O; mkdir -p $O
for p in ${PANELS:-0 1}; do
  SP_PANEL2=$p timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$p -- python3 bench.py --no-cpu --no-extras --steps 60 --warmup 5 > $O/p$p.log 2>&1 || exit 1
  echo "== SP_PANEL2=$p"; python3 tools/kstats.py $O/p$p 485 | tee $O/p$p.kstats.txt
  grep -o '"value": [0-9.]*' $O/p$p.log | head -1
done
