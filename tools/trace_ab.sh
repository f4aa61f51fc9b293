#!/bin/bash
# Kernel trace of one step at a time in the one-launch-per-panel mode, old (SP_PANEL2=0) and new
# panel kernel; prints the launches of one step in order.  Run ON THE GPU BOX from the repo root.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/trace_ab
rm -rf $O; mkdir -p $O
for p in ${PANELS:-0 1}; do
  SP_PANEL2=$p SP_ONELAUNCH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$p -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 20 --warmup 5 > $O/p$p.log 2>&1 || exit 1
  f=$(ls $O/p$p/*/*kernel_trace.csv | head -1)
  echo "== SP_PANEL2=$p"; python3 tools/trace_step.py $f > $O/p$p.trace.txt; cat $O/p$p.trace.txt
  python3 tools/kstats.py $O/p$p 25 > $O/p$p.kstats.txt
  grep -o '"one_step_at_a_time": {[^}]*' $O/p$p.log | head -1
  grep -o '"value": [0-9.]*' $O/p$p.log | head -1
done
