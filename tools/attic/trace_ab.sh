#!/bin/bash
# Kernel trace of one step at a time: prints the launches of one step in order, the per-kernel
# totals, and the bench line's one-at-a-time figure.  Run ON THE GPU BOX from the repo root.
#   VARIANTS="" (the library) or e.g. VARIANTS=" trace" (library variants of tools/ab_build.sh)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/trace_ab
rm -rf $O; mkdir -p $O
for v in ${VARIANTS:-main}; do
  lib=$v; [ "$v" = main ] && lib=""
  SP_LIB_VARIANT=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 20 --warmup 5 > $O/$v.log 2>&1 || exit 1
  f=$(ls $O/$v/*/*kernel_trace.csv | head -1)
  echo "== variant $v"; python3 tools/trace_step.py $f > $O/$v.trace.txt; cat $O/$v.trace.txt
  python3 tools/kstats.py $O/$v 25 > $O/$v.kstats.txt
  grep -o '"value": [0-9.]*' $O/$v.log | head -1
done
