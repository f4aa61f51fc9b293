#!/bin/bash
# (round 4) per-kernel durations of one step at a time for library variants: bash tools/kstats_r04.sh "<variants>" [in-flight]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
F=${2:-1}
for V in $1; do
  if [ "$V" = "-" ]; then unset SP_LIB_VARIANT; else export SP_LIB_VARIANT=$V; fi
  O=gpurun_out/ks_${V/-/default}_F$F; rm -rf $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu --no-extras --in-flight $F --steps 20 --warmup 5 > $O.log 2>&1
  echo "== variant $V (in flight $F)"; python3 tools/kstats.py $O 25 2>/dev/null | head -16
done
