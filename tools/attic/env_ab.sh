#!/bin/bash
# A/B of environment settings on one box: bash tools/env_ab.sh "VAR=a VAR=b ..." [rounds]; "-" = nothing set
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
SETTINGS=${1:-"-"}; ROUNDS=${2:-2}
for r in $(seq $ROUNDS); do
for S in $SETTINGS; do
  for F in 1 4; do
    echo -n "$S in-flight $F: "
    if [ "$S" = "-" ]; then E=""; else E="$S"; fi
    env $E python bench.py --steps 100 --warmup 10 --in-flight $F --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d.get('parity_ok'))"
  done
done
done
