#!/bin/bash
# A/B of library variants (tools/ab_build.sh; "-" = the default build) on one box: bash tools/variant_ab.sh "- a b" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
VARIANTS=${1:-"-"}; ROUNDS=${2:-2}
for r in $(seq $ROUNDS); do
for V in $VARIANTS; do
  for F in 1 4; do
    echo -n "variant $V in-flight $F: "
    if [ "$V" = "-" ]; then unset SP_LIB_VARIANT; else export SP_LIB_VARIANT=$V; fi
    python bench.py --steps 100 --warmup 10 --in-flight $F --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d.get('parity_ok'), d.get('max_rel_err_vs_oracle'))"
  done
done
done
