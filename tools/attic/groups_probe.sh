#!/bin/bash
# star groups on concurrent streams inside one call (SP_GROUPS) x steps in flight: the bench line's value and ms/step
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
for G in 1 2 4; do
  for F in 1 4; do
    echo "== SP_GROUPS=$G in-flight $F" 
    SP_GROUPS=$G python bench.py --steps 60 --warmup 10 --in-flight $F --no-cpu --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('max_rel_err_vs_oracle'))"
  done
done
