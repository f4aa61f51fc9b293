#!/bin/bash
# steps in flight: the bench line's value for F = 2, 3, 4, 5, alternating on one box: bash tools/depth_ab.sh [rounds] [steps]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
ROUNDS=${1:-3}; STEPS=${2:-240}
for r in $(seq $ROUNDS); do
for F in 2 3 4 5; do
  echo -n "in-flight $F: "
  python bench.py --steps $STEPS --warmup 20 --in-flight $F --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d.get('parity_ok'))"
done
done
