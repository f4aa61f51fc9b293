#!/bin/bash
# like env_ab.sh, four steps in flight only, longer runs: bash tools/env_ab_inflight.sh "VAR=a VAR=b" [rounds] [steps]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
SETTINGS=${1:-"-"}; ROUNDS=${2:-3}; STEPS=${3:-240}
for r in $(seq $ROUNDS); do
for S in $SETTINGS; do
  echo -n "$S in-flight 4: "
  if [ "$S" = "-" ]; then E=""; else E="$S"; fi
  env $E python bench.py --steps $STEPS --warmup 20 --in-flight 4 --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d.get('parity_ok'))"
done
done
