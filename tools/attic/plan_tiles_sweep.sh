#!/bin/bash
# SP_PLAN_TILES (written tiles per workgroup of the planned assembly): in flight and one step at a time.  On the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
for T in ${@:-2 3 4 6 8 12}; do
  SP_PLAN_TILES=$T timeout -k 10 300 python3 bench.py --no-cpu --no-extras --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SP_PLAN_TILES=$T  value %.0f  ms %.4f  one at a time %.4f ms  parity %s' % (d['value'], d['ms_per_step'], d['one_step_at_a_time']['ms_per_step'], d['parity_ok']))"
done
