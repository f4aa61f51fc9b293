#!/bin/bash
# A/B of library variants on ONE box: alternating rounds of the in-flight bench.
#   bash tools/ab_run.sh "<variant> <variant> ..." [rounds] [extra bench flags]
vars=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do for v in $vars; do
  SP_LIB_VARIANT=$([ "$v" = base ] && echo "" || echo $v) timeout 250 python bench.py --steps 40 --warmup 5 --no-extras --cpu-stars 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r', round(d['value']), round(d['ms_per_step'],4), round(d.get('one_step_at_a_time',{}).get('ms_per_step',0),4))"
done; done
