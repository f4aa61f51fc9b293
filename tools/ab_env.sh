#!/bin/bash
# A/B of environment settings on ONE box: alternating rounds of the in-flight bench.
#   bash tools/ab_env.sh "<VAR=val[,VAR=val]> <...> ..." [rounds] [extra bench flags]   ("none" = no setting)
sets=$1; rounds=${2:-2}; shift; shift
for r in $(seq $rounds); do for s in $sets; do
  envs=$([ "$s" = none ] && echo "" || echo $s | tr ',' ' ')
  env $envs timeout 250 python bench.py --steps 40 --warmup 5 --no-extras --cpu-stars 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s round $r', round(d['value']), round(d['ms_per_step'],4), round(d.get('one_step_at_a_time',{}).get('ms_per_step',0),4))"
done; done
