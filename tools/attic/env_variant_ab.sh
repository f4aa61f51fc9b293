#!/bin/bash
# (round 4) A/B of "variant:ENV=value" settings on one box: bash tools/env_variant_ab.sh "-: base: -:SP_ASM_TILES=1" [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
for r in $(seq ${2:-2}); do
for cfg in $1; do
  V=${cfg%%:*}; E=${cfg#*:}
  for F in 1 4; do
    echo -n "variant $V env '$E' in-flight $F: "
    ( if [ "$V" != "-" ]; then export SP_LIB_VARIANT=$V; fi; if [ -n "$E" ]; then export $E; fi
      python bench.py --steps 100 --warmup 10 --in-flight $F --no-cpu --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d.get('parity_ok'))" )
  done
done
done
