#!/bin/bash
# (round 4) the assembly kernel under different chunkings / variants: kernel stats, one step at a time
#   bash tools/asm_ab.sh "<variant:wgs> ..."
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for cfg in $1; do
  V=${cfg%%:*}; W=${cfg##*:}
  if [ "$V" = "-" ]; then unset SP_LIB_VARIANT; else export SP_LIB_VARIANT=$V; fi
  export SP_ASM_TILES=$W
  O=gpurun_out/asm_${V}_$W; rm -rf $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 20 --warmup 5 > $O.log 2>&1
  echo "== variant $V SP_ASM_TILES=$W: $(python3 tools/kstats.py $O 25 2>/dev/null | grep assemble_sums)  $(grep '^{' $O.log | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],4), d.get("parity_ok"))')"
done
