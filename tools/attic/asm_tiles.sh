#!/bin/bash
# the assembly kernel's duration under different chunkings (tiles per workgroup), one step at a time:
#   bash tools/asm_tiles.sh "4 8 17 34"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for W in ${1:-8 17}; do
  O=gpurun_out/asmt_$W; rm -rf $O
  SP_ASM_TILES=$W timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/cfg3_sweep.py 1 30 > $O.log 2>&1
  echo "SP_ASM_TILES=$W: $(python3 tools/kstats.py $O 36 2>/dev/null | grep assemble_sums)  $(grep '^{' $O.log | tail -1)"
done
