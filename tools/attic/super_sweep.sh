#!/bin/bash
# cfg3's step under different super-panel widths, one at a time and four in flight, on one box:
#   bash tools/super_sweep.sh "4 6 8 16"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
for w in ${1:-4 6 8}; do
  for F in 1 4; do
    echo -n "SP_SUPER=$w F=$F  "
    SP_SUPER=$w python3 tools/cfg3_sweep.py $F 120 2>/dev/null | tail -1
  done
done
