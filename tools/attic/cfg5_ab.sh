#!/bin/bash
# (round 4) cfg5's shape under super-panel widths: bash tools/cfg5_ab.sh "4 6 8 12 16"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for w in $1; do
  for F in 1 4; do
    echo -n "SP_SUPER=$w F=$F: "; SP_SUPER=$w python3 tools/cfg5_sweep.py $F 2>/dev/null | tail -1
  done
done
O=gpurun_out/ks_cfg5; rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/cfg5_sweep.py 1 6 > $O.log 2>&1
python3 tools/kstats.py $O 12 | head -14
