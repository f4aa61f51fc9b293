#!/bin/bash
# LDS bank-conflict counters of the step's kernels, one step at a time (separate --pmc pass, kernel trace only):
#   bash tools/pmc_lds.sh [variant ...]        ("" = the shipped library)        on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for V in "${@:-}"; do
  O=gpurun_out/pmc_lds_${V:-base}; rm -rf $O
  SP_LIB_VARIANT=$V timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $O -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 4 --warmup 1 --prewarm-ms 0 > $O.log 2>&1
  echo "== variant '${V:-base}'"
  python3 tools/pmc.py $O | python3 -c "
import sys
name=None; vals={}
def flush():
    if name and 'SQ_LDS_IDX_ACTIVE' in vals and vals['SQ_LDS_IDX_ACTIVE']>0:
        print('%-46s conflicts / active = %.3f   LDS wait / wave cycles = %.3f' % (name[:46], vals.get('SQ_LDS_BANK_CONFLICT',0)/vals['SQ_LDS_IDX_ACTIVE'], vals.get('SQ_WAIT_INST_LDS',0)/max(vals.get('SQ_WAVE_CYCLES',1),1)))
for line in sys.stdin:
    if not line.startswith('   '):
        flush(); name=line.split(' dispatches')[0]; vals={}
    else:
        p=line.split(); vals[p[0]]=float(p[4])
flush()"
  rm -rf $O/*/*kernel_trace.csv
done
