#!/bin/bash
# SQ counters of the one-kernel path alone.  Separate --pmc passes, kernel trace only.       on the GPU box:
#   bash tools/small_k_pmc.sh [K]                  one workgroup a CU (S = 256) against four (S = 3072): what co-resident
#                                                  pivot blocks contend for
#   bash tools/small_k_pmc.sh K phases             instruction counts by phase, S = 3072: the variants stop1 .. stop6
#                                                  (bash tools/ab_build.sh stop$k -DSMK_STOP=$k for k in 1 2 3 8 6) return at a
#                                                  phase boundary; differences of SQ_INSTS_* between them
#                                                  (SP_STOPS="stop1 stop2 ...": another list, e.g. K > 64's phases)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
K=${1:-64}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/small_pmc; rm -rf $O; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64"
P3="SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_IFETCH"
run() {   # variant S pass-name counters
  SP_LIB_VARIANT=$1 SP_PROBE_S=$2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $4 -d $O/$3 -- python3 tools/small_k_probe.py $K > $O/$3.log 2>&1 || { echo "$3 failed"; tail -5 $O/$3.log; }
  echo "== variant '$1' S $2 $3"
  grep "^K" $O/$3.log
  python3 tools/pmc.py $O/$3 small_lnlike
  rm -rf $O/$3/*/*kernel_trace.csv
}
if [ "$2" = phases ]; then
  for V in ${SP_STOPS:-stop1 stop2 stop3 stop8 stop6} ""; do
    run "$V" 3072 v${V:-full}_p1 "$P1"
    run "$V" 3072 v${V:-full}_p2 "$P2"
  done
else
  for S in 256 3072; do
    run "" $S s${S}_p1 "$P1"; run "" $S s${S}_p2 "$P2"; run "" $S s${S}_p3 "$P3"
  done
fi
