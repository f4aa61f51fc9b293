#!/bin/bash
# What each kind of launch costs with four steps in flight: the step without it (library built with -DSP_PROBE;
# results are garbage, only the timing means anything):  bash tools/probe_inflight.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
export SP_LIB_VARIANT=probe
for F in 4 1; do
  echo "== F=$F"
  echo -n "all          "; python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
  echo -n "no assembly  "; SP_PROBE_SKIP_ASM=1 python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
  echo -n "no MM2       "; SP_PROBE_SKIP_MM2=1 python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
  echo -n "no panels SP1"; SP_PROBE_SKIP_PANELS=1 python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
  echo -n "no panels SP2"; SP_PROBE_SKIP_PANELS=2 python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
  echo -n "no panels    "; SP_PROBE_SKIP_PANELS=3 python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
  echo -n "nothing      "; SP_PROBE_SKIP_ASM=1 SP_PROBE_SKIP_MM2=1 SP_PROBE_SKIP_PANELS=3 python3 tools/cfg3_sweep.py $F 240 2>/dev/null | tail -1 | cut -c1-110
done
