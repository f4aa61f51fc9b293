#!/bin/bash
# Where the planned assembly's time goes: its average duration (one step at a time, rocprofv3 kernel trace) in the
# shipped library and in probe variants (tools/ab_build.sh <name> -DP_PLAN_...; garbage results, timing only).
# ONLY the probed kernel's own duration means anything: a probe leaves NaNs in the systems, and the factorisation of
# matrices full of NaNs runs 5 % FASTER with steps in flight (less switching, higher clocks) -- an in-flight rate
# measured with a probe build says nothing (round 5 fell for it once more: DESIGN.md 4.11).
#   bash tools/plan_asm_probe.sh "" noeval nostore ...        (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for V in "$@"; do
  O=gpurun_out/planprobe_${V:-base}; rm -rf $O
  SP_LIB_VARIANT=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 20 --warmup 5 > $O.log 2>&1
  f=$(ls $O/*/*kernel_stats.csv | head -1)
  python3 - "$f" "${V:-base}" "${SP_PLAN_TILES:-default}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "assemble_planned" in r["Name"] or "panel_kernel<1, true, false>" in r["Name"] or "panel_kernel<0" in r["Name"]:
        print("variant %-8s SP_PLAN_TILES=%s  %-32s calls %s avg %.1f us min %.1f max %.1f" % (
            sys.argv[2], sys.argv[3], r["Name"].replace("void (anonymous namespace)::", "")[:32], r["Calls"],
            float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $O/*/*kernel_trace.csv
done
