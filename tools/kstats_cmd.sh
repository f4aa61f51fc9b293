#!/bin/bash
# kernel statistics of any command under rocprofv3 (on the GPU box):  bash tools/kstats_cmd.sh <name> <python script and args>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=$1; shift
O=gpurun_out/ks_$N
rm -rf $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 "$@" > $O.log 2>&1
f=$(find $O -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/${N}_kernel_stats.csv
cp $(find $O -name '*kernel_trace.csv' | head -1) gpurun_out/${N}_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${N}_kernel_stats.csv")))
for r in rows[:22]:
    print("%-100s %7s calls %9.1f us avg %6.2f %%" % (r["Name"].replace("(anonymous namespace)::","")[:100], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
