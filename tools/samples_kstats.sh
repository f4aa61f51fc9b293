#!/bin/bash
# kernel statistics of the batched-samples step (tools/bench_samples.py) under rocprofv3; on the GPU box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ks_samples
rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/bench_samples.py ${1:-4} > $O.log 2>&1
f=$(find $O -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/samples_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/samples_kernel_stats.csv")))
for r in rows[:24]:
    print("%-90s %7s calls %9.1f us avg %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
