#!/bin/bash
# One GPU call of the edit-measure loop (round 5): the planned-path tests, the kernel trace of one step at a time and the
# driver's command without the CPU leg.  Run ON THE GPU BOX from the repository root:  bash tools/quick_r05.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
T=${1:-q}
O=gpurun_out/r05_$T
rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_planned.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 20 --warmup 5 > $O/stats_one.log 2>&1
f=$(ls $O/stats_one/*/*kernel_trace.csv | head -1); python3 tools/trace_step.py $f > $O/one_step_trace.txt; rm -f $O/stats_one/*/*kernel_trace.csv
head -8 $O/one_step_trace.txt; tail -1 $O/one_step_trace.txt
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu ${QUICK_BENCH_ARGS} > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms %.4f parity %s" % (d["value"], d["ms_per_step"], d["parity_ok"]))
print("one %.4f sustained %s" % (d["one_step_at_a_time"]["ms_per_step"], d["sustained"] and "%.0f" % d["sustained"]["evals_per_s"]))
u=d.get("unplanned")
if u: print("unplanned sustained %.0f one %.4f diff %.2e" % (u["evals_per_s"], u["one_step_at_a_time_ms"], u["max_rel_diff_planned"]))
o=d.get("other_shapes") or {}
for k in ("cfg5_shape","cfg5_shape_unplanned","cfg3_conditional"):
    if k in o: print(k, "%.0f %.3f" % (o[k]["evals_per_s"], o[k]["whole_step_frac"]))
if "error" in o: print("extras error", o["error"])
print("roof frac %.3f whole %.3f" % (d["roofline"]["frac"], d["roofline"]["whole_step"]["frac"]))
PY
