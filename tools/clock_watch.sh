#!/bin/bash
# Samples the GPU clocks / power while the benchmark runs with many timed steps (is the in-flight regime
# power-limited?).  Run ON THE GPU BOX from the repo root.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
( for i in $(seq 1 40); do
    echo "t=$i $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -3 | tr '\n' ' ')"
    sleep 0.5
  done ) > gpurun_out/clock_watch.txt 2>&1 &
W=$!
sleep 1
timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 6000 --warmup 5 --in-flight ${F:-3} > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err
sleep 1
timeout -k 10 300 python bench.py --no-cpu --no-extras --steps 3000 --warmup 5 --in-flight 1 > gpurun_out/clock_bench1.json 2> gpurun_out/clock_bench1.err
wait $W
python - <<'PY'
import json
for f in ("gpurun_out/clock_bench.json","gpurun_out/clock_bench1.json"):
    r=json.load(open(f)); print(f, r["value"], r["ms_per_step"], r["steps_in_flight"])
PY
