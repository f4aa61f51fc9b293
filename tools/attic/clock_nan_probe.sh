#!/bin/bash
# Is the in-flight regime power / clock limited?  Clocks and power sampled while the benchmark runs four steps in flight
# on real data and on the NaN-filled systems of a probe build (tools/ab_build.sh pnostore -DP_PLAN_NOSTORE), which read
# 5 % faster.  On the GPU box: bash tools/clock_nan_probe.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT" || exit 1
mkdir -p gpurun_out
for V in "" pnostore; do
  ( for i in $(seq 1 14); do
      echo "t=$i $(rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' ')"
      sleep 0.5
    done ) > gpurun_out/clock_${V:-real}.txt 2>&1 &
  W=$!
  sleep 1
  SP_LIB_VARIANT=$V timeout -k 10 300 python3 bench.py --no-cpu --no-extras --steps 8000 --warmup 5 --in-flight 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('variant \"${V:-real}\": %.0f evals/s, %.4f ms per step' % (d['value'], d['ms_per_step']))"
  wait $W
  sed -n 5,9p gpurun_out/clock_${V:-real}.txt
done
