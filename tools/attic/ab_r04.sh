#!/bin/bash
# (round 4) one GPU call: parity tests of the product engines, then an A/B of library variants
#   bash tools/ab_r04.sh "<variants>" [rounds] [pytest -k expression]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
mkdir -p gpurun_out
K=${3:-"gemm or chol or lnlike or drivers or cond"}
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_drivers.py tests/test_gpu_lnlike.py -x -q -m gpu -k "$K" > gpurun_out/ab_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/ab_tests.log
tail -3 gpurun_out/ab_tests.log
bash tools/variant_ab.sh "$1" ${2:-2} 2>&1 | tee gpurun_out/ab_variants.log
