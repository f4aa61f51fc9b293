#!/bin/bash
# kernel trace of the ensemble gradient (tools/grad_timing.py) under rocprofv3; on the GPU box: bash tools/prof_grad.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
rm -rf gpurun_out/ks_grad
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_grad -- python3 tools/grad_timing.py > gpurun_out/ks_grad.log 2>&1
python3 tools/kstats.py gpurun_out/ks_grad 1 | head -30
