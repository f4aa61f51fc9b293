#!/bin/bash
# Build a variant of the library for an A/B run on one box:
#   bash tools/ab_build.sh <name> "<extra hipcc flags>"      ->  starry_process_amd/libsp_hip_<name>.so
# select it at run time with SP_LIB_VARIANT=<name> (starry_process_amd/_lib.py; debug only).
set -e
cd "$(dirname "$0")/../starry_process_amd/csrc"
name=$1; shift
tmp=$(mktemp -d)
for f in sp_host.cpp sp_wigner.hip sp_gemm.hip sp_panel.hip sp_cond.hip sp_upstream.hip sp_samples.hip sp_table.hip sp_assemble.hip sp_plan.hip sp_planasm.hip sp_small.hip sp_cholesky.hip sp_grad.hip sp_api.hip; do
  fl="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-fast-math"
  case $f in sp_wigner.hip|sp_assemble.hip|sp_plan.hip|sp_table.hip|sp_host.cpp) fl="$fl -ffp-contract=off";; esac
  case $f in sp_cholesky.hip|sp_gemm.hip|sp_panel.hip|sp_cond.hip|sp_planasm.hip|sp_small.hip) fl="$fl -mllvm -amdgpu-mfma-vgpr-form=1";; esac
  /opt/rocm/bin/hipcc $fl "$@" -I. -c $f -o $tmp/${f%.*}.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done      # (set -e: a failed compile stops the build)
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o ../libsp_hip_$name.so $tmp/*.o
rm -rf $tmp
echo built ../libsp_hip_$name.so
