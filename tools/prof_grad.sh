cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks_grad
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_grad -- python3 tools/grad_timing.py > gpurun_out/ks_grad.log 2>&1
python3 tools/kstats.py gpurun_out/ks_grad 1 | head -30
