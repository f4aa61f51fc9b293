#!/bin/bash
# Refresh the evidence under profiles/: run ON THE GPU BOX from the repository root
#   bash tools/profile_all.sh
# Every rocprofv3 pass has the program directly after `--` and its own timeout; the PMC
# passes use --kernel-trace only (tools/README.md).  Kernel-level evidence (stats, PMC) is taken one
# step at a time (--in-flight 1): with several steps in flight a kernel's duration includes
# the time it shares the GPU with the other steps' kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu --in-flight 1 --steps 10 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -- python3 bench.py --no-cpu --in-flight 1 --steps 5 --warmup 1 > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_fetch -- python3 bench.py --no-cpu --in-flight 1 --steps 5 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_write -- python3 bench.py --no-cpu --in-flight 1 --steps 5 --warmup 1 > $O/pmc_write.log 2>&1
# the one-launch-per-panel mode (what the default's handles run), one step at a time
SP_ONELAUNCH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_onelaunch -- python3 bench.py --no-cpu --in-flight 1 --steps 10 > $O/stats_onelaunch.log 2>&1
# the default command (three steps in flight) under the kernel trace as well
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflight -- python3 bench.py --no-cpu --steps 30 > $O/stats_inflight.log 2>&1
timeout 300 python3 tools/bench_cfg5.py 32 > $O/cfg5.json 2> $O/cfg5.err
ls $O
