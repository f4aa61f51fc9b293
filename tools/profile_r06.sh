#!/bin/bash
# Round 6's evidence in one GPU call: run ON THE GPU BOX from the repository root
#   bash tools/profile_r06.sh [a|b|c|d]        (parts: gpurun gives a call 20 minutes; without an argument: everything)
# Raw output goes to gpurun_out/r06 (scratch); tools/collect_r06.py copies the judged summaries into
# profiles/ (tracked).  Every rocprofv3 pass has the program directly after `--` and its own timeout;
# the PMC passes use --kernel-trace only.  Kernel-level PMC evidence is taken one step at a time (with
# several steps in flight a dispatch shares its counters' window with the other steps' kernels).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
O=gpurun_out/r06
PART=${1:-abcd}
mkdir -p $O
if [[ $PART == *a* ]]; then
rm -rf $O; mkdir -p $O
# the driver's exact command
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2> $O/bench.err
# the same command under the kernel trace (no CPU leg, no extra shapes: same GPU work in the timed region)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflight -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > $O/stats_inflight.log 2>&1
# one step at a time
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_one -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 20 --warmup 5 > $O/stats_one.log 2>&1
f=$(ls $O/stats_one/*/*kernel_trace.csv | head -1); python3 tools/trace_step.py $f > $O/one_step_trace.txt
fi
if [[ $PART == *b* ]]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_$c -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 4 --warmup 1 --prewarm-ms 0 > $O/pmc_$c.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 4 --warmup 1 --prewarm-ms 0 > $O/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $O/pmc_grbm -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 4 --warmup 1 --prewarm-ms 0 > $O/pmc_grbm.log 2>&1
# LDS bank conflicts per kernel (its own pass), planned against unplanned on this box, the unplanned step's kernels in flight
bash tools/pmc_lds.sh "" > $O/pmc_lds.txt 2>&1
# executed MFMA work per kernel (its own counter pass) and the budget against the algorithmic count
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 -d $O/pmc_mfma -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 4 --warmup 1 --prewarm-ms 0 > $O/pmc_mfma.log 2>&1
n=$(grep -c assemble_planned_kernel $(ls $O/pmc_mfma/*/*kernel_trace.csv | head -1))
python3 tools/mfma_budget.py $O/pmc_mfma $n > $O/mfma_budget.txt 2>&1
fi
if [[ $PART == *c* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_unpl -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --unplanned > $O/stats_unpl.log 2>&1
python3 tools/kstats.py $O/stats_unpl 25 > $O/stats_unplanned_inflight.txt 2>&1
# hyperparameter samples in batches (cfg2 the way the reference is driven): the bench lines alone, their kernels
timeout -k 10 300 python3 tools/bench_samples.py 4 > $O/samples_bench.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_samples -- python3 tools/bench_samples.py 4 40 > $O/stats_samples.log 2>&1
# the K sweep alone (the driver's line carries it too)
timeout -k 10 600 python3 tools/k_sweep.py 4 > $O/k_sweep.txt 2>&1
# cfg5's shape: kernel statistics and the launches of one step, one step at a time
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -- python3 tools/cfg5_run.py 1 6 > $O/stats_cfg5.log 2>&1
f=$(ls $O/stats_cfg5/*/*kernel_trace.csv | head -1); python3 tools/trace_step.py $f > $O/cfg5_one_step_trace.txt 2>&1
# short light curves: the one-kernel path against the blocked one, and its scaling with the number of stars
timeout -k 10 300 python3 tools/small_k_probe.py 64 128 > $O/small_k_probe.txt 2>&1
SP_SMALL_K=0 timeout -k 10 300 python3 tools/k_sweep.py 4 64 128 > $O/k_sweep_blocked_small.txt 2>&1
fi
if [[ $PART == *d* ]]; then
# the sampler level and the concurrency probe
timeout -k 10 300 python3 tools/prof_elp.py > $O/prof_elp.txt 2>&1
timeout -k 10 300 python3 tools/inflight_probe.py 1 2 3 4 > $O/inflight_probe.txt 2>&1
# the ensemble gradient and cfg5's shape
timeout -k 10 300 python3 tools/grad_timing.py > $O/grad_timing.txt 2>&1
timeout -k 10 600 python3 tools/cfg5_planned_ab.py 2 > $O/cfg5_shape.txt 2>/dev/null
fi
python3 tools/collect_r06.py $O
ls $O
