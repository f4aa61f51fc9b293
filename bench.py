#!/usr/bin/env python
"""
bench.py -- log_likelihood evaluations per second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one-GPU share of configs[3]):
  ydeg = 15, K = 1000 cadences, 64 independent stars per GPU, marginalised over
  inclination, normalised, covpts = 300, fp64.  Weak scaling: rank r owns stars
  64 r .. 64 r + 63, so 8 GPUs evaluate the 512-star calibrate ensemble.

One "step" = one hyperparameter sample of an MCMC / nested-sampling loop:
  polar-frame moments of (mu_y, Sigma_y)   (sp_set_ylm_moments_dev)
  -> inclination-marginal kernel table      (sp_kernel_table)
  -> covariance assembly + Cholesky + solve + reduction for every star
                                            (sp_lnlike_ensemble)
  -> N > 1: RCCL all-gather of the per-star log-likelihoods (torch.distributed).
Consecutive steps are independent samples (the walkers / live points a sampler
evaluates per iteration), so --in-flight F of them (default 3) are kept in flight:
step i runs on stream i mod F with its own library handle, workspace and outputs.
The latency-bound phases of one step (diagonal blocks, panel solves) then overlap
the throughput-bound phases of its neighbours (assembly, trailing updates); every
step still does all of its work, and K steps are timed between the same barriers.
All inputs are resident in HBM when the timed region starts.  (mu_y, Sigma_y) come
from tests/golden (the upstream integrals are outside the hot path).

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

YDEG, UDEG, K, STARS_PER_GPU, COVPTS = 15, 2, 1000, 64, 300
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix = vector peak (AMD CDNA4 datasheet)


def cpu_baseline(nstars, timeout=240.0):
    """The oracle (a NumPy/SciPy/C port of the reference's CPU path, LAPACK potrf and
    trtrs exactly like reference math.py:75-100) timed on the host cores: the stars are
    farmed out to single-threaded worker processes (oracle/cpu_worker.py, plain child
    processes that never touch the GPU), the way an ensemble would run on a CPU node;
    value = stars / slowest worker's compute time, cores = workers."""
    import subprocess

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(ncpu, 32, nstars))
    per = -(-nstars // workers)
    spans = [(w * per, min(nstars, (w + 1) * per)) for w in range(workers)]
    spans = [sp for sp in spans if sp[0] < sp[1]]
    mom = os.path.join(ROOT, "tests", "golden", "moments_L15.npz")
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env.pop("LD_PRELOAD", None)   # (a profiler's preload has no business in the CPU workers)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), mom,
                               str(a), str(b), str(K), str(YDEG), str(UDEG)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
             for a, b in spans]
    vals = np.full(nstars, np.nan)
    slowest, done = 0.0, 0
    deadline = time.perf_counter() + timeout
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, deadline - time.perf_counter()))
            rec = json.loads(out.decode().strip().splitlines()[-1])
            vals[rec["stars"]] = rec["values"]
            slowest = max(slowest, rec["seconds"])
            done += len(rec["stars"])
        except Exception:
            pr.kill()
    if done == 0:
        return None, vals
    return dict(value=done / slowest, unit="evals/s", cores=len(spans), kind="port",
                sample="%d stars (ydeg=15, K=1000, marginal, normalized) over %d single-threaded worker "
                       "processes, oracle.OracleProcess.log_likelihood with SciPy/LAPACK potrf/trtrs"
                       % (done, len(spans))), vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--in-flight", type=int, default=3,
                    help="independent steps kept in flight on separate streams / handles")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed pre-warm after the --warmup steps: whole steps keep running until this "
                         "much wall time has passed and every slot has run 3 times (clocks and fabric "
                         "at their loaded state); the timed region is exactly --steps steps either way")
    ap.add_argument("--cpu-stars", type=int, default=128)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    # SP_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL communicator, all-gather,
    # barriers, max over ranks) with however many ranks there are, even one -- lets a
    # single-GPU box exercise it
    use_dist = world > 1 or os.environ.get("SP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from starry_process_amd.engine import engine_slots, get_engine, make_stars
    from starry_process_amd.synthetic import synthetic_star

    mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
    mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]

    e = get_engine(YDEG, UDEG, local_rank)
    S = STARS_PER_GPU
    first = rank * S
    sts = [synthetic_star(s, K) for s in range(first, first + S)]
    t_d = e.f64(np.array([s["t"] for s in sts]))
    f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
    stars_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
    mu_d, Sig_d = e.f64(mu), e.f64(Sig)
    rta1_d = e.f64(e.rTA1L([0.0, 0.0]))          # one flux operator: u = [0, 0]

    # one slot per step in flight: library handle, stream, workspace, outputs (the inputs
    # above are read-only and shared)
    F = max(1, args.in_flight)
    slots = []
    for ek, stream in engine_slots(YDEG, UDEG, local_rank, F):
        ek.set_moments(mu, Sig)  # first call allocates / uploads the lag grid
        slots.append(dict(e=ek, stream=stream, ws=ek.workspace(S, K, 1),
                          out=ek.empty(S), status=torch.zeros(S, dtype=torch.int32, device=ek.device),
                          gathered=ek.empty(world * S) if use_dist else None))
    torch.cuda.synchronize()

    def run_step(c):
        ek = c["e"]
        ek.set_moments_dev(mu_d, Sig_d)
        tab, mv = ek.kernel_table(rta1_d, COVPTS)
        ek.lnlike_ensemble(t_d, f_d, stars_d, covpts=COVPTS, tab=tab, meanvar=mv,
                           normalized=True, out=c["out"], status=c["status"], workspace=c["ws"])
        if use_dist:
            dist.all_gather_into_tensor(c["gathered"], c["out"])
            return c["gathered"].sum()
        return c["out"].sum()

    def step(i=0):
        c = slots[i % F]
        with torch.cuda.stream(c["stream"]):
            return run_step(c)

    if use_dist:
        # communicator set-up (lazy in RCCL) must not land in the timed region even with --warmup 0
        dist.all_gather_into_tensor(slots[0]["gathered"], slots[0]["out"])
    nwarm = max(args.warmup, F if args.warmup else 0)
    for i in range(nwarm):
        step(i)
    # Stated, untimed pre-warm: a fresh process reaches the timed region
    # 5 ms after its first launch otherwise, with the memory / fabric clocks still at their idle
    # state, and the bandwidth-bound in-flight mode then reads like the latency-bound one.
    # Whole steps of the same workload, every slot at least 3 times, at least --prewarm-ms.
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        torch.cuda.synchronize()
        tw = time.perf_counter()
        while prewarm_steps < 3 * F or 1e3 * (time.perf_counter() - tw) < args.prewarm_ms:
            for _ in range(F):
                step(nwarm + prewarm_steps)
                prewarm_steps += 1
            torch.cuda.synchronize()
    nsyrk = (K + 63) // 64  # upper bound on timed launches per step
    for c in slots:
        c["e"].profile_begin((args.steps // F + 1) * nsyrk)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        total = step(i)
    host_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    launches, kern_ms, kern_flops = 0, 0.0, 0.0
    for c in slots:
        a, b, f = c["e"].profile_end()
        launches, kern_ms, kern_flops = launches + a, kern_ms + b, kern_flops + f
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=e.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out, status = slots[0]["out"], slots[0]["status"]
    torch.cuda.synchronize()
    lnl_timed = out.cpu().numpy().copy()
    status_timed = status.cpu().numpy().copy()
    slots_agree = all(torch.equal(c["out"], out) for c in slots[1:min(F, args.steps + nwarm + prewarm_steps)])

    # the same steps one at a time on one stream (not `value`): what a strictly sequential
    # caller gets, and the trailing update's rate when it has the GPU to itself
    one = None
    if world == 1:
        c0 = dict(slots[0])
        if F > 1:   # one evaluation at a time: the shared engine, in its latency-oriented mode
            c0["e"] = e
            c0["ws"] = e.workspace(S, K, 1)
            # (its own outputs: the two panel modes agree to rounding, not bit for bit, and the
            #  slots of the timed region are compared bit for bit below)
            c0["out"] = e.empty(S)
            c0["status"] = torch.zeros(S, dtype=torch.int32, device=e.device)
            e.set_moments(mu, Sig)
        nrep = max(10, min(50, args.steps))
        for _ in range(5):
            run_step(c0)
        c0["e"].profile_begin(nrep * nsyrk)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nrep):
            run_step(c0)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        a, b, f = c0["e"].profile_end()
        one = {"evals_per_s": S * nrep / dt1, "ms_per_step": 1e3 * dt1 / nrep,
               "trailing_update_TFLOPs": (f / (b * 1e-3)) / 1e12 if b > 0 else 0.0}
        one["trailing_update_frac"] = one["trailing_update_TFLOPs"] / FP64_PEAK_TFLOPS

    nran = min(F, args.steps + nwarm + prewarm_steps)   # slots that ran
    lnl = lnl_timed   # (the slots of the timed region: same inputs, same bits)
    ok = bool(np.all(np.isfinite(lnl))) and not bool(status_timed.any()) and slots_agree
    if not ok and rank == 0:
        print("parity check failed: finite %s, status bits set on %d stars, slots agree %s; stars that differ "
              "between slots: %s" % (bool(np.all(np.isfinite(lnl))), int(np.count_nonzero(status_timed)),
                                     slots_agree, [int((c["out"] != out).sum().item()) for c in slots[1:nran]]),
              file=sys.stderr)

    # PCIe-inclusive rate (never `value`): the same step fed from pinned host buffers
    # (t, flux up, log-likelihoods down) -- what a caller without resident data would see
    pcie_rate = None
    if world == 1:
        t_h, f_h = t_d.cpu().pin_memory(), f_d.cpu().pin_memory()
        out_h = torch.empty(S, dtype=torch.float64).pin_memory()
        nrep = max(3, min(20, args.steps))

        def fed_step():
            t_d.copy_(t_h, non_blocking=True)
            f_d.copy_(f_h, non_blocking=True)
            run_step(c0)
            out_h.copy_(c0["out"], non_blocking=True)

        for _ in range(2):          # first use of the pinned buffers maps them (tens of ms, once)
            fed_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nrep):
            fed_step()
        torch.cuda.synchronize()
        pcie_rate = S * nrep / (time.perf_counter() - t1)

    # the hyperparameter-level step (mu_y, Sigma_y from r, a, b, c, n) is outside the timed
    # region by definition of the metric; its cost is reported beside it
    upstream_ms = None
    if world == 1 and rank == 0:
        from starry_process_amd.upstream_device import ylm_moments_device
        from starry_process_amd import upstream as host_upstream

        ylm_moments_device(e, r=20.0, a=0.40, b=0.27, c=0.1, n=10.0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for k in range(10):
            ylm_moments_device(e, r=20.0 + 0.01 * k, a=0.40, b=0.27, c=0.1, n=10.0)
        torch.cuda.synchronize()
        dev_ms = 1e2 * (time.perf_counter() - t2)
        host_upstream.ylm_moments(r=20.0, a=0.40, b=0.27, c=0.1, n=10.0, ydeg=YDEG)
        t2 = time.perf_counter()
        for k in range(2):
            host_upstream.ylm_moments(r=20.0 + 0.01 * k, a=0.40, b=0.27, c=0.1, n=10.0, ydeg=YDEG)
        upstream_ms = {"device_quadrature": dev_ms, "host_reference_algorithm": 5e2 * (time.perf_counter() - t2)}

    if rank == 0:
        evals = world * S * args.steps
        achieved = (kern_flops / (kern_ms * 1e-3)) / 1e12 if kern_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("gemm_nt_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "log_likelihood evals/sec (ydeg=15, K=1000)",
            "value": evals / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "cfg3: ydeg=15, K=1000, 64 stars per GPU batched, marginalize_over_inclination, "
                            "normalized, covpts=300 (cfg4 = the same at 8 GPUs: 512 stars, RCCL all-gather)",
                "stars_per_gpu": S, "ydeg": YDEG, "K": K, "parallelism": "stars sharded %d-way" % world,
                "steps_in_flight": F,
            },
            "parity_ok": ok,
            "prewarm": {"steps": prewarm_steps, "min_ms": args.prewarm_ms, "timed": False},
            "steps_in_flight": F,
            "one_step_at_a_time": one,
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "pcie_inclusive_evals_per_s": pcie_rate,
            "upstream_ms_per_sample": upstream_ms,
            "roofline": {
                "kernel": "gemm_nt_kernel (Cholesky trailing update, v_mfma_f64_16x16x4_f64)",
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "launches": launches,
                "avg_launch_ms": kern_ms / max(launches, 1),
                "algorithmic_flops_per_launch": kern_flops / max(launches, 1),
                # with F > 1 the launches of F steps share the GPU: a launch's duration then
                # includes the time its workgroups wait for CUs held by the other steps' kernels,
                # and `frac` is that launch's share of the machine, not the kernel's efficiency;
                # the same kernel with the GPU to itself (one step at a time, measured in this run):
                "steps_in_flight": F,
                "frac_alone": one["trailing_update_frac"] if one else None,
                "achieved_alone": one["trailing_update_TFLOPs"] if one else None,
            },
        }
        if not args.no_cpu and world == 1:   # reported at N = 1 only (rank 0 would hold the others up)
            base, ref_vals = cpu_baseline(args.cpu_stars)
            line["cpu_baseline"] = base
            n = min(len(ref_vals), S)
            ok_ref = np.isfinite(ref_vals[:n])
            if ok_ref.any():
                line["max_rel_err_vs_oracle"] = float(np.max(np.abs(lnl[:n][ok_ref] / ref_vals[:n][ok_ref] - 1)))
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
