#!/usr/bin/env python
"""
bench.py -- log_likelihood evaluations per second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (launch_ranks: N fresh child processes of this script, one per GPU, rendezvous on
127.0.0.1; the parent never touches a GPU and relays rank 0's line and the first non-zero exit
code).  Under a launcher WORLD_SIZE must equal --gpus, anything else is an error (exit code 2).

Workload (BASELINE.json configs[2], the one-GPU share of configs[3]):
  ydeg = 15, K = 1000 cadences, 64 independent stars per GPU, marginalised over
  inclination, normalised, covpts = 300, fp64.  Weak scaling: rank r owns stars
  64 r .. 64 r + 63, so 8 GPUs evaluate the 512-star calibrate ensemble.

One "step" = one hyperparameter sample of an MCMC / nested-sampling loop:
  polar-frame moments of (mu_y, Sigma_y)   (sp_set_ylm_moments_dev)
  -> inclination-marginal kernel table      (sp_kernel_table)
  -> covariance assembly + Cholesky + solve + reduction for every star
                                            (sp_lnlike_ensemble_planned: the data set -- t, flux, variances,
                                             periods -- is PLANNED once, untimed, by sp_plan_data, as the
                                             reference fixes it when calibrate.get_log_prob is built;
                                             `unplanned` reports sp_lnlike_ensemble, which plans nothing)
  -> N > 1: RCCL all-gather of the per-star log-likelihoods (torch.distributed).
Consecutive steps are independent samples (the walkers / live points a sampler
evaluates per iteration), so --in-flight F of them (default 4) are kept in flight:
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
# one hardware queue per stream in flight (engine._want_hw_queues would do the same when the slots are asked
# for; here before anything can have touched the GPU, and visible in the file the driver runs)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

YDEG, UDEG, K, STARS_PER_GPU, COVPTS = 15, 2, 1000, int(os.environ.get("SP_BENCH_STARS", "64")), 300
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix = vector peak (AMD CDNA4 datasheet)
HBM_PEAK_TBS = 8.0       # MI355X_MICROARCH.md (spec; 6.3 measured for a streaming copy)


def step_work(S, Kc, M=1, N=0):
    """Algorithmic work of one step of S stars: (flops, HBM bytes).
    N > 0: conditional branch (adds the design-matrix products 2 K N^2 + 2 K^2 N)."""
    flops = S * (Kc ** 3 / 3.0 + 2.0 * M * Kc ** 2 + 20.0 * Kc ** 2
                 + (2.0 * Kc * N * N + 2.0 * Kc * Kc * N if N else 0.0))
    return flops, S * 24.0 * Kc ** 2


def cpu_baseline(nstars, timeout=240.0, engine="c"):
    """The CPU restatement of the reference's path timed on the host cores: the stars are farmed
    out to single-threaded worker processes (oracle/cpu_worker.py, plain child processes that
    never touch the GPU), the way an ensemble would run on a CPU node; value = stars / slowest
    worker's compute time, cores = workers.
      engine "c":     oracle/cpu_pipeline.c -- the per-star pipeline (assembly, normalisation,
                      LAPACK potrf / trtrs, reduction) in C, kernel table from the NumPy oracle
                      (a process per core rather than OpenMP threads because the
                      SciPy-bundled OpenBLAS serialises concurrent callers of one process);
      engine "numpy": oracle.OracleProcess.log_likelihood, NumPy / SciPy exactly as the reference
                      composes them (math.py:75-100)."""
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
                               str(a), str(b), str(K), str(YDEG), str(UDEG), engine],
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
    what = ("oracle/cpu_pipeline.c (assembly, normalisation, LAPACK potrf/trtrs, reduction in C; kernel table "
            "from the NumPy oracle once per worker)") if engine == "c" else \
        "oracle.OracleProcess.log_likelihood with SciPy/LAPACK potrf/trtrs"
    return dict(value=done / slowest, unit="evals/s", cores=len(spans), kind="port",
                sample="%d stars (ydeg=15, K=1000, marginal, normalized) over %d single-threaded worker "
                       "processes, %s" % (done, len(spans), what)), vals


# ---------------------------------------------------------------------------------------------
# The timed loop, independent of what a step is: the GPU benchmark below and the CPU dry run of
# the multi-rank control flow (tests/test_bench_dist_gloo.py: gloo backend, a stub evaluator)
# both go through it.
# ---------------------------------------------------------------------------------------------
class Harness(object):
    """What the loop needs from the platform.  `slots[i].run()` enqueues one whole step on slot
    i (its own stream / handle / outputs); `sync()` waits for every slot; `barrier()` and
    `max_over_ranks(x)` are the two inter-rank operations of the timing."""

    def __init__(self, slots, sync, barrier=None, max_over_ranks=None):
        self.slots = slots
        self.sync = sync
        self.barrier = barrier or (lambda: None)
        self.max_over_ranks = max_over_ranks or (lambda x: x)


def timed_steps(h, steps, warmup, prewarm_ms=0.0, before_timed=None):
    """W untimed warm-up steps, the stated untimed pre-warm, then EXACTLY `steps` steps between
    barrier + sync on both sides; returns (elapsed seconds: max over ranks, host enqueue seconds,
    pre-warm steps)."""
    F = len(h.slots)
    nwarm = max(warmup, F if warmup else 0)
    for i in range(nwarm):
        h.slots[i % F].run()
    # Stated, untimed pre-warm: a fresh process reaches the timed region
    # 5 ms after its first launch otherwise, with the memory / fabric clocks still at their idle
    # state, and the bandwidth-bound in-flight mode then reads like the latency-bound one.
    # Whole steps of the same workload, every slot at least 3 times, at least prewarm_ms.
    prewarm_steps = 0
    if prewarm_ms > 0:
        h.sync()
        tw = time.perf_counter()
        spent = 0.0
        while prewarm_steps < 3 * F or spent < prewarm_ms:
            for _ in range(F):
                h.slots[(nwarm + prewarm_steps) % F].run()
                prewarm_steps += 1
            h.sync()
            # (the same verdict on every rank: each step issues a collective, so the ranks must run
            #  the same number of them -- a per-rank clock would deadlock the all-gathers)
            spent = h.max_over_ranks(1e3 * (time.perf_counter() - tw))
    if before_timed:
        before_timed()
    h.barrier()
    h.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        h.slots[i % F].run()
    host_enqueue = time.perf_counter() - t0
    h.sync()
    h.local_elapsed = time.perf_counter() - t0      # this rank's own clock to its own last step (N > 1: reported per rank)
    h.barrier()
    elapsed = h.max_over_ranks(time.perf_counter() - t0)
    return elapsed, host_enqueue, prewarm_steps


class Slot(object):
    """One step in flight on the GPU: handle, stream, workspace, outputs (inputs shared)."""

    def __init__(self, torch, engine, stream, S, Kc, world, use_dist, dist, covpts, conditional=False):
        self.torch, self.e, self.stream, self.dist, self.use_dist = torch, engine, stream, dist, use_dist
        self.ws = engine.workspace(S, Kc, 1)
        self.out = engine.empty(S)
        self.status = torch.zeros(S, dtype=torch.int32, device=engine.device)
        self.gathered = engine.empty(world * S) if use_dist else None
        self.covpts, self.conditional = covpts, conditional
        self.inputs = None
        self.plan = None          # a DataPlan: the step goes through sp_lnlike_ensemble_planned

    def bind(self, **inputs):
        self.inputs = inputs

    def step(self):
        a, e = self.inputs, self.e
        e.set_moments_dev(a["mu_d"], a["Sig_d"])
        tab = mv = None
        if not self.conditional:
            tab, mv = e.kernel_table(a["rta1_d"], self.covpts)
        if self.plan is not None:
            e.lnlike_ensemble_planned(self.plan, a["t_d"], a["f_d"], a["stars_d"], tab, mv, out=self.out,
                                      status=self.status, workspace=self.ws)
        else:
            e.lnlike_ensemble(a["t_d"], a["f_d"], a["stars_d"], conditional=self.conditional, covpts=self.covpts,
                              tab=tab, meanvar=mv, rta1=a["rta1_d"], temporal=a.get("temporal"), normalized=True,
                              out=self.out, status=self.status, workspace=self.ws)
        if self.use_dist:
            # issued from this step's stream: torch orders the collectives of one communicator in
            # program order (the same on every rank), and only this step waits for its own
            self.dist.all_gather_into_tensor(self.gathered, self.out)

    def run(self):
        with self.torch.cuda.stream(self.stream):
            self.step()


def prof_summary(engines, kinds):
    """Per kind: launches, summed ms, summed algorithmic flops over the given engines."""
    out = {}
    for k in kinds:
        n = ms = fl = flp = 0.0
        for e in engines:
            a, b, c, d = e.profile_kind(k, padded=True)
            n, ms, fl, flp = n + a, ms + b, fl + c, flp + d
        out[k] = dict(launches=int(n), ms=ms, flops=fl, flops_padded=flp)
    return out


PANEL_KERNEL = ("panel_kernel (csrc/sp_panel.hip; one launch per 64-column panel: left-looking block-column product + "
                "triangular solve + eager rank-64 diagonal updates on v_mfma_f64_16x16x4_f64, next diagonal block)")
KERNEL_OF_KIND = {
    "panels": PANEL_KERNEL,
    "chain": PANEL_KERNEL,
    "syrk": "mm_nt_kernel<MM2<64,64,8,6,4>> (csrc/sp_gemm.hip; symmetric rank-512 trailing update, "
            "v_mfma_f64_16x16x4_f64)",
}


def rocprof_average_us(kernel_prefix):
    """Average duration of a kernel in the committed rocprofv3 kernel-trace summary of THIS round's
    one-step-at-a-time run (profiles/r05_one_kernel_stats.csv), over all its template instantiations
    (the panel kernel is one per kind of launch), or None: printed beside the event average so that the
    two can be compared; it is a file of the repository, not of this run."""
    import csv

    for name in ("r06_one_kernel_stats.csv", "r05_one_kernel_stats.csv", "r04_one_kernel_stats.csv"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            calls = total = 0.0
            for r in csv.DictReader(open(path)):
                kn = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                if kn.startswith(kernel_prefix):
                    calls += float(r["Calls"])
                    total += float(r["TotalDurationNs"])
            if calls:
                return total / calls / 1e3, "profiles/" + name
        except Exception:
            pass
    return None, None


def roofline_entry(kind, rec, extra=None):
    ach = (rec["flops"] / (rec["ms"] * 1e-3)) / 1e12 if rec["ms"] > 0 else 0.0
    d = {"kernel": KERNEL_OF_KIND[kind], "bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS,
         "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "launches": rec["launches"],
         "avg_launch_ms": rec["ms"] / max(rec["launches"], 1),
         "algorithmic_flops_per_launch": rec["flops"] / max(rec["launches"], 1),
         # (what the launches execute: every row of the padded system, every block 64 wide -- rounds 1-5 reported this
         #  count as the algorithmic one; `achieved` / `frac` are on the algorithmic count since round 6)
         "executed_system_flops_per_launch": rec.get("flops_padded", rec["flops"]) / max(rec["launches"], 1)}
    if extra:
        d.update(extra)
    return d


def bench_shape(torch, dist, ydeg, Kc, S, tspan, tau, u, conditional, F, steps, device, planned=False):
    """evals/s and whole-step roofline fraction of another BASELINE shape on this GPU (same
    in-flight scheme, its own handles; 6 untimed steps and 150 ms of untimed pre-warm steps first).  planned: the data
    set planned once (untimed)."""
    from starry_process_amd.engine import engine_slots, make_stars
    from starry_process_amd.synthetic import synthetic_star

    mom = np.load(os.path.join(ROOT, "tests", "golden", "moments_L%d.npz" % ydeg))
    mu, Sig = mom["default_mean_ylm"], mom["default_cov_ylm"]
    sts = [synthetic_star(s, Kc, tspan) for s in range(S)]
    pairs = engine_slots(ydeg, UDEG, device, F)
    e0 = pairs[0][0]
    stars_h = make_stars(S, period=[s["p"] for s in sts], inc_deg=[s["i"] for s in sts], tau=tau or 0.0,
                         data_var=1e-6)
    inputs = dict(
        t_d=e0.f64(np.array([s["t"] for s in sts])),
        f_d=e0.f64(np.array([s["flux"] for s in sts])[:, None, :]),
        stars_d=e0.stars_to_device(stars_h),
        mu_d=e0.f64(mu), Sig_d=e0.f64(Sig), rta1_d=e0.f64(e0.rTA1L(list(u))),
        temporal="matern32" if tau else None)
    slots = []
    for ek, stream in pairs:
        ek.set_moments(mu, Sig)
        sl = Slot(torch, ek, stream, S, Kc, 1, False, dist, COVPTS, conditional)
        sl.bind(**inputs)
        slots.append(sl)
    if planned and not conditional:
        plan = e0.plan_data(inputs["t_d"], inputs["f_d"], inputs["stars_d"], covpts=COVPTS, temporal=inputs["temporal"],
                            workspace=slots[0].ws)
        for sl in slots:
            sl.plan = plan
    h = Harness(slots, torch.cuda.synchronize)
    # (the same stated, untimed pre-warm as the headline's, 150 ms: a shape's timed region must not start on idle clocks)
    elapsed, _, _ = timed_steps(h, steps, 6, 150.0, None)
    ms = 1e3 * elapsed / steps
    N = (ydeg + 1) ** 2 if conditional else 0
    fl, by = step_work(S, Kc, 1, N)
    res = {"ydeg": ydeg, "K": Kc, "stars": S, "conditional": bool(conditional),
           "temporal": "matern32" if tau else None, "u": list(u), "steps": steps, "steps_in_flight": F,
           "planned_data": bool(planned and not conditional),
           "evals_per_s": S * steps / elapsed, "ms_per_step": ms,
           "whole_step_TFLOPs": fl / (ms * 1e-3) / 1e12,
           "whole_step_frac": fl / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
           # (SURVEY 8d's algorithmic bytes, 24 K^2 per star, over the step: the bound of short light curves -- K / 72 flop
           #  per byte is below the ridge of ~10 for K < 700)
           "algorithmic_TBs": by / (ms * 1e-3) / 1e12, "hbm_frac": by / (ms * 1e-3) / 1e12 / HBM_PEAK_TBS,
           "finite": bool(torch.isfinite(slots[0].out).all().item())}
    return res


def k_sweep_stars(Kc):
    """Stars per step of the K sweep: about 0.5 GB of padded systems (a multiple of 8 where that is at least 8)."""
    Kp = (Kc + 3 + 63) // 64 * 64
    S = max(1, int(round(0.5e9 / (8.0 * Kp * Kp))))
    return S // 8 * 8 if S >= 8 else S


def bench_k_sweep(torch, dist, F, device, Ks=(64, 128, 256, 512, 1000, 2048, 4096)):
    """The reference's own benchmark protocol is a sweep over the number of cadences (joss/figures/speed.py:22-37:
    npts = logspace(0, 4, 20), both branches; tests/test_timing.py:14-145): evaluations/s and whole-step fraction of
    the fp64 peak for K = 64 ... 4096 at ydeg 15, marginal (planned data) and conditional, each with the stars that
    hold about 0.5 GB of systems, F steps in flight."""
    out = {}
    for Kc in Ks:
        S = k_sweep_stars(Kc)
        fl, _ = step_work(S, Kc, 1, 0)
        # (about 60 ms of timed steps at a guessed 30 % of the peak, at least 8)
        steps = int(min(400, max(8, 0.06 / (fl / (0.3 * FP64_PEAK_TFLOPS * 1e12)))))
        steps = (steps + F - 1) // F * F
        rec = {"stars": S}
        for name, cond in (("marginal", False), ("conditional", True)):
            try:
                r = bench_shape(torch, dist, ydeg=15, Kc=Kc, S=S, tspan=4.0, tau=None, u=(0.0, 0.0), conditional=cond, F=F,
                                steps=steps, device=device, planned=not cond)
                rec[name] = {k: r[k] for k in ("evals_per_s", "ms_per_step", "whole_step_frac", "hbm_frac", "steps", "finite",
                                               "planned_data")}
            except Exception as exc:
                rec[name] = {"error": repr(exc)}
        out["K%d" % Kc] = rec
    return out


def bench_samples(torch, ydeg, Kc, Sd, F, steps, device):
    """BASELINE cfg2 the way the reference is driven -- ONE light curve (Sd = 1; or a small ensemble) evaluated at many
    hyperparameter samples (sp.py:1052-1062 inside calibrate/sample.py:95-107) -- with the samples batched ceil(64 / Sd)
    to a library call (calibrate.SampleBatches): per step the samples' polar moments FROM THE HYPERPARAMETERS
    (sp_polar_moments_samples: the upstream is per-sample work here, so it is INSIDE the timed step), their kernel
    tables, one planned likelihood call on (sample, star) systems.  Fresh samples every step; F steps in flight."""
    from starry_process_amd.calibrate import SampleBatches
    from starry_process_amd.engine import engine_slots, make_stars
    from starry_process_amd.synthetic import synthetic_star

    sts = [synthetic_star(s, Kc) for s in range(Sd)]
    slots = engine_slots(ydeg, UDEG, device, F)
    e0 = slots[0][0]
    stars = make_stars(Sd, period=[s["p"] for s in sts], data_var=1e-6)
    sb = SampleBatches(slots, e0.f64(np.array([s["t"] for s in sts])), e0.f64(np.array([s["flux"] for s in sts])[:, None, :]),
                       stars, e0.f64(e0.rTA1L([0.0, 0.0])), COVPTS)
    g = sb.group
    rng = np.random.RandomState(7)

    def draw(n):      # around the defaults (defaults.py:7-12), inside z < zmax
        return np.column_stack([rng.uniform(15.0, 25.0, n), rng.uniform(0.3, 0.5, n), rng.uniform(0.2, 0.35, n),
                                rng.uniform(0.08, 0.12, n), rng.uniform(5.0, 12.0, n)])

    # (untimed pre-warm like the headline's: whole steps of the same workload until 300 ms have passed -- a timed region
    #  that starts 12 ms after the shape's first launch reads 52-58k in one process out of three and 96-100k in the others)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.3:
        sb(draw(3 * F * g))
        torch.cuda.synchronize()
    smp = draw(steps * g)
    t0 = time.perf_counter()
    out = sb(smp)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals = out.cpu().numpy()
    # one step at a time, and the share of the moments + tables in it (the same calls, on one slot)
    (e, stream), b = sb._slots[0], sb._buf[0]
    one = draw(g)

    def upstream_only():
        e.polar_moments_samples(one, ez=b["ez"], Ez=b["Ez"])
        e.kernel_table_samples(b["ez"], b["Ez"], sb._rta1, COVPTS, tab=b["tab"], meanvar=b["mv"])

    def whole():
        upstream_only()
        e.lnlike_ensemble_planned(sb._plan, None, None, sb._stars, b["tab"], b["mv"], workspace=b["ws"])

    res = {}
    with torch.cuda.stream(stream):
        for name, fn, reps in (("upstream_ms", upstream_only, 30), ("one_step_at_a_time_ms", whole, 30)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            res[name] = 1e3 * (time.perf_counter() - t1) / reps
    fl, _ = step_work(g * Sd, Kc, 1, 0)
    ms = 1e3 * dt / steps
    res.update({"ydeg": ydeg, "K": Kc, "stars": Sd, "samples_per_call": g, "systems_per_call": g * Sd, "steps": steps,
                "steps_in_flight": F, "evals_per_s": g * Sd * steps / dt, "samples_per_s": g * steps / dt,
                "ms_per_step": ms, "host_enqueue_ms_per_step": 1e3 * host / steps,
                "upstream_share_one_at_a_time": res["upstream_ms"] / res["one_step_at_a_time_ms"],
                "whole_step_TFLOPs": fl / (ms * 1e-3) / 1e12, "whole_step_frac": fl / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                "finite": bool(np.isfinite(vals).all()),
                "note": "hyperparameters -> polar moments (device Gauss-Jacobi + rotations, sp_polar_moments_samples) -> "
                        "kernel tables -> planned likelihood of (sample, star) systems, all inside the timed step; "
                        "upstream_ms = moments + tables of one call alone"})
    return res


def bench_grad(torch, S, Kc, device, forward_ms):
    """The ensemble gradient d sum_s lnL_s / d(r, a, b, c, n) at cfg3's shape (grad.EnsembleGradient: one device
    sweep for the whole batch -- C^-1 by the factorisation's machinery, the kernel table's adjoint on the device),
    never in `value`.  Algorithmic work of the sweep: S K^3 flop (factorisation K^3/3 + triangular inverse K^3/3 +
    L^-T L^-1 K^3/3)."""
    from starry_process_amd.grad import EnsembleGradient
    from starry_process_amd.synthetic import synthetic_star
    from starry_process_amd.upstream_device import ylm_moments_device

    sts = [synthetic_star(s, Kc) for s in range(S)]
    eg = EnsembleGradient(np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), ferr=1e-3,
                          p=np.array([s["p"] for s in sts]), device=device)
    for k in range(3):
        total, g = eg()
    n = 10
    t1 = time.perf_counter()
    for k in range(n):
        total, g = eg(r=20.0 + 0.01 * k)
    ms_call = 1e3 * (time.perf_counter() - t1) / n
    e = eg._e
    mu, Sig = ylm_moments_device(e)
    e.set_moments_dev(mu, Sig)
    tab, mv = e.kernel_table(eg._rta1, COVPTS)
    for _ in range(3):
        e.lnlike_grad_marginal(eg._t, eg._flux, eg._stars, tab, mv, workspace=eg._ws)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        e.lnlike_grad_marginal(eg._t, eg._flux, eg._stars, tab, mv, workspace=eg._ws)
    torch.cuda.synchronize()
    ms_sweep = 1e3 * (time.perf_counter() - t1) / n
    fl = S * float(Kc) ** 3
    return {"stars": S, "K": Kc, "ms_per_gradient": ms_call, "device_sweep_ms": ms_sweep,
            "gradients_per_s": 1e3 / ms_call, "star_gradients_per_s": S * 1e3 / ms_call,
            "sweep_over_forward": ms_sweep / forward_ms if forward_ms else None,
            "call_over_forward": ms_call / forward_ms if forward_ms else None,
            "sweep_TFLOPs": fl / (ms_sweep * 1e-3) / 1e12, "sweep_frac": fl / (ms_sweep * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            "lnlike_sum": total, "gradient": g,
            "note": "d sum lnL / d(r, a, b, c, n), one call for the batch; forward = one step at a time "
                    "(one_step_at_a_time.ms_per_step); sweep flops S K^3 (factor + triangular inverse + L^-T L^-1)"}


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Start the n ranks of `bench.py --gpus n` as child processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, 127.0.0.1 rendezvous).  The caller has made no GPU call and makes none; rank 0
    inherits stdout (its JSON line is the job's), every rank inherits stderr.  Returns the exit code:
    0 only if every rank returned 0 -- the first failure ends the others."""
    import subprocess

    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    def stop_all(grace=5.0):
        """terminate every rank still running; kill what ignores it after `grace` seconds"""
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + grace
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
                p.wait()

    import signal

    def on_signal(signum, frame):
        # (a `timeout`, Ctrl-C or the driver's SIGTERM on the parent must not leave N ranks holding their GPUs and
        #  the rendezvous)
        raise KeyboardInterrupt("signal %d" % signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        live = set(range(n))
        deadline = None                   # set once a rank has failed: the others get a few seconds to leave
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    # (a rank killed by signal k reports -k: 128 + k, as a shell would)
                    rc = code if code > 0 else 128 - code
                    print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for o in live:
                        procs[o].terminate()
                    deadline = time.time() + 10.0
            if live and deadline is not None and time.time() > deadline:
                for o in live:
                    procs[o].kill()       # a survivor stuck in a collective ignores SIGTERM
                deadline = time.time() + 1e9
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt as e:
        print("bench.py: interrupted (%s); stopping the ranks" % e, file=sys.stderr)
        rc = rc or 130
    finally:
        stop_all()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


def _stub_lnlike(s, step_id):
    return -0.5 * (s + 1) ** 1.5 + np.cos(s) + 1e-3 * step_id


class StubSlot(object):
    """SP_BENCH_BACKEND=gloo: a slot whose "device step" is a deterministic function of (star, step) on
    the CPU -- the per-star values of this rank's shard, then the all-gather Slot.step issues.  Lets the
    whole multi-rank flow of this script (launch, rendezvous, shards, collectives, timing bracket, the
    JSON line) run without a GPU (tests/test_bench_dist_gloo.py).  Never a measurement."""

    def __init__(self, dist, torch, rank, world, S, slot_id):
        self.dist, self.torch, self.rank, self.world, self.S = dist, torch, rank, world, S
        self.slot_id = slot_id
        self.calls = 0
        self.out = torch.zeros(S, dtype=torch.float64)
        self.gathered = torch.zeros(world * S, dtype=torch.float64)
        self.history = []

    def run(self):
        first = self.rank * self.S
        # (the step id is a function of the call count only: the same on every rank)
        step_id = self.calls * 16 + self.slot_id
        self.out[:] = self.torch.tensor([_stub_lnlike(s, step_id) for s in range(first, first + self.S)])
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.gathered, self.out)
        else:
            self.gathered[:] = self.out
        self.history.append((step_id, self.gathered.clone()))
        self.calls += 1


def per_rank_ms(torch, dist, world, local_elapsed, steps, device):
    """{"min", "max", "all"}: every rank's own ms_per_step over the timed region (its clock from the common
    barrier to the completion of ITS last step; `ms_per_step` of the line is the job's: to the barrier behind
    the slowest rank)."""
    mine = 1e3 * local_elapsed / steps
    if world == 1:
        return {"min": mine, "max": mine, "all": [mine]}
    tt = torch.tensor([mine], dtype=torch.float64, device=device)
    every = [torch.zeros_like(tt) for _ in range(world)]
    dist.all_gather(every, tt)
    vals = [float(x.item()) for x in every]
    return {"min": min(vals), "max": max(vals), "all": vals}


def main_stub(args, rank, world):
    """The control flow of main() on gloo with StubSlot evaluators (no GPU, no library)."""
    import torch
    import torch.distributed as dist

    if os.environ.get("SP_BENCH_STUB_FAIL_RANK") == str(rank):   # (test hook: a rank that dies at start-up)
        return 7
    if os.environ.get("SP_BENCH_STUB_SLEEP"):                    # (test hook: ranks that outlive their parent's patience)
        time.sleep(float(os.environ["SP_BENCH_STUB_SLEEP"]))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    S, F = STARS_PER_GPU, max(1, args.in_flight)
    slots = [StubSlot(dist, torch, rank, world, S, i) for i in range(F)]

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    h = Harness(slots, lambda: None, dist.barrier if world > 1 else None, max_over_ranks)
    elapsed, _, prewarm_steps = timed_steps(h, args.steps, args.warmup, min(args.prewarm_ms, 5.0), None)
    sid, g = slots[(args.steps - 1) % F].history[-1]
    ref = np.array([_stub_lnlike(s, sid) for s in range(world * S)])
    ok = bool(np.array_equal(g.numpy(), ref))
    per_rank = per_rank_ms(torch, dist, world, h.local_elapsed, args.steps, None)
    if rank == 0:
        print(json.dumps({
            "metric": "log_likelihood evals/sec (ydeg=15, K=1000)", "value": world * S * args.steps / elapsed,
            "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "STUB evaluator on gloo (SP_BENCH_BACKEND=gloo): control flow only, not a "
                                   "measurement", "stars_per_gpu": S, "parallelism": "stars sharded %d-way" % world,
                       "steps_in_flight": F},
            "backend": "gloo-stub", "parity_ok": ok, "gathered_values": int(g.numel()),
            "per_rank_ms_per_step": per_rank, "cpu_baseline": "N = 1 only" if world > 1 else None,
            "prewarm": {"steps": prewarm_steps, "timed": False}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--in-flight", type=int, default=4,
                    help="independent steps kept in flight on separate streams / handles")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed pre-warm after the --warmup steps: whole steps keep running until this "
                         "much wall time has passed and every slot has run 3 times (clocks and fabric "
                         "at their loaded state); the timed region is exactly --steps steps either way")
    ap.add_argument("--cpu-stars", type=int, default=1024,
                    help="stars of the CPU-baseline sample (about 25 core-seconds of the C pipeline)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--unplanned", action="store_true",
                    help="the headline through sp_lnlike_ensemble (nothing planned: the pre-pass over the covariance's "
                         "entries in every step) instead of sp_plan_data + sp_lnlike_ensemble_planned")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the cfg5-shape and conditional-branch measurements (after the headline)")
    args = ap.parse_args()

    if args.gpus < 1:
        print("bench.py: --gpus must be at least 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: be the launcher (this process makes no GPU call, before or after)
        return launch_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (rank %d): the line would not describe the job that ran"
              % (args.gpus, world, rank), file=sys.stderr)
        return 2
    if os.environ.get("SP_BENCH_BACKEND", "nccl") == "gloo":
        return main_stub(args, rank, world)

    import torch
    import torch.distributed as dist

    if torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d wants GPU %d, %d visible" % (rank, local_rank, torch.cuda.device_count()),
              file=sys.stderr)
        return 3
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
    inputs = dict(
        t_d=e.f64(np.array([s["t"] for s in sts])),
        f_d=e.f64(np.array([s["flux"] for s in sts])[:, None, :]),
        stars_d=e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6)),
        mu_d=e.f64(mu), Sig_d=e.f64(Sig),
        rta1_d=e.f64(e.rTA1L([0.0, 0.0])))          # one flux operator: u = [0, 0]

    # one slot per step in flight: library handle, stream, workspace, outputs (the inputs
    # above are read-only and shared)
    F = max(1, args.in_flight)
    slots = []
    for ek, stream in engine_slots(YDEG, UDEG, local_rank, F):
        ek.set_moments(mu, Sig)  # first call allocates / uploads the lag grid
        sl = Slot(torch, ek, stream, S, K, world, use_dist, dist, COVPTS)
        sl.bind(**inputs)
        slots.append(sl)
    torch.cuda.synchronize()
    # The data plan (sp_plan_data): once per data set, UNTIMED -- phases, the kernel table's weights in the covariance's
    # sum, sums of flux and variances.  One plan, read-only, shared by every slot.  Its cost is reported (second call:
    # the first one loads the kernels' code).
    plan, plan_ms = None, None
    if not args.unplanned:
        plan = e.plan_data(inputs["t_d"], inputs["f_d"], inputs["stars_d"], covpts=COVPTS, workspace=slots[0].ws)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        plan = e.plan_data(inputs["t_d"], inputs["f_d"], inputs["stars_d"], covpts=COVPTS, workspace=slots[0].ws)
        torch.cuda.synchronize()
        plan_ms = 1e3 * (time.perf_counter() - tp)
        for sl in slots:
            sl.plan = plan
    if use_dist:
        # communicator set-up (lazy in RCCL) must not land in the timed region even with --warmup 0
        dist.all_gather_into_tensor(slots[0].gathered, slots[0].out)

    def max_over_ranks(x):
        if not use_dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=e.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    harness = Harness(slots, torch.cuda.synchronize, dist.barrier if use_dist else None, max_over_ranks)
    # kinds bracketed with HIP events INSIDE the timed region (on the launch streams): the symmetric
    # trailing updates (one pair each) and the panel kernels of each super-panel (one pair per
    # super-panel); the per-launch figures are taken outside it
    timed_kinds = ("syrk", "panels")
    nev = (args.steps // F + 2) * 8

    def arm():
        for sl in slots:
            sl.e.profile_begin(nev, timed_kinds)

    elapsed, host_enqueue, prewarm_steps = timed_steps(harness, args.steps, args.warmup, args.prewarm_ms, arm)
    timed_prof = prof_summary([sl.e for sl in slots], timed_kinds)
    per_rank = per_rank_ms(torch, dist, world, harness.local_elapsed, args.steps, e.device) if use_dist else None
    # N > 1: what the all-gather costs a step (untimed repeat, the collective of every step between two events on
    # the step's own stream: its span there, the wait for the slowest rank's shard included)
    allgather = None
    if use_dist:
        nag = min(args.steps, 4 * F)
        evs = []
        dist.barrier()
        for i in range(nag):
            sl = slots[i % F]
            with torch.cuda.stream(sl.stream):
                sl.use_dist = False
                sl.step()
                sl.use_dist = True
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(sl.stream)
                dist.all_gather_into_tensor(sl.gathered, sl.out)
                a1.record(sl.stream)
                evs.append((a0, a1))
        torch.cuda.synchronize()
        ag = [a0.elapsed_time(a1) for a0, a1 in evs]
        allgather = {"launches": nag, "avg_ms": float(np.mean(ag)), "max_ms": float(np.max(ag)),
                     "bytes_per_rank": 8 * S, "note": "span of ncclAllGather (RCCL) on the step's stream, " 
                     "untimed repeat with %d steps in flight; includes the wait for the slowest rank" % F}
    # Untimed repeat of the same steps in flight with EVERY panel kernel under its own pair of
    # events (the pairs cost a few per cent of a step, hence not in the timed region)
    launch_prof = None
    if F > 1:
        for sl in slots:
            sl.e.profile_begin((args.steps // F + 2) * 24, ("panel_launch",))
        for i in range(args.steps):
            slots[i % F].run()
        torch.cuda.synchronize()
        launch_prof = prof_summary([sl.e for sl in slots], ("panel_launch",))["panel_launch"]
    nran = min(F, args.steps + max(args.warmup, F if args.warmup else 0) + prewarm_steps)   # slots that ran
    out, status = slots[0].out, slots[0].status
    lnl = out.cpu().numpy().copy()
    status_timed = status.cpu().numpy().copy()
    slots_agree = all(torch.equal(sl.out, out) for sl in slots[1:nran])   # same inputs, same bits
    ok = bool(np.all(np.isfinite(lnl))) and not bool(status_timed.any()) and slots_agree
    if not ok and rank == 0:
        print("parity check failed: finite %s, status bits set on %d stars, slots agree %s; stars that differ "
              "between slots: %s" % (bool(np.all(np.isfinite(lnl))), int(np.count_nonzero(status_timed)),
                                     slots_agree, [int((sl.out != out).sum().item()) for sl in slots[1:nran]]),
              file=sys.stderr)

    # the same steps one at a time on one stream (not `value`): what a strictly sequential
    # caller gets, with every launch of the factorisation bracketed (kernel-level breakdown)
    one = None
    one_prof = None
    c0 = None
    if world == 1:
        c0 = Slot(torch, e, torch.cuda.current_stream(), S, K, world, False, dist, COVPTS)
        c0.bind(**inputs)
        c0.plan = plan
        e.set_moments(mu, Sig)
        nrep = 100    # (0.1 s: a 20-step sample of this latency-bound leg scatters by 8 % from run to run)
        for _ in range(10):
            c0.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nrep):
            c0.step()
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        one = {"evals_per_s": S * nrep / dt1, "ms_per_step": 1e3 * dt1 / nrep}
        # (a separate short pass with every launch bracketed: the events cost ~10 % of a step)
        kinds1 = ("syrk", "chain")
        e.profile_begin(10 * 64, kinds1)
        for _ in range(10):
            c0.step()
        torch.cuda.synchronize()
        one_prof = prof_summary([e], kinds1)
        fl1, _ = step_work(S, K)
        one["whole_step_TFLOPs"] = fl1 / (one["ms_per_step"] * 1e-3) / 1e12
        one["whole_step_frac"] = one["whole_step_TFLOPs"] / FP64_PEAK_TFLOPS
        one["per_kind"] = {k: {"ms_per_step": v["ms"] / 10, "launches_per_step": v["launches"] / 10,
                               "TFLOPs": (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0)}
                           for k, v in one_prof.items()}

    # PCIe-inclusive rate (never `value`): the same step fed from pinned host buffers
    # (t, flux up, log-likelihoods down) -- what a caller without resident data would see
    pcie_rate = None
    if world == 1:
        t_d, f_d = inputs["t_d"], inputs["f_d"]
        t_h, f_h = t_d.cpu().pin_memory(), f_d.cpu().pin_memory()
        out_h = torch.empty(S, dtype=torch.float64).pin_memory()
        nrep = max(3, min(20, args.steps))

        def fed_step():
            t_d.copy_(t_h, non_blocking=True)
            f_d.copy_(f_h, non_blocking=True)
            c0.step()
            out_h.copy_(c0.out, non_blocking=True)

        for _ in range(2):          # first use of the pinned buffers maps them (tens of ms, once)
            fed_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nrep):
            fed_step()
        torch.cuda.synchronize()
        pcie_rate = S * nrep / (time.perf_counter() - t1)

    # the same steps in flight over a longer run (not `value`): a timed region of K = 20 steps holds five
    # rounds of F = 4, of which the first starts with the streams in lockstep and the last drains with
    # nothing left to overlap; a sampler's thousands of evaluations see the sustained rate
    sustained = None
    if world == 1 and not args.no_extras:
        nlong = 240
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nlong):
            slots[i % F].run()
        torch.cuda.synchronize()
        dtl = time.perf_counter() - t1
        sustained = {"steps": nlong, "steps_in_flight": F, "evals_per_s": S * nlong / dtl,
                     "ms_per_step": 1e3 * dtl / nlong}
        fls, _ = step_work(S, K)
        sustained["whole_step_frac"] = fls / (sustained["ms_per_step"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS

    # the same workload with nothing planned (sp_lnlike_ensemble: phases, the pre-pass over all K^2 entries for the
    # normalisation's sums, in every step) -- what a caller whose periods change from sample to sample gets; not `value`
    unplanned = None
    if world == 1 and plan is not None and not args.no_extras:
        for sl in slots + [c0]:
            sl.plan = None
        for i in range(3 * F):
            slots[i % F].run()
        torch.cuda.synchronize()
        nun = 120
        t1 = time.perf_counter()
        for i in range(nun):
            slots[i % F].run()
        torch.cuda.synchronize()
        dtu = time.perf_counter() - t1
        for _ in range(5):
            c0.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(50):
            c0.step()
        torch.cuda.synchronize()
        dt1u = time.perf_counter() - t1
        lnl_un = slots[0].out.cpu().numpy().copy()
        unplanned = {"steps": nun, "steps_in_flight": F, "evals_per_s": S * nun / dtu, "ms_per_step": 1e3 * dtu / nun,
                     "one_step_at_a_time_ms": 1e3 * dt1u / 50,
                     "max_rel_diff_planned": float(np.max(np.abs(lnl_un / lnl - 1))),
                     "note": "sp_lnlike_ensemble on the same inputs: theta, assembly pre-pass (every entry evaluated for "
                             "the normalisation's sums, then again at first touch), normalisation vectors per step"}
        for sl in slots + [c0]:
            sl.plan = plan

    # the other shapes of BASELINE.json, after the headline and never in `value`
    extras = None
    if world == 1 and not args.no_extras:
        extras = {}
        try:
            # (the unplanned form first: the first shape measured after the headline also warms this shape's buffers up)
            if plan is not None:
                extras["cfg5_shape_unplanned"] = bench_shape(torch, dist, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0,
                                                             u=(0.4, 0.2), conditional=False, F=F, steps=24,
                                                             device=local_rank)
            extras["cfg5_shape"] = bench_shape(torch, dist, ydeg=20, Kc=3000, S=32, tspan=30.0, tau=3.0,
                                               u=(0.4, 0.2), conditional=False, F=F, steps=24, device=local_rank,
                                               planned=plan is not None)
            # BASELINE configs[1]: ONE light curve (the reference's own use: sp.log_likelihood inside a sampler), the
            # latency of an evaluation and what four of them in flight give
            extras["cfg2_single_star"] = bench_shape(torch, dist, ydeg=15, Kc=1000, S=1, tspan=4.0, tau=None, u=(0.0, 0.0),
                                                     conditional=False, F=F, steps=200, device=local_rank,
                                                     planned=plan is not None)
            extras["cfg2_single_star_one_at_a_time"] = bench_shape(torch, dist, ydeg=15, Kc=1000, S=1, tspan=4.0, tau=None,
                                                                   u=(0.0, 0.0), conditional=False, F=1, steps=100,
                                                                   device=local_rank, planned=plan is not None)
            # ... and the same light curve with the hyperparameter samples batched 64 to a call, upstream included
            # (six streams: a third of such a step is the samples' moments and tables, light kernels -- 88-92k evaluations/s
            #  with four, 95-99k with six; calibrate.MAX_STREAMS_SAMPLES)
            Fs = F if F < 4 else 6
            extras["cfg2_batched_samples"] = bench_samples(torch, 15, 1000, 1, Fs, 120, local_rank)
            extras["ensemble8_batched_samples"] = bench_samples(torch, 15, 1000, 8, Fs, 120, local_rank)
            extras["cfg3_conditional"] = bench_shape(torch, dist, ydeg=15, Kc=1000, S=64, tspan=4.0, tau=None,
                                                     u=(0.0, 0.0), conditional=True, F=F, steps=24,
                                                     device=local_rank)
            extras["k_sweep"] = bench_k_sweep(torch, dist, F, local_rank)
            extras["cfg3_grad"] = bench_grad(torch, S, K, local_rank, one["ms_per_step"] if one else None)
        except Exception as exc:   # (never lose the headline over an extra)
            extras["error"] = repr(exc)

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
        ms_per_step = 1e3 * elapsed / args.steps
        # the dominant kernel of the timed region: the bracketed kind with the largest summed duration
        cand = {k: v for k, v in timed_prof.items() if v["launches"] > 0}
        dom = max(cand, key=lambda k: cand[k]["ms"]) if cand else "syrk"
        traffic = traffic_source = None
        for name in ("r06_step_traffic.json", "r05_step_traffic.json", "r04_step_traffic.json", "r03_step_traffic.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("dominant_kernel_bytes_per_launch")
                    traffic_source = "profiles/" + name + " (separate FETCH_SIZE / WRITE_SIZE passes of an earlier run, " \
                                                          "one step at a time; NOT measured by this run)"
                    break
                except Exception:
                    traffic = None
        shared = roofline_entry(dom, cand.get(dom, dict(launches=0, ms=0.0, flops=0.0)))
        alone = None
        if one_prof is not None:
            key = "chain" if dom == "panels" else dom
            if one_prof.get(key, {}).get("launches", 0) > 0:
                alone = roofline_entry(key, one_prof[key])
        fl, by = step_work(S, K)
        whole = {"flops": fl, "bytes": by, "ms": ms_per_step,
                 "achieved_TFLOPs": fl / (ms_per_step * 1e-3) / 1e12,
                 "frac": fl / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                 "algorithmic_TBs": by / (ms_per_step * 1e-3) / 1e12,
                 "hbm_frac": by / (ms_per_step * 1e-3) / 1e12 / HBM_PEAK_TBS,
                 "note": "per GPU: S (K^3/3 + 2 M K^2 + 20 K^2) flop and 24 S K^2 bytes (SURVEY 8d) over "
                         "the timed ms_per_step (steps_in_flight steps share the GPU)"}
        # `frac` is the dominant kernel's fraction of the fp64 peak with the GPU to itself: one step at
        # a time, every launch of the kernel under its own pair of HIP events on the launch stream,
        # measured in this run (reproducible from the committed rocprofv3 kernel trace of the same
        # command: flops / summed duration / 78.6e12).  With F steps in flight a launch shares the
        # GPU with the other steps' kernels and its bracket includes the time its workgroups wait
        # for CUs: that figure is `frac_shared`, a share of the machine, not an efficiency.
        main = alone if alone else shared
        rp_us, rp_src = rocprof_average_us("panel_kernel" if dom in ("panels", "chain") else "mm_nt_kernel")
        roof = dict(main)
        roof.update({
            "measured": ("one step at a time, every launch under its own pair of HIP events on its stream"
                         if alone else "in flight (no one-at-a-time leg in this run): a share of the machine"),
            "algorithmic_flops_note": "per panel launch of pivot block j, q-th of its super-panel: S (2 rows na (64 q) + rows "
                                      "na^2 + sum over the eager diagonal blocks of n_i^2 na + na^3 / 3) with rows = the "
                                      "K cadences' rows below the block + the M residual rows, na = min(64, K - 64 j): "
                                      "left-looking product, triangular solve, eager updates, the diagonal block "
                                      "(1.33e10 per 64-star K = 1000 step); executed_system_flops_per_launch is the same "
                                      "count on the padded 1 024-row system with every block 64 wide (what rounds 1-5 "
                                      "reported: 7 % more); the flops the look-ahead items move from one launch to the "
                                      "one before are counted where the sum has them",
            "rocprof_avg_launch_us": rp_us, "rocprof_source": rp_src,
            "traffic": traffic, "traffic_source": traffic_source,
            "steps_in_flight": F,
            "frac_shared": shared["frac"],
            "shared": {"achieved": shared["achieved"], "launches": shared["launches"],
                       "avg_launch_ms": shared["avg_launch_ms"],
                       "note": "the same kernel in the timed region (one pair of events per super-panel)"},
            "event_ms_by_kind": {k: v["ms"] for k, v in timed_prof.items()},
            "per_launch_shared": (None if not launch_prof or not launch_prof["launches"] else {
                "launches": launch_prof["launches"],
                "avg_launch_ms": launch_prof["ms"] / launch_prof["launches"],
                "achieved": launch_prof["flops"] / (launch_prof["ms"] * 1e-3) / 1e12,
                "frac": launch_prof["flops"] / (launch_prof["ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}),
            "secondary": {k: {"shared": roofline_entry(k, v),
                              "alone": (roofline_entry(k, one_prof[k]) if one_prof and one_prof.get(k, {}).get("launches", 0)
                                        else None)}
                          for k, v in cand.items() if k != dom},
            "whole_step": whole,
        })
        line = {
            "metric": "log_likelihood evals/sec (ydeg=15, K=1000)",
            "value": evals / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
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
                "planned_data": plan is not None,
            },
            "plan": (None if plan is None else {
                "ms": plan_ms, "timed": False,
                "note": "sp_plan_data, once per data set (t, flux, variances, periods), outside every timed region: "
                        "phases, the kernel table's weights in the covariance's sum (covpts + 4 per star), sums of "
                        "flux / variances; one plan shared by all slots"}),
            "unplanned": unplanned,
            "parity_ok": ok,
            "prewarm": {"steps": prewarm_steps, "min_ms": args.prewarm_ms, "timed": False},
            "steps_in_flight": F,
            "one_step_at_a_time": one,
            "sustained": sustained,
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / args.steps,
            "pcie_inclusive_evals_per_s": pcie_rate,
            "upstream_ms_per_sample": upstream_ms,
            "other_shapes": extras,
            "roofline": roof,
        }
        if use_dist:      # (N > 1, or SP_BENCH_FORCE_DIST=1: the same code path with one RCCL rank)
            line["per_rank_ms_per_step"] = per_rank
            if allgather:
                allgather["share_of_step"] = allgather["avg_ms"] / ms_per_step
            line["allgather"] = allgather
        if world > 1:
            line["cpu_baseline"] = "N = 1 only (rank 0 would hold the other ranks up; see the N = 1 line)"
            roof["measured"] += "; N > 1: no one-at-a-time leg (`alone` figures are the N = 1 line's), " \
                                "whole_step is per GPU over the job's ms_per_step"
        if not args.no_cpu and world == 1:   # reported at N = 1 only (rank 0 would hold the others up)
            base, ref_vals = cpu_baseline(args.cpu_stars, engine="c")
            line["cpu_baseline"] = base
            base2, _ = cpu_baseline(min(args.cpu_stars, 64), engine="numpy")
            if base2:
                line["cpu_baseline_numpy_scipy"] = base2
            n = min(len(ref_vals), S)
            ok_ref = np.isfinite(ref_vals[:n])
            if ok_ref.any():
                line["max_rel_err_vs_oracle"] = float(np.max(np.abs(lnl[:n][ok_ref] / ref_vals[:n][ok_ref] - 1)))
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
