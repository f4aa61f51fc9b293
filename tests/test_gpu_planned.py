"""
GPU parity of the PLANNED likelihood step (sp_plan_data + sp_lnlike_ensemble_planned, round 5): the per-sample
call on a data set whose data-only quantities -- phases, the kernel table's weights in the covariance's sum, sums of
the flux and of the variances -- were taken once.  What the reference fixes when the log-probability is built
(calibrate/log_prob.py:7-55); the values are sp.log_likelihood's (sp.py:1129-1188 with sp.py:705-727).

  * every star of bench.py's workloads (cfg3: 64 stars, K 1000; cfg5's share: 32 stars, K 3000, Matern-3/2) within
    1e-8 of the value the executed reference gave for it (tests/golden/lnlike_full.npz);
  * the plan's weights against their definition evaluated in NumPy; the plan is the same bits every time;
  * planned against the direct normalisation (sp_set_defer_norm(0): row sums, normalised matrix assembled and
    factored as such) and against the unplanned deferred form over the input variants of the path (1e-10);
  * awkward sizes, calibrate's covpts = K - 1, tiles formed at first touch or all assembled (same bits), a plan
    shared by several handles with steps in flight, a stale plan, bad arguments.
"""
import ctypes

import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu
TOL = 1e-8


def make_engine(L=15, defer=1):
    from starry_process_amd.engine import Engine

    e = Engine(L, 2, 0)
    mom = golden("moments_L%d" % L)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    if not defer:
        e.set_defer_norm(0)
    return e


@pytest.fixture(scope="module")
def engines():
    cache = {}

    def get(L):
        if L not in cache:
            cache[L] = make_engine(L)
        return cache[L]

    return get


def star_batch(K, idx, tspan=4.0, M=1):
    sts = [synthetic_star(int(s), K, tspan) for s in idx]
    t = np.array([st["t"] for st in sts])
    if M == 1:
        flux = np.array([st["flux"] for st in sts])[:, None, :]
    else:
        flux = np.array([[np.roll(st["flux"], 7 * m) * (1.0 + 0.01 * m) for m in range(M)] for st in sts])
    return sts, t, flux


def run(e, t, flux, stars, planned, u=(0.0, 0.0), covpts=300, temporal=None, diag=None, plan=None):
    """(values, status[, plan]) of the planned or the unplanned call on the same inputs"""
    t_d, f_d, s_d = e.f64(t), e.f64(flux), e.stars_to_device(stars)
    d_d = None if diag is None else e.f64(diag)
    tab, mv = e.kernel_table(e.f64(e.rTA1L(u)), covpts)
    if planned:
        if plan is None:
            plan = e.plan_data(t_d, f_d, s_d, diag=d_d, covpts=covpts, temporal=temporal)
            out, status = e.lnlike_ensemble_planned(plan, t_d, f_d, s_d, tab, mv, diag=d_d)
        else:
            # (a plan of an earlier call: its data are the arrays of plan time -- the call takes them from the plan;
            #  fresh copies of the same numbers are refused, sp_lnlike_ensemble_planned)
            out, status = e.lnlike_ensemble_planned(plan, None, None, s_d, tab, mv)
    else:
        out, status = e.lnlike_ensemble(t_d, f_d, s_d, diag=d_d, covpts=covpts, tab=tab, meanvar=mv,
                                        temporal=temporal, normalized=True)
    return out.cpu().numpy(), status.cpu().numpy(), plan


def planned_lnl(e, K, idx, tspan=4.0, u=(0.0, 0.0), tau=None, M=1):
    from starry_process_amd.engine import make_stars

    sts, t, flux = star_batch(K, idx, tspan, M)
    stars = make_stars(len(sts), period=[st["p"] for st in sts], tau=tau or 0.0, data_var=1e-6)
    v, st, _ = run(e, t, flux, stars, True, u=u, temporal="matern32" if tau else None)
    return v, st


def test_planned_cfg3_every_star(engines):
    """bench.py's workload through the planned call: every star within 1e-8 of the reference's own value; a star's
    value does not depend on the batch it is planned and evaluated in (bit for bit)."""
    ref = golden("lnlike_full")["cfg3_L15_K1000"]
    e = engines(15)
    v, st = planned_lnl(e, 1000, range(64))
    assert not st.any() and np.all(np.isfinite(v))
    assert np.max(np.abs(v / ref - 1)) < TOL
    v8, _ = planned_lnl(e, 1000, range(8))
    assert np.array_equal(v8, v[:8])
    v1, _ = planned_lnl(e, 1000, [63])
    assert v1[0] == v[63]
    v5, _ = planned_lnl(e, 1000, range(20, 25))
    assert np.array_equal(v5, v[20:25])


def test_planned_cfg5_share_every_star(engines):
    """cfg5's share of one GPU: 32 stars, ydeg 20, K 3000, Matern-3/2 (tau 3), u = [0.4, 0.2]."""
    ref = golden("lnlike_full")["cfg5_L20_K3000"]
    e = engines(20)
    v, st = planned_lnl(e, 3000, range(32), tspan=30.0, u=(0.4, 0.2), tau=3.0)
    assert not st.any() and np.all(np.isfinite(v))
    assert np.max(np.abs(v / ref - 1)) < TOL
    v3, _ = planned_lnl(e, 3000, range(3), tspan=30.0, u=(0.4, 0.2), tau=3.0)
    assert np.array_equal(v3, v[:3])


def numpy_wbar(t, p, covpts, nobs=None, tau=None):
    """sum_ij [s_ij + k = n] b_k(x0_ij) T_ij by the definition (flux.py:256-276, 322-330; temporal.py:8-11)"""
    from oracle import sp_oracle as so

    n = len(t) if nobs is None else nobs
    t = t[:n]
    theta = so.phase(t, p)
    dx, xp = so.lag_grid(covpts)
    x = np.abs(theta[:, None] - theta[None, :]).reshape(-1)
    inds = np.floor(x / dx).astype("int64")
    x0 = (x - xp[inds + 1]) / dx
    x2, x3 = x0 * x0, x0 ** 3
    b = np.array([-x0 / 3 + x2 / 2 - x3 / 6, 1 - x0 / 2 - x2 + x3 / 2, x0 + x2 / 2 - x3 / 2, -x0 / 6 + x3 / 6])
    if tau is not None:
        b = b * so.Matern32Kernel(t, t, tau).reshape(-1)[None, :]
    w = np.zeros(covpts + 4)
    for k in range(4):
        np.add.at(w, inds + k, b[k])
    return w


def test_plan_weights_against_their_definition(engines):
    """wbar per star against NumPy (ragged, with and without a temporal kernel); its entries sum to nobs^2 without
    one (the cubic's weights are a partition of unity); planning twice gives the same bits."""
    from starry_process_amd.engine import make_stars

    e = engines(15)
    K = 500
    sts, t, flux = star_batch(K, range(40, 44))
    nobs = [K, K - 1, 130, 64]
    for tau in (None, 1.7):
        stars = make_stars(4, period=[st["p"] for st in sts], tau=tau or 0.0, data_var=1e-6, nobs=nobs)
        t_d, f_d, s_d = e.f64(t), e.f64(flux), e.stars_to_device(stars)
        temporal = "matern32" if tau else None
        w1 = e.plan_data(t_d, f_d, s_d, covpts=300, temporal=temporal).wbar()
        w2 = e.plan_data(t_d, f_d, s_d, covpts=300, temporal=temporal).wbar()
        assert np.array_equal(w1, w2)
        for s in range(4):
            ref = numpy_wbar(t[s], sts[s]["p"], 300, nobs[s], tau)
            assert np.max(np.abs(w1[s] - ref)) < 1e-12 * np.max(np.abs(ref)), (tau, s)
            if tau is None:
                assert abs(w1[s].sum() / nobs[s] ** 2 - 1) < 1e-13


def test_planned_matches_direct_and_deferred():
    """The planned call against the direct normalisation (row sums, the normalised matrix assembled and factored as
    such) and against the unplanned deferred form, over the input variants of the path."""
    from starry_process_amd.engine import make_stars

    e_dir, e_def = make_engine(15, defer=0), make_engine(15)
    rng = np.random.RandomState(3)
    K, S = 333, 7
    sts = [synthetic_star(20 + s, K, 6.0) for s in range(S)]
    t = np.array([st["t"] for st in sts])
    cases = {
        "plain": dict(),
        "baseline": dict(baseline_var=3e-5, baseline_mean=2e-3),
        "temporal": dict(tau=2.5, temporal="matern32"),
        "temporal_out_of_order": dict(tau=2.5, temporal="matern32", swap=True),
        "temporal_short_tau": dict(tau=1.0e-3, temporal="matern32"),
        "temporal_expsquared": dict(tau=2.0, temporal="expsquared"),
        "temporal_ragged": dict(tau=2.5, temporal="matern32", nobs=[K, K - 1, 200, 65, 64, 63, 2]),
        "ragged": dict(nobs=[K, K - 1, 200, 65, 64, 63, 2]),
        "multi": dict(M=4),
        "multi_baseline": dict(M=3, baseline_var=1e-5, baseline_mean=-1e-3),
        "vector_variance": dict(diag=1e-6 * (1.0 + rng.rand(S, K))),
        "vector_variance_ragged_baseline": dict(diag=1e-6 * (1.0 + rng.rand(S, K)), baseline_var=2e-5,
                                                nobs=[K, 300, 200, 65, 64, 63, 5]),
        "not_pd": dict(data_var=[1e-6, -1.0, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6]),
    }
    for name, kw in cases.items():
        M = kw.get("M", 1)
        flux = np.array([[np.roll(st["flux"], 5 * m) for m in range(M)] for st in sts])
        tt = t.copy()
        if kw.get("swap"):
            tt[:, [10, 11]] = tt[:, [11, 10]]
            tt[2, [200, 100]] = tt[2, [100, 200]]
        stars = make_stars(S, period=[st["p"] for st in sts], tau=kw.get("tau", 0.0),
                           data_var=kw.get("data_var", 1e-6), baseline_var=kw.get("baseline_var", 0.0),
                           baseline_mean=kw.get("baseline_mean", 0.0), nobs=kw.get("nobs", 0))
        args = dict(u=(0.3, 0.1), temporal=kw.get("temporal"), diag=kw.get("diag"))
        a, sa, _ = run(e_dir, tt, flux, stars, False, **args)
        b, sb, _ = run(e_def, tt, flux, stars, False, **args)
        c, sc, _ = run(e_def, tt, flux, stars, True, **args)
        assert np.array_equal(sa, sc) and np.array_equal(sb, sc), name
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(c)), name
        if name == "not_pd":
            assert not fin[1] and (sc[1] & 1) and fin[[0, 2, 3, 4, 5, 6]].all()
        else:
            assert fin.all(), name
        scale = np.maximum(np.abs(a[fin]), np.abs(a[fin]).max())
        assert np.max(np.abs(a[fin] - c[fin]) / scale) < 1e-10, (name, a, c)
        assert np.max(np.abs(b[fin] - c[fin]) / scale) < 1e-10, (name, b, c)
    # z > zmax: -inf with the ZMAX bit
    g = golden("lnlike")
    st = synthetic_star(0, 100)
    e_def.set_moments(g["zmax_guard_mean_ylm"], g["zmax_guard_cov_ylm"])
    out, status, _ = run(e_def, st["t"][None, :], st["flux"][None, None, :], make_stars(1, period=1.0, data_var=1e-6), True)
    assert out[0] == -np.inf and (status[0] & 2)


@pytest.mark.parametrize("K,M", [(40, 1), (64, 1), (65, 1), (127, 1), (129, 1), (200, 1), (257, 3), (513, 1),
                                 (960, 70), (1000, 1), (1023, 1), (1024, 1), (1100, 5), (1345, 1), (2100, 1)])
def test_planned_awkward_sizes(engines, K, M):
    """Partial last blocks, residual rows in the last pivot block's rows or in blocks of their own, a single pivot
    block, one or several super-panels, a first trailing update on 128 x 64 tiles: planned equals unplanned."""
    from starry_process_amd.engine import make_stars

    e = engines(15)
    S = 5 if K > 600 else 9
    sts, t, flux = star_batch(K, range(3, 3 + S), M=M)
    stars = make_stars(S, period=[st["p"] for st in sts], data_var=1e-6)
    a, sa, _ = run(e, t, flux, stars, False)
    b, sb, _ = run(e, t, flux, stars, True)
    assert not sa.any() and not sb.any() and np.all(np.isfinite(b))
    scale = np.maximum(np.abs(a), np.abs(a).max())
    assert np.max(np.abs(a - b) / scale) < 1e-10


def test_planned_with_calibrate_style_covpts(engines):
    """covpts = K - 1 (calibrate/log_prob.py:37-38): a table of 700 segments."""
    from starry_process_amd.engine import make_stars

    e = engines(15)
    K = 700
    sts, t, flux = star_batch(K, range(6))
    stars = make_stars(6, period=[st["p"] for st in sts], data_var=1e-6)
    a, sa, _ = run(e, t, flux, stars, False, covpts=K - 1)
    b, sb, _ = run(e, t, flux, stars, True, covpts=K - 1)
    e.kernel_table(e.f64(e.rTA1L((0.0, 0.0))), 300)     # (the module's engine back on the usual lag grid)
    assert not sa.any() and not sb.any()
    assert np.max(np.abs(a / b - 1)) < 1e-10


@pytest.mark.parametrize("K,kw", [(1000, {}), (640, dict(nobs=[640, 639, 500, 130, 65, 3])), (1345, {})])
def test_planned_tiles_at_first_touch_or_all_assembled(K, kw):
    """sp_set_lazy_cov under the planned call: the tiles below the diagonal formed by the kernel that touches them
    first, or written by the assembly -- the same bits."""
    from starry_process_amd.engine import make_stars

    res = []
    for lazy in (1, 0):
        e = make_engine(15)
        e.set_lazy_cov(lazy)
        sts, t, flux = star_batch(K, range(30, 36))
        stars = make_stars(6, period=[st["p"] for st in sts], data_var=1e-6, nobs=kw.get("nobs", 0))
        v, st, _ = run(e, t, flux, stars, True)
        assert not st.any() and np.all(np.isfinite(v))
        res.append(v)
    assert np.array_equal(res[0], res[1])


def test_a_stale_plan_is_loud(engines):
    """A plan fixes period, nobs and tau: a call whose stars differ returns NaN (not -inf: it must not pass for a
    rejected sample) and SP_STAR_STALE_PLAN for those stars only.  Table, baseline and scalar variance may change."""
    from starry_process_amd.engine import make_stars

    e = engines(15)
    K = 300
    sts, t, flux = star_batch(K, range(4))
    per = [st["p"] for st in sts]
    stars = make_stars(4, period=per, data_var=1e-6)
    good, st0, plan = run(e, t, flux, stars, True)
    assert not st0.any()
    per2 = list(per)
    per2[2] *= 1.01
    bad, st1, _ = run(e, t, flux, make_stars(4, period=per2, data_var=1e-6), True, plan=plan)
    assert np.isnan(bad[2]) and st1[2] & 8
    assert np.array_equal(np.delete(bad, 2), np.delete(good, 2)) and not np.delete(st1, 2).any()
    bad, st1, _ = run(e, t, flux, make_stars(4, period=per, data_var=1e-6, nobs=[0, 299, 0, 0]), True, plan=plan)
    assert np.isnan(bad[1]) and st1[1] & 8
    # what a plan does not fix
    other = make_stars(4, period=per, data_var=2e-6, baseline_var=1e-5, baseline_mean=1e-3)
    a, sa, _ = run(e, t, flux, other, True, plan=plan)
    b, sb, _ = run(e, t, flux, other, False)
    assert not sa.any() and np.max(np.abs(a / b - 1)) < 1e-10


def test_one_plan_many_handles_in_flight():
    """bench.py's configuration: four (handle, stream) pairs share ONE plan; steps enqueued before anything is
    synchronised give exactly the bits of the same step run alone."""
    import torch
    from starry_process_amd.engine import engine_slots, make_stars

    S, K = 64, 1000
    mom = golden("moments_L15")
    slots = engine_slots(15, 2, None, 4)
    e0 = slots[0][0]
    sts, t, flux = star_batch(K, range(S))
    t_d, f_d = e0.f64(t), e0.f64(flux)
    stars = e0.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
    rta1 = e0.f64(e0.rTA1L([0.0, 0.0]))
    names = ["default", "hilat", "spread", "default"]
    wss = [e.workspace(S, K, 1) for e, _ in slots]
    for e, _ in slots:
        e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
        e.kernel_table(rta1, 300)
    plan = e0.plan_data(t_d, f_d, stars, covpts=300)
    torch.cuda.synchronize()

    def evaluate(e, name, out, ws):
        e.set_moments_dev(e.f64(mom[name + "_mean_ylm"]), e.f64(mom[name + "_cov_ylm"]))
        tab, mv = e.kernel_table(rta1, 300)
        e.lnlike_ensemble_planned(plan, t_d, f_d, stars, tab, mv, out=out, workspace=ws)

    alone = []
    for (e, stream), name, ws in zip(slots, names, wss):
        out = e.empty(S)
        with torch.cuda.stream(stream):
            evaluate(e, name, out, ws)
        torch.cuda.synchronize()
        alone.append(out.clone())
    outs = [e.empty(S) for e, _ in slots]
    for rep in range(4):
        for (e, stream), name, out, ws in zip(slots, names, outs, wss):
            with torch.cuda.stream(stream):
                evaluate(e, name, out, ws)
    torch.cuda.synchronize()
    for a, b in zip(alone, outs):
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    assert torch.equal(alone[0], alone[3]) and len({float(a[0]) for a in alone}) == 3
    ref = golden("lnlike_full")["cfg3_L15_K1000"]
    assert np.max(np.abs(alone[0].cpu().numpy() / ref - 1)) < TOL


def test_planned_bad_arguments(engines):
    """Invalid arguments are status codes, never faults: NULL pointers, a plan of another shape of variance, a lag
    grid that is not the plan's."""
    from starry_process_amd import _lib
    from starry_process_amd.engine import make_stars

    e = engines(15)
    L = _lib.lib()
    K = 128
    sts, t, flux = star_batch(K, range(2))
    t_d, f_d = e.f64(t), e.f64(flux)
    s_d = e.stars_to_device(make_stars(2, period=1.0, data_var=1e-6))
    tab, mv = e.kernel_table(e.f64(e.rTA1L((0.0, 0.0))), 300)
    ws = e.workspace(2, K, 1)
    out = e.empty(2)
    p = ctypes.c_void_p()
    st = e._stream()
    assert L.sp_plan_data(e._h, 2, 1, 1, e._p(t_d), e._p(f_d), None, e._p(s_d), 300, 0, e._p(ws), st, ctypes.byref(p)) == -1
    assert L.sp_plan_data(e._h, 2, K, 1, None, e._p(f_d), None, e._p(s_d), 300, 0, e._p(ws), st, ctypes.byref(p)) == -1
    assert L.sp_plan_data(e._h, 2, K, 1, e._p(t_d), e._p(f_d), None, e._p(s_d), 300, 7, e._p(ws), st, ctypes.byref(p)) == -1
    plan = e.plan_data(t_d, f_d, s_d, covpts=300)

    def call(plan_ptr, diag=None, tabp=tab):
        return L.sp_lnlike_ensemble_planned(e._h, plan_ptr, e._p(t_d), e._p(f_d), diag, e._p(s_d), e._p(tabp), e._p(mv),
                                            20, 0.023, e._p(ws), e._p(out), None, st)

    assert call(plan.ptr) == 0
    assert call(None) == -1
    assert call(plan.ptr, diag=e._p(t_d)) == -1          # planned without per-cadence variances
    assert call(plan.ptr, tabp=None) == -1
    plan200 = e.plan_data(t_d, f_d, s_d, covpts=200)
    assert call(plan200.ptr) == -4                        # the handle's lag grid is covpts = 300
    L.sp_plan_destroy(None)


@pytest.mark.parametrize("seed", [1, 2])
def test_planned_random_configurations(seed):
    """Randomised sweep of the planned call against the ORACLE (one star of each case) and against the unplanned call (all
    of them): sizes around the 64-column panel boundaries, 1-4 light curves per star, both temporal kernels, limb
    darkening, scalar / per-cadence variances, baseline mean and variance, irregular cadences, ragged light curves."""
    from oracle import sp_oracle as orc
    from starry_process_amd.engine import make_stars

    rng = np.random.RandomState(100 + seed)
    e = make_engine(15)
    mom = golden("moments_L15")
    worst_o = worst_u = 0.0
    for case in range(10):
        K = int(rng.choice([2, 3, 17, 63, 64, 65, 100, 127, 128, 129, 200, 257, 320, 400, 640]))
        M = int(rng.choice([1, 1, 2, 4]))
        S = 3
        tau = None if rng.rand() < 0.5 else float(rng.uniform(0.5, 5.0))
        tk = str(rng.choice(["matern32", "expsquared"]))
        u = (0.0, 0.0) if rng.rand() < 0.5 else tuple(rng.uniform(0, 0.4, 2))
        per = rng.uniform(0.3, 3.0, S)
        t = np.array([np.sort(rng.uniform(0, 6, K)) if rng.rand() < 0.5 else np.linspace(0, 4, K) for _ in range(S)])
        flux = np.array([[1e-2 * np.sin(2 * np.pi * t[s] / per[s]) * rng.rand() + 1e-3 * rng.randn(K) for _ in range(M)]
                         for s in range(S)])
        vec = rng.rand() < 0.5
        diag = 1e-6 * (1 + rng.rand(S, K)) if vec else None
        bvar = float(rng.choice([0.0, 1e-6, 1e-3]))
        bmean = float(rng.choice([0.0, 1e-3]))
        nobs = [0, 0, 0] if (rng.rand() < 0.6 or K < 4) else [K, int(rng.randint(2, K)), int(rng.randint(2, K))]
        stars = make_stars(S, period=per, tau=tau or 0.0, data_var=1e-6, baseline_var=bvar, baseline_mean=bmean, nobs=nobs)
        temporal = tk if tau else None
        a, sa, _ = run(e, t, flux, stars, True, u=u, temporal=temporal, diag=diag)
        b, sb, _ = run(e, t, flux, stars, False, u=u, temporal=temporal, diag=diag)
        assert np.array_equal(sa, sb) and np.array_equal(np.isfinite(a), np.isfinite(b)), (seed, case)
        fin = np.isfinite(a)
        scale = np.maximum(np.abs(b[fin]), 1.0)
        worst_u = max(worst_u, float(np.max(np.abs(a[fin] - b[fin]) / scale)) if fin.any() else 0.0)
        okw = dict(tau=tau, temporal_kernel=orc.Matern32Kernel if tk == "matern32" else orc.ExpSquaredKernel) if tau else {}
        o = orc.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15, **okw)
        ref = o.log_likelihood(t[0], flux[0, 0] if M == 1 else flux[0], diag[0] if vec else 1e-6, p=float(per[0]), u=u,
                               baseline_mean=bmean, baseline_var=bvar)
        assert np.isfinite(ref) == np.isfinite(a[0]), (seed, case, ref, a[0])
        if np.isfinite(ref):
            worst_o = max(worst_o, abs(a[0] - ref) / max(1.0, abs(ref)))
    assert worst_u < 1e-9 and worst_o < 1e-8, (worst_u, worst_o)
