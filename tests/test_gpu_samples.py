"""
Hyperparameter samples in batches (round 6): sp_polar_moments_samples, sp_kernel_table_samples, sp_plan_replicate and
the planned likelihood call on (sample, star) systems -- how ONE light curve (BASELINE cfg2; the reference's own call
pattern, sp.py:1052-1062 driven by calibrate/sample.py:95-107) is evaluated at the rate of a 64-star ensemble.

What is asserted:
  * the polar-frame moments of a batch equal those of the per-sample path (sp_ylm_moments_quadrature +
    sp_set_ylm_moments_dev) to rounding, over the reference's whole prior box (this pins the Gauss-Jacobi rule found
    on the device and the size integral taken there), and the EXTENDED-PRECISION arbiter of
    tests/golden/upstream_extended.npz rotated into the polar frame to 1e-12;
  * B samples in one call carry the BITS of B one-sample calls: moments, kernel tables, log-likelihoods;
  * 64 samples x 1 star (K = 1000, ydeg 15) in one call: bit-equal to the 64 one-sample calls, within 1e-8 of the
    oracle evaluated on the same moments, within 1e-9 of the per-sample planned path;
  * 8 samples x 8 stars on a replicated plan = the 8 per-sample planned calls;
  * EnsembleLogProb / StarryProcess.log_likelihood_samples agree with their one-sample-at-a-time forms;
  * bad arguments are status codes.
"""
import ctypes

import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu

TOL = 1e-8     # BASELINE.json: fp64 log-likelihood within 1e-8 relative of the reference


def random_samples(ns, seed=0, box=False):
    """(r, a, b, c, n) rows: around the defaults, or (box) over the reference's whole prior box in (a, b)."""
    rng = np.random.RandomState(seed)
    out = np.empty((ns, 5))
    out[:, 0] = rng.uniform(10.0, 30.0, ns)
    out[:, 1] = rng.uniform(0.0, 1.0, ns) if box else rng.uniform(0.2, 0.6, ns)
    out[:, 2] = rng.uniform(0.0, 1.0, ns) if box else rng.uniform(0.1, 0.5, ns)
    out[:, 3] = rng.uniform(0.05, 0.2, ns)
    out[:, 4] = rng.uniform(1.0, 20.0, ns)
    return out


def same(a, b, tol):
    """Equal to tol where finite; -inf (z > zmax, sp.py:1178-1183) must be -inf on both sides."""
    a, b = np.atleast_1d(np.asarray(a, dtype=float)), np.atleast_1d(np.asarray(b, dtype=float))
    fin = np.isfinite(b)
    return np.array_equal(np.isfinite(a), fin) and np.array_equal(a[~fin], b[~fin]) and \
        (not fin.any() or np.max(np.abs(a[fin] / b[fin] - 1)) < tol)


def per_sample_polar(e, sample):
    """(ez, Ez) by the per-sample path: quadrature of rotations -> resident moments -> polar frame."""
    from starry_process_amd.upstream_device import ylm_moments_device

    r, a, b, c, n = sample
    mu, S = ylm_moments_device(e, r=r, a=a, b=b, c=c, n=n)
    e.set_moments_dev(mu, S)
    e.synchronize()
    return e.polar_moments()


@pytest.fixture(scope="module")
def e15():
    from starry_process_amd.engine import Engine

    return Engine(15, 2, 0)


@pytest.mark.parametrize("ydeg", [5, 15, 20])
def test_polar_moments_of_a_batch_equal_the_per_sample_path(ydeg):
    from starry_process_amd.engine import Engine

    e = Engine(ydeg, 2, 0)
    sm = random_samples(12, seed=ydeg, box=True)
    # the corners and edges of the (a, b) box (latitude.py:176-197: alpha <= exp(5), beta <= exp(10))
    sm[:6, 1:3] = [(0.0, 0.0), (1.0, 1.0), (0.0, 1.0), (1.0, 0.0), (0.5, 0.74), (0.4, 0.27)]
    ez, Ez = e.polar_moments_samples(sm)
    ez, Ez = ez.cpu().numpy(), Ez.cpu().numpy()
    for k, s in enumerate(sm):
        ez1, Ez1 = per_sample_polar(e, s)
        assert np.abs(ez[k] - ez1).max() <= 2e-11 * np.abs(ez1).max(), (k, s)
        assert np.abs(Ez[k] - Ez1).max() <= 2e-11 * np.abs(Ez1).max(), (k, s)
        assert np.array_equal(Ez[k], Ez[k].T)


@pytest.mark.parametrize("name", ["default", "hilat"])     # ("spread" has a spread of radii: the per-sample path's case)
def test_polar_moments_match_the_extended_precision_arbiter(e15, name):
    from oracle import sp_oracle as orc

    g, x = golden("moments_L15"), golden("upstream_extended")
    r, dr, a, b, c, n = g[name + "_hyper"]
    assert np.isnan(dr)
    ez, Ez = e15.polar_moments_samples(np.array([[r, a, b, c, n]]))
    ez_x, Ez_x = orc.polar_moments(15, x[name + "_mean_ylm"], x[name + "_cov_ylm"])
    assert np.abs(ez[0].cpu().numpy() - ez_x.ravel()).max() < 1e-12 * np.abs(ez_x).max()
    assert np.abs(Ez[0].cpu().numpy() - Ez_x).max() < 1e-12 * np.abs(Ez_x).max()


def test_a_batch_carries_the_bits_of_one_sample_calls(e15):
    import torch

    e = e15
    sm = random_samples(64, seed=3)
    ez, Ez = e.polar_moments_samples(sm)
    rta1 = e.f64(e.rTA1L(np.array([[0.0, 0.0], [0.4, 0.2]])))
    tab, mv = e.kernel_table_samples(ez, Ez, rta1, 300)
    assert tuple(tab.shape) == (128, 5, 304) and tuple(mv.shape) == (128, 2)
    for k in (0, 1, 17, 63):
        ez1, Ez1 = e.polar_moments_samples(sm[k:k + 1])
        assert torch.equal(ez1[0], ez[k]) and torch.equal(Ez1[0], Ez[k])
        tab1, mv1 = e.kernel_table_samples(ez1, Ez1, rta1, 300)
        assert torch.equal(tab1, tab[2 * k:2 * k + 2]) and torch.equal(mv1, mv[2 * k:2 * k + 2])
    # the resident-moments table kernel on the same (ez, Ez): set the polar moments through the per-sample path
    ez1, Ez1 = per_sample_polar(e, sm[5])
    tab_r, mv_r = e.kernel_table(rta1, 300)
    assert np.abs(tab_r.cpu().numpy() - tab[10:12].cpu().numpy()).max() <= 1e-10 * np.abs(tab_r.cpu().numpy()).max()


def _oracle_process(ez, Ez, ydeg=15, covpts=300, **kw):
    """The oracle's process on given POLAR moments (the marginal branch reads nothing else of the Ylm moments:
    oracle/sp_oracle.py, OracleProcess.flux_mean_cov)."""
    from oracle import sp_oracle as orc

    N = (ydeg + 1) ** 2
    op = orc.OracleProcess(np.zeros(N), np.eye(N), ydeg=ydeg, covpts=covpts, **kw)
    op.ez, op.Ez = np.ascontiguousarray(ez).reshape(-1, 1), np.ascontiguousarray(Ez)
    return op


def test_cfg2_64_samples_of_one_light_curve_in_one_call(e15):
    """BASELINE cfg2's light curve (ydeg 15, K = 1000, star 0) at 64 hyperparameter samples."""
    import torch

    from starry_process_amd.engine import make_stars, stars_for_samples

    e = e15
    K, B = 1000, 64
    st = synthetic_star(0, K)
    t_d, f_d = e.f64(st["t"][None, :]), e.f64(st["flux"][None, None, :])
    stars = make_stars(1, period=st["p"], data_var=1e-6)
    s_d = e.stars_to_device(stars)
    rta1 = e.f64(e.rTA1L(np.array([0.0, 0.0])))
    plan = e.plan_data(t_d, f_d, s_d, covpts=300)
    rep = e.replicate_plan(plan, B)
    assert rep.S == B
    sm = random_samples(B, seed=11)
    ez, Ez = e.polar_moments_samples(sm)
    tab, mv = e.kernel_table_samples(ez, Ez, rta1, 300)
    sB = e.stars_to_device(stars_for_samples(stars, B, 1))
    out, status = e.lnlike_ensemble_planned(rep, None, None, sB, tab, mv, workspace=e.workspace(B, K, 1))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    # (a sample whose normalisation parameter z exceeds zmax is -inf with SP_STAR_ZMAX, sp.py:1178-1183: a few of 64)
    flags = status.cpu().numpy()
    assert np.array_equal(np.isfinite(got), flags == 0) and set(flags) <= {0, 2} and (flags == 0).sum() >= 48
    # bit for bit the 64 one-sample calls (the same entry points with B = 1), and the per-sample planned path's
    # values (resident moments, sp_kernel_table, the un-replicated plan) to rounding
    rep1 = e.replicate_plan(plan, 1)
    s1 = e.stars_to_device(stars_for_samples(stars, 1, 1))
    ws1 = e.workspace(B, K, 1)
    for k in range(B):
        ez1, Ez1 = e.polar_moments_samples(sm[k:k + 1])
        tab1, mv1 = e.kernel_table_samples(ez1, Ez1, rta1, 300)
        o1, _ = e.lnlike_ensemble_planned(rep1, None, None, s1, tab1, mv1, workspace=ws1)
        assert float(o1[0]) == got[k], k
    for k in (0, 9, 33):
        per_sample_polar(e, sm[k])
        tab_r, mv_r = e.kernel_table(rta1, 300)
        o2, _ = e.lnlike_ensemble_planned(plan, t_d, f_d, s_d, tab_r, mv_r, workspace=ws1)
        assert same(float(o2[0]), got[k], 1e-9), k
    # within 1e-8 of the oracle on the same moments (and -inf where the oracle says z > zmax)
    ezh, Ezh = ez.cpu().numpy(), Ez.cpu().numpy()
    for k in (0, 21, 63, int(np.argmax(flags != 0))):
        ref = _oracle_process(ezh[k], Ezh[k]).log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"])
        assert same(got[k], ref, TOL), (k, got[k], ref)


def test_8_samples_of_8_stars_on_a_replicated_plan(e15):
    import torch

    from starry_process_amd.engine import make_stars, stars_for_samples

    e = e15
    K, S, B = 320, 8, 8
    sts = [synthetic_star(s, K) for s in range(S)]
    t_d = e.f64(np.array([s["t"] for s in sts]))
    f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
    rng = np.random.RandomState(5)
    d_d = e.f64(1e-6 * (1 + rng.rand(S, K)))
    stars = make_stars(S, period=[s["p"] for s in sts], table=[0, 1] * 4, baseline_var=1e-6, baseline_mean=1e-4)
    s_d = e.stars_to_device(stars)
    rta1 = e.f64(e.rTA1L(np.array([[0.0, 0.0], [0.4, 0.2]])))
    plan = e.plan_data(t_d, f_d, s_d, diag=d_d, covpts=300)
    rep = e.replicate_plan(plan, B)
    sm = random_samples(B, seed=2)
    ez, Ez = e.polar_moments_samples(sm)
    tab, mv = e.kernel_table_samples(ez, Ez, rta1, 300)
    sB = e.stars_to_device(stars_for_samples(stars, B, 2))
    out, status = e.lnlike_ensemble_planned(rep, None, None, sB, tab, mv, workspace=e.workspace(B * S, K, 1))
    got = out.cpu().numpy().reshape(B, S)
    assert np.array_equal(np.isfinite(got).ravel(), status.cpu().numpy() == 0) and np.isfinite(got).sum() >= 48
    ws = e.workspace(B * S, K, 1)
    for b in range(B):
        o, _ = e.lnlike_ensemble_planned(plan, t_d, f_d, s_d, tab[2 * b:2 * b + 2].contiguous(),
                                         mv[2 * b:2 * b + 2].contiguous(), diag=d_d, workspace=ws)
        assert same(o.cpu().numpy(), got[b], 1e-11), b
    # against the oracle: star 3 under sample 5
    op = _oracle_process(ez[5].cpu().numpy(), Ez[5].cpu().numpy())
    ref = op.log_likelihood(sts[3]["t"], sts[3]["flux"], d_d[3].cpu().numpy(), p=sts[3]["p"], u=[0.4, 0.2],
                            baseline_mean=1e-4, baseline_var=1e-6)
    assert np.isfinite(ref) and same(got[5, 3], ref, TOL)


def test_ensemble_log_prob_packs_samples(e15):
    """EnsembleLogProb with 8 stars: the packed form (8 samples per call) = one sample per step."""
    from starry_process_amd.calibrate import EnsembleLogProb

    K, S = 256, 8
    sts = [synthetic_star(s, K) for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([s["flux"] for s in sts])
    per = [s["p"] for s in sts]
    sm = random_samples(21, seed=8)          # (not a multiple of the group: the last call is filled up)
    packed = EnsembleLogProb(t, flux, ferr=1e-3, p=per)(sm)
    single = EnsembleLogProb(t, flux, ferr=1e-3, p=per, batch_samples=False)(sm)
    assert np.isfinite(packed).sum() >= 18 and same(packed, single, 1e-9)
    # 64 stars and more: one sample per call, the same entry points
    sts64 = [synthetic_star(s, 128) for s in range(70)]
    t64, f64, p64 = np.array([s["t"] for s in sts64]), np.array([s["flux"] for s in sts64]), [s["p"] for s in sts64]
    big = EnsembleLogProb(t64, f64, ferr=1e-3, p=p64)
    assert big._batch is not None and big._batch.group == 1
    assert same(big(sm[:5]), EnsembleLogProb(t64, f64, ferr=1e-3, p=p64, batch_samples=False)(sm[:5]), 1e-9)
    soft = EnsembleLogProb(t, flux, ferr=1e-3, p=per, out_of_bounds="inf")
    mixed = np.vstack([sm[:2], [[20.0, -0.2, 0.3, 0.1, 10.0]], sm[2:4]])
    vm = soft(mixed)
    assert vm[2] == -np.inf and same(vm[[0, 1, 3, 4]], packed[:4], 1e-12)
    with pytest.raises(ValueError):
        EnsembleLogProb(t, flux, ferr=1e-3, p=per)(mixed)
    one = EnsembleLogProb(t[:1], flux[:1], ferr=1e-3, p=per[:1])
    assert one._batch is not None and one._batch.group == 64
    v = one(sm)
    w = EnsembleLogProb(t[:1], flux[:1], ferr=1e-3, p=per[:1], batch_samples=False)(sm)
    assert same(v, w, 1e-9)


def test_stream_depth_is_clamped_and_large_lag_grids_fall_back():
    """depth = 6 runs on MAX_STREAMS concurrent streams (a fifth loses 10-25 %, calibrate.MAX_STREAMS) with the values
    of depth = 3; a lag grid of 4 000 points (no room for the star's phases beside the table in the assembly's LDS, no
    tiles formed at first touch) goes through the planned and the batched calls all the same."""
    import warnings

    from starry_process_amd import calibrate
    from starry_process_amd.calibrate import EnsembleLogProb, get_log_prob_ensemble

    K, S = 128, 3
    sts = [synthetic_star(s, K) for s in range(S)]
    t, flux, per = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), [s["p"] for s in sts]
    sm = random_samples(7, seed=12)
    calibrate._warned_depth[0] = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        deep = EnsembleLogProb(t, flux, ferr=1e-3, p=per, depth=6, batch_samples=False)
    assert any("concurrent streams" in str(w.message) for w in rec)
    assert len(deep._slots) + 1 <= calibrate.MAX_STREAMS
    assert same(deep(sm), EnsembleLogProb(t, flux, ferr=1e-3, p=per, depth=3, batch_samples=False)(sm), 1e-12)
    big = EnsembleLogProb(t, flux, ferr=1e-3, p=per, covpts=4000)
    assert big._plan is not None and big._batch is not None
    v = big(sm[:2])
    for k in range(2):
        r, a, b, c, n = sm[k]
        ref = get_log_prob_ensemble(t, flux, ferr=1e-3, p=per, covpts=4000, upstream="device")(r, a, b, c, n)
        assert same(v[k], ref, 1e-9), k


def test_log_likelihood_samples_of_a_process():
    from starry_process_amd import StarryProcess

    K = 200
    st = synthetic_star(1, K)
    sm = random_samples(70, seed=4)
    sp = StarryProcess()
    got = np.asarray(sp.log_likelihood_samples(st["t"], st["flux"], 1e-6, sm, p=st["p"]))
    assert got.shape == (70,)
    for k in (0, 34, 69):
        r, a, b, c, n = sm[k]
        ref = float(StarryProcess(r=r, a=a, b=b, c=c, n=n, upstream="device").log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"]))
        assert same(got[k], ref, 1e-9), k
    assert np.isfinite(got).sum() >= 60
    again = np.asarray(sp.log_likelihood_samples(st["t"], st["flux"], 1e-6, sm[:3], p=st["p"]))
    assert np.array_equal(again, got[:3])
    # a Matern-3/2 process with per-cadence variances and two light curves on one covariance
    spt = StarryProcess(tau=2.0)
    F = np.vstack([st["flux"], st["flux"][::-1]])
    dv = 1e-6 * (1 + np.random.RandomState(0).rand(K))
    got = np.asarray(spt.log_likelihood_samples(st["t"], F, dv, sm[:5], p=st["p"], u=[0.3, 0.1]))
    for k in (0, 4):
        r, a, b, c, n = sm[k]
        ref = float(StarryProcess(r=r, a=a, b=b, c=c, n=n, tau=2.0, upstream="device").log_likelihood(
            st["t"], F, dv, p=st["p"], u=[0.3, 0.1]))
        assert same(got[k], ref, 1e-9), k
    assert np.isfinite(got).sum() >= 3
    with pytest.raises(ValueError):
        sp.log_likelihood_samples(st["t"], st["flux"], 1e-6, [[95.0, 0.4, 0.27, 0.1, 10.0]], p=st["p"])
    # out_of_bounds="inf": a walker outside the box gets -inf, the others their values
    mixed = np.vstack([sm[:2], [[95.0, 0.4, 0.27, 0.1, 10.0]], sm[2:3], [[20.0, 0.4, 1.5, 0.1, 10.0]]])
    v = np.asarray(sp.log_likelihood_samples(st["t"], st["flux"], 1e-6, mixed, p=st["p"], out_of_bounds="inf"))
    w = np.asarray(sp.log_likelihood_samples(st["t"], st["flux"], 1e-6, sm[:3], p=st["p"]))
    assert v[2] == -np.inf and v[4] == -np.inf and np.array_equal(v[[0, 1, 3]], w)


def test_samples_bad_arguments(e15):
    from starry_process_amd import _lib
    from starry_process_amd.engine import Engine, make_stars

    L = _lib.lib()
    e = e15
    st = e._stream()
    ez, Ez = e.empty(2, e.N), e.empty(2, e.N, e.N)
    good = np.ascontiguousarray([[0.3, 50.0, 9.0, 0.1, 10.0], [0.2, 1.0, 0.5, 0.1, 1.0]])
    e.set_size_basis()
    call = lambda arr, B=2, ezp=ez: L.sp_polar_moments_samples(e._h, B, _lib.hptr(arr) if arr is not None else None,
                                                             1e-12, 1e-9, e._p(ezp), e._p(Ez), st)
    assert call(good) == 0
    assert call(None) == -1 and call(good, ezp=None) == -1 and call(good, B=-1) == -1
    for col, val in ((0, 2.0), (0, -0.1), (1, 0.0), (2, -1.0), (4, -1.0), (3, np.nan)):
        bad = good.copy()
        bad[1, col] = val
        assert call(bad) == -1, (col, val)
    fresh = Engine(5, 2, 0)
    assert L.sp_polar_moments_samples(fresh._h, 1, _lib.hptr(good), 1e-12, 1e-9, e._p(ez), e._p(Ez), st) == -4
    # replicas: the planned call takes no data pointers of the caller's
    K = 128
    s0 = synthetic_star(0, K)
    t_d, f_d = e.f64(s0["t"][None, :]), e.f64(s0["flux"][None, None, :])
    s_d = e.stars_to_device(make_stars(1, period=1.0, data_var=1e-6))
    plan = e.plan_data(t_d, f_d, s_d, covpts=300)
    p = ctypes.c_void_p()
    assert L.sp_plan_replicate(e._h, plan.ptr, 0, st, ctypes.byref(p)) == -1
    assert L.sp_plan_replicate(e._h, None, 2, st, ctypes.byref(p)) == -1
    assert L.sp_plan_systems(plan.ptr) == 1 and L.sp_plan_systems(None) == -1
    rep = e.replicate_plan(plan, 2)
    tab, mv = e.kernel_table_samples(ez, Ez, e.f64(e.rTA1L(np.array([0.0, 0.0]))), 300)
    ws, out = e.workspace(2, K, 1), e.empty(2)
    s2 = e.stars_to_device(make_stars(2, period=1.0, data_var=1e-6, table=[0, 1]))
    f = lambda tp, fp: L.sp_lnlike_ensemble_planned(e._h, rep.ptr, tp, fp, None, e._p(s2), e._p(tab), e._p(mv), 20, 0.023,
                                                    e._p(ws), e._p(out), None, st)
    assert f(None, None) == 0
    assert f(e._p(t_d), e._p(f_d)) == -1          # not the replica's own arrays
    # ... and an un-replicated plan refuses arrays other than the planned ones
    other = t_d.clone()
    g = lambda tp: L.sp_lnlike_ensemble_planned(e._h, plan.ptr, tp, e._p(f_d), None, e._p(s_d), e._p(tab), e._p(mv), 20,
                                                0.023, e._p(ws), e._p(out), None, st)
    assert g(e._p(t_d)) == 0 and g(e._p(other)) == -1
