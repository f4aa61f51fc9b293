"""
Short light curves in one kernel (csrc/sp_small.hip, round 6): sp_lnlike_ensemble_planned at K <= 128 evaluates a star
in one workgroup's LDS -- assembly, factorisation, riding rows, reduction -- instead of the blocked path's launches.

Asserted: the same values as the blocked planned step (sp_debug_set_small_k(0)) to 1e-10 and as the CPU oracle to 1e-8
(BASELINE.json) over sizes around the 64-row block boundary, one or two light curves per star, scalar and per-cadence
variances, baseline terms, limb darkening, both temporal kernels, ragged light curves; failure semantics (z > zmax, a
matrix that is not positive definite, a stale plan); a star's value independent of its batch.
"""
import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def engine():
    from starry_process_amd.engine import Engine

    e = Engine(15, 2, 0)
    mom = golden("moments_L15")
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    return e


def small_k(on):
    from starry_process_amd import _lib

    _lib.check(_lib.lib().sp_debug_set_small_k(int(on)))


@pytest.fixture(autouse=True)
def restore_default():
    yield
    small_k(-1)


def both_paths(e, t, flux, stars, tab, mv, diag=None, temporal=None, zmax=0.023):
    t_d, f_d, s_d = e.f64(t), e.f64(flux), e.stars_to_device(stars)
    d_d = None if diag is None else e.f64(diag)
    plan = e.plan_data(t_d, f_d, s_d, diag=d_d, covpts=300, temporal=temporal)
    out = []
    for on in (1, 0):
        small_k(on)
        v, st = e.lnlike_ensemble_planned(plan, t_d, f_d, s_d, tab, mv, diag=d_d, zmax=zmax)
        out.append((v.cpu().numpy().copy(), st.cpu().numpy().copy()))
    return out


@pytest.mark.parametrize("K", [2, 3, 17, 40, 63, 64, 65, 100, 127, 128])
def test_small_k_equals_the_blocked_path_and_the_oracle(engine, K):
    from oracle.sp_oracle import OracleProcess
    from starry_process_amd.engine import make_stars

    e = engine
    S = 13
    sts = [synthetic_star(s, K) for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([s["flux"] for s in sts])[:, None, :]
    stars = make_stars(S, period=[s["p"] for s in sts], data_var=1e-6)
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    (v1, s1), (v0, s0) = both_paths(e, t, flux, stars, tab, mv)
    assert not s1.any() and not s0.any() and np.all(np.isfinite(v1))
    assert np.max(np.abs(v1 / v0 - 1)) < 1e-10
    mom = golden("moments_L15")
    op = OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15)
    for s in (0, 7, 12):
        ref = op.log_likelihood(sts[s]["t"], sts[s]["flux"], 1e-6, p=sts[s]["p"])
        assert abs(v1[s] / ref - 1) < TOL, (K, s, v1[s], ref)
    # a star's value does not depend on its batch
    small_k(1)
    t1, f1, st1 = e.f64(t[5:6]), e.f64(flux[5:6]), e.stars_to_device(stars[5:6])
    one, _ = e.lnlike_ensemble_planned(e.plan_data(t1, f1, st1, covpts=300), t1, f1, st1, tab, mv)
    assert float(one[0]) == v1[5]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_small_k_random_configurations(engine, seed):
    """Two light curves per star, per-cadence variances, baseline terms, limb darkening, temporal kernels, irregular and
    ragged cadences: small-K kernel = blocked path (1e-10) = oracle (1e-8, one star per case)."""
    from oracle import sp_oracle as orc
    from starry_process_amd.engine import make_stars

    rng = np.random.RandomState(300 + seed)
    e = engine
    mom = golden("moments_L15")
    for case in range(8):
        K = int(rng.choice([5, 31, 64, 66, 90, 128]))
        M = int(rng.choice([1, 1, 2]))
        S = 4
        tau = None if rng.rand() < 0.5 else float(rng.uniform(0.5, 5.0))
        tk = str(rng.choice(["matern32", "expsquared"]))
        u = (0.0, 0.0) if rng.rand() < 0.5 else tuple(rng.uniform(0, 0.4, 2))
        per = rng.uniform(0.3, 3.0, S)
        t = np.array([np.sort(rng.uniform(0, 6, K)) if rng.rand() < 0.5 else np.linspace(0, 4, K) for _ in range(S)])
        flux = np.array([[1e-2 * np.sin(2 * np.pi * t[s] / per[s]) * rng.rand() + 1e-3 * rng.randn(K) for _ in range(M)]
                         for s in range(S)])
        vec = rng.rand() < 0.5 and M == 1          # (M + 2 riding rows at most four: two light curves take a scalar variance)
        diag = 1e-6 * (1 + rng.rand(S, K)) if vec else None
        bvar = float(rng.choice([0.0, 1e-6, 1e-3]))
        bmean = float(rng.choice([0.0, 1e-3]))
        nobs = [0] * S if (rng.rand() < 0.6 or K < 6) else [K, int(rng.randint(2, K)), int(rng.randint(2, K)), 0]
        stars = make_stars(S, period=per, tau=tau or 0.0, data_var=1e-6, baseline_var=bvar, baseline_mean=bmean, nobs=nobs)
        tab, mv = e.kernel_table(e.f64(e.rTA1L(u)), 300)
        (v1, s1), (v0, s0) = both_paths(e, t, flux, stars, tab, mv, diag=diag, temporal=tk if tau else None)
        assert np.array_equal(s1, s0), (case, s1, s0)
        fin = np.isfinite(v0)
        assert np.array_equal(np.isfinite(v1), fin) and fin.any()
        assert np.max(np.abs(v1[fin] / v0[fin] - 1)) < 1e-10, (case, K, M, tau, v1, v0)
        s = int(np.argmax(fin))
        n = nobs[s] if nobs[s] else K
        op = orc.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15, tau=tau,
                               temporal_kernel=orc.Matern32Kernel if tk == "matern32" else orc.ExpSquaredKernel)
        ref = op.log_likelihood(t[s][:n], flux[s][:, :n], diag[s][:n] if vec else 1e-6, p=per[s], u=u, baseline_mean=bmean,
                                baseline_var=bvar)
        assert abs(v1[s] / ref - 1) < TOL, (case, K, M, s, v1[s], ref)


def test_small_k_failure_semantics(engine):
    from starry_process_amd.engine import make_stars

    e = engine
    K, S = 96, 3
    sts = [synthetic_star(s, K) for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([s["flux"] for s in sts])[:, None, :]
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    # z > zmax: -inf and SP_STAR_ZMAX on both paths
    stars = make_stars(S, period=[s["p"] for s in sts], data_var=1e-6)
    (v1, s1), (v0, s0) = both_paths(e, t, flux, stars, tab, mv, zmax=1e-9)
    assert np.all(v1 == -np.inf) and np.all(s1 & 2) and np.array_equal(s1, s0)
    # not positive definite (a negative data variance larger than the kernel's): -inf and SP_STAR_NOT_PD
    bad = make_stars(S, period=[s["p"] for s in sts], data_var=[1e-6, -1.0, 1e-6])
    (v1, s1), (v0, s0) = both_paths(e, t, flux, bad, tab, mv)
    assert v1[1] == -np.inf and (s1[1] & 1) and np.isfinite(v1[0]) and np.isfinite(v1[2])
    assert np.array_equal(np.isfinite(v1), np.isfinite(v0))
    # a plan that does not belong to the stars: NaN and SP_STAR_STALE_PLAN
    small_k(1)
    t_d, f_d = e.f64(t), e.f64(flux)
    plan = e.plan_data(t_d, f_d, e.stars_to_device(stars), covpts=300)
    other = make_stars(S, period=[1.01 * s["p"] if k == 2 else s["p"] for k, s in enumerate(sts)], data_var=1e-6)
    v, st = e.lnlike_ensemble_planned(plan, None, None, e.stars_to_device(other), tab, mv)
    v, st = v.cpu().numpy(), st.cpu().numpy()
    assert np.isnan(v[2]) and (st[2] & 8) and np.isfinite(v[0]) and not st[0]


@pytest.mark.parametrize("K", [48, 64, 100, 128])
def test_small_k_four_riding_rows(engine, K):
    """Two light curves per star AND per-cadence variances: four riding rows (residuals, ones, variances), the most the
    kernel serves -- every wavefront substitutes a row (K <= 64 and the second block of K > 64) -- against the blocked
    path (1e-10) and the oracle (1e-8)."""
    from oracle.sp_oracle import OracleProcess
    from starry_process_amd.engine import make_stars

    e = engine
    S, M = 5, 2
    rng = np.random.RandomState(K)
    sts = [synthetic_star(s, K) for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([[s["flux"], s["flux"][::-1] * 0.5 + 1e-3 * rng.randn(K)] for s in sts])
    diag = 1e-6 * (1 + rng.rand(S, K))
    stars = make_stars(S, period=[s["p"] for s in sts], data_var=1e-6, baseline_var=1e-4)
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    (v1, s1), (v0, s0) = both_paths(e, t, flux, stars, tab, mv, diag=diag)
    assert not s1.any() and not s0.any() and np.all(np.isfinite(v1))
    assert np.max(np.abs(v1 / v0 - 1)) < 1e-10
    mom = golden("moments_L15")
    op = OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15)
    ref = op.log_likelihood(t[2], flux[2], diag[2], p=sts[2]["p"], baseline_var=1e-4)
    assert abs(v1[2] / ref - 1) < TOL, (K, v1[2], ref)


@pytest.mark.parametrize("K", [40, 96])
def test_small_k_more_stars_than_cus(engine, K):
    """1 100 stars: workgroups 256 apart in the grid share a CU and take the pivot block's roles in rotated order
    (sp_small.hip: tid_rot), four rotations in all -- every star equals its value in the blocked path, and the stars
    are copies of 11 different ones, so a star's value must not depend on its rotation either."""
    from starry_process_amd.engine import make_stars

    e = engine
    base = [synthetic_star(s, K) for s in range(11)]
    S = 1100
    sts = [base[s % 11] for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([s["flux"] for s in sts])[:, None, :]
    stars = make_stars(S, period=[s["p"] for s in sts], data_var=1e-6)
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    (v1, s1), (v0, s0) = both_paths(e, t, flux, stars, tab, mv)
    assert not s1.any() and not s0.any()
    assert np.max(np.abs(v1 / v0 - 1)) < 1e-10
    for k in range(11):
        assert np.unique(v1[k::11]).size == 1, k


@pytest.mark.parametrize("K", [60, 100])
def test_small_k_large_table(K):
    """covpts = 1000: the largest spline tables the one-kernel path serves (K <= 64: in the pivot block's place in the
    LDS, K > 64: a region of its own, 77 KB per workgroup) -- equal to the blocked path, within 1e-8 of the oracle."""
    from oracle.sp_oracle import OracleProcess
    from starry_process_amd.engine import Engine, make_stars

    e = Engine(15, 2, 0)
    mom = golden("moments_L15")
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    S, covpts = 6, 1000
    sts = [synthetic_star(s, K) for s in range(S)]
    t_d = e.f64(np.array([s["t"] for s in sts]))
    f_d = e.f64(np.array([s["flux"] for s in sts])[:, None, :])
    s_d = e.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), covpts)
    plan = e.plan_data(t_d, f_d, s_d, covpts=covpts)
    out = []
    for on in (1, 0):
        small_k(on)
        v, st = e.lnlike_ensemble_planned(plan, None, None, s_d, tab, mv)
        assert not st.cpu().numpy().any()
        out.append(v.cpu().numpy().copy())
    assert np.max(np.abs(out[0] / out[1] - 1)) < 1e-10
    op = OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15, covpts=covpts)
    ref = op.log_likelihood(sts[3]["t"], sts[3]["flux"], 1e-6, p=sts[3]["p"])
    assert abs(out[0][3] / ref - 1) < TOL
