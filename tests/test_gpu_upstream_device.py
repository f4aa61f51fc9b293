"""
Upstream moments on the device (upstream_device.py): exact quadrature of
rotations with the path's own Wigner kernels.

What is asserted, and why these tolerances:
  * ydeg = 5, where the reference's algorithm is still well conditioned: agreement with
    the reference fixture to 1e-12 -- the two methods compute the same integrals;
  * ydeg = 15: against the EXTENDED-PRECISION evaluation of the defining integrals
    (tests/golden/upstream_extended.npz, made by tools/upstream_extended.py: 80-bit arithmetic,
    50-digit nodes) mu_y and every degree of Sigma_y to 1e-12 max|Sigma_y| -- the arbiter says
    the quadrature is right to rounding and the reference's own output is off by 1e-12 (l = 0)
    to 1e-2 (l = 15) of max|Sigma_y|, which is the only reason the comparison
    with the reference fixture below has loose bounds in the top degrees;
  * independently of the reference: the same expectation by brute-force quadrature
    (40x more nodes, plain trapezoid / Gauss-Jacobi refinement) agrees to 1e-13 -- the
    node counts are sufficient for exactness, as the degree argument says;
  * the likelihood built on these moments stays within 1e-5 of the one built on the
    reference's moments.
"""
import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu


def _moments(L, hp, **kw):
    from starry_process_amd.engine import get_engine
    from starry_process_amd.upstream_device import ylm_moments_device

    r, dr, a, b, c, n = hp
    mu, S = ylm_moments_device(get_engine(L, 2), r=r, dr=None if np.isnan(dr) else dr, a=a, b=b, c=c, n=n, **kw)
    return mu.cpu().numpy(), S.cpu().numpy()


def test_low_degree_matches_reference_exactly():
    g = golden("moments_L5")
    mu, S = _moments(5, g["default_hyper"])
    assert np.abs(mu - g["default_mean_ylm"]).max() < 1e-12 * np.abs(g["default_mean_ylm"]).max()
    assert np.abs(S - g["default_cov_ylm"]).max() < 1e-11 * np.abs(g["default_cov_ylm"]).max()
    assert np.array_equal(S, S.T)


@pytest.mark.parametrize("name", ["default", "hilat", "spread"])
def test_ydeg15_within_reference_noise(name):
    g = golden("moments_L15")
    mu, S = _moments(15, g[name + "_hyper"])
    mr, Sr = g[name + "_mean_ylm"], g[name + "_cov_ylm"]
    assert np.abs(mu - mr).max() < 1e-9 * np.abs(mr).max()
    scale = np.abs(Sr).max()
    d = np.abs(S - Sr)
    assert d[:25].max() < 1e-7 * scale          # rows of degree <= 4 (their columns reach l = 15)
    assert d[:81].max() < 1e-5 * scale          # l <= 8
    assert d.max() < 5e-2 * scale               # top degrees: the reference's own noise
    # Sigma_y must be a covariance: symmetric, positive definite with the eps added
    assert np.array_equal(S, S.T)
    assert np.linalg.eigvalsh(S).min() > 0


@pytest.mark.parametrize("name", ["default", "hilat", "spread"])
def test_ydeg15_matches_extended_precision(name):
    """The arbiter: the same integrals in 80-bit arithmetic (tools/upstream_extended.py)."""
    g = golden("moments_L15")
    x = golden("upstream_extended")
    mu, S = _moments(15, g[name + "_hyper"])
    me, Se = x[name + "_mean_ylm"], x[name + "_cov_ylm"]
    assert np.abs(mu - me).max() < 1e-12 * np.abs(me).max()
    scale = np.abs(Se).max()
    for l in range(16):
        blk = slice(l * l, (l + 1) ** 2)
        assert np.abs(S[blk] - Se[blk]).max() < 1e-12 * scale, l
    # ... and the reference's own output is further from it than we are, degree by degree
    Sr = g[name + "_cov_ylm"]
    for l in range(16):
        blk = slice(l * l, (l + 1) ** 2)
        assert np.abs(S[blk] - Se[blk]).max() <= np.abs(Sr[blk] - Se[blk]).max() + 1e-15 * scale, l


def test_quadrature_is_exact():
    """More nodes change nothing: refine both rules 3x and compare."""
    from starry_process_amd import upstream_device as ud

    g = golden("moments_L15")
    mu, S = _moments(15, g["default_hyper"])
    orig = ud.quadrature_nodes

    def finer(ydeg, alpha, beta):
        from scipy.special import roots_jacobi

        t, w = roots_jacobi(3 * (ydeg + 2), beta - 1.0, alpha - 1.0)
        x = 0.5 * (1 + t)
        w = w / w.sum()
        phi = np.arccos(x)
        nl = 3 * (2 * ydeg + 3) + 1
        return np.concatenate([phi, -phi]), 0.5 * np.concatenate([w, w]), 2 * np.pi * np.arange(nl) / nl

    ud.quadrature_nodes = finer
    try:
        mu2, S2 = _moments(15, g["default_hyper"])
    finally:
        ud.quadrature_nodes = orig
    assert np.abs(mu - mu2).max() < 1e-13 * np.abs(mu).max()
    assert np.abs(S - S2).max() < 1e-13 * np.abs(S).max()


def test_likelihood_and_calibrate_with_device_upstream():
    from starry_process_amd import StarryProcess
    from starry_process_amd.calibrate import get_log_prob

    st = synthetic_star(0, 1000)
    ref = StarryProcess(ydeg=15)
    dev = StarryProcess(ydeg=15, upstream="device")
    a = float(ref.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]))
    b = float(dev.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]))
    assert abs(b / a - 1) < 1e-5
    assert np.abs(np.array(dev.mean_ylm) - np.array(ref.mean_ylm)).max() < 1e-9 * np.abs(np.array(ref.mean_ylm)).max()
    g = golden("calibrate")
    f = get_log_prob(g["t"], g["flux"], upstream="device")
    assert abs(f(*g["default_hyper"]) / float(g["default"]) - 1) < 1e-4
    with pytest.raises(ValueError):
        StarryProcess(ydeg=15, upstream="nope")


@pytest.mark.parametrize("L,name", [(5, "default"), (15, "default"), (15, "hilat"), (15, "spread")])
def test_device_equals_cpu_quadrature(L, name):
    """Same nodes, same rotations, on the CPU (oracle.ylm_moments_quadrature): the device
    result must agree to rounding -- this pins the device kernels free of the reference's noise."""
    from oracle import sp_oracle as orc
    from starry_process_amd import upstream

    g = golden("moments_L%d" % L)
    hp = g[name + "_hyper"]
    r, dr, a, b, c, n = hp
    dr = None if np.isnan(dr) else dr
    mu, S = _moments(L, hp)
    s1, eigS = upstream.size_moments(r, dr, L)
    cols = eigS.T[np.abs(eigS).sum(axis=0) > 0.0] if dr is not None else s1[None, :]
    alpha, beta = upstream.ab_to_alphabeta(a, b)
    mu_o, S_o = orc.ylm_moments_quadrature(s1, cols, alpha, beta, c, n, L)
    assert np.abs(mu - mu_o).max() < 1e-13 * np.abs(mu_o).max()
    assert np.abs(S - S_o).max() < 1e-12 * np.abs(S_o).max()


@pytest.mark.parametrize("name", ["default", "hilat", "spread"])
def test_native_call_matches_the_composed_ops(name):
    """sp_ylm_moments_quadrature (one library call: staged upload + nine launches, csrc/sp_upstream.hip)
    against the same method composed from the path's ops in Python: the same rotations and sums up to
    their order (the longitudes' cos / sin come from a table instead of a Chebyshev recurrence)."""
    g = golden("moments_L15")
    mu_n, S_n = _moments(15, g[name + "_hyper"])
    mu_c, S_c = _moments(15, g[name + "_hyper"], native=False)
    assert np.abs(mu_n - mu_c).max() < 1e-13 * np.abs(mu_c).max()
    assert np.abs(S_n - S_c).max() < 1e-13 * np.abs(S_c).max()
    assert np.array_equal(S_n, S_n.T)


# ---- the whole prior box (round 4): latitude.py:176-197 lets beta reach exp(10) ------------------
def test_device_moments_over_the_prior_box():
    """(mu_y, Sigma_y) of the device upstream against the executed reference on a in {0, .5, 1} x
    b in {0, .25, .5, .74, .9, 1} (tests/golden/upstream_grid.npz), with the per-degree bounds of
    test_ydeg15_within_reference_noise -- round 3 returned NaN for b >= 0.737."""
    import warnings

    from test_upstream_grid import grid_cov_errors

    g = golden("upstream_grid")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for i, a in enumerate(g["a"]):
            for j, b in enumerate(g["b"]):
                mu, S = _moments(15, (20.0, np.nan, a, b, 0.1, 10.0))
                assert np.all(np.isfinite(mu)) and np.all(np.isfinite(S)), (a, b)
                mr = g["mean_ylm"][i, j]
                assert np.abs(mu - mr).max() < 1e-9 * np.abs(mr).max(), (a, b)
                scale, e4, e8, elow = grid_cov_errors(S, g, i, j)
                assert e4 < 1e-7 * scale and e8 < 1e-5 * scale and elow < 5e-2 * scale, (a, b)
                assert np.array_equal(S, S.T) and np.linalg.eigvalsh(S).min() > 0


@pytest.mark.parametrize("k", range(5))
def test_device_moments_against_extended_precision_at_the_corners_of_the_box(k):
    """at the corners of the prior box and at b = 0.9 the DEVICE moments are right to rounding in
    every degree (1e-12 of max |Sigma_y|, as at the three interior points of test_ydeg15_matches_extended_precision)
    and closer to the extended-precision arbiter than the reference's own moments are -- the distance between the two
    log-likelihoods over the box (LNLIKE_BOX_TOL) is the reference's error, not the device's."""
    from test_upstream_grid import extended_box_errors

    x = golden("upstream_extended_box")
    mu, S = _moments(15, (20.0, np.nan, float(x["a"][k]), float(x["b"][k]), 0.1, 10.0))
    ab, emu, own, ref = extended_box_errors(mu, S, k)
    assert emu[0] < 1e-12 and emu[0] <= max(emu[1], 1e-12), (ab, emu)
    assert np.all(own < 1e-12), (ab, own)
    assert np.all(own <= np.maximum(ref, 1e-12)), (ab, own, ref)


@pytest.mark.parametrize("branch", [0, 1])
def test_likelihood_over_the_prior_box(branch):
    """log_likelihood(upstream="device") against the reference's own value at the 18 grid points and on
    the 100-point b scan of tests/test_lnlike.py:84-88: finite everywhere, no RuntimeWarning, and within
    the distance the reference's high-degree noise moves its own likelihood."""
    import warnings

    from starry_process_amd import StarryProcess

    g = golden("upstream_grid")
    kw = dict() if branch == 0 else dict(marginalize_over_inclination=False, normalized=False)
    ckw = dict() if branch == 0 else dict(i=60.0)
    t, flux, dc, p = g["t"], g["flux"], float(g["data_cov"]), float(g["p"])
    worst = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for i, a in enumerate(g["a"]):
            for j, b in enumerate(g["b"]):
                sp = StarryProcess(ydeg=15, a=a, b=b, upstream="device", **kw)
                got = float(sp.log_likelihood(t, flux, dc, p=p, **ckw))
                ref = g["lnlike"][i, j, branch]
                assert np.isfinite(got), (a, b)
                worst = max(worst, abs(got - ref) / abs(ref))
                print("a=%.2f b=%.2f  device upstream %.10g  reference %.10g  (%.1e)" % (a, b, got, ref, abs(got / ref - 1)))
                assert abs(got - ref) < LNLIKE_BOX_TOL * abs(ref), (a, b, got, ref)
        ll = np.array([float(StarryProcess(ydeg=15, b=b, upstream="device", **kw).log_likelihood(t, flux, dc, p=p, **ckw))
                       for b in g["scan_b"]])
    ref = g["scan_lnlike"][:, branch]
    assert np.all(np.isfinite(ll)) and np.all(np.isfinite(ref))           # ALL 100: the reference's are
    assert np.abs(ll - ref).max() < LNLIKE_BOX_TOL * np.abs(ref).max()
    assert np.argmax(ll) == np.argmax(ref)
    print("worst relative distance to the reference over the box: %.2e; scan %.2e"
          % (worst, (np.abs(ll - ref) / np.abs(ref)).max()))


# (what the reference's rounding noise in the l >= 9 rows of Sigma_y -- up to 1e-2 of max|Sigma_y| at the
#  corners of the box, tests/test_upstream_grid.py -- moves its own log-likelihood by; at the three
#  hyperparameter sets with an extended-precision arbiter the device moments are the accurate ones)
LNLIKE_BOX_TOL = 5e-5     # measured: 1.6e-5 (marginal, normalised), 1.3e-6 (conditional)


def test_ensemble_log_prob_is_finite_at_high_b():
    """calibrate.EnsembleLogProb has no other upstream than the device's: b = 0.9 and the corners of the box
    must be finite and equal to get_log_prob_ensemble(upstream="device") within 1e-8."""
    import warnings

    from starry_process_amd.calibrate import EnsembleLogProb, get_log_prob_ensemble

    S, K = 4, 150
    sts = [synthetic_star(s, K) for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([s["flux"] for s in sts])
    p = np.array([s["p"] for s in sts])
    samples = np.array([[20.0, 0.40, 0.90, 0.10, 10.0], [20.0, 0.0, 1.0, 0.10, 10.0], [20.0, 1.0, 1.0, 0.10, 10.0],
                        [20.0, 0.5, 0.74, 0.10, 10.0], [20.0, 1.0, 0.0, 0.10, 10.0], [20.0, 0.0, 0.0, 0.10, 10.0]])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        # (without the Jacobian of (a, b) -> (mu, sigma): latitude.py:281-316 is NaN / -inf ON the box's edges
        #  a = 1 / b = 0, in the reference too -- the likelihood itself is what must be finite there)
        one = get_log_prob_ensemble(t, flux, ferr=1e-3, p=p, upstream="device", apply_jac=False)
        many = EnsembleLogProb(t, flux, ferr=1e-3, p=p, depth=3, apply_jac=False)
        ref = np.array([one(*s) for s in samples])
        got = many(samples)
        assert np.all(np.isfinite(got)) and np.all(np.isfinite(ref)), (got, ref)
        assert np.abs(got - ref).max() < 1e-8 * np.abs(ref).max()
        # ... and with it inside the box
        one = get_log_prob_ensemble(t, flux, ferr=1e-3, p=p, upstream="device")
        many = EnsembleLogProb(t, flux, ferr=1e-3, p=p, depth=3)
        ref = np.array([one(*s) for s in samples[:4]])
        got = many(samples[:4])
    assert np.all(np.isfinite(got)) and np.all(np.isfinite(ref))
    assert np.abs(got - ref).max() < 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("L,hp", [(15, dict(r=20.0, a=0.40, b=0.27, c=0.1, n=10.0)),
                                  (15, dict(r=12.0, a=0.85, b=0.60, c=0.2, n=3.0)),
                                  (15, dict(r=30.0, a=0.05, b=0.95, c=0.05, n=20.0)),
                                  (20, dict(r=15.0, a=0.62, b=0.11, c=0.2, n=5.0)),
                                  (5, dict(r=25.0, a=0.3, b=0.5, c=0.1, n=1.0))])
def test_exact_tangents_of_the_moments(L, hp):
    """ylm_moments_device_grad: the value is ylm_moments_device's, and d(mu_y, Sigma_y)/d(r, a, b) -- the tangents
    that ride through the rotations, with the quadrature rule differentiated with respect to its exponents -- agree
    with Richardson-extrapolated central differences of the device moments to the differences' own noise.  The
    reference has these derivatives analytically (ops/include/latitude.h:21-173, tests/test_latitude.py:90-129)."""
    from starry_process_amd.engine import get_engine
    from starry_process_amd.upstream_device import ylm_moments_device, ylm_moments_device_grad

    e = get_engine(L, 2)
    mu, Sig, dmu, dSig = [x.cpu().numpy() for x in ylm_moments_device_grad(e, **hp)]
    m0, S0 = [x.cpu().numpy() for x in ylm_moments_device(e, **hp)]
    assert np.abs(mu - m0).max() <= 1e-14 * np.abs(m0).max()
    assert np.abs(Sig - S0).max() <= 1e-14 * np.abs(S0).max()
    for k, (name, h) in enumerate((("r", 2e-2), ("a", 2e-3), ("b", 2e-3))):
        def at(d):
            q = dict(hp)
            q[name] = hp[name] + d
            return [x.cpu().numpy() for x in ylm_moments_device(e, **q)]

        def central(step):
            (m1, S1), (m2, S2) = at(step), at(-step)
            return (m1 - m2) / (2 * step), (S1 - S2) / (2 * step)

        (ma, Sa), (mb, Sb) = central(h), central(h / 2)
        fm, fS = (4 * mb - ma) / 3, (4 * Sb - Sa) / 3
        # (the differences carry the moments' rounding divided by the step: 1e-14 max|.| / h -- what decides where the
        #  derivative is small, a = 0.05 in the third case)
        em, eS = np.abs(dmu[k] - fm).max(), np.abs(dSig[k] - fS).max()
        assert em < 1e-7 * np.abs(fm).max() + 1e-14 * np.abs(m0).max() / h, (name, em, np.abs(fm).max())
        assert eS < 1e-7 * np.abs(fS).max() + 1e-14 * np.abs(S0).max() / h, (name, eS, np.abs(fS).max())
        assert np.array_equal(dSig[k], dSig[k].T)
