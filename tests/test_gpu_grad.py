"""
End-to-end gradient of the log-likelihood (starry_process_amd/grad.py), pinned the way
the reference pins its own (tests/test_lnlike.py:100-136, ``theano.gradient.verify_grad``: the analytic
gradient against finite differences of the function) -- with the ORACLE as the function:

  * the value the differentiable graph computes is the oracle's log-likelihood (1e-9 relative);
  * d lnL / d(mu_y, Sigma_y): against central differences of the oracle's log-likelihood along random
    directions in (mu_y, Sigma_y), both branches, normalised or not, with and without a temporal kernel;
  * d lnL / d(r, a, b, c, n), d/dp, d/di, d/dtau: against central differences of the oracle's log-likelihood
    evaluated on the oracle's own upstream quadrature (oracle.ylm_moments_quadrature) -- the reference test's
    parameters, data set (t = linspace(0, 3, 100), unit-variance noise, normalized=False) and both branches.

Tolerances: central differences with one Richardson step (truncation O(h^4), rounding O(eps lnL / h)) leave
about 1e-8 of the gradient's scale; 2e-6 relative is asserted.
"""
import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu

YDEG = 15


def _oracle_lnlike(mu, Sig, t, flux, data_var, marg=True, normalized=True, tau=None, i=60.0, p=1.0,
                   kernel="matern32"):
    import oracle.sp_oracle as orc

    tk = orc.Matern32Kernel if kernel == "matern32" else orc.ExpSquaredKernel
    op = orc.OracleProcess(mu, Sig, ydeg=YDEG, udeg=2, marginalize_over_inclination=marg, normalized=normalized,
                           tau=tau, temporal_kernel=tk)
    return op.log_likelihood(t, flux, data_var, i=i, p=p)


def _central(f, h):
    """Central difference of f about 0 with one Richardson step: error O(h^4) (the normalised likelihood of a
    low-noise light curve has third derivatives large enough that the plain O(h^2) rule needs h < 1e-7)."""
    d1 = (f(h) - f(-h)) / (2 * h)
    d2 = (f(0.5 * h) - f(-0.5 * h)) / h
    return (4.0 * d2 - d1) / 3.0


def _oracle_moments(r, a, b, c, n):
    import oracle.sp_oracle as orc
    from starry_process_amd.upstream import ab_to_alphabeta, size_moments

    s1, _ = size_moments(r, None, YDEG)
    alpha, beta = ab_to_alphabeta(a, b)
    return orc.ylm_moments_quadrature(s1, s1[None, :], alpha, beta, c, n, YDEG)


def _reference_data():
    """tests/test_lnlike.py:108-112."""
    rng = np.random.RandomState(42)
    t = np.linspace(0, 3, 100)
    return t, rng.randn(len(t)), 1.0


def _moments():
    g = golden("moments_L15")
    return g["default_mean_ylm"], g["default_cov_ylm"]


CASES = [
    dict(marg=True, normalized=False, tau=None),
    dict(marg=True, normalized=True, tau=None),
    dict(marg=False, normalized=False, tau=None),
    dict(marg=False, normalized=True, tau=None),
    dict(marg=True, normalized=True, tau=0.7),
    dict(marg=False, normalized=False, tau=0.7),
]


def _data(case):
    if case["normalized"]:
        st = synthetic_star(3, 96)
        return st["t"], st["flux"], 1e-6, st["p"], st["i"]
    t, f, v = _reference_data()
    return t, f, v, 1.3, 70.0


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s-tau%s" % ("marg" if c["marg"] else "cond",
                                                                         "norm" if c["normalized"] else "raw", c["tau"]))
def test_value_and_moment_gradient_match_the_oracle(case):
    from starry_process_amd.grad import log_likelihood_with_grad

    mu, Sig = _moments()
    t, flux, dv, p, inc = _data(case)
    kw = dict(marginalize_over_inclination=case["marg"], normalized=case["normalized"], tau=case["tau"], i=inc, p=p)
    lnl, g = log_likelihood_with_grad(mu, Sig, t, flux, dv, ydeg=YDEG, **kw)
    okw = dict(marg=case["marg"], normalized=case["normalized"], tau=case["tau"], i=inc, p=p)
    ref = _oracle_lnlike(mu, Sig, t, flux, dv, **okw)
    assert np.isfinite(ref)
    assert abs(lnl - ref) < 1e-9 * abs(ref)
    rng = np.random.RandomState(7)
    for trial in range(2):
        dmu = rng.randn(mu.shape[0]) * np.abs(mu).max()
        B = rng.randn(*Sig.shape)
        dS = (B + B.T) * 0.5 * np.abs(Sig).max() / 16.0
        if trial == 1:      # the low degrees carry the signal: a direction confined to them
            dmu[25:] = 0.0
            dS[25:, :] = 0.0
            dS[:, 25:] = 0.0
        analytic = g["mean_ylm"] @ dmu + np.sum(g["cov_ylm"] * dS)
        f = lambda h: _oracle_lnlike(mu + h * dmu, Sig + h * dS, t, flux, dv, **okw)    # noqa: E731
        # two step sizes: their disagreement is the differences' own uncertainty (rounding noise of the oracle's
        # likelihood divided by h -- 1e-5 absolute on the low-noise normalised light curve, which matters where
        # the directional derivative happens to be small: the conditional normalised case)
        fd1, fd2 = _central(f, 4e-6), _central(f, 8e-6)
        fd, unc = 0.5 * (fd1 + fd2), abs(fd1 - fd2)
        print("directional derivative %.10g, differences %.10g +- %.1g" % (analytic, fd, unc))
        assert unc < 1e-4 * abs(fd)
        assert abs(analytic - fd) < 2e-6 * abs(fd) + 2 * unc, (trial, analytic, fd1, fd2)


@pytest.mark.parametrize("marg", [True, False], ids=["marg", "cond"])
@pytest.mark.parametrize("param", ["r", "a", "b", "c", "n", "i", "p", "tau"])
def test_hyper_gradient_against_finite_differences_of_the_oracle(param, marg):
    """The reference's test_lnlike_grad, parameter by parameter (plus tau)."""
    from starry_process_amd.defaults import defaults
    from starry_process_amd.grad import hyper_gradient

    if param == "i" and marg:
        pytest.skip("the inclination is integrated out")
    t, flux, dv = _reference_data()
    hp = {k: float(defaults[k]) for k in ("r", "a", "b", "c", "n")}
    ex = dict(i=65.0, p=1.1, tau=0.9 if param == "tau" else None)
    lnl, g = hyper_gradient(t, flux, dv, ydeg=YDEG, normalized=False, marginalize_over_inclination=marg, **hp, **ex)

    def f(x):
        q, e2 = dict(hp), dict(ex)
        (q if param in q else e2)[param] = x
        mu, Sig = _oracle_moments(**q)
        return _oracle_lnlike(mu, Sig, t, flux, dv, marg=marg, normalized=False, **e2)

    x0 = hp[param] if param in hp else ex[param]
    assert abs(lnl - f(x0)) < 1e-9 * abs(lnl)
    fd = _central(lambda h: f(x0 + h), 1e-4 * max(abs(x0), 0.1))
    assert abs(g[param] - fd) < 2e-6 * max(abs(fd), 1e-3 * abs(lnl)), (param, g[param], fd)


def test_normalised_hyper_gradient_and_failure():
    """Normalised process on a realistic light curve; and a star the likelihood rejects returns -inf with a
    zero gradient rather than NaNs."""
    from starry_process_amd.defaults import defaults
    from starry_process_amd.grad import hyper_gradient, log_likelihood_with_grad

    st = synthetic_star(5, 128)
    hp = {k: float(defaults[k]) for k in ("r", "a", "b", "c", "n")}
    lnl, g = hyper_gradient(st["t"], st["flux"], st["data_cov"], p=st["p"], ydeg=YDEG, **hp)
    for param in ("a", "c"):
        def f(x):
            q = dict(hp)
            q[param] = x
            mu, Sig = _oracle_moments(**q)
            return _oracle_lnlike(mu, Sig, st["t"], st["flux"], st["data_cov"], p=st["p"])
        fd = _central(lambda h: f(hp[param] + h), 1e-3 * max(hp[param], 0.1))   # (below 1e-4: rounding noise)
        assert abs(g[param] - fd) < 2e-6 * max(abs(fd), 1e-3 * abs(lnl)), (param, g[param], fd)
    mu, Sig = _moments()
    bad, gb = log_likelihood_with_grad(mu, -Sig, st["t"], st["flux"], 1e-12, ydeg=YDEG, normalized=False)
    assert bad == -np.inf and not np.any(gb["cov_ylm"]) and gb["p"] == 0.0


def test_facade_log_likelihood_grad():
    """StarryProcess.log_likelihood_grad: the value is log_likelihood's, the gradient hyper_gradient's; a process
    built from explicit moments (a sum of two populations) returns the gradient with respect to the moments."""
    from starry_process_amd import StarryProcess
    from starry_process_amd.grad import hyper_gradient

    st = synthetic_star(2, 80)
    sp = StarryProcess(r=20.0, a=0.5, b=0.3, c=0.08, n=5.0, tau=1.5, upstream="device")
    lnl, g = sp.log_likelihood_grad(st["t"], st["flux"], st["data_cov"], p=st["p"])
    ref = float(sp.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]).eval())
    assert abs(float(lnl.eval()) - ref) < 1e-9 * abs(ref)
    _, g2 = hyper_gradient(st["t"], st["flux"], st["data_cov"], r=20.0, a=0.5, b=0.3, c=0.08, n=5.0, tau=1.5,
                           p=st["p"])
    assert set(g) == {"r", "a", "b", "c", "n", "p", "tau"} and all(g[k] == g2[k] for k in g)
    # the DEFAULT constructor (upstream="reference") is a process built from hyperparameters too: same keys, the
    # value is its own log_likelihood, the gradient the device chain's to the accuracy the two upstreams agree
    spr = StarryProcess(r=20.0, a=0.5, b=0.3, c=0.08, n=5.0, tau=1.5)
    lnl_r, g_r = spr.log_likelihood_grad(st["t"], st["flux"], st["data_cov"], p=st["p"])
    ref_r = float(spr.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]).eval())
    assert abs(float(lnl_r.eval()) - ref_r) < 1e-9 * abs(ref_r)
    assert set(g_r) == set(g)
    for k in g:
        assert abs(g_r[k] - g[k]) < 1e-3 * max(abs(g[k]), 1e-3 * max(abs(v) for v in g.values())), k
    # custom stabilisers reach the chain rule; hyperparameters on the bounds of the prior box do not leave it
    spe = StarryProcess(r=20.0, a=0.0, b=1.0, c=0.08, n=5.0, epsy=1e-10, epsy15=1e-8, upstream="device")
    lnl_e, g_e = spe.log_likelihood_grad(st["t"], st["flux"], st["data_cov"], p=st["p"])
    ref_e = float(spe.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]).eval())
    assert abs(float(lnl_e.eval()) - ref_e) < 1e-9 * abs(ref_e) and all(np.isfinite(v) for v in g_e.values())
    sp0 = StarryProcess(r=20.0, c=0.0, upstream="device")
    _, g_0 = sp0.log_likelihood_grad(st["t"], st["flux"], st["data_cov"], p=st["p"])
    assert all(np.isfinite(v) for v in g_0.values()) and g_0["n"] == 0.0
    # per-point data variance, a spread of radii, the conditional branch
    sp = StarryProcess(r=20.0, dr=5.0, marginalize_over_inclination=False, normalized=False, upstream="device")
    var = np.linspace(1e-6, 2e-6, 80)
    lnl, g = sp.log_likelihood_grad(st["t"], st["flux"], var, i=st["i"], p=st["p"], baseline_var=1e-4)
    ref = float(sp.log_likelihood(st["t"], st["flux"], var, i=st["i"], p=st["p"], baseline_var=1e-4).eval())
    assert abs(float(lnl.eval()) - ref) < 1e-9 * abs(ref)
    assert set(g) == {"r", "dr", "a", "b", "c", "n", "p", "i"} and all(np.isfinite(v) for v in g.values())
    both = StarryProcess(r=15.0, upstream="device") + StarryProcess(r=25.0, a=0.2, upstream="device")
    lnl, g = both.log_likelihood_grad(st["t"], st["flux"], st["data_cov"], p=st["p"])
    ref = float(both.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]).eval())
    assert abs(float(lnl.eval()) - ref) < 1e-9 * abs(ref)
    assert g["mean_ylm"].shape == (256,) and g["cov_ylm"].shape == (256, 256)


# ---- the ensemble gradient (round 4): one device sweep for a batch of stars -------------------------------------
def _ensemble(S, K, seed0=0):
    sts = [synthetic_star(seed0 + s, K) for s in range(S)]
    return (np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]),
            np.array([s["p"] for s in sts]), sts)


@pytest.mark.parametrize("normalized", [True, False])
def test_ensemble_gradient_equals_the_sum_of_single_star_gradients(normalized):
    """EnsembleGradient (C^-1 by the factorisation's machinery, adjoint of the kernel table on the device) against
    hyper_gradient star by star (the reverse-mode kernels of round 3, pinned on the oracle above): same value,
    same gradient -- two independent routes through the chain rule."""
    from starry_process_amd.grad import EnsembleGradient, hyper_gradient

    S, K = 5, 120
    t, flux, p, sts = _ensemble(S, K)
    hp = dict(r=18.0, a=0.45, b=0.3, c=0.12, n=6.0)
    eg = EnsembleGradient(t, flux, ferr=1e-3, p=p, normalized=normalized)
    total, g = eg(**hp)
    ref_l, ref_g = 0.0, {k: 0.0 for k in hp}
    for s in range(S):
        l1, g1 = hyper_gradient(t[s], flux[s], 1e-6, p=float(p[s]), normalized=normalized, **hp)
        assert abs(eg.lnlike[s] - l1) < 1e-9 * abs(l1)
        ref_l += l1
        for k in hp:
            ref_g[k] += g1[k]
    assert abs(total - ref_l) < 1e-9 * abs(ref_l)
    scale = max(abs(v) for v in ref_g.values())
    for k in hp:
        assert abs(g[k] - ref_g[k]) < 2e-6 * max(abs(ref_g[k]), 1e-3 * scale), (k, g[k], ref_g[k])


def test_ensemble_gradient_against_finite_differences_of_the_oracle():
    """The reference's test_lnlike_grad for a batch (tests/test_lnlike.py:100-136): d sum_s lnL_s / d(r, a, b, c, n)
    against central differences of the ORACLE's log-likelihood on the oracle's own upstream quadrature."""
    from starry_process_amd.grad import EnsembleGradient

    S, K = 3, 100
    t, flux, p, sts = _ensemble(S, K, seed0=7)
    hp = dict(r=20.0, a=0.40, b=0.27, c=0.10, n=10.0)
    eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
    total, g = eg(**hp)

    def f_of(name):
        def f(d):
            q = dict(hp)
            q[name] = hp[name] + d
            mu, Sig = _oracle_moments(q["r"], q["a"], q["b"], q["c"], q["n"])
            return sum(_oracle_lnlike(mu, Sig, t[s], flux[s], 1e-6, p=float(p[s])) for s in range(S))
        return f

    ref0 = f_of("r")(0.0)
    assert abs(total - ref0) < 1e-8 * abs(ref0)
    for name, h in (("r", 1e-3), ("a", 1e-4), ("b", 1e-4), ("c", 1e-5), ("n", 1e-3)):
        fd = _central(f_of(name), h)
        assert abs(g[name] - fd) < 2e-5 * max(abs(fd), 1.0), (name, g[name], fd)


@pytest.mark.parametrize("S,K,tau,tspan", [(2, 1000, None, 4.0), (1, 3000, 3.0, 30.0)])
def test_ensemble_gradient_at_full_size_against_finite_differences_of_the_oracle(S, K, tau, tspan):
    """The same at bench.py's sizes: K = 1000 -- two super-panels, a trailing update and a second
    super-panel inside sp_spd_inverse_batched, which K = 100 (one pivot block) never reaches -- and K = 3000 with a
    Matern-3/2 kernel (47 pivot blocks, trailing updates on 128 x 64 tiles).  Central differences of the ORACLE's summed
    log-likelihood on the oracle's own upstream, one Richardson step, for every parameter of r, a, b, c, n."""
    from starry_process_amd.grad import EnsembleGradient

    sts = [synthetic_star(11 + s, K, tspan) for s in range(S)]
    t, flux, p = np.array([s["t"] for s in sts]), np.array([s["flux"] for s in sts]), np.array([s["p"] for s in sts])
    hp = dict(r=20.0, a=0.40, b=0.27, c=0.10, n=10.0)
    eg = EnsembleGradient(t, flux, ferr=1e-3, p=p, tau=tau)
    total, g = eg(**hp)

    def f_of(name):
        def f(d):
            q = dict(hp)
            q[name] = hp[name] + d
            mu, Sig = _oracle_moments(q["r"], q["a"], q["b"], q["c"], q["n"])
            return sum(_oracle_lnlike(mu, Sig, t[s], flux[s], 1e-6, p=float(p[s]), tau=tau) for s in range(S))
        return f

    ref0 = f_of("r")(0.0)
    assert abs(total - ref0) < 1e-8 * abs(ref0)
    # (steps: the oracle's value carries ~1e-10 relative of rounding at these sizes (cond C ~ 1e6), the differences
    #  divide it by h -- ten times the steps of the K = 100 test keep that under the tolerance, and the Richardson step
    #  keeps the truncation there too)
    for name, h in (("r", 1e-2), ("a", 1e-3), ("b", 1e-3), ("c", 1e-4), ("n", 1e-2)):
        fd = _central(f_of(name), h)
        assert abs(g[name] - fd) < 2e-5 * max(abs(fd), 1.0), (name, g[name], fd)


def test_ensemble_gradient_cfg3_shape_and_options():
    """cfg3's shape (64 stars, K = 1000): values equal sp_lnlike_ensemble's to 1e-9, the gradient is finite; per-
    cadence variances, a baseline variance, two limb-darkening tables and a temporal kernel take the same path."""
    from starry_process_amd.engine import get_engine, make_stars
    from starry_process_amd.grad import EnsembleGradient
    from starry_process_amd.upstream_device import ylm_moments_device

    S, K = 64, 1000
    t, flux, p, sts = _ensemble(S, K)
    eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
    total, g = eg()
    e = get_engine(15, 2)
    mu, Sig = ylm_moments_device(e)
    e.set_moments_dev(mu, Sig)
    rta1 = e.f64(e.rTA1L([0.0, 0.0]))
    tab, mv = e.kernel_table(rta1, 300)
    stars = e.stars_to_device(make_stars(S, period=p, data_var=1e-6))
    ref, status = e.lnlike_ensemble(e.f64(t), e.f64(flux[:, None, :]), stars, tab=tab, meanvar=mv)
    ref = ref.cpu().numpy()
    assert not status.cpu().numpy().any() and not eg.status.any()
    assert np.abs(eg.lnlike / ref - 1).max() < 1e-9
    assert set(g) == {"r", "a", "b", "c", "n"} and all(np.isfinite(v) for v in g.values())
    # options: small batch against hyper_gradient
    from starry_process_amd.grad import hyper_gradient

    S2, K2 = 3, 90
    t2, f2, p2, _ = _ensemble(S2, K2, seed0=3)
    var = np.linspace(1e-6, 3e-6, K2)[None, :] * np.ones((S2, 1))
    u = np.array([[0.0, 0.0], [0.4, 0.2], [0.4, 0.2]])
    eg2 = EnsembleGradient(t2, f2, ferr=np.sqrt(var), p=p2, u=u, baseline_var=1e-5, tau=2.0)
    tot2, g2 = eg2(r=22.0, a=0.3, b=0.5, c=0.08, n=4.0)
    ref_g = {k: 0.0 for k in g2}
    for s in range(S2):
        l1, g1 = hyper_gradient(t2[s], f2[s], var[s], p=float(p2[s]), u=u[s], baseline_var=1e-5, tau=2.0,
                                r=22.0, a=0.3, b=0.5, c=0.08, n=4.0)
        assert abs(eg2.lnlike[s] - l1) < 1e-9 * abs(l1)
        for k in ref_g:
            ref_g[k] += g1[k]
    scale = max(abs(v) for v in ref_g.values())
    for k in ref_g:
        assert abs(g2[k] - ref_g[k]) < 5e-6 * max(abs(ref_g[k]), 1e-3 * scale), (k, g2[k], ref_g[k])


def test_ensemble_gradient_spread_of_radii_and_rejected_stars():
    """dr joins the gradient when a spread of radii is given (against hyper_gradient); an evaluation the likelihood
    rejects (z > normalization_zmax, sp.py:1178-1183) returns -inf for those stars, flags them, and they add nothing
    to the gradient -- no NaN reaches the caller."""
    from starry_process_amd.grad import EnsembleGradient, hyper_gradient

    S, K = 3, 100
    t, flux, p, _ = _ensemble(S, K, seed0=11)
    eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
    hp = dict(r=22.0, dr=4.0, a=0.35, b=0.4, c=0.1, n=8.0)
    total, g = eg(**hp)
    assert set(g) == {"r", "dr", "a", "b", "c", "n"}
    ref = {k: 0.0 for k in g}
    for s in range(S):
        l1, g1 = hyper_gradient(t[s], flux[s], 1e-6, p=float(p[s]), **hp)
        assert abs(eg.lnlike[s] - l1) < 1e-9 * abs(l1)
        for k in ref:
            ref[k] += g1[k]
    scale = max(abs(v) for v in ref.values())
    for k in ref:
        assert abs(g[k] - ref[k]) < 5e-6 * max(abs(ref[k]), 1e-3 * scale), (k, g[k], ref[k])
    # a contrast for which the normalisation's expansion parameter is out of range: every star is rejected
    total, g = eg(r=20.0, a=0.4, b=0.27, c=0.9, n=20.0)
    assert total == -np.inf and np.all(eg.lnlike == -np.inf) and np.all(eg.status & 2)
    assert all(v == 0.0 for v in g.values())


def test_ensemble_gradient_on_the_boundary_of_the_contrast_box():
    """c = 0 or n = 0: the analytic limits, as hyper_gradient returns them -- not NaN."""
    from starry_process_amd.grad import EnsembleGradient, hyper_gradient

    S, K = 2, 90
    t, flux, p, _ = _ensemble(S, K, seed0=5)
    eg = EnsembleGradient(t, flux, ferr=1e-3, p=p)
    for cn in (dict(c=0.0, n=10.0), dict(c=0.1, n=0.0)):
        hp = dict(r=20.0, a=0.4, b=0.27, **cn)
        total, g = eg(**hp)
        ref = {k: 0.0 for k in hp}
        for s in range(S):
            _, g1 = hyper_gradient(t[s], flux[s], 1e-6, p=float(p[s]), **hp)
            for k in hp:
                ref[k] += g1[k]
        assert all(np.isfinite(g[k]) for k in hp), (cn, g)
        scale = max(abs(v) for v in ref.values())
        for k in ("c", "n"):
            assert abs(g[k] - ref[k]) < 2e-6 * max(abs(ref[k]), 1e-3 * scale), (cn, k, g[k], ref[k])


def test_gradient_sweep_refuses_ragged_stars_loudly():
    """sp_lnlike_grad_marginal takes every cadence as valid: a star with 0 < nobs < K gets NaN and SP_STAR_NAN, its
    neighbours their values (it used to return a silently wrong number)."""
    from starry_process_amd.engine import get_engine, make_stars

    S, K = 3, 100
    t, flux, p, _ = _ensemble(S, K, seed0=9)
    e = get_engine(15, 2)
    mu, Sig = _moments()
    e.set_moments(mu, Sig)
    tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
    good = e.lnlike_grad_marginal(e.f64(t), e.f64(flux), e.stars_to_device(make_stars(S, period=p, data_var=1e-6)),
                                  tab, mv)
    bad = e.lnlike_grad_marginal(e.f64(t), e.f64(flux),
                                 e.stars_to_device(make_stars(S, period=p, data_var=1e-6, nobs=[0, K - 1, K])), tab, mv)
    lg, lb = good[0].cpu().numpy(), bad[0].cpu().numpy()
    assert np.isnan(lb[1]) and (bad[3].cpu().numpy()[1] & 4)
    assert lb[0] == lg[0] and lb[2] == lg[2] and not good[3].cpu().numpy().any()


def test_exact_upstream_derivatives_against_the_differenced_tables():
    """r, a, b through the EXACT tangents of the moments (the default since round 5) against round 4's central
    differences of the kernel table (exact=False): the same gradient to the differences' O(h^2), for the ensemble
    sweep and for the one-star chain; c and n do not change."""
    from starry_process_amd.grad import EnsembleGradient, hyper_gradient

    S, K = 4, 150
    t, flux, p, sts = _ensemble(S, K, seed0=3)
    for hp in (dict(r=20.0, a=0.40, b=0.27, c=0.10, n=10.0), dict(r=12.0, a=0.8, b=0.6, c=0.2, n=3.0)):
        tot_e, g_e = EnsembleGradient(t, flux, ferr=1e-3, p=p)(**hp)
        tot_f, g_f = EnsembleGradient(t, flux, ferr=1e-3, p=p, exact=False)(**hp)
        assert abs(tot_e - tot_f) < 1e-11 * abs(tot_f)
        scale = max(abs(v) for v in g_f.values())
        for k in ("r", "a", "b"):
            assert abs(g_e[k] - g_f[k]) < 3e-6 * max(abs(g_f[k]), 1e-3 * scale), (k, g_e[k], g_f[k])
        for k in ("c", "n"):
            assert abs(g_e[k] - g_f[k]) < 1e-10 * max(abs(g_f[k]), 1e-3 * scale), (k, g_e[k], g_f[k])
        l_e, h_e = hyper_gradient(t[0], flux[0], 1e-6, p=float(p[0]), **hp)
        l_f, h_f = hyper_gradient(t[0], flux[0], 1e-6, p=float(p[0]), exact=False, **hp)
        assert abs(l_e - l_f) < 1e-11 * abs(l_f)
        scale = max(abs(v) for v in h_f.values())
        for k in ("r", "a", "b"):
            assert abs(h_e[k] - h_f[k]) < 3e-6 * max(abs(h_f[k]), 1e-3 * scale), (k, h_e[k], h_f[k])


@pytest.mark.parametrize("normalized,tau", [(True, None), (False, None), (True, 0.7)])
def test_ensemble_gradient_with_several_light_curves_per_star(normalized, tau):
    """M light curves on a star's ONE covariance (flux [S, M, K]; sp.py:1162-1171, what calibrate.get_log_prob sums):
    value and gradient are the sums over m of the one-curve sweeps -- an identity, since the curves share C -- and the
    value is the forward path's for the same stacked input.  One factorisation and one inverse per star whatever M
    (VERDICT r04 "missing" #3, the M > 1 half)."""
    from starry_process_amd.engine import make_stars
    from starry_process_amd.grad import EnsembleGradient

    S, M, K = 3, 5, 130
    t, _, p, sts = _ensemble(S, K, seed0=21)
    flux = np.array([[synthetic_star(100 * s + m, K)["flux"] for m in range(M)] for s in range(S)])
    hp = dict(r=17.0, a=0.35, b=0.4, c=0.15, n=4.0)
    kw = dict(ferr=1e-3, p=p, normalized=normalized, tau=tau, baseline_var=1e-4)
    eg = EnsembleGradient(t, flux, **kw)
    total, g = eg(**hp)
    ref_t, ref_g, ref_l = 0.0, {k: 0.0 for k in hp}, np.zeros(S)
    for m in range(M):
        e1 = EnsembleGradient(t, flux[:, m, :], **kw)
        t1, g1 = e1(**hp)
        ref_t += t1
        ref_l += e1.lnlike
        for k in hp:
            ref_g[k] += g1[k]
    assert abs(total - ref_t) < 1e-10 * abs(ref_t)
    assert np.abs(eg.lnlike - ref_l).max() < 1e-10 * np.abs(ref_l).max()
    scale = max(abs(v) for v in ref_g.values())
    for k in hp:
        assert abs(g[k] - ref_g[k]) < 1e-8 * max(abs(ref_g[k]), 1e-3 * scale), (k, g[k], ref_g[k])
    # the forward path on the stacked input, at the same tables
    from starry_process_amd.upstream_device import ylm_moments_device

    e = eg._e
    mu, Sig = ylm_moments_device(e, **hp)
    e.set_moments_dev(mu, Sig)
    tab, mv = e.kernel_table(eg._rta1, eg._covpts)
    fwd, _ = e.lnlike_ensemble(eg._t, eg._flux, eg._stars, tab=tab, meanvar=mv, covpts=eg._covpts,
                               temporal=eg._temporal, normalized=normalized)
    assert np.abs(fwd.cpu().numpy() - eg.lnlike).max() < 1e-9 * np.abs(eg.lnlike).max()


@pytest.mark.parametrize("normalized", [False, True])
def test_conditional_ensemble_gradient_against_finite_differences_of_the_oracle(normalized):
    """The conditional branch for an ensemble, each star at its own inclination (tests/test_lnlike.py:100-136 checks
    both branches; VERDICT r04 "missing" #3): the stars' moment adjoints summed, the chain through the upstream taken
    once with the exact tangents -- against central differences of the ORACLE's summed conditional log-likelihood on the
    oracle's own upstream, every hyperparameter, and per star in i and p."""
    from starry_process_amd.grad import ensemble_gradient_conditional

    S, K = 3, 80
    t, flux, p, sts = _ensemble(S, K, seed0=31)
    inc = np.array([35.0, 60.0, 80.0])
    hp = dict(r=20.0, a=0.40, b=0.27, c=0.10, n=10.0)
    total, g, lnl = ensemble_gradient_conditional(t, flux, ferr=1e-3, p=p, i=inc, normalized=normalized, **hp)

    def f_of(name, star=None):
        def f(d):
            q, ii, pp = dict(hp), inc.copy(), p.copy()
            if name == "i":
                ii[star] += d
            elif name == "p":
                pp[star] += d
            else:
                q[name] = hp[name] + d
            mu, Sig = _oracle_moments(q["r"], q["a"], q["b"], q["c"], q["n"])
            return sum(_oracle_lnlike(mu, Sig, t[s], flux[s], 1e-6, marg=False, normalized=normalized,
                                      i=float(ii[s]), p=float(pp[s])) for s in range(S))
        return f

    ref0 = f_of("r")(0.0)
    assert abs(total - ref0) < 1e-8 * abs(ref0) and abs(lnl.sum() - total) < 1e-12 * abs(total)
    for name, h in (("r", 1e-3), ("a", 1e-4), ("b", 1e-4), ("c", 1e-5), ("n", 1e-3)):
        fd = _central(f_of(name), h)
        assert abs(g[name] - fd) < 2e-5 * max(abs(fd), 1.0), (name, g[name], fd)
    for s in range(S):
        fd = _central(f_of("i", s), 1e-3)
        assert abs(g["i"][s] - fd) < 2e-5 * max(abs(fd), 1.0), ("i", s, g["i"][s], fd)
        fd = _central(f_of("p", s), 1e-6)
        assert abs(g["p"][s] - fd) < 2e-5 * max(abs(fd), 1.0), ("p", s, g["p"][s], fd)


def test_conditional_ensemble_gradient_on_the_boundary_of_the_box():
    """c = 0 (no contrast: where a bounded optimiser sits): mu = c n m1, Sigma - eps = c^2 n S1 give d/dc = n gmu.m1 there,
    not zero -- the three gradient entry points agree at the boundary."""
    from starry_process_amd.grad import ensemble_gradient_conditional

    S, K = 2, 80
    t, flux, p, sts = _ensemble(S, K, seed0=41)
    inc = np.array([40.0, 70.0])
    hp = dict(r=20.0, a=0.40, b=0.27, c=0.0, n=10.0)
    total, g, lnl = ensemble_gradient_conditional(t, flux, ferr=1e-3, p=p, i=inc, normalized=True, **hp)

    def f(d, name="c"):
        q = dict(hp)
        q[name] = hp[name] + d
        mu, Sig = _oracle_moments(q["r"], q["a"], q["b"], q["c"], q["n"])
        return sum(_oracle_lnlike(mu, Sig, t[s], flux[s], 1e-6, marg=False, normalized=True, i=float(inc[s]),
                                  p=float(p[s])) for s in range(S))

    assert abs(total - f(0.0)) < 1e-8 * abs(total)
    # (mu is odd and Sigma even in c: the oracle evaluates c < 0 as well.  The step is small: c^2 n S overtakes the
    #  noise variance at c ~ 1e-3, the series in c converges slowly beyond a tenth of that)
    fd = _central(f, 3e-6)
    assert abs(fd) > 0.05 and abs(g["c"] - fd) < 1e-4 * max(abs(fd), 1.0), (g["c"], fd)
    fdn = _central(lambda d: f(d, "n"), 1e-3)
    assert abs(g["n"] - fdn) < 1e-6 * max(abs(fd), 1.0), (g["n"], fdn)


def test_new_entry_points_refuse_bad_arguments():
    """The C ABI does not throw (include/starry_process_amd.h): the round's new entry points -- the moments with their
    tangents, the rule's derivatives, the gradient sweep for M light curves -- hand bad arguments back as status codes."""
    import ctypes
    import torch
    from starry_process_amd import _lib
    from starry_process_amd.engine import get_engine, make_stars
    from starry_process_amd.upstream_device import quadrature_nodes_grad

    e = get_engine(5, 2)
    L, N = _lib.lib(), e.N
    hp = _lib.hptr
    # sp_gauss_jacobi_grad
    buf = [np.empty(7) for _ in range(6)]
    assert L.sp_gauss_jacobi_grad(7, 0.5, 0.5, *[hp(x) for x in buf]) == 0
    assert L.sp_gauss_jacobi_grad(0, 0.5, 0.5, *[hp(x) for x in buf]) == -1
    assert L.sp_gauss_jacobi_grad(7, -1.5, 0.5, *[hp(x) for x in buf]) == -1
    assert L.sp_gauss_jacobi_grad(7, 0.5, 0.5, hp(buf[0]), hp(buf[1]), None, hp(buf[3]), hp(buf[4]), hp(buf[5])) == -1
    # sp_ylm_moments_quadrature_grad: odd P, latitudes that are not mirror pairs, null outputs
    phi, w, dphi, dw, lam = quadrature_nodes_grad(5, 3.0, 2.0)
    s0, ds = np.ones(N), np.zeros(N)
    mean, cov, dmean, dcov = e.empty(N), e.empty(N, N), e.empty(3, N), e.empty(3, N, N)
    arrs = [np.ascontiguousarray(x) for x in (s0, ds, phi, w, dphi, dw)]

    def call(P=len(phi), arrays=arrs, cov_p=e._p(cov)):
        return L.sp_ylm_moments_quadrature_grad(e._h, *[hp(x) for x in arrays], P, len(lam), 1.0, 1.0, 1e-12, 1e-5,
                                                e._p(mean), cov_p, e._p(dmean), e._p(dcov), e._stream())

    assert call() == 0
    assert call(P=len(phi) - 1) == -1
    assert call(cov_p=None) == -1
    bad = [a.copy() for a in arrs]
    bad[2][-1] *= 0.5                                  # the last latitude no longer mirrors its partner
    assert call(arrays=bad) == -1
    torch.cuda.synchronize()
    assert np.all(np.isfinite(cov.cpu().numpy()))
    # sp_lnlike_grad_marginal_multi
    K, S = 40, 1
    rta1 = e.f64(e.rTA1L([0.0, 0.0]))
    e.set_moments_dev(mean, cov)
    tab, mv = e.kernel_table(rta1, 300)
    st = synthetic_star(0, K)
    t, f = e.f64(st["t"][None, :]), e.f64(np.stack([st["flux"], st["flux"][::-1]])[None])
    stars = e.stars_to_device(make_stars(S, period=1.0, data_var=1e-6))
    ws = e.grad_workspace(S, K, 300, 2)
    out, yb, mb = e.empty(S), e.empty(S, 304), e.empty(S)

    def sweep(S_=S, K_=K, M_=2, covpts=300, ws_p=e._p(ws)):
        return L.sp_lnlike_grad_marginal_multi(e._h, S_, K_, M_, e._p(t), e._p(f), None, e._p(stars), covpts, e._p(tab),
                                               e._p(mv), 0, 1, 20, ctypes.c_double(0.023), ws_p, e._p(out), e._p(yb),
                                               e._p(mb), None, e._stream())

    assert sweep() == 0
    assert sweep(M_=0) == -1
    assert sweep(K_=1) == -1
    assert sweep(ws_p=None) == -1
    assert sweep(covpts=123) == -4                     # the table was built for another lag grid: SP_ERR_STATE
    assert sweep(S_=0) == 0
    assert L.sp_lnlike_grad_workspace_bytes_multi(e._h, 1, 40, 0, 300) == 0
    assert L.sp_lnlike_grad_workspace_bytes_multi(e._h, 1, 40, 3, 300) > L.sp_lnlike_grad_workspace_bytes(e._h, 1, 40, 300)
    torch.cuda.synchronize()
