"""
GPU tests of the source-compatible front end (StarryProcess / FluxIntegral /
ops), written the way the reference's own tests read
(tests/test_lnlike.py, test_variance.py, test_ld.py, test_sum.py).
"""
import numpy as np
import pytest

from conftest import golden
from oracle import sp_oracle as orc
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu


def SP(L=15, **kw):
    from starry_process_amd import StarryProcess

    mom = golden("moments_L%d" % L)
    return StarryProcess(ydeg=L, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"], **kw)


def test_log_likelihood_matches_reference():
    g = golden("lnlike")
    sp = SP(15)
    st = synthetic_star(0, 1000)
    v = sp.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"])
    assert abs(float(v.eval()) / g["cfg2_L15_K1000"][0] - 1) < 1e-8
    # Appendix-B anchor of the survey
    t = np.linspace(0, 4, 1000)
    fl = 1e-2 * np.sin(2 * np.pi * t) + 1e-3 * np.random.RandomState(0).randn(1000)
    assert abs(float(sp.log_likelihood(t, fl, 1e-6)) - 5455.083646979640) < 5e-5
    # conditional, not normalised
    spc = SP(15, marginalize_over_inclination=False, normalized=False)
    st1 = synthetic_star(1, 1000)
    v = spc.log_likelihood(st1["t"], st1["flux"], st1["data_cov"], p=st1["p"], i=st1["i"])
    assert abs(float(v) / g["L15_K1000_cond"][1] - 1) < 1e-8


def test_data_cov_forms_agree():
    """scalar == vector == matrix data covariance; matrix baseline (sp.py:1135-1151)."""
    sp = SP(15)
    st = synthetic_star(3, 150)
    K = 150
    a = float(sp.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"]))
    b = float(sp.log_likelihood(st["t"], st["flux"], 1e-6 * np.ones(K), p=st["p"]))
    c = float(sp.log_likelihood(st["t"], st["flux"], 1e-6 * np.eye(K), p=st["p"]))
    assert abs(b / a - 1) < 1e-12 and abs(c / a - 1) < 1e-10
    d = float(sp.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"], baseline_var=1e-4, baseline_mean=1e-3))
    e = float(sp.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"], baseline_var=1e-4 * np.ones((K, K)),
                                baseline_mean=1e-3 * np.ones(K)))
    assert abs(e / d - 1) < 1e-10
    mom = golden("moments_L15")
    op = orc.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15)
    ref = op.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"], baseline_var=1e-4, baseline_mean=1e-3)
    assert abs(d / ref - 1) < 1e-8


def test_multi_lightcurve_batch():
    g = golden("lnlike")
    sp = SP(15)
    sts = [synthetic_star(s, 200) for s in range(5)]
    F = np.array([st["flux"] for st in sts])
    v = float(sp.log_likelihood(sts[0]["t"], F, 1e-6, p=1.3))
    assert abs(v / float(g["L15_K200_batchM5"]) - 1) < 1e-8


def test_ensemble_equals_per_star_calls():
    sp = SP(15)
    S, K = 6, 200
    sts = [synthetic_star(s, K) for s in range(10, 10 + S)]
    F = np.array([st["flux"] for st in sts])
    p = np.array([st["p"] for st in sts])
    us = np.array([[0.0, 0.0], [0.4, 0.2], [0.0, 0.0], [0.1, 0.3], [0.4, 0.2], [0.0, 0.0]])
    ens = np.asarray(sp.log_likelihood_ensemble(sts[0]["t"], F, 1e-6, p=p, u=us))
    one = np.array([float(sp.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"], u=uu)) for st, uu in zip(sts, us)])
    assert np.max(np.abs(ens / one - 1)) < 1e-12
    g = golden("lnlike")
    ens2 = np.asarray(sp.log_likelihood_ensemble(
        sts[0]["t"], np.array([synthetic_star(s, K)["flux"] for s in range(6)]), 1e-6,
        p=np.array([synthetic_star(s, K)["p"] for s in range(6)])))
    assert np.max(np.abs(ens2 / g["L15_K200"][:6] - 1)) < 1e-8


def test_mean_cov_and_variance_special_case():
    g = golden("cov_L15")
    sp = SP(15, normalized=False)
    k1 = np.asarray(sp.cov(np.array([0.3])))
    k2 = np.asarray(sp.cov(np.array([0.0, 0.1])))
    assert np.max(np.abs(k1 - g["k1_cov"])) < 1e-11 * np.abs(g["k1_cov"]).max()
    assert np.max(np.abs(k2 - g["k2_cov"])) < 1e-11 * np.abs(g["k2_cov"]).max()
    assert abs(k2[0, 0] - k1[0, 0]) < 1e-12 * abs(k1[0, 0])     # reference tests/test_variance.py:5-11
    t = g["t"]
    spn = SP(15, normalized=True, tau=2.5)
    cov = np.asarray(spn.cov(t, p=float(g["marg_mat32_p"])))
    assert np.max(np.abs(cov - g["marg_mat32_cov"])) < 5e-11 * np.abs(g["marg_mat32_cov"]).max()
    assert np.all(np.asarray(spn.mean(t)) == 0)
    m = np.asarray(sp.mean(t, p=1.37))
    assert np.allclose(m, g["marg_raw_mean"], rtol=1e-12)


def test_null_limb_darkening_and_bounds():
    """reference tests/test_ld.py:44-49; ops/exceptions.py:30-48."""
    from starry_process_amd import StarryProcess

    mom = golden("moments_L15")
    st = synthetic_star(2, 120)
    a = StarryProcess(ydeg=15, udeg=2, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"])
    b = StarryProcess(ydeg=15, udeg=0, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"])
    va = float(a.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"], u=[0.0, 0.0]))
    vb = float(b.log_likelihood(st["t"], st["flux"], 1e-6, p=st["p"], u=[]))
    assert abs(va / vb - 1) < 1e-9
    with pytest.raises(ValueError):
        a.log_likelihood(st["t"], st["flux"], 1e-6, i=95.0)
    with pytest.raises(ValueError):
        a.log_likelihood(st["t"], st["flux"], 1e-6, p=-1.0)
    with pytest.raises(ValueError):
        StarryProcess(ydeg=15, tau=-1.0, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"])
    assert float(a.log_likelihood(st["t"], st["flux"], -1.0)) == -np.inf


def test_ops_module_signatures():
    from starry_process_amd import ops

    g = golden("ops_L5")
    R, dR = ops.RxOp(5)(g["Rx_theta"][1])
    assert np.array_equal(R.eval(), g["Rx_R"][1]) and np.array_equal(dR.eval(), g["Rx_dR"][1])
    with pytest.raises(ValueError):
        ops.RxOp(5)(np.zeros(3))
    rng = np.random.RandomState(int(g["tdRz_seed"]))
    M = rng.randn(50, 36)
    th = rng.uniform(-7, 7, 50)
    f = ops.tensordotRzOp(5)(M, th)
    assert np.max(np.abs(f.eval() - g["tdRz_f"])) < 1e-13 * np.abs(g["tdRz_f"]).max()
    assert np.array_equal(ops.rTA1Op(5)().eval(), g["rTA1"])
    assert np.max(np.abs(ops.rTA1LOp(5, 2)(g["rTA1L_u"][1]).eval() - g["rTA1L"][1])) < 1e-13
    al, be, _, _ = ops.AlphaBetaOp(20)(4.2904487674796314e-4)
    assert abs(float(al) - 1.00128990414774) < 1e-14
    A = np.array([[4.0, 2.0], [2.0, 3.0]])
    L = ops.cho_factor(A).eval()
    assert np.allclose(L @ L.T, A, atol=1e-14) and L[0, 1] == 0
    x = ops.cho_solve(L, np.array([1.0, 2.0])).eval()
    assert np.allclose(A @ x, [1.0, 2.0], atol=1e-13)
    with pytest.raises(ValueError):
        ops.CheckBoundsOp(name="p", lower=0, upper=np.inf)(-1.0)


def test_flux_integral_mirror():
    from starry_process_amd.flux import FluxIntegral

    g = golden("cov_L15")
    mom = golden("moments_L15")
    fi = FluxIntegral(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15,
                      marginalize_over_inclination=False)
    A = np.asarray(fi.design_matrix(g["t"], float(g["cond_raw_i"]), float(g["cond_raw_p"]), g["cond_raw_u"]))
    assert np.max(np.abs(A - g["cond_raw_A"])) < 4e-12 * np.abs(g["cond_raw_A"]).max()
    cov = np.asarray(fi.cov(g["t"], float(g["cond_raw_i"]), float(g["cond_raw_p"]), g["cond_raw_u"]))
    assert np.max(np.abs(cov - g["cond_raw_cov"])) < 5e-11 * np.abs(g["cond_raw_cov"]).max()
    fm = FluxIntegral(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15)
    k = np.asarray(fm.kernel(np.array([0.0, 0.1, 0.7]), 60.0, 1.0, [0.0, 0.0]))
    tab = mom["default_u0_yp"]
    assert abs(k[0] - tab[1]) < 1e-12 * abs(tab[1])   # zero lag = yp at x = 0


def test_from_hyperparameters_end_to_end():
    """StarryProcess(r, a, b, c, n) with no injected moments: upstream.ylm_moments
    feeds the GPU path; the end-to-end value matches the executed reference
    (README quick-start call)."""
    from starry_process_amd import StarryProcess

    g = golden("lnlike")
    sp = StarryProcess(ydeg=15)  # all defaults, like the reference README
    mom = golden("moments_L15")
    mu, Sig = sp.mean_ylm.eval(), sp.cov_ylm.eval()
    assert np.abs(mu - mom["default_mean_ylm"]).max() < 1e-12
    # Sigma_y: the reference's eigen-truncated square roots times its polynomial Wigner
    # matrices amplify LAPACK rounding differences between CPUs up to ~1e-3 of max|Sigma|
    # in the l >= 12 rows; identical inputs, measured EPYC 9575F vs fixture host
    assert np.abs(Sig - mom["default_cov_ylm"]).max() < 5e-3 * np.abs(Sig).max()
    st = synthetic_star(0, 1000)
    v = float(sp.log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"]))
    # parity of the device path for exactly these moments: oracle on the same (mu, Sigma)
    ref = orc.OracleProcess(mu, Sig, ydeg=15).log_likelihood(st["t"], st["flux"], st["data_cov"], p=st["p"])
    assert abs(v / ref - 1) < 1e-8
    # and the end-to-end value stays close to the one the reference produced on its host
    assert abs(v / g["cfg2_L15_K1000"][0] - 1) < 1e-3
    up = golden("upstream")
    assert np.isclose(float(sp.log_jac()), float(up["default_log_jac"]), rtol=1e-12)
    # mu / sigma parametrisation (sp.py:243-255)
    sp2 = StarryProcess(ydeg=5, mu=30.0, sigma=5.0)
    assert np.isfinite(float(sp2.log_likelihood(st["t"][:100], st["flux"][:100], 1e-6)))
    with pytest.raises(ValueError):
        StarryProcess(ydeg=5, mu=30.0)


def test_calibrate_get_log_prob():
    """calibrate.get_log_prob.  Two references: (i) values produced by
    the reference's own classes evaluated along calibrate/log_prob.py:37-91 on the fixture
    host (make_golden.py: gen_calibrate) -- Sigma_y carries the cross-platform noise of
    DESIGN.md 8, hence 1e-4; (ii) the oracle fed with the moments computed on THIS host,
    which isolates the device path and the callable's plumbing: 1e-8."""
    from starry_process_amd import upstream
    from starry_process_amd.calibrate import get_log_prob, get_log_prob_ensemble

    g = golden("calibrate")
    t, flux = g["t"], g["flux"]
    K = len(t)

    def expected(hyper, m=0.0, v=0.0, i=60.0, p=1.0, ferr=1e-3, u=(0.0, 0.0), jac=True, **kw):
        r, a, b, c, n = hyper
        mu, Sig = upstream.ylm_moments(r=r, a=a, b=b, c=c, n=n, ydeg=15)
        o = orc.OracleProcess(mu, Sig, ydeg=15, covpts=K - 1, normalization_zmax=np.inf, **kw)
        ll = o.log_likelihood(t, flux, ferr ** 2, i=i, p=p, u=u, baseline_mean=m, baseline_var=10.0 ** v)
        return ll + (upstream.log_jac(a, b) if jac else 0.0)

    cases = [
        ("default", dict(), (), dict()),
        ("nojac_p", dict(apply_jac=False, p=1.3, ferr=2e-3), (), dict(jac=False, p=1.3, ferr=2e-3)),
        ("free_baseline", dict(baseline_mean=None, baseline_log_var=None), (1e-3, -5.0), dict(m=1e-3, v=-5.0)),
        ("cond_i", dict(marginalize_over_inclination=False, u=[0.3, 0.1]), (70.0,),
         dict(i=70.0, u=(0.3, 0.1), marginalize_over_inclination=False)),
        ("unnormalized", dict(normalized=False, baseline_log_var=-6.0), (), dict(v=-6.0, normalized=False)),
    ]
    for name, kw, extra, okw in cases:
        hyper = g[name + "_hyper"]
        val = get_log_prob(t, flux, **kw)(*hyper, *extra)
        assert abs(val / float(g[name]) - 1) < 1e-4, name
        assert abs(val / expected(hyper, **okw) - 1) < 1e-8, name
    # free flux: first positional argument, like the reference
    f0 = get_log_prob(t, flux)
    assert get_log_prob(t)(flux, *g["default_hyper"]) == f0(*g["default_hyper"])
    with pytest.raises(TypeError):
        f0(*g["default_hyper"], 1.0)
    # per-star ensemble with identical settings = the shared-covariance value
    # (the ensemble path applies the z > zmax guard of sp.py:1178-1183; not triggered here)
    fe = get_log_prob_ensemble(t, flux, ferr=1e-3, p=1.0, covpts=K - 1, apply_jac=False)
    fs = get_log_prob(t, flux, apply_jac=False)
    assert abs(fe(*g["default_hyper"]) / fs(*g["default_hyper"]) - 1) < 1e-9


def test_predict_and_sample_conditional():
    """StarryProcess.predict / sample_conditional against the reference's
    own predict (sp.py:767-903) on the fixture moments (tests/golden/make_golden.py:
    gen_predict): conditional mean and covariance for the marginal, conditional and
    time-variable branches, on a separate sample grid and on the observed grid."""
    from starry_process_amd import StarryProcess

    g = golden("predict")
    mom = golden("moments_L15")
    t, ts, flux = g["t"], g["ts"], g["flux"]
    cases = [
        ("marg", dict(marginalize_over_inclination=True), dict(p=0.9, u=[0.0, 0.0]), ts),
        ("marg_same_t", dict(marginalize_over_inclination=True), dict(p=0.9, u=[0.3, 0.1]), None),
        ("cond", dict(marginalize_over_inclination=False), dict(p=1.1, i=55.0, u=[0.4, 0.2]), ts),
        ("marg_tau", dict(marginalize_over_inclination=True, tau=2.0), dict(p=0.9, u=[0.0, 0.0]), ts),
    ]
    for name, ckw, kw, tsamp in cases:
        sp = StarryProcess(ydeg=15, normalized=False, mean_ylm=mom["default_mean_ylm"],
                           cov_ylm=mom["default_cov_ylm"], **ckw)
        mu, Kp = sp.predict(t, flux, 2.5e-7, t_sample=tsamp, baseline_mean=1e-4, baseline_var=1e-6, **kw)
        mu, Kp = np.array(mu), np.array(Kp)
        # the posterior covariance is a difference of nearly equal matrices (prior 1e-6
        # scale, posterior 1e-8): errors are measured against the prior scale
        scale = 1e-6 + np.abs(g[name + "_K"]).max()
        assert np.abs(mu - g[name + "_mu"]).max() < 1e-9 * np.abs(g[name + "_mu"]).max() + 1e-12, name
        assert np.abs(Kp - g[name + "_K"]).max() < 1e-9 * scale, name
    # samples: right shape, reproducible, and distributed around the conditional mean
    sp = StarryProcess(ydeg=15, normalized=False, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"])
    s1 = np.array(sp.sample_conditional(t, flux, 2.5e-7, t_sample=ts, p=0.9, nsamples=400, seed=3))
    s2 = np.array(sp.sample_conditional(t, flux, 2.5e-7, t_sample=ts, p=0.9, nsamples=400, seed=3))
    assert s1.shape == (400, len(ts)) and np.array_equal(s1, s2)
    mu, Kp = sp.predict(t, flux, 2.5e-7, t_sample=ts, p=0.9)
    sig = np.sqrt(np.diag(np.array(Kp)) + 1e-12)
    assert np.all(np.abs(s1.mean(0) - np.array(mu)) < 5 * sig / np.sqrt(400))
    with pytest.raises(NotImplementedError):
        StarryProcess(ydeg=15, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"]).predict(t, flux, 1e-6)


def test_ragged_ensemble_equals_per_star_calls():
    """Light curves of different lengths in one device call (sp_star.nobs): every star's
    value equals the one of a single-star call on its own cadences -- marginal and
    conditional branches, scalar and per-cadence noise, normalised and not."""
    rng = np.random.RandomState(11)
    lens = [300, 257, 64, 129, 1, 200]
    ts, fs, ps, dcs = [], [], [], []
    for s, n in enumerate(lens):
        st = synthetic_star(40 + s, 300)
        ts.append(st["t"][:n].copy())
        fs.append(st["flux"][:n].copy())
        ps.append(st["p"])
        dcs.append(1e-6 * (1 + rng.rand(n)))
    inc = [60.0, 35.0, 80.0, 15.0, 50.0, 70.0]
    for kw in (dict(), dict(marginalize_over_inclination=False), dict(normalized=False, tau=2.0), dict(tau=2.0)):
        sp = SP(15, **kw)
        for data_cov in (1e-6, dcs):
            ens = np.array(sp.log_likelihood_ensemble(ts, fs, data_cov, p=ps, i=inc, baseline_var=1e-7))
            for s, n in enumerate(lens):
                dc = data_cov if np.isscalar(data_cov) else data_cov[s]
                one = float(sp.log_likelihood(ts[s], fs[s], dc, p=ps[s], i=inc[s], baseline_var=1e-7))
                assert abs(ens[s] - one) <= 1e-10 * abs(one), (kw, s, n, ens[s], one)


def test_ragged_ensemble_through_calibrate():
    """get_log_prob_ensemble with light curves of different lengths = sum of single-star values."""
    from starry_process_amd import StarryProcess
    from starry_process_amd.calibrate import get_log_prob_ensemble

    lens = [120, 64, 97]
    ts, fs, ps = [], [], []
    for s, n in enumerate(lens):
        st = synthetic_star(70 + s, 120)
        ts.append(st["t"][:n].copy())
        fs.append(st["flux"][:n].copy())
        ps.append(st["p"])
    hyper = (20.0, 0.40, 0.27, 0.1, 10.0)
    f = get_log_prob_ensemble(ts, fs, ferr=1e-3, p=ps, apply_jac=False)
    tot = f(*hyper)
    sp = StarryProcess(ydeg=15, r=hyper[0], a=hyper[1], b=hyper[2], c=hyper[3], n=hyper[4])
    # (baseline_log_var defaults to 0, i.e. a baseline variance of 1, as in the reference)
    ref = sum(float(sp.log_likelihood(ts[s], fs[s], 1e-6, p=ps[s], baseline_var=1.0)) for s in range(3))
    assert abs(tot / ref - 1) < 1e-10


def test_prior_samples_and_flux():
    """sample / sample_ylm / flux (sp.py:489-516, 729-765, 1237-1282): the random stream of the
    reference cannot be reproduced, so the samples are pinned by their first two moments
    against the process mean / covariance, and `flux` against the design matrix."""
    sp = SP(15, normalized=False)
    t = np.linspace(0, 2.0, 40)
    n = 4000
    S = np.array(sp.sample(t, p=0.8, nsamples=n, seed=1))
    assert S.shape == (n, 40)
    mean, cov = np.array(sp.mean(t, p=0.8)), np.array(sp.cov(t, p=0.8))
    sig = np.sqrt(np.diag(cov))
    assert np.all(np.abs(S.mean(0) - mean) < 5 * sig / np.sqrt(n))
    emp = np.cov(S.T)
    assert np.abs(emp - cov).max() < 0.15 * np.abs(cov).max()
    Y = np.array(sp.sample_ylm(nsamples=3000, seed=2))
    assert Y.shape == (3000, 256)
    mom = golden("moments_L15")
    # l = 0, 1 coefficients: compare the empirical covariance block with Sigma_y
    blk = np.cov(Y[:, :4].T)
    assert np.abs(blk - mom["default_cov_ylm"][:4, :4]).max() < 0.15 * np.abs(mom["default_cov_ylm"][:4, :4]).max()
    L = np.array(sp.cho_cov_ylm)
    assert np.abs(L @ L.T - mom["default_cov_ylm"]).max() < 1e-13 * np.abs(mom["default_cov_ylm"]).max()
    # flux of given maps = design matrix product; conditional (fixed inclination) process
    spc = SP(15, normalized=False, marginalize_over_inclination=False)
    A = np.array(spc._flux.design_matrix(t, 65.0, 0.8, [0.2, 0.1]))
    F = np.array(spc.flux(Y[:5], t, i=65.0, p=0.8, u=[0.2, 0.1]))
    assert np.abs(F - Y[:5] @ A.T).max() < 1e-14 * max(1.0, np.abs(F).max())


def test_sum_of_processes():
    """sp1 + sp2 (sp.py:1190-1197, 1335-1400; the reference's tests/test_sum.py) against the
    executed reference (tests/golden/sum.npz): children from the fixture moments, so nothing
    upstream of the path enters the comparison."""
    from starry_process_amd import StarryProcess, StarryProcessSum

    mom = golden("moments_L15")
    g = golden("sum")
    for tag, kw in (("marg_norm", dict()),
                    ("cond_raw", dict(marginalize_over_inclination=False, normalized=False))):
        sp1 = StarryProcess(ydeg=15, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"], **kw)
        sp2 = StarryProcess(ydeg=15, mean_ylm=mom["hilat_mean_ylm"], cov_ylm=mom["hilat_cov_ylm"], **kw)
        sp = sp1 + sp2
        assert isinstance(sp, StarryProcessSum) and sp._children == [sp1, sp2]
        assert sum([sp1, sp2])._children == [sp1, sp2]
        assert (sp + sp1)._children == [sp1, sp2, sp1]
        args = dict(i=50.0, p=0.7, u=[0.3, 0.1])
        ll = float(sp.log_likelihood(g["t"], g["flux"], 1e-6, **args))
        assert abs(ll - float(g[tag + "_lnlike"])) < 1e-8 * abs(float(g[tag + "_lnlike"]))
        cov = np.array(sp.cov(g["t"][:40], **args))
        assert np.abs(cov - g[tag + "_cov"]).max() < 1e-10 * np.abs(g[tag + "_cov"]).max()
        mean = np.array(sp.mean(g["t"][:40], **args))
        assert np.abs(mean - g[tag + "_mean"]).max() < 1e-12 + 1e-10 * np.abs(g[tag + "_mean"]).max()
    with pytest.raises(AssertionError):
        sp1 + StarryProcess(ydeg=15, mean_ylm=mom["default_mean_ylm"], cov_ylm=mom["default_cov_ylm"])
    with pytest.raises(AssertionError):
        sp1 + 3


@pytest.mark.parametrize("marginalize_over_inclination", [True, False])
def test_lnlike_array(marginalize_over_inclination):
    """The reference's own end-to-end check (tests/test_lnlike.py:57-97): draw a light curve
    from the process at the default hyperparameters, scan the latitude parameter b over [0, 1]
    and require the likelihood to peak within 0.10 of the truth.  Hyperparameters -> moments on
    the device (upstream="device"), 100 evaluations of the K = 1000 likelihood."""
    from starry_process_amd import StarryProcess
    from starry_process_amd.defaults import defaults

    kw = dict(marginalize_over_inclination=marginalize_over_inclination, normalized=False,
              upstream="device")
    params = dict(r=defaults["r"], a=defaults["a"], b=defaults["b"], c=defaults["c"], n=defaults["n"])
    t = np.linspace(0, 1, 1000)
    gp = StarryProcess(**params, **kw)
    flux = np.array(gp.sample(t, p=defaults["p"], i=defaults["i"], seed=42)).reshape(-1)
    flux_err = 1e-3
    rng = np.random.RandomState(42)
    flux = flux + rng.randn(t.size) * flux_err
    import warnings

    b_arr = np.linspace(0.0, 1.0, 100)
    ll = np.empty(b_arr.size)
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # round 3: SciPy's Gauss-Jacobi overflowed for b >= 0.737
        for k, b in enumerate(b_arr):
            p2 = dict(params, b=b)
            ll[k] = float(StarryProcess(**p2, **kw).log_likelihood(t, flux, flux_err ** 2, p=defaults["p"],
                                                                  i=defaults["i"]))
    assert not np.any(np.isnan(ll))             # the reference's scan has no NaN (tests/golden/upstream_grid.npz)
    assert np.isfinite(ll).sum() >= 95          # (-inf: a covariance the factorisation rejects, math.py:82-91)
    assert abs(b_arr[np.argmax(ll)] - defaults["b"]) < 0.10


def test_ensemble_log_prob_many_samples():
    """calibrate.EnsembleLogProb (data resident, samples in flight on separate streams) gives,
    sample by sample, the value of get_log_prob_ensemble(upstream="device")."""
    from starry_process_amd.calibrate import EnsembleLogProb, get_log_prob_ensemble

    S, K = 6, 150
    sts = [synthetic_star(s, K) for s in range(S)]
    t = np.array([s["t"] for s in sts])
    flux = np.array([s["flux"] for s in sts])
    p = np.array([s["p"] for s in sts])
    inc = np.linspace(30.0, 80.0, S)
    samples = np.array([[20.0, 0.40, 0.27, 0.10, 10.0], [15.0, 0.62, 0.11, 0.20, 5.0],
                        [25.0, 0.30, 0.50, 0.05, 20.0], [12.0, 0.10, 0.90, 0.15, 3.0],
                        [20.0, 0.40, 0.27, 0.10, 10.0], [30.0, 0.80, 0.05, 0.02, 40.0],
                        [18.0, 0.55, 0.35, 0.12, 8.0]])
    for kw in (dict(), dict(marginalize_over_inclination=False, normalized=False, i=inc)):
        one = get_log_prob_ensemble(t, flux, ferr=1e-3, p=p, upstream="device", **kw)
        many = EnsembleLogProb(t, flux, ferr=1e-3, p=p, depth=3, **kw)
        ref = np.array([one(*s) for s in samples])
        got = many(samples)
        assert got.shape == (len(samples),)
        fin = np.isfinite(ref)                        # (z > zmax or a failed factorisation: -inf)
        assert fin.sum() >= 4 and np.array_equal(fin, np.isfinite(got))
        assert np.array_equal(got[~fin], ref[~fin])
        assert np.abs(got[fin] - ref[fin]).max() < 1e-12 * np.abs(ref[fin]).max()
        assert got[0] == got[4]                       # same sample on different slots: same bits
        assert np.abs(many(samples[:2]) - ref[:2]).max() < 1e-12 * np.abs(ref[:2]).max()
