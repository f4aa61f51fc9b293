"""
CPU check (no GPU) of the two identities the planned likelihood step rests on, with the oracle as
the judge -- the NumPy statement of tools/planned_identities.py:

  * the covariance's mean is linear in the kernel table:  m = sum_n yp[n] wbar[n] / K^2,  wbar a function of the
    cadences' phases (and the temporal factor) alone  (flux.py:256-276, 322-330);
  * the normalised likelihood (sp.py:705-727, 1129-1188) from ONE factorisation of B = Sigma + D / c1 and the rows
    L^-1 1, L^-1 d, L^-1 r -- no row sums of Sigma, no vectors p, q:
        1' B^-1 q = (K - u_1.u_d) / (K m),   q' B^-1 q = (K^2 m - sum d + u_d.u_d) / (K m)^2,
        r' B^-1 q = (sum r - y.u_d) / (K m).

Against OracleProcess.log_likelihood (the reference's own order of operations) and the golden values of the executed
reference.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden

sys.path.insert(0, os.path.join(ROOT, "tools"))
from planned_identities import planned_lnlike, wbar_of  # noqa: E402

from oracle import sp_oracle as so  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402


@pytest.mark.parametrize("s,tau,vec,bvar", [(0, None, False, 0.0), (3, None, True, 1e-5), (7, 3.0, False, 0.0),
                                            (9, 0.7, True, 2e-6)])
def test_planned_form_equals_the_oracle(s, tau, vec, bvar):
    import scipy.linalg

    K = 300
    mom = golden("moments_L15")
    st = synthetic_star(s, K)
    t, flux, p = st["t"], st["flux"], st["p"]
    proc = so.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15, udeg=2, tau=tau)
    rng = np.random.RandomState(s)
    dv = 1e-6 * (1 + rng.rand(K)) if vec else np.full(K, 1e-6)
    ref = proc.log_likelihood(t, flux, dv if vec else 1e-6, p=p, baseline_var=bvar)
    mean, cov = proc.flux_mean_cov(t, 60.0, p)
    T = so.Matern32Kernel(t, t, tau) if tau is not None else None
    Sig = cov * T if T is not None else cov
    wbar = wbar_of(t, p, proc.covpts, T)
    if T is None:
        assert abs(wbar.sum() / K ** 2 - 1) < 1e-13           # the cubic's weights are a partition of unity
    m_plan = float(proc.tab["yp"] @ wbar) / K ** 2
    assert abs(m_plan / np.mean(Sig) - 1) < 1e-12
    mu = 1.0 + mean
    alpha, _, _, _ = so.alpha_beta(m_plan / mu ** 2, 20)
    d = dv / (alpha / mu ** 2)
    L = scipy.linalg.cholesky(Sig + np.diag(d), lower=True)
    val, z = planned_lnlike(L, K, flux, d, m_plan, mu, 20, bvar, float(np.sum(flux)), float(np.sum(d)))
    assert abs(val / ref - 1) < 1e-11
    assert abs(z - proc.z) < 1e-14


def test_planned_form_reproduces_the_reference():
    """cfg1 of BASELINE.json (ydeg 5, K 100) through the planned form: the executed reference's value."""
    import scipy.linalg

    mom = golden("moments_L5")
    K = 100
    t = np.linspace(0, 4, K)
    flux = 1e-2 * np.sin(2 * np.pi * t) + 1e-3 * np.random.RandomState(0).randn(K)
    proc = so.OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=5, udeg=2)
    ref = proc.log_likelihood(t, flux, 1e-6)
    assert abs(ref / 520.866380643320 - 1) < 1e-9               # SURVEY Appendix B, the executed reference
    mean, cov = proc.flux_mean_cov(t, 60.0, 1.0)
    wbar = wbar_of(t, 1.0, proc.covpts)
    m = float(proc.tab["yp"] @ wbar) / K ** 2
    mu = 1.0 + mean
    alpha, _, _, _ = so.alpha_beta(m / mu ** 2, 20)
    d = np.full(K, 1e-6) / (alpha / mu ** 2)
    L = scipy.linalg.cholesky(cov + np.diag(d), lower=True)
    val, _ = planned_lnlike(L, K, flux, d, m, mu, 20, 0.0, float(np.sum(flux)), float(np.sum(d)))
    assert abs(val / 520.866380643320 - 1) < 1e-9
