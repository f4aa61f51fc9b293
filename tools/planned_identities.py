#!/usr/bin/env python3
"""The identities behind the planned likelihood step, checked in NumPy against the oracle (CPU only).

Normalised marginal likelihood (sp.py:705-727): C = c1 Sigma + z ((alpha + beta) p p^T - alpha q q^T) + D + b 1 1^T
with q = Sigma 1 / (K m), p = 1 - q, m = mean(Sigma).  The device factors B = Sigma + D / c1 = L L^T and applies
the rank-2 (+ baseline) part afterwards; what it needs of q is only its Gram entries under B^-1, and those follow
from rows that ride anyway:

    Sigma 1 = B 1 - d        (d = diag(D) / c1)
    1^T B^-1 q = (K - 1^T B^-1 d) / (K m)
    q^T B^-1 q = (K^2 m - sum(d) + d^T B^-1 d) / (K m)^2
    r^T B^-1 q = (sum(r) - d^T B^-1 r) / (K m)

and m itself is linear in the kernel table: cov_ij = sum_k yp[s_ij + k] b_k(x0_ij) (flux.py:256-276, 322-330), so
m = sum_n yp[n] wbar[n] / K^2 with wbar a function of the cadences' phases alone.

usage: python tools/planned_identities.py [K]
"""
import os
import sys

import numpy as np
import scipy.linalg

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

from oracle import sp_oracle as so  # noqa: E402
from starry_process_amd.synthetic import synthetic_star  # noqa: E402


def spline_weights(x0):
    """b_k(x0): the cubic a0 + a1 x0 + a2 x0^2 + a3 x0^3 of flux.py:322-330 as weights of yp[s .. s + 3]."""
    x2, x3 = x0 * x0, x0 * x0 * x0
    return np.array([
        -x0 / 3.0 + 0.5 * x2 - x3 / 6.0,
        1.0 - 0.5 * x0 - x2 + 0.5 * x3,
        x0 + 0.5 * x2 - 0.5 * x3,
        -x0 / 6.0 + x3 / 6.0,
    ])


def wbar_of(t, p, covpts, T=None):
    theta = so.phase(t, p)
    dx, xp = so.lag_grid(covpts)
    x = np.abs(theta[:, None] - theta[None, :]).reshape(-1)
    inds = np.floor(x / dx).astype("int64")
    x0 = (x - xp[inds + 1]) / dx
    b = spline_weights(x0)
    if T is not None:
        b = b * T.reshape(-1)[None, :]
    w = np.zeros(covpts + 4)
    for k in range(4):
        np.add.at(w, inds + k, b[k])
    return w


def planned_lnlike(L, K, r, d, m, mu, order, baseline_var, sr, sd):
    """everything after the factorisation of B = Sigma + diag(d) (d already divided by c1)"""
    z = m / mu ** 2
    alpha, beta, _, _ = so.alpha_beta(z, order)
    c1 = alpha / mu ** 2
    one = np.ones(K)
    u1 = scipy.linalg.solve_triangular(L, one, lower=True)
    ud = scipy.linalg.solve_triangular(L, d, lower=True)
    y = scipy.linalg.solve_triangular(L, r, lower=True)
    G11, G1d, Gdd = u1 @ u1, u1 @ ud, ud @ ud
    km = K * m
    H1q = (K - G1d) / km
    Hqq = (K * K * m - sd + Gdd) / km ** 2
    Hpp = G11 - 2 * H1q + Hqq
    Hp1 = G11 - H1q
    Hpq = H1q - Hqq
    H = np.array([[Hpp, Hp1, Hpq], [Hp1, G11, H1q], [Hpq, H1q, Hqq]])
    h1 = y @ u1
    hq = (sr - y @ ud) / km
    h = np.array([h1 - hq, h1, hq])
    g = y @ y
    dd = [z * (alpha + beta) / c1, baseline_var / c1, -z * alpha / c1]
    logs = 0.0
    for k in range(3):
        if dd[k] == 0.0:
            continue
        col = H[:, k].copy()
        piv = 1.0 + dd[k] * H[k, k]
        logs += np.log(piv)
        f = dd[k] / piv
        hk = h[k]
        g -= f * hk * hk
        h = h - f * hk * col
        H = H - f * np.outer(col, col)
    logdet = np.sum(np.log(np.diag(L)))
    return -0.5 * g / c1 - (logdet + 0.5 * K * np.log(c1) + 0.5 * logs) - 0.5 * K * np.log(2 * np.pi), z


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    g = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
    keys = list(g.keys())
    mu_y = g["mean_ylm_default"] if "mean_ylm_default" in keys else g[[k for k in keys if "mean" in k][0]]
    cov_y = g["cov_ylm_default"] if "cov_ylm_default" in keys else g[[k for k in keys if "cov" in k][0]]
    worst = 0.0
    for s, tau, vec, bvar in [(0, None, False, 0.0), (1, None, False, 0.0), (5, None, True, 1e-5), (7, 3.0, False, 0.0),
                              (9, 0.7, True, 2e-6)]:
        st = synthetic_star(s, K)
        t, flux, p = st["t"], st["flux"], st["p"]
        proc = so.OracleProcess(mu_y, cov_y, ydeg=15, udeg=2, tau=tau)
        rng = np.random.RandomState(s)
        dv = 1e-6 * (1 + rng.rand(K)) if vec else np.full(K, 1e-6)
        ref = proc.log_likelihood(t, flux, dv if vec else 1e-6, p=p, baseline_var=bvar)
        # the planned form
        mean, cov = proc.flux_mean_cov(t, 60.0, p)
        T = so.Matern32Kernel(t, t, tau) if tau is not None else None
        Sig = cov * T if T is not None else cov
        wbar = wbar_of(t, p, proc.covpts, T)
        m_plan = float(proc.tab["yp"] @ wbar) / K ** 2
        m_sum = float(np.mean(Sig))
        mu = 1.0 + mean
        z = m_plan / mu ** 2
        alpha, beta, _, _ = so.alpha_beta(z, 20)
        c1 = alpha / mu ** 2
        d = dv / c1
        L = scipy.linalg.cholesky(Sig + np.diag(d), lower=True)
        r = flux - 0.0
        val, z = planned_lnlike(L, K, r, d, m_plan, mu, 20, bvar, float(np.sum(r)), float(np.sum(d)))
        rel = abs(val - ref) / abs(ref)
        worst = max(worst, rel)
        print("star %d tau %s vec %d bvar %g: reference %.10f planned %.10f rel %.2e   m: plan %.15e sum %.15e (rel %.1e) z %.3e"
              % (s, tau, vec, bvar, ref, val, rel, m_plan, m_sum, abs(m_plan - m_sum) / abs(m_sum), z))
    print("worst relative difference: %.2e" % worst)
    return 0 if worst < 1e-9 else 1


if __name__ == "__main__":
    sys.exit(main())
