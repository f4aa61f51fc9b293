"""
Upstream of the hot path ("next #1"): hyperparameters
(r, dr, a, b, c, n) -> mean and covariance of the spherical-harmonic
coefficients (mu_y, Sigma_y), i.e. the inputs of the GPU log-likelihood path.

Host NumPy / SciPy, like the reference (size.py, latitude.py, longitude.py,
contrast.py, integrals.py, math.py:121-139); the one native piece of the
reference, ``LatitudeIntegralOp`` (ops/include/latitude.h), is the host C++
entry point ``sp_latitude_integrals`` of libsp_hip.so.  One evaluation costs
two symmetric eigendecompositions of an N x N matrix (~20-40 ms at ydeg = 15),
once per hyperparameter sample -- not per star.

Checked against (mu_y, Sigma_y) produced by the executed reference for three
hyperparameter sets (tests/test_upstream.py, tests/golden/moments_L*.npz).
"""
import numpy as np
from scipy.special import gamma
from scipy.special import legendre as _legendre_poly

from . import _lib
from .defaults import defaults
from .hostconst import wigner_poly
from .ops import CheckBoundsOp

__all__ = ["ylm_moments", "gauss2beta", "beta2gauss", "latitude_integrals", "log_jac", "log_jac_samples"]

_cache = {}
_ANG = np.pi / 180


def matrix_sqrt(Q, neig=None, mindiff=1e-15):
    """U with U U^T = Q from the (top-neig) symmetric eigendecomposition
    (math.py:121-139, ops/eigh/eigh.py:11-24)."""
    N = Q.shape[0]
    neig = N if neig is None else neig
    try:
        w, V = np.linalg.eigh(Q)
    except np.linalg.LinAlgError:
        return np.full((N, neig), np.nan)
    w = w[-neig:]
    V = V[:, -neig:]
    sqrtw = np.where(w > mindiff, np.sqrt(np.where(w > mindiff, w, 1.0)), 0.0)
    return V * sqrtw[None, :]


# -- spot size (size.py:9-134) -------------------------------------------------
def _spot_basis(ydeg, spts=1000, eps4=1e-9, smoothing=0.075):
    key = ("spot", ydeg, spts, eps4, smoothing)
    if key not in _cache:
        theta = np.linspace(0, np.pi, spts)
        cost = np.cos(theta)
        B = np.hstack([
            np.sqrt(2 * l + 1) * _legendre_poly(l)(cost).reshape(-1, 1) for l in range(ydeg + 1)
        ])
        A = np.linalg.solve(B.T @ B + eps4 * np.eye(ydeg + 1), B.T)
        l = np.arange(ydeg + 1)
        idx = l * (l + 1)
        S = np.exp(-0.5 * idx * smoothing ** 2)
        _cache[key] = (theta, S[:, None] * A, idx)
    return _cache[key]


def size_moments(r_deg, dr_deg, ydeg, sfac=300, cutoff=1.5, **kw):
    """(q, eigQ): first moment (N) and square root of the second moment of the
    spot-size integral (size.py:92-134)."""
    theta, Bp, idx = _spot_basis(ydeg, **{k: kw[k] for k in ("spts", "eps4", "smoothing") if k in kw})
    N = (ydeg + 1) ** 2
    r = CheckBoundsOp(name="r", lower=0, upper=0.5 * np.pi)(r_deg * _ANG)
    if dr_deg is None:
        b = 1 / (1 + np.exp(-sfac * (theta - r))) - 1
        q = np.zeros(N)
        q[idx] = Bp @ b
        if kw.get("deriv"):
            # d/dr [degrees] of the sigmoid profile (size.py:92-101), for the exact gradient (upstream_device.py)
            sg = b + 1.0
            dq = np.zeros(N)
            dq[idx] = Bp @ (-sfac * sg * (1.0 - sg)) * _ANG
            return q, dq
        return q, q.reshape(-1, 1)
    dr = CheckBoundsOp(name="dr", lower=0, upper=0.5 * np.pi)(dr_deg * _ANG)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        chim = np.exp(sfac * (r - dr - theta))
        chip = np.exp(sfac * (r + dr - theta))
        c = 1.0 / (2 * dr * sfac) * np.log((1 + chim) / (1 + chip))
        e = np.zeros(N)
        e[idx] = Bp @ c
        kmax = int(np.argmax(theta / (r + dr) > cutoff))
        t = theta[:kmax].reshape(1, -1)
        chim = np.exp(sfac * (r - dr - t))
        chip = np.exp(sfac * (r + dr - t))
        ex = np.exp(sfac * (t - t.T))
        term = np.log(1 + chim) - np.log(1 + chip)
        C0 = (ex * term - term.T) / (1 - ex + 1.0e-15)
        k = np.arange(kmax)
        C0[k, k] = (1 / (1 + chip) + chim / (1 + chim) - term - 1).reshape(-1)
        C0 /= 2 * dr * sfac
    C = np.zeros((theta.shape[0], theta.shape[0]))
    C[:kmax, :kmax] = C0
    Etilde = Bp @ C @ Bp.T
    eigEtilde = matrix_sqrt(Etilde)
    eigE = np.zeros((N, N))
    eigE[np.ix_(idx, idx)] = eigEtilde
    return e, eigE


# -- Wigner integrals (integrals.py:109-156) ---------------------------------------
def _wigner_operators(ydeg, q, Q, Rp):
    U = matrix_sqrt(Q, neig=2 * ydeg + 1)
    t, T = [], []
    for l in range(ydeg + 1):
        blk = slice(l * l, (l + 1) ** 2)
        t.append(Rp[l] @ q[blk])
        T.append(np.swapaxes(Rp[l] @ U[blk], 1, 2))
    return t, T


def _first_moment(ydeg, t, e):
    mu = np.zeros((ydeg + 1) ** 2)
    for l in range(ydeg + 1):
        blk = slice(l * l, (l + 1) ** 2)
        mu[blk] = t[l] @ e[blk]
    return mu


def _second_moment(ydeg, T, eigE):
    N = (ydeg + 1) ** 2
    neig = T[0].shape[1]
    sqrtC = np.zeros((N, neig, eigE.shape[-1]))
    for l in range(ydeg + 1):
        blk = slice(l * l, (l + 1) ** 2)
        sqrtC[blk] = np.dot(T[l], eigE[blk])
    sqrtC = sqrtC.reshape(N, -1)
    if sqrtC.shape[1] > N:
        sqrtC = matrix_sqrt(sqrtC @ sqrtC.T)
    return sqrtC


def latitude_integrals(ydeg, alpha, beta):
    """LatitudeIntegralOp values q (N), Q (N, N) (ops/include/latitude.h:21-173)."""
    N = (ydeg + 1) ** 2
    q = np.empty(N)
    Q = np.empty((N, N))
    _lib.check(_lib.lib().sp_latitude_integrals(int(ydeg), float(alpha), float(beta),
                                                _lib.hptr(q), _lib.hptr(Q)))
    return q, Q


def _longitude_integrals(ydeg):
    """longitude.py:25-52 (constants of ydeg)."""
    key = ("lon", ydeg)
    if key not in _cache:
        n = 4 * ydeg + 1
        i = np.arange(n).reshape(-1, 1)
        j = np.arange(0, n, 2).reshape(1, -1)
        term = np.zeros((n, n))
        term[:, ::2] = gamma(0.5 * (i + 1)) * gamma(0.5 * (j + 1)) / gamma(0.5 * (2 + i + j))
        term /= np.pi
        l = np.floor(np.sqrt(np.arange((ydeg + 1) ** 2))).astype(int)
        m = np.arange((ydeg + 1) ** 2) - l * l - l
        j1, i1 = m + l, l - m
        q = term[j1, i1]
        Q = term[j1[:, None] + j1[None, :], i1[:, None] + i1[None, :]]
        _cache[key] = (q, Q)
    return _cache[key]


def ab_to_alphabeta(a, b, **kwargs):
    """latitude.py:176-197."""
    abmin = kwargs.get("abmin", defaults["abmin"])
    a = CheckBoundsOp(name="a", lower=0, upper=1)(a)
    b = CheckBoundsOp(name="b", lower=0, upper=1)(b)
    a = abmin if a < abmin else a
    b = abmin if b < abmin else b
    lam = kwargs.get("log_alpha_max", defaults["log_alpha_max"])
    lbm = kwargs.get("log_beta_max", defaults["log_beta_max"])
    return np.exp(a * lam), np.exp(np.log(0.5) + b * (lbm - np.log(0.5)))


def ylm_moments(r=defaults["r"], dr=defaults["dr"], a=defaults["a"], b=defaults["b"],
                c=defaults["c"], n=defaults["n"], ydeg=defaults["ydeg"], **kwargs):
    """(mu_y, Sigma_y) exactly as StarryProcess.__init__ builds them
    (sp.py:257-266; contrast.py:18-33)."""
    N = (ydeg + 1) ** 2
    n = CheckBoundsOp(name="n", lower=0, upper=np.inf)(n)
    skw = {k: kwargs[k] for k in ("spts", "eps4", "smoothing", "sfac", "cutoff") if k in kwargs}
    e, eigE = size_moments(r, dr, ydeg, **skw)
    # latitude
    alpha, beta = ab_to_alphabeta(a, b, **kwargs)
    q, Q = latitude_integrals(ydeg, alpha, beta)
    t, T = _wigner_operators(ydeg, q, Q, wigner_poly(ydeg, 0, 1, 0, -1))
    e = _first_moment(ydeg, t, e)
    eigE = _second_moment(ydeg, T, eigE)
    # longitude
    key = ("lonops", ydeg)
    if key not in _cache:
        ql, Ql = _longitude_integrals(ydeg)
        _cache[key] = _wigner_operators(ydeg, ql, Ql, wigner_poly(ydeg, 1, 0, 1, 0))
    t, T = _cache[key]
    mom1 = _first_moment(ydeg, t, e)
    eig2 = _second_moment(ydeg, T, eigE)
    mom2 = eig2 @ eig2.T
    mean = np.pi * c * n * mom1
    cov = (np.pi * c) ** 2 * n * (mom2 - np.outer(mom1, mom1))
    lam = np.ones(N) * kwargs.get("epsy", defaults["epsy"])
    lam[15 ** 2:] = kwargs.get("epsy15", defaults["epsy15"])
    return mean, cov + np.diag(lam)


# -- latitude parametrisations (latitude.py:13-167, 281-316) -------------------------
def gauss2beta(mu, sigma, log_alpha_max=defaults["log_alpha_max"], log_beta_max=defaults["log_beta_max"]):
    is_vector = hasattr(mu, "__len__")
    m = np.atleast_1d(mu) * np.pi / 180
    v = (np.atleast_1d(sigma) * np.pi / 180) ** 2
    c1, c2, c3 = np.cos(m), np.cos(2 * m), np.cos(3 * m)
    term = 1.0 / (16 * v * np.cos(0.5 * m) ** 4)
    alpha = (2 + 4 * v + (3 + 8 * v) * c1 + 2 * c2 + c3) * term
    beta = (c1 + 2 * v * (3 + c2) - c3) * term
    a = np.log(alpha) / log_alpha_max
    b = np.maximum(0.0, (np.log(beta) - np.log(0.5)) / (log_beta_max - np.log(0.5)))
    return (a, b) if is_vector else (a[0], b[0])


def _mu_sigma(alpha, beta):
    term = 4 * alpha ** 2 - 8 * alpha - 6 * beta + 4 * alpha * beta + beta ** 2 + 5
    mu = 2 * np.arctan(np.sqrt(2 * alpha + beta - 2 - np.sqrt(term)))
    term = 1 - alpha + beta + (beta - 1) * np.cos(mu) + (alpha - 1) / np.cos(mu) ** 2
    return mu, np.sin(mu) / np.sqrt(term)


def beta2gauss(a, b, log_alpha_max=defaults["log_alpha_max"], log_beta_max=defaults["log_beta_max"]):
    is_vector = hasattr(a, "__len__")
    alpha = np.atleast_1d(np.exp(np.asarray(a) * log_alpha_max))
    beta = np.atleast_1d(np.exp(np.log(0.5) + np.asarray(b) * (log_beta_max - np.log(0.5))))
    with np.errstate(invalid="ignore"):
        mu, sigma = _mu_sigma(alpha, beta)
    bad = (alpha <= 1) | (beta <= 0.5)
    mu = np.where(bad, np.nan, mu)
    sigma = np.where(bad, np.nan, sigma)
    mu, sigma = mu / (np.pi / 180), sigma / (np.pi / 180)
    return (mu, sigma) if is_vector else (mu[0], sigma[0])


def log_jac(a, b, **kwargs):
    """Log |Jacobian| of (a, b) -> (mu, sigma) (latitude.py:281-316)."""
    alpha, beta = ab_to_alphabeta(a, b, **kwargs)
    sigma_max = kwargs.get("sigma_max", defaults["sigma_max"]) * _ANG
    with np.errstate(invalid="ignore", divide="ignore"):
        mu, sigma = _mu_sigma(alpha, beta)
        val = np.log(np.abs(
            (alpha * beta * (1 + np.cos(mu)) ** 3 * np.sin(2 * mu) ** 3)
            / (sigma
               * (-3 + 2 * alpha + beta + (-1 + 2 * alpha + beta) * np.cos(mu))
               * (2 * (-1 + alpha + beta) + 3 * (-1 + beta) * np.cos(mu)
                  - 2 * (-1 + alpha - beta) * np.cos(2 * mu) + (-1 + beta) * np.cos(3 * mu)) ** 2)))
    return float(-np.inf) if sigma > sigma_max else float(val)


def log_jac_samples(a, b, **kwargs):
    """``log_jac`` for arrays of (a, b): one NumPy pass for the samples of a batch (a Python call per sample costs more
    than the sample's share of a batched likelihood step)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    CheckBoundsOp(name="a", lower=0, upper=1)(a)
    CheckBoundsOp(name="b", lower=0, upper=1)(b)
    abmin = kwargs.get("abmin", defaults["abmin"])
    lam = kwargs.get("log_alpha_max", defaults["log_alpha_max"])
    lbm = kwargs.get("log_beta_max", defaults["log_beta_max"])
    alpha = np.exp(np.maximum(a, abmin) * lam)
    beta = np.exp(np.log(0.5) + np.maximum(b, abmin) * (lbm - np.log(0.5)))
    sigma_max = kwargs.get("sigma_max", defaults["sigma_max"]) * _ANG
    with np.errstate(invalid="ignore", divide="ignore"):
        mu, sigma = _mu_sigma(alpha, beta)
        val = np.log(np.abs(
            (alpha * beta * (1 + np.cos(mu)) ** 3 * np.sin(2 * mu) ** 3)
            / (sigma
               * (-3 + 2 * alpha + beta + (-1 + 2 * alpha + beta) * np.cos(mu))
               * (2 * (-1 + alpha + beta) + 3 * (-1 + beta) * np.cos(mu)
                  - 2 * (-1 + alpha - beta) * np.cos(2 * mu) + (-1 + beta) * np.cos(3 * mu)) ** 2)))
    return np.where(sigma > sigma_max, -np.inf, val)
