"""
Host-side (NumPy/SciPy) constants of the inclination marginalisation -- the
product's counterpart of ``FluxIntegral._precompute`` (reference
flux.py:107-179), which the reference also runs in Python, once, at
graph-construction time.

PARITY NOTE.  These constants are numerically ill-conditioned at the degrees the
reference targets: the half-angle integrals G span 19 decades and the reference
evaluates the small ones as a difference of O(1) terms (flux.py:115-119), and
the polynomial Wigner coefficients reach 1e8, so Wnp carries ~1e-4 (relative)
of *deterministic* rounding noise at l = 15 that depends on the exact operation
order.  An independently more accurate evaluation (closed-form d-matrices +
incomplete Beta functions) therefore does NOT reproduce the reference's
log-likelihoods to 1e-8.  What is done here instead: the same formulas and the
same per-entry floating-point operation order as the reference (SciPy's
``gamma`` / ``hyp2f1`` for G, the Alvarez-Collado three-term recursion for the
polynomial d-matrices), but organised as whole-row array operations instead of
per-entry Python loops, and with the second-moment contraction reduced to the
m' = 0 rows that actually enter W (flux.py:181-187) instead of the 4-index
tensor Q of flux.py:151-171.  tests/test_host.py (test_marginal_constants_match_reference) checks the outcome
against golden vectors from the executed reference.
"""
import numpy as np
from scipy.special import gamma as _gamma
from scipy.special import hyp2f1 as _hyp2f1

__all__ = ["G_matrix", "wigner_poly", "wigner_poly_rows", "marginal_constants", "lag_grid"]

_cache = {}

def G_matrix(ydeg):
    """G[a, b] = int_0^{pi/2} cos(x/2)^a sin(x/2)^b sin x dx, a, b = 0..4*ydeg,
    in the reference's layout and with the reference's expression
    (flux.py:107-119, 131-137)."""
    n = 4 * ydeg + 1
    i = np.arange(n, dtype=float).reshape(-1, 1)  # cos power (row)
    j = np.arange(n, dtype=float).reshape(1, -1)  # sin power (column)
    return 2 * _gamma(1 + 0.5 * i) * _gamma(1 + 0.5 * j) / _gamma(
        0.5 * (4 + i + j)
    ) - (2 ** (1 - 0.5 * i) / (2 + i)) * _hyp2f1(1 + 0.5 * i, -0.5 * j, 2 + 0.5 * i, 0.5)


def _shift(x, k):
    """Coefficient vectors (last axis) shifted up by k places, zero filled."""
    out = np.zeros_like(x)
    out[..., k:] = x[..., : x.shape[-1] - k]
    return out


def _dpoly_next(l, D1, D2):
    """d^l from d^(l-1) (D1) and d^(l-2) (D2): arrays [m'+l, m+l, i] of
    coefficients over sin(b/2)^(2l-i) cos(b/2)^i (reference wigner.py:192-263,
    same per-entry operation order, one array operation per row)."""
    w = 2 * l + 1
    D = np.zeros((w, w, w))
    pad1 = np.zeros((w - 2, w - 2, w))
    pad1[:, :, : w - 2] = D1  # degree l-1 vectors padded to length 2l+1
    # last row, m' = l
    D[2 * l, 2 * l] = _shift(pad1[2 * l - 2, 2 * l - 2], 2)   # times c^2
    D[2 * l, 0] = pad1[2 * l - 2, 0]                          # times s^2
    for m in range(l - 1, -l, -1):
        v = -np.sqrt((l + m + 1.0) / (l - m)) * D[2 * l, m + 1 + l]
        D[2 * l, m + l] = np.append(v[1:], [0])
    # rows m' = l-1 .. 0, columns |m| <= m'
    if l >= 2:
        pad2 = np.zeros((w - 4, w - 4, w))
        pad2[:, :, : w - 4] = D2
    a = l * (l - 1)
    for mp in range(l - 1, -1, -1):
        m = np.arange(-mp, mp + 1)
        laux, lbux = l + mp, l - mp
        aux = 1.0 / ((l - 1) * np.sqrt(laux * lbux))
        cux = np.sqrt((laux - 1) * (lbux - 1)) * l
        lauz, lbuz = l + m, l - m
        auz = 1.0 / np.sqrt(lauz * lbuz)
        fact = aux * auz
        b = -(m * mp) / a
        x1 = (fact * (2 * l - 1) * a)[:, None] * pad1[mp + l - 1, m + l - 1]
        row = _shift(x1, 2) * (b + 1)[:, None] + x1 * (b - 1)[:, None]
        if lbux != 1:
            sel = lbuz != 1
            ms = m[sel]
            cuz = np.sqrt((lauz[sel] - 1) * (lbuz[sel] - 1))
            x2 = pad2[mp + l - 2, ms + l - 2]
            p2 = (_shift(x2, 4) * 1 + _shift(x2, 2) * 2) + x2 * 1
            row[sel] = row[sel] - (fact[sel] * cux * cuz)[:, None] * p2
        D[mp + l, m + l] = row
    # reflection, then inversion (signs (-1)^(m+m'))
    mm = np.arange(-l, l + 1)
    sgn = np.where((mm[:, None] + mm[None, :]) % 2 == 0, 1.0, -1.0)
    for m in range(1, l + 1):
        mp = np.arange(-m, m)
        D[mp + l, m + l] = sgn[mp + l, m + l][:, None] * D[m + l, mp + l]
    low = (mm[:, None] + mm[None, :]) < 0
    ii, jj = np.nonzero(low)
    D[ii, jj] = sgn[ii, jj][:, None] * D[2 * l - ii, 2 * l - jj]
    return D


def _trig_multiples(c, s, n):
    """cos(k a), sin(k a) for k = 1..n by the reference's angle-addition
    recurrence (wigner.py:289-291); exact for the 0 / +-1 inputs used here."""
    cs, sn = [None], [None]
    ck, sk = c, s
    for _ in range(n):
        cs.append(ck)
        sn.append(sk)
        ck, sk = ck * c - sk * s, sk * c + ck * s
    return cs, sn


def wigner_poly(ydeg, cos_alpha=0, sin_alpha=1, cos_gamma=0, sin_gamma=-1):
    """Polynomial real rotation matrices for the given Euler angles alpha, gamma
    (defaults: the ones flux.py:49-51 and latitude.py:201-203 request; the
    longitude integral uses (1, 0, 1, 0), longitude.py:21-23): list over l of
    arrays [m', m, i] (reference wigner.py:265-372)."""
    key = ("R", ydeg, cos_alpha, sin_alpha, cos_gamma, sin_gamma)
    if key in _cache:
        return _cache[key]
    r2 = np.sqrt(2.0)
    Ds = [np.ones((1, 1, 1))]
    if ydeg >= 1:
        D1 = np.zeros((3, 3, 3))
        D1[2, 2] = [0, 0, 1]
        D1[2, 1] = [0, -r2, 0]
        D1[2, 0] = [1, 0, 0]
        D1[1, 2] = -D1[2, 1]
        D1[1, 1] = D1[2, 2] - D1[2, 0]
        D1[1, 0] = D1[2, 1]
        D1[0, 2] = D1[2, 0]
        D1[0, 1] = D1[1, 2]
        D1[0, 0] = D1[2, 2]
        Ds.append(D1)
    for l in range(2, ydeg + 1):
        Ds.append(_dpoly_next(l, Ds[l - 1], Ds[l - 2]))
    cal, sal = _trig_multiples(cos_alpha, sin_alpha, ydeg)
    cga, sga = _trig_multiples(cos_gamma, sin_gamma, ydeg)
    out = []
    for l, D in enumerate(Ds):
        w = 2 * l + 1
        R = np.zeros((w, w, w))
        R[l, l] = D[l, l]
        for mp in range(1, l + 1):
            ca, sa = cal[mp], sal[mp]
            sg = -1 if mp & 1 else 1
            aux = r2 * D[l, l + mp]
            R[l + mp, l] = aux * ca
            R[l - mp, l] = aux * sa
            m = np.arange(1, l + 1)
            cg = np.array([cga[k] for k in m], dtype=float)[:, None]
            sgm = np.array([sga[k] for k in m], dtype=float)[:, None]
            auxm = r2 * D[l + m, l]
            R[l, l + m] = auxm * cg
            R[l, l - m] = -auxm * sgm
            d1 = D[l - mp, l - m]
            d2 = sg * D[l + mp, l - m]
            cag, cagm = ca * cg - sa * sgm, ca * cg + sa * sgm
            sag, sagm = sa * cg + ca * sgm, sa * cg - ca * sgm
            R[l + mp, l + m] = d1 * cag + d2 * cagm
            R[l + mp, l - m] = -d1 * sag + d2 * sagm
            R[l - mp, l + m] = d1 * sag + d2 * sagm
            R[l - mp, l - m] = d1 * cag - d2 * cagm
        out.append(R)
    _cache[key] = out
    return out


def wigner_poly_rows(ydeg):
    """The m' = 0 rows only: P[l][m + l, i]."""
    return [R[l] for l, R in enumerate(wigner_poly(ydeg))]


def marginal_constants(ydeg):
    """(wnp_packed [NWIG], Wnp [N, N]) -- flux.py:139-179."""
    key = ("W", ydeg)
    if key in _cache:
        return _cache[key]
    G = G_matrix(ydeg)
    Rp = wigner_poly(ydeg)
    N = (ydeg + 1) ** 2
    # first moment: wnp[l] = R[l] @ G[l - m, l + m]
    parts = []
    for l in range(ydeg + 1):
        i = np.arange(2 * l + 1)
        parts.append((Rp[l] @ G[2 * l - i, i]).reshape(-1))
    wnp = np.concatenate(parts)
    # second moment: Wnp[n1, n2] = sum_ab P1[n1, a] G[a + b, 2(l1+l2) - a - b] P2[n2, b]
    P = wigner_poly_rows(ydeg)
    Wnp = np.empty((N, N))
    for l1 in range(ydeg + 1):
        a = np.arange(2 * l1 + 1).reshape(-1, 1)
        for l2 in range(ydeg + 1):
            b = np.arange(2 * l2 + 1).reshape(1, -1)
            Gs = G[a + b, 2 * (l1 + l2) - a - b]
            Wnp[l1 * l1 : (l1 + 1) ** 2, l2 * l2 : (l2 + 1) ** 2] = P[l1] @ Gs @ P[l2].T
    _cache[key] = (wnp, Wnp)
    return _cache[key]


def lag_grid(covpts):
    """dx and the covpts+4 grid points, built exactly like flux.py:311-314 so
    that floor(x / dx) and xp[inds + 1] agree bit for bit with the reference."""
    dx = 2 * np.pi / covpts
    xp = np.arange(-dx, 2 * np.pi + 2.5 * dx, dx)
    return dx, xp
