"""
Upstream of the hot path ON THE DEVICE ("next #1"): hyperparameters
(r, dr, a, b, c, n) -> (mu_y, Sigma_y), evaluated as what the integrals of the
reference ARE -- expectations of rotated spot expansions,

    mu_y    = pi c n  E[ Ry(lambda) Rx(phi) s ]
    Sigma_y = (pi c)^2 n ( E[ (Ry Rx s)(Ry Rx s)^T ] - E[.] E[.]^T ) + diag(eps),

with cos(phi) ~ Beta(alpha, beta) (either sign of phi) and lambda uniform on the
circle (latitude.py:199-212 / longitude.py:19-24: the polynomial Wigner matrices
with Euler angles (pi/2, phi, -pi/2) and (0, lambda, 0) are Rx(phi) and Ry(lambda);
the reference's own tests integrate exactly this numerically, tests/test_latitude.py).

Instead of closed-form moment matrices, eigen-square-roots and polynomial Wigner
tensors with entries up to 1e8 (integrals.py:109-156), the expectations are taken by
EXACT quadrature of actual rotations:
  * every surviving integrand is a polynomial of degree <= 2 ydeg in x = cos(phi)
    (the terms odd in phi cancel between +phi and -phi), so ydeg + 2 Gauss-Jacobi
    nodes for the weight x^(alpha-1) (1-x)^(beta-1) integrate it exactly;
  * in lambda it is a trigonometric polynomial of degree <= 2 ydeg: 2 ydeg + 3
    equispaced angles integrate it exactly.
The (2 (ydeg + 2)) x (2 ydeg + 3) ~ 1100 rotations are applied by the path's own
kernels (sp_Rx, sp_dotRx, sp_tensordotRz; Ry = Rx(-pi/2) Rz Rx(pi/2)), the second
moment is one product on the matrix cores (sp_gemm_nt).  The result stays on the
device for sp_set_ylm_moments_dev.

Numerically this is well conditioned, which the reference's route is not: its Sigma_y
carries rounding noise of 1e-3 max|Sigma_y| in the l >= 12 rows that differs from CPU to
CPU.  Against the reference fixtures the two agree to 2e-12 in mu_y and,
in Sigma_y, to 1e-12 for l <= 4, 1e-10 for l = 8, 1e-8 for l = 12, 1.3e-6 for l = 15 --
the same per-degree profile as reference-vs-reference on two hosts.  ``upstream.py``
(the reference's algorithm, bit-comparable on the same host) stays the default.
"""
import numpy as np

from . import _lib
from .defaults import defaults
from .ops import CheckBoundsOp
from .upstream import ab_to_alphabeta, size_moments

__all__ = ["ylm_moments_device", "ylm_moments_device_grad", "quadrature_nodes", "quadrature_nodes_grad"]


def gauss_jacobi(n, a, b):
    """(nodes, weights summing to 1) of the n-point Gauss-Jacobi rule for (1 - t)^a (1 + t)^b:
    the library's host routine (Golub-Welsch, csrc/sp_host.cpp).  scipy.special.roots_jacobi
    normalises by the zeroth moment, which overflows once a + b exceeds ~1 020 -- a quarter of
    the reference's prior box in b (latitude.py:176-197) -- and costs 0.3 ms per call."""
    t, w = np.empty(n), np.empty(n)
    _lib.check(_lib.lib().sp_gauss_jacobi(int(n), float(a), float(b), _lib.hptr(t), _lib.hptr(w)))
    return t, w


_memo = {}


def _memoised(kind, key, make):
    """The last few results of a host-side preparation step (size moments, quadrature nodes): a gradient evaluation
    asks for the same (r, dr) with six different (a, b) and the same (a, b) with four different r, and each costs
    20 us of NumPy -- a tenth of the device sweep they are meant to hide behind (grad.EnsembleGradient)."""
    d = _memo.setdefault(kind, {})
    v = d.get(key)
    if v is None:
        if len(d) >= 16:
            d.pop(next(iter(d)))
        v = d[key] = make()
    return v


def quadrature_nodes(ydeg, alpha, beta):
    """(phi [P], w_phi [P], lam [Q]): latitude angles with their weights (sum 1) and the
    equispaced longitudes (weight 1 / Q each)."""
    return _memoised("nodes", (int(ydeg), float(alpha), float(beta)), lambda: _quadrature_nodes(ydeg, alpha, beta))


def _quadrature_nodes(ydeg, alpha, beta):
    t, w = gauss_jacobi(ydeg + 2, beta - 1.0, alpha - 1.0)   # weight (1 - t)^(beta-1) (1 + t)^(alpha-1)
    x = 0.5 * (1.0 + t)                                 # cos(phi) in (0, 1)
    phi = np.arccos(x)
    nl = 2 * ydeg + 3
    return (np.concatenate([phi, -phi]), 0.5 * np.concatenate([w, w]),
            2.0 * np.pi * np.arange(nl) / nl)


def gauss_jacobi_grad(n, a, b):
    """(nodes, weights, d nodes/da, d weights/da, d nodes/db, d weights/db) of the same rule: sp_gauss_jacobi_grad."""
    out = [np.empty(n) for _ in range(6)]
    _lib.check(_lib.lib().sp_gauss_jacobi_grad(int(n), float(a), float(b), *[_lib.hptr(x) for x in out]))
    return out


def quadrature_nodes_grad(ydeg, alpha, beta):
    """(phi [P], w_phi [P], dphi [2, P], dw [2, P], lam [Q]): the nodes of ``quadrature_nodes`` with the derivatives
    of the latitude angles and weights with respect to (alpha, beta)."""
    # weight (1 - t)^(beta-1) (1 + t)^(alpha-1): the rule's first exponent is beta's, its second alpha's
    t, w, t_b, w_b, t_a, w_a = gauss_jacobi_grad(ydeg + 2, beta - 1.0, alpha - 1.0)
    x = 0.5 * (1.0 + t)
    phi = np.arccos(x)
    # phi = arccos((1 + t) / 2):  dphi = -dt / (2 sin(phi))
    sn = np.sqrt((1.0 - x) * (1.0 + x))
    dphi = np.stack([-0.5 * t_a / sn, -0.5 * t_b / sn])
    dw = 0.5 * np.stack([w_a, w_b])
    nl = 2 * ydeg + 3
    return (np.concatenate([phi, -phi]), 0.5 * np.concatenate([w, w]), np.concatenate([dphi, -dphi], axis=1),
            np.concatenate([dw, dw], axis=1), 2.0 * np.pi * np.arange(nl) / nl)


def ylm_moments_device_grad(engine, r=defaults["r"], a=defaults["a"], b=defaults["b"], c=defaults["c"],
                            n=defaults["n"], **kwargs):
    """(mu_y [N], Sigma_y [N, N], dmu [3, N], dSigma [3, N, N]): the moments of ``ylm_moments_device`` (one radius,
    dr = None) and their EXACT derivatives with respect to (r [degrees], a, b), as device tensors -- one library
    call, sp_ylm_moments_quadrature_grad (csrc/sp_upstream.hip): the tangents ride through the same rotations.
    The reference differentiates the same integrals analytically (ops/include/latitude.h:21-173 returns d/d alpha,
    d/d beta; tests/test_latitude.py:90-129 checks them).  The derivatives with respect to c and n are closed forms
    of the moments themselves (mu ~ c n, Sigma - eps ~ c^2 n: contrast.py:21-33)."""
    from ._lib import check, hptr

    e = engine
    ydeg, N = e.ydeg, e.N
    n = CheckBoundsOp(name="n", lower=0, upper=np.inf)(n)
    skw = {k: kwargs[k] for k in ("spts", "eps4", "smoothing", "sfac", "cutoff") if k in kwargs}
    s, ds = size_moments(r, None, ydeg, deriv=True, **skw)
    alpha, beta = ab_to_alphabeta(a, b, **kwargs)
    # alpha = exp(a log_alpha_max), beta = exp(log(1/2) + b (log_beta_max - log(1/2))) (latitude.py:176-197);
    # below abmin the parameter is clamped: no dependence
    abmin = kwargs.get("abmin", defaults["abmin"])
    lam_a = kwargs.get("log_alpha_max", defaults["log_alpha_max"])
    lam_b = kwargs.get("log_beta_max", defaults["log_beta_max"]) - np.log(0.5)
    da = lam_a * alpha if a >= abmin else 0.0
    db = lam_b * beta if b >= abmin else 0.0
    phi, wphi, dphi, dw, lam = quadrature_nodes_grad(ydeg, alpha, beta)
    P, Q = phi.shape[0], lam.shape[0]
    mean, cov, dmean, dcov = e.empty(N), e.empty(N, N), e.empty(3, N), e.empty(3, N, N)
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (s, ds, phi, wphi, dphi, dw)]
    check(e._L.sp_ylm_moments_quadrature_grad(
        e._h, *[hptr(x) for x in arrs], int(P), int(Q), float(np.pi * float(c) * np.sqrt(float(n))),
        float(np.sqrt(float(n))), float(kwargs.get("epsy", defaults["epsy"])),
        float(kwargs.get("epsy15", defaults["epsy15"])), e._p(mean), e._p(cov), e._p(dmean), e._p(dcov), e._stream()))
    # (alpha, beta) -> (a, b)
    dmean[1] *= da
    dcov[1] *= da
    dmean[2] *= db
    dcov[2] *= db
    return mean, cov, dmean, dcov


def ylm_moments_device(engine, r=defaults["r"], dr=defaults["dr"], a=defaults["a"],
                       b=defaults["b"], c=defaults["c"], n=defaults["n"], native=True, **kwargs):
    """(mu_y [N], Sigma_y [N, N]) as device tensors of ``engine``'s GPU.

    native (default): the host computes the size moments and the quadrature nodes and hands them to
    ONE library call, sp_ylm_moments_quadrature (csrc/sp_upstream.hip), which enqueues the rotations
    and the two moment products; native=False composes the same computation from the path's ops
    through this module (Rx, dotRx, tensordotRz, gemm_nt) -- 0.48 ms of host time per call against
    0.15, kept as the readable statement of the method and as a cross-check."""
    e = engine
    ydeg, N = e.ydeg, e.N
    n = CheckBoundsOp(name="n", lower=0, upper=np.inf)(n)
    skw = {k: kwargs[k] for k in ("spts", "eps4", "smoothing", "sfac", "cutoff") if k in kwargs}
    s1, eigS = _memoised("size", (float(r), None if dr is None else float(dr), int(ydeg), tuple(sorted(skw.items()))),
                         lambda: size_moments(r, dr, ydeg, **skw))   # first moment [N], factor of the second [N, m]
    alpha, beta = ab_to_alphabeta(a, b, **kwargs)
    phi, wphi, lam = quadrature_nodes(ydeg, alpha, beta)
    P, Q = phi.shape[0], lam.shape[0]
    # vectors to rotate: the size first moment and the columns of the second-moment factor
    # (one and the same vector when dr is None)
    # (size.py:121-134 embeds the (ydeg + 1)-column factor in an N x N matrix of zeros)
    cols = eigS.T[np.abs(eigS).sum(axis=0) > 0.0] if dr is not None else s1[None, :]
    m = cols.shape[0]
    first_is_col = dr is None
    vecs = cols if first_is_col else np.vstack([s1[None, :], cols])      # [mv, N]
    mv = vecs.shape[0]
    if native:
        from ._lib import check, hptr

        vecs_c = np.ascontiguousarray(vecs, dtype=np.float64)
        phi_c, w_c = np.ascontiguousarray(phi), np.ascontiguousarray(wphi)
        mean, cov = e.empty(N), e.empty(N, N)
        check(e._L.sp_ylm_moments_quadrature(
            e._h, hptr(vecs_c), int(mv), int(first_is_col), hptr(phi_c), hptr(w_c), int(P), int(Q),
            float(np.pi * float(c) * np.sqrt(float(n))), float(np.sqrt(float(n))),
            float(kwargs.get("epsy", defaults["epsy"])), float(kwargs.get("epsy15", defaults["epsy15"])),
            e._p(mean), e._p(cov), e._stream()))
        return mean, cov
    # sqrt of the joint weights, with the contrast scale g = pi c sqrt(n) folded in:
    #   Sigma_y = sum (g sqrt(W) row)^T (g sqrt(W) row) - m1 m1^T,   m1 = g mom1,   mu_y = sqrt(n) m1
    g = np.pi * float(c) * np.sqrt(float(n))
    sw = g * np.sqrt(wphi / Q)
    # rows (k, j): g sqrt(W_k) vecs_j, rotated about x by phi_k   (row vectors: v^T R)
    # (two small uploads, the outer product on the device: 2 x 2 KB instead of 70 KB of pageable memory)
    sw_d = e.f64(np.ascontiguousarray(sw))
    M0 = (sw_d[:, None, None] * e.f64(np.ascontiguousarray(vecs))[None, :, :]).contiguous()   # [P, mv, N]
    Rphi, _ = e.Rx(phi, deriv=False)
    V = e.dotRx(M0, Rphi)                                                # [P, mv, N]
    # rotation about y by lambda_q:  Rx(pi/2), Rz(lambda), Rx(-pi/2)
    Rq = getattr(e, "_Rx_quarter_turns", None)          # constant: once per engine
    if Rq is None:
        Rq, _ = e.Rx(np.array([0.5 * np.pi, -0.5 * np.pi]), deriv=False)
        e._Rx_quarter_turns = Rq
    U = e.dotRx(V.reshape(P * mv, N), Rq[0])                             # [P mv, N]
    U = U.unsqueeze(0).expand(Q, P * mv, N).reshape(Q * P * mv, N).contiguous()
    key = (P * mv, Q)
    cache = e.__dict__.setdefault("_lam_rows", {})
    th = cache.get(key)                                   # the longitudes, row by row: constant
    if th is None:
        th = cache[key] = e.f64(np.repeat(lam, P * mv))
    U = e.tensordotRz(U, th)
    A = e.dotRx(U, Rq[1]).reshape(Q, P, mv, N)                           # g sqrt(W) Ry Rx v
    # first moment: sum_k W_k (.) = sum_k sqrt(W_k) (sqrt(W_k) row)
    swd = (sw_d / g).repeat(Q)[None, :]                                  # [1, Q P]: sqrt(W), tiled over lambda
    A1 = A[:, :, 0, :].reshape(Q * P, N).t().contiguous()                # [N, Q P]
    m1 = e.gemm_nt(A1, swd)                                              # [N, 1] = g mom1
    # second moment: sum over rotations and factor columns of row^T row, minus m1 m1^T
    A2 = (A if first_is_col else A[:, :, 1:, :]).reshape(Q * P * m, N).t().contiguous()
    cov = e.gemm_nt(A2, A2)                                              # [N, N]
    e.gemm_nt(m1, m1, C=cov, alpha=-1.0)
    eps = (float(kwargs.get("epsy", defaults["epsy"])), float(kwargs.get("epsy15", defaults["epsy15"])))
    cache = e.__dict__.setdefault("_epsy_diag", {})
    lamd = cache.get(eps)                                 # constant per engine
    if lamd is None:
        lam_h = np.ones(N) * eps[0]
        lam_h[15 ** 2:] = eps[1]
        lamd = cache[eps] = e.f64(lam_h)
    cov.diagonal().add_(lamd)
    mean = m1[:, 0] * float(np.sqrt(float(n)))
    return mean, cov
