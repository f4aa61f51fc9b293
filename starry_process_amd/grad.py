"""
End-to-end gradient of the log-likelihood (the part beyond the single ops): what
the reference obtains by letting Theano chain the ``grad`` / ``L_op`` methods of its Ops through the
graph of ``StarryProcess.log_likelihood`` (sp.py:1052-1188; verified by finite differences in the
reference's tests/test_lnlike.py:100-136).

Here the graph is ``torch.autograd`` over fp64 tensors on the GPU, and its heavy nodes are the
library's own reverse-mode kernels:

  * ``lnL(C, r)``                the factorisation ``sp_cho_factor`` forward; backward in closed form,
                                 ``C_bar = 1/2 (alpha alpha^T - C^-1)``, ``r_bar = -alpha`` with
                                 ``alpha = C^-1 r`` (what chaining ``Cholesky.L_op`` and ``Solve.L_op``,
                                 math.py:40-100, gives; tests/test_gpu_linalg_rev.py shows the two agree),
                                 both solves on the device (``sp_cho_solve``);
  * ``special_tensordotRz``      backward ``sp_special_tensordotRz_rev`` (wigner.h:464-531);
  * the glue between them -- rotation of the moments to the polar frame (flux.py:54-62), mean / variance
    (flux.py:297-308), spline build (flux.py:310-330), the K x K gather interpolation (flux.py:256-276),
    the normalisation correction with its alpha(z), beta(z) series (sp.py:705-727, ops/norm/norm.py:26-44),
    noise and baseline (sp.py:1135-1151) -- is elementwise / small-matrix torch arithmetic that autograd
    differentiates by itself.

``log_likelihood_with_grad`` returns d lnL / d(mu_y, Sigma_y): the gradient with respect to the hot path's
own inputs.  ``hyper_gradient`` takes it on to (r, a, b, c, n): c and n enter the moments as plain scale
factors (contrast.py:21-33), r, a, b through the upstream integrals, whose directional derivatives are
taken by central differences of the device quadrature (upstream_device.py: smooth and accurate to
rounding in its parameters, 0.3 ms per evaluation) -- one reverse sweep through the expensive part, six
cheap upstream evaluations, instead of ten full likelihoods for finite differences of everything.

One star, the marginalised branch (with or without normalisation); this is a diagnostic / optimisation
aid (``mci.optimize()``-style callers), not part of the timed hot path.
"""
import numpy as np

from .defaults import defaults
from .engine import get_engine
from .temporal import kernel_id

__all__ = ["EnsembleGradient", "ensemble_gradient", "log_likelihood_with_grad", "hyper_gradient"]

_cache = {}


def _torch():
    import torch

    return torch


def _constants(e, u):
    """Device constants for limb darkening ``u``: dense blockdiag Rx(pi/2), w, W (flux.py:181-231), rTA1L(u)."""
    from . import _lib, hostconst

    torch = _torch()
    key = (e.ydeg, e.device_index, tuple(np.asarray(u, dtype=float).reshape(-1)))
    if key in _cache:
        return _cache[key]
    ydeg, N = e.ydeg, e.N
    Rpk = e.Rx(np.array([0.5 * np.pi]), deriv=False)[0][0].cpu().numpy()
    D = np.zeros((N, N))
    off = 0
    for l in range(ydeg + 1):
        w_ = 2 * l + 1
        D[l * l:(l + 1) ** 2, l * l:(l + 1) ** 2] = Rpk[off:off + w_ * w_].reshape(w_, w_)
        off += w_ * w_
    wnp, Wnp = hostconst.marginal_constants(ydeg)
    rta1 = np.asarray(e.rTA1L(np.asarray(u, dtype=float).reshape(-1)[: e.udeg])).reshape(-1)
    idx = _lib.index_tables(ydeg)
    rho = rta1[idx["m0"]][idx["l_of"]]
    W = Wnp * rho[:, None] * rho[None, :]
    # w (N): w[l-block] = rTA1[l-block] . wnp[l]  (flux.py:196-198): mean = sum_n w[n] ez[n]
    wv = np.zeros(N)
    off = 0
    for l in range(ydeg + 1):
        w_ = 2 * l + 1
        wv[l * l:(l + 1) ** 2] = rta1[l * l:(l + 1) ** 2] @ wnp[off:off + w_ * w_].reshape(w_, w_)
        off += w_ * w_
    _cache[key] = (e.f64(D), e.f64(wv), e.f64(W), e.f64(rta1))
    return _cache[key]


def _functions(e):
    torch = _torch()

    class SpecialTensordotRz(torch.autograd.Function):
        @staticmethod
        def forward(ctx, W, M, theta):
            ctx.save_for_backward(W, M, theta)
            return e.special_tensordotRz(W, M.contiguous(), theta)

        @staticmethod
        def backward(ctx, bf):
            W, M, theta = ctx.saved_tensors
            bM, _ = e.special_tensordotRz_rev(W, M.contiguous(), theta, bf.contiguous())
            return None, bM, None

    class GaussianLogLike(torch.autograd.Function):
        """-1/2 r^T C^-1 r - 1/2 log det C - K/2 log 2 pi; -inf when C is not positive definite."""

        @staticmethod
        def forward(ctx, C, r):
            K = C.shape[0]
            L, info = e.cho_factor(C.contiguous())
            if int(info.max().item()) != 0:
                ctx.failed = True
                return C.new_tensor(-float("inf"))
            ctx.failed = False
            alpha = e.cho_solve(L, r.reshape(K, 1).contiguous()).reshape(K)
            ctx.save_for_backward(L, alpha)
            return -0.5 * torch.dot(r, alpha) - torch.log(torch.diagonal(L)).sum() - 0.5 * K * np.log(2 * np.pi)

        @staticmethod
        def backward(ctx, g):
            if ctx.failed:
                return None, None
            L, alpha = ctx.saved_tensors
            K = L.shape[0]
            Cinv = e.cho_solve(L, torch.eye(K, dtype=L.dtype, device=L.device))
            return g * 0.5 * (torch.outer(alpha, alpha) - Cinv), -g * alpha

    class DesignMatrix(torch.autograd.Function):
        """A(theta, inc) = ((1_K x rTA1) Rx(-inc)) Rz(theta) Rx(pi/2)  (flux.py:88-105, 278-281); backward with
        the library's reverse rotation (tensordotRz_rev) and the derivative matrices of sp_Rx."""

        @staticmethod
        def forward(ctx, theta, inc, rta1):
            K = theta.shape[0]
            R, dR = e.Rx(np.array([-float(inc.item()), 0.5 * np.pi, -0.5 * np.pi]), deriv=True)
            M0 = rta1.reshape(1, -1).expand(K, -1).contiguous()
            M1 = e.dotRx(M0, R[0])
            A = e.dotRx(e.tensordotRz(M1, theta.contiguous()), R[1])
            ctx.save_for_backward(theta, M0, M1, R, dR)
            return A

        @staticmethod
        def backward(ctx, bA):
            theta, M0, M1, R, dR = ctx.saved_tensors
            bM2 = e.dotRx(bA.contiguous(), R[2])          # Rx(pi/2)^T = Rx(-pi/2)
            bM1, btheta = e.tensordotRz_rev(M1, theta.contiguous(), bM2)
            # M1 = M0 Rx(-inc): d/d inc = -M0 Rx'(-inc)
            binc = -(bM1 * e.dotRx(M0, dR[0])).sum().reshape(1)
            return btheta, binc, None

    return SpecialTensordotRz, GaussianLogLike, DesignMatrix


def _alpha_beta(z, order):
    fac = z * 0.0 + 1.0
    alpha, beta = z * 0.0, z * 0.0
    for n in range(order + 1):
        alpha = alpha + fac
        beta = beta + 2 * n * fac
        fac = fac * z * (2 * n + 3)
    return alpha, beta


def log_likelihood_with_grad(mean_ylm, cov_ylm, t, flux, data_var, i=defaults["i"], p=defaults["p"], u=None,
                             tau=None, temporal_kernel="matern32", baseline_mean=0.0, baseline_var=0.0,
                             marginalize_over_inclination=True, normalized=True, covpts=defaults["covpts"],
                             ydeg=defaults["ydeg"], udeg=defaults["udeg"],
                             norm_order=defaults["normalization_order"],
                             zmax=defaults["normalization_zmax"], device=None):
    """(lnL, grads) for ONE light curve; NumPy in, NumPy out.  ``grads`` holds d lnL / d of
    "mean_ylm" [N], "cov_ylm" [N, N] (its N^2 entries treated as independent: symmetrise it for a symmetric
    perturbation), "p", "tau" (when a timescale is given) and, on the conditional branch, "i" (per degree)."""
    torch = _torch()
    e = get_engine(ydeg, udeg, device)
    u = np.zeros(e.udeg) if u is None else np.asarray(u, dtype=float).reshape(-1)[: e.udeg]
    D, wv, W, rta1 = _constants(e, u)
    Special, LogLike, Design = _functions(e)
    t = np.asarray(t, dtype=np.float64).reshape(-1)
    K = t.shape[0]
    td = e.f64(t)
    leaf = lambda v: e.f64(np.atleast_1d(np.asarray(v, dtype=np.float64))).clone().requires_grad_(True)  # noqa: E731
    mu = leaf(np.asarray(mean_ylm).reshape(-1))
    Sig = e.f64(np.asarray(cov_ylm, dtype=np.float64)).clone().requires_grad_(True)
    per, inc = leaf(p), leaf(i)
    # theta = 2 pi mod(t / p, 1) (flux.py:262, 279): the integer part is data
    tp = td / per
    theta = 2 * np.pi * (tp - torch.floor(tp.detach()))
    if marginalize_over_inclination:
        # polar frame (flux.py:54-62)
        ez = D.t() @ mu
        Ez = D.t() @ (Sig + torch.outer(mu, mu)) @ D
        mean = torch.dot(wv, ez)
        # kernel on the lag grid and its cubic coefficients (flux.py:297-330)
        dx = 2 * np.pi / covpts
        xp = np.arange(-dx, 2 * np.pi + 2.5 * dx, dx)
        yp = Special.apply(W, Ez, e.f64(xp)) - mean ** 2
        y0, y1, y2, y3 = yp[:-3], yp[1:-2], yp[2:-1], yp[3:]
        a0 = y1
        a1 = -y0 / 3.0 - 0.5 * y1 + y2 - y3 / 6.0
        a2 = 0.5 * (y0 + y2) - y1
        a3 = 0.5 * ((y1 - y2) + (y3 - y0) / 3.0)
        # K x K interpolation (flux.py:256-276); the interval index is data
        if K == 1:
            cov = (torch.sum(W * Ez) - mean ** 2).reshape(1, 1)
        else:
            x = torch.abs(theta[:, None] - theta[None, :]).reshape(-1)
            ii = torch.floor(x.detach() / dx).to(torch.int64)
            x0 = (x - e.f64(xp)[ii + 1]) / dx
            cov = (a0[ii] + a1[ii] * x0 + a2[ii] * x0 ** 2 + a3[ii] * x0 ** 3).reshape(K, K)
    else:
        # A = ((1_K x rTA1) Rx(-i)) Rz(theta) Rx(pi/2); mean = (A mu)_0, cov = A Sigma A^T (flux.py:278-281, 337-343)
        A = Design.apply(theta, inc * (np.pi / 180.0), rta1)
        mean = (A @ mu)[0]
        cov = A @ Sig @ A.t()
    tau_leaf = None
    if tau is not None:
        # temporal.py:8-16
        tau_leaf = leaf(tau)
        dt = torch.abs(td[:, None] - td[None, :])
        if temporal_kernel == "matern32":
            xx = np.sqrt(3.0) * dt / tau_leaf
            cov = cov * ((1 + xx) * torch.exp(-xx))
        elif temporal_kernel == "expsquared":
            cov = cov * torch.exp(-(dt ** 2) / (2 * tau_leaf))
        else:
            raise ValueError("temporal_kernel must be 'matern32' or 'expsquared'")
    gp_mean = mean
    zval = 0.0
    if normalized:
        # sp.py:705-727 with mu = 1 + mean; the GP mean of the normalised process is zero (sp.py:669-670)
        mu1 = 1.0 + mean
        m = cov.mean()
        q = cov.sum(dim=1) / (K * m)
        z = m / mu1 ** 2
        pp = 1.0 - q
        alpha, beta = _alpha_beta(z, int(norm_order))
        cov = (alpha / mu1 ** 2) * cov + z * ((alpha + beta) * torch.outer(pp, pp) - alpha * torch.outer(q, q))
        gp_mean = mean * 0.0
        zval = float(z.item())
    dv = np.asarray(data_var, dtype=np.float64)
    noise = e.f64(np.full(K, float(dv)) if dv.ndim == 0 else dv.reshape(-1))
    C = cov + torch.diag(noise) + float(baseline_var)
    r = e.f64(np.asarray(flux, dtype=np.float64).reshape(-1)) - (gp_mean + float(baseline_mean))
    lnl = LogLike.apply(C, r)
    leaves = {"mean_ylm": mu, "cov_ylm": Sig, "p": per}
    if not marginalize_over_inclination:
        leaves["i"] = inc
    if tau_leaf is not None:
        leaves["tau"] = tau_leaf
    if not bool(torch.isfinite(lnl)) or (normalized and zval > zmax):
        return -np.inf, {k: np.zeros(tuple(v.shape)) if v.numel() > 1 else 0.0 for k, v in leaves.items()}
    lnl.backward()
    grads = {}
    for k, v in leaves.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        grads[k] = g.cpu().numpy() if v.numel() > 1 else float(g.item())
    return float(lnl.item()), grads


def hyper_gradient(t, flux, data_var, r=defaults["r"], dr=defaults["dr"], a=defaults["a"], b=defaults["b"],
                   c=defaults["c"], n=defaults["n"], h=1e-4, upstream_kwargs=None, moments0=None, exact=True,
                   **kwargs):
    """(lnL, {"r": ., "a": ., "b": ., "c": ., "n": ., "p": ., ...}): the log-likelihood of one light curve and its
    gradient with respect to the spot hyperparameters (moments by the device quadrature, upstream_device.py;
    "dr" too when a spread of radii is given) and whatever else ``log_likelihood_with_grad`` differentiates
    (p; i on the conditional branch; tau).
    kwargs: as for ``log_likelihood_with_grad`` (i, p, u, tau, normalized, marginalize_over_inclination, ...);
    moments0: (mu_y, Sigma_y) the value and the moment gradient are taken at, when they are not the device
    quadrature's own (a process built with upstream="reference": same integrals, the reference's rounding);
    upstream_kwargs: the constructor's numerical keywords of the moment integrals (epsy, epsy15, sfac, ...).
    r, a and b: with one radius (dr = None) the moments' EXACT tangents (the quadrature rule differentiated with
    respect to its exponents, the sigmoid profile with respect to its radius: ylm_moments_device_grad -- what the
    reference's analytic latitude derivatives are, ops/include/latitude.h:21-173); with a spread of radii, or
    exact=False, central differences of the moments, one-sided within a step of the bounds [0, 1]
    (ops/exceptions.py:30-48 raises outside); c = 0 or n = 0 (no spots: Sigma_y = diag(eps), mu_y = 0)
    has a zero gradient in the other of the two and is returned as such."""
    from .upstream_device import ylm_moments_device

    ydeg = kwargs.get("ydeg", defaults["ydeg"])
    e = get_engine(ydeg, kwargs.get("udeg", defaults["udeg"]), kwargs.get("device"))

    ukw = dict(upstream_kwargs or {})

    def moments(r_, dr_, a_, b_, c_=c, n_=n):
        mu, Sig = ylm_moments_device(e, r=r_, dr=dr_, a=a_, b=b_, c=c_, n=n_, **ukw)
        return mu.cpu().numpy(), Sig.cpu().numpy()

    mu, Sig = moments(r, dr, a, b)
    mu_at, Sig_at = (mu, Sig) if moments0 is None else (np.asarray(moments0[0]), np.asarray(moments0[1]))
    lnl, g = log_likelihood_with_grad(mu_at, Sig_at, t, flux, data_var, **kwargs)
    gmu, gSig = g["mean_ylm"], g["cov_ylm"]
    N = mu.shape[0]
    eps = np.ones(N) * float(ukw.get("epsy", defaults["epsy"]))
    eps[15 ** 2:] = float(ukw.get("epsy15", defaults["epsy15"]))
    S0 = Sig - np.diag(eps)                       # the part of Sigma_y that scales with c and n
    out = {k: v for k, v in g.items() if k not in ("mean_ylm", "cov_ylm")}
    if c != 0 and n != 0:
        # mu ~ c n, Sigma ~ c^2 n (contrast.py:21-33)
        out.update({"c": float(gmu @ mu / c + 2.0 * np.sum(gSig * S0) / c),
                    "n": float(gmu @ mu / n + np.sum(gSig * S0) / n)})
    else:
        # at c = 0 (n = 0) the moments are linear (quadratic) in the vanishing parameter: take the
        # derivative from the moments at unit value of it
        m1, S1 = moments(r, dr, a, b, c_=c if c != 0 else 1.0, n_=n if n != 0 else 1.0)
        S1 = S1 - np.diag(eps)
        out.update({"c": float(gmu @ m1) if (c == 0 and n != 0) else 0.0,     # d mu / dc = mu(c = 1); d Sigma / dc = 0
                    "n": float(gmu @ m1 + np.sum(gSig * S1)) if (n == 0 and c != 0) else 0.0})
    x0 = {"r": r, "dr": dr, "a": a, "b": b}
    bounds = {"r": (0.0, 90.0), "dr": (0.0, 90.0), "a": (0.0, 1.0), "b": (0.0, 1.0)}
    if dr is None and exact:
        # one radius: the moments' exact tangents (upstream_device.ylm_moments_device_grad)
        from .upstream_device import ylm_moments_device_grad

        _, _, dmu, dSig = ylm_moments_device_grad(e, r=r, a=a, b=b, c=c, n=n, **ukw)
        dmu, dSig = dmu.cpu().numpy(), dSig.cpu().numpy()
        for k, name in enumerate(("r", "a", "b")):
            out[name] = float(gmu @ dmu[k] + np.sum(gSig * dSig[k]))
        return lnl, out
    for name in ("r", "dr", "a", "b"):
        if x0[name] is None:
            continue
        x = float(x0[name])
        step = h * max(abs(x), 0.1)
        lo_b, hi_b = bounds[name]
        xl, xh = max(x - step, lo_b), min(x + step, hi_b)       # one-sided within a step of a bound
        lo, hi = dict(x0), dict(x0)
        lo[name], hi[name] = xl, xh
        m0, S0_ = (mu, Sig) if xl == x else moments(lo["r"], lo["dr"], lo["a"], lo["b"])
        m1, S1_ = (mu, Sig) if xh == x else moments(hi["r"], hi["dr"], hi["a"], hi["b"])
        out[name] = float(gmu @ ((m1 - m0) / (xh - xl)) + np.sum(gSig * ((S1_ - S0_) / (xh - xl))))
    return lnl, out


class EnsembleGradient(object):
    """Log-likelihood of an ENSEMBLE of light curves and its gradient with respect to the spot hyperparameters
    (r, a, b, c, n[, dr]) in ONE device sweep per evaluation -- what ``theano.grad`` of the summed
    ``sp.log_likelihood`` is in the reference (tests/test_lnlike.py:100-136, calibrate/log_prob.py:53-91), for every
    star of the batch at once.  Marginal branch, scalar or per-cadence data variance; one light curve per star
    (flux [S, K]) or M of them on the star's one covariance (flux [S, M, K]: the shared-covariance form of
    sp.py:1162-1171 -- with S = 1 the gradient of what ``calibrate.get_log_prob`` evaluates).

        eg = EnsembleGradient(t, flux, ferr=1e-3, p=periods)       # data -> GPU, once
        lnl, grad = eg(r=20., a=.4, b=.27, c=.1, n=10.)             # lnl: sum over stars; grad: dict
        eg.lnlike                                                   # per-star values of the last call

    How: d lnL / dC = (alpha alpha^T - C^-1) / 2 with C^-1 from the factorisation's own machinery
    (sp_spd_inverse_batched: the identity rides through the blocked Cholesky), pulled back on the device through the
    normalisation and the cubic interpolation to the adjoint of each star's kernel TABLE (304 numbers) and flux mean
    (sp_lnlike_grad_marginal).  The chain from the hyperparameters to the table is short and cheap -- moments by the
    device quadrature, then the table kernels -- and is differentiated there: exactly in c and n (the moments are
    mu_y = c n m, Sigma_y = c^2 n S + eps, contrast.py:21-33, and the table is linear in Sigma_y + mu_y mu_y^T), and
    since round 5 exactly in r, a, b too: the moments come with their tangents (ylm_moments_device_grad: the
    quadrature rule differentiated with respect to its exponents -- the reference's analytic latitude derivatives,
    ops/include/latitude.h:21-173), and one table evaluation per parameter turns a tangent of the moments into the
    tangent of the table, on two more streams while the sweep runs.  With a spread of radii (dr) or exact=False:
    central differences of the table (step h), as in round 4."""

    def __init__(self, t, flux, ferr=1.0e-3, p=1.0, u=None, ydeg=15, baseline_var=0.0, baseline_mean=0.0,
                 normalized=True, covpts=None, tau=None, temporal_kernel="matern32", device=None, h=1.0e-4,
                 upstream_kwargs=None, exact=True):
        import torch

        from .engine import engine_slots, make_stars

        flux = np.asarray(flux, dtype=np.float64)
        if flux.ndim not in (2, 3):
            raise ValueError("flux must be (S, K) or (S, M, K)")
        S, K = flux.shape[0], flux.shape[-1]
        M = flux.shape[1] if flux.ndim == 3 else 1
        if K < 2:
            raise ValueError("at least two cadences")
        t = np.asarray(t, dtype=np.float64)
        t = np.broadcast_to(t, (S, K)) if t.ndim == 1 else t
        udeg = defaults["udeg"]
        per = lambda x: np.broadcast_to(np.asarray(x, dtype=np.float64), (S,))
        uu = np.asarray(defaults["u"][:udeg] if u is None else u, dtype=np.float64)
        if uu.ndim == 1:
            utab, table = uu[None, :udeg], np.zeros(S, dtype=np.int32)
        else:
            utab, table = np.unique(uu[:, :udeg], axis=0, return_inverse=True)
            table = table.astype(np.int32).reshape(-1)
        if np.any(per(p) < -1e-6):
            raise ValueError("p out of bounds")
        var = np.asarray(ferr, dtype=np.float64) ** 2
        stars = make_stars(S, period=per(p), tau=float(tau) if tau else 0.0, baseline_var=per(baseline_var),
                           baseline_mean=per(baseline_mean), data_var=per(var) if var.ndim < 2 else 0.0, table=table)
        # the sweep's handle + three more for the tables' finite differences: a table evaluation is a latency chain
        # of a dozen small kernels (0.3 ms); nine of them in a row on ONE stream outlast the sweep they should hide
        # behind (3.2 against 2.5 ms), three streams of three do not
        slots = engine_slots(ydeg, udeg, device, 4)
        (self._e, self._stream), self._side = slots[0], slots[1:]
        e = self._e
        self.S, self.K, self._ntab = S, K, utab.shape[0]
        self._t, self._flux = e.f64(np.ascontiguousarray(t)), e.f64(np.ascontiguousarray(flux))
        self._diag = e.f64(np.ascontiguousarray(np.broadcast_to(var, (S, K)))) if var.ndim == 2 else None
        self._stars = e.stars_to_device(stars)
        self._rta1 = e.f64(e.rTA1L(utab))
        self._table = torch.as_tensor(table.astype(np.int64), device=e.device)
        self._covpts = int(defaults["covpts"] if covpts is None else covpts)
        self._temporal = (temporal_kernel if isinstance(temporal_kernel, str) else kernel_id(temporal_kernel)) if tau else None
        self._normalized, self._h, self._ukw = bool(normalized), float(h), dict(upstream_kwargs or {})
        self._exact = bool(exact)
        self._ws = e.grad_workspace(S, K, self._covpts, M)
        self.lnlike = None
        torch.cuda.synchronize(e.device)

    def _tables(self, eng, **hp):
        """(yp [ntab, np], mean [ntab]) of the kernel tables at the given hyperparameters, on eng's stream."""
        from .upstream_device import ylm_moments_device

        mu, Sig = ylm_moments_device(eng, **hp, **self._ukw)
        eng.set_moments_dev(mu, Sig)
        tab, mv = eng.kernel_table(self._rta1, self._covpts)
        return tab[:, 0, :], mv[:, 0], (mu, Sig, tab, mv)

    def __call__(self, r=defaults["r"], a=defaults["a"], b=defaults["b"], c=defaults["c"], n=defaults["n"], dr=None):
        import torch

        e = self._e
        x0 = {"r": float(r), "dr": dr, "a": float(a), "b": float(b)}
        hp0 = dict(x0, c=float(c), n=float(n))
        torch.cuda.synchronize(e.device)
        exact = self._exact and dr is None
        # main stream: the tables at the point, then the sweep
        with torch.cuda.stream(self._stream):
            yp0, mean0, (mu, Sig, tab, mv) = self._tables(e, **hp0)
            at_point = torch.cuda.Event()
            at_point.record(self._stream)
            lnl, ybar, mbar, status = e.lnlike_grad_marginal(
                self._t, self._flux, self._stars, tab, mv, diag=self._diag, covpts=self._covpts,
                temporal=self._temporal, normalized=self._normalized, workspace=self._ws)
        # three more streams, meanwhile: the tables' derivatives
        bounds = {"r": (0.0, 90.0), "dr": (0.0, 90.0), "a": (0.0, 1.0), "b": (0.0, 1.0)}
        dy, dm, events = {}, {}, []

        def central(eng, name):
            x = float(x0[name])
            step = self._h * max(abs(x), 0.1)
            lo_b, hi_b = bounds[name]
            xl, xh = max(x - step, lo_b), min(x + step, hi_b)        # one-sided within a step of a bound
            yl, ml, _ = self._tables(eng, **dict(hp0, **{name: xl}))
            yh, mh, _ = self._tables(eng, **dict(hp0, **{name: xh}))
            dy[name], dm[name] = (yh - yl) / (xh - xl), (mh - ml) / (xh - xl)

        def tangent(eng, stream, tang, k, name):
            # The table is linear in Sigma_y + mu_y mu_y^T and its mean in mu_y (flux.py:297-320): with the moments
            # (dmu, X - dmu dmu^T), X = dSigma + dmu mu^T + mu dmu^T, the table kernels return f[X] - dmean^2 and
            # dmean, and d yp = f[X] - 2 mean dmean.  One table evaluation per parameter, no step size.
            mu1, dmu, dSig = tang
            d1 = dmu[k]
            cross = torch.outer(d1, mu1)
            eng.set_moments_dev(d1, dSig[k] + cross + cross.t() - torch.outer(d1, d1))
            tk, mk = eng.kernel_table(self._rta1, self._covpts)
            dmean = mk[:, 0]
            stream.wait_event(at_point)         # (mean0 is the main stream's; recorded before the sweep was enqueued)
            dy[name] = tk[:, 0, :] + (dmean * (dmean - 2.0 * mean0))[:, None]
            dm[name] = dmean

        (e1, s1), (e2, s2), (e3, s3) = self._side
        with torch.cuda.stream(s1):
            if exact:
                # the moments again, with their tangents, beside the main stream (which only waits for the value)
                from .upstream_device import ylm_moments_device_grad

                mu1, _, dmu, dSig = ylm_moments_device_grad(e1, r=float(r), a=float(a), b=float(b), c=float(c),
                                                            n=float(n), **self._ukw)
                tang = (mu1, dmu, dSig)
                have_tangents = torch.cuda.Event()
                have_tangents.record(s1)
                tangent(e1, s1, tang, 0, "r")
                tangent(e1, s1, tang, 1, "a")
            else:
                central(e1, "r")
                if x0["dr"] is not None:
                    central(e1, "dr")
            events.append(torch.cuda.Event())
            events[-1].record(s1)
        with torch.cuda.stream(s2):
            if exact:
                s2.wait_event(have_tangents)
                tangent(e2, s2, tang, 2, "b")
            else:
                central(e2, "a")
                central(e2, "b")
            events.append(torch.cuda.Event())
            events[-1].record(s2)
        with torch.cuda.stream(s3):
            eu = e3
            # c and n, exactly: mu_y = c n m, Sigma_y = c^2 n S + eps; the second moment f = yp + mean^2 is linear in
            # Sigma_y + mu_y mu_y^T:  f = c^2 n f_S + c^2 n^2 f_mm + f_eps,  mean = c n m1
            # (the tables at the point are the main stream's: they are ready long before the sweep is)
            s3.wait_event(at_point)
            ypA, meanA, muA, SigA = yp0, mean0, mu, Sig
            N = muA.shape[0]
            eps = torch.full((N,), float(self._ukw.get("epsy", defaults["epsy"])), dtype=torch.float64, device=eu.device)
            eps[15 ** 2:] = float(self._ukw.get("epsy15", defaults["epsy15"]))
            zero_mu = torch.zeros_like(muA)
            eu.set_moments_dev(muA, torch.zeros_like(SigA))
            t_mm, _ = eu.kernel_table(self._rta1, self._covpts)             # f_mm part: yp = c^2 n^2 f_mm - mean^2
            eu.set_moments_dev(zero_mu, torch.diag(eps))
            t_eps, _ = eu.kernel_table(self._rta1, self._covpts)            # f_eps (mean 0)
            f = ypA + meanA[:, None] ** 2
            f_mm = t_mm[:, 0, :] + meanA[:, None] ** 2
            f_eps = t_eps[:, 0, :]
            f_S = f - f_mm - f_eps
            if c != 0 and n != 0:
                dy["c"] = 2.0 * (f - f_eps) / c - 2.0 * meanA[:, None] ** 2 / c
                dm["c"] = meanA / c
                dy["n"] = (f_S + 2.0 * f_mm) / n - 2.0 * meanA[:, None] ** 2 / n
                dm["n"] = meanA / n
            else:
                # On the boundary c = 0 or n = 0 the tables at the point hold nothing to divide by: take them at unit
                # contrast and unit number of spots instead (f is a polynomial in c and n: f = c^2 n f_S' + c^2 n^2 f_mm'
                # + f_eps, mean = c n m1), one more upstream evaluation -- what hyper_gradient returns there too.
                from .upstream_device import ylm_moments_device

                mu1, Sig1 = ylm_moments_device(eu, **dict(hp0, c=1.0, n=1.0), **self._ukw)
                eu.set_moments_dev(mu1, torch.zeros_like(Sig1))
                t1, mv1 = eu.kernel_table(self._rta1, self._covpts)
                m1 = mv1[:, 0]
                f_mm1 = t1[:, 0, :] + m1[:, None] ** 2
                eu.set_moments_dev(zero_mu, Sig1 - torch.diag(eps))
                tS, _ = eu.kernel_table(self._rta1, self._covpts)
                f_S1 = tS[:, 0, :]
                cc, nn = float(c), float(n)
                dy["c"] = 2.0 * cc * nn * f_S1 + 2.0 * cc * nn * nn * f_mm1 - 2.0 * cc * nn * nn * m1[:, None] ** 2
                dm["c"] = nn * m1
                dy["n"] = cc * cc * f_S1 + 2.0 * cc * cc * nn * f_mm1 - 2.0 * cc * cc * nn * m1[:, None] ** 2
                dm["n"] = cc * m1
            events.append(torch.cuda.Event())
            events[-1].record(s3)
        with torch.cuda.stream(self._stream):
            for ev in events:
                self._stream.wait_event(ev)
            # adjoints per table: the stars that share a flux operator add up
            Yb = torch.zeros(self._ntab, ybar.shape[1], dtype=torch.float64, device=e.device).index_add_(0, self._table, ybar)
            Mb = torch.zeros(self._ntab, dtype=torch.float64, device=e.device).index_add_(0, self._table, mbar)
            names = [k for k in ("r", "dr", "a", "b", "c", "n") if k in dy]
            # ONE transfer for everything the host wants: [gradient | per-star values | per-star status]
            parts = [lnl, status.to(torch.float64)]
            if names:
                DY, DM = torch.stack([dy[k] for k in names]), torch.stack([dm[k] for k in names])
                parts.insert(0, (DY * Yb).sum(dim=(1, 2)) + (DM * Mb).sum(dim=1))
            host = torch.cat(parts).cpu().numpy()    # (on the main stream, which has waited for the others)
        ng = len(names)
        self.lnlike = host[ng:ng + self.S].copy()
        self.status = host[ng + self.S:].astype(np.uint32)
        total = float(self.lnlike.sum())
        grad = {k: float(v) for k, v in zip(names, host[:ng])}
        return total, grad


def ensemble_gradient(t, flux, ferr=1.0e-3, p=1.0, r=defaults["r"], a=defaults["a"], b=defaults["b"],
                      c=defaults["c"], n=defaults["n"], dr=None, **kwargs):
    """One-shot form of ``EnsembleGradient``: (sum of log-likelihoods, {"r": ., "a": ., "b": ., "c": ., "n": .})."""
    return EnsembleGradient(t, flux, ferr=ferr, p=p, **kwargs)(r=r, a=a, b=b, c=c, n=n, dr=dr)


def ensemble_gradient_conditional(t, flux, ferr=1.0e-3, p=1.0, i=defaults["i"], r=defaults["r"], a=defaults["a"],
                                  b=defaults["b"], c=defaults["c"], n=defaults["n"], upstream_kwargs=None, **kwargs):
    """The CONDITIONAL branch for an ensemble (each star at its own inclination i_s; tests/test_lnlike.py:100-136 verifies
    the gradient on both branches): (sum_s lnL_s, {"r", "a", "b", "c", "n": floats, "i", "p": arrays [S]}, lnL [S]).

    The hyperparameters enter every star through the SAME (mu_y, Sigma_y), so the chain through the upstream is taken
    once: the moments' adjoints of the stars (one reverse sweep per star over the library's reverse-mode kernels,
    ``log_likelihood_with_grad``: C = A Sigma_y A^T, the design matrix's adjoint through sp_dotRx / sp_tensordotRz_rev)
    are summed, then contracted with the moments' exact tangents (``ylm_moments_device_grad``) -- one upstream
    evaluation per gradient instead of one per star and parameter.  Star by star on the host (6.5 ms each at K = 1000):
    an optimiser's aid, not a timed path; the marginal branch has the one-sweep device form (``EnsembleGradient``).
    kwargs: as for ``log_likelihood_with_grad`` (u, tau, normalized, baseline_*, ydeg, ...)."""
    from .upstream_device import ylm_moments_device_grad

    flux = np.asarray(flux, dtype=np.float64)
    if flux.ndim != 2:
        raise ValueError("flux must be (S, K)")
    S, K = flux.shape
    t = np.asarray(t, dtype=np.float64)
    t = np.broadcast_to(t, (S, K)) if t.ndim == 1 else t
    per = lambda x: np.broadcast_to(np.asarray(x, dtype=np.float64), (S,))          # noqa: E731
    var = np.asarray(ferr, dtype=np.float64) ** 2
    e = get_engine(kwargs.get("ydeg", defaults["ydeg"]), kwargs.get("udeg", defaults["udeg"]), kwargs.get("device"))
    ukw = dict(upstream_kwargs or {})
    mu, Sig, dmu, dSig = [x.cpu().numpy() for x in ylm_moments_device_grad(e, r=r, a=a, b=b, c=c, n=n, **ukw)]
    gmu, gSig = np.zeros_like(mu), np.zeros_like(Sig)
    lnl, gi, gp = np.zeros(S), np.zeros(S), np.zeros(S)
    for s in range(S):
        v = var if var.ndim == 0 else (var[s] if var.ndim == 2 else var)
        lnl[s], g = log_likelihood_with_grad(mu, Sig, t[s], flux[s], v, i=float(per(i)[s]), p=float(per(p)[s]),
                                             marginalize_over_inclination=False, **kwargs)
        gmu += g["mean_ylm"]
        gSig += g["cov_ylm"]
        gi[s], gp[s] = g["i"], g["p"]
    N = mu.shape[0]
    eps = np.ones(N) * float(ukw.get("epsy", defaults["epsy"]))
    eps[15 ** 2:] = float(ukw.get("epsy15", defaults["epsy15"]))
    S0 = Sig - np.diag(eps)
    out = {name: float(gmu @ dmu[k] + np.sum(gSig * dSig[k])) for k, name in enumerate(("r", "a", "b"))}
    if c != 0 and n != 0:
        # mu ~ c n, Sigma - eps ~ c^2 n (contrast.py:21-33)
        out["c"] = float(gmu @ mu / c + 2.0 * np.sum(gSig * S0) / c)
        out["n"] = float(gmu @ mu / n + np.sum(gSig * S0) / n)
    else:
        # On the boundary c = 0 or n = 0 the moments at the point hold nothing to divide by, but the derivative is not
        # zero there: with the moments at unit contrast and unit number of spots, mu = c n m1 and Sigma - eps = c^2 n S1
        # give d/dc = n gmu.m1 + 2 c n <gSig, S1> and d/dn = c gmu.m1 + c^2 <gSig, S1> -- what hyper_gradient and
        # EnsembleGradient return there too (one more upstream evaluation).
        from .upstream_device import ylm_moments_device

        m1, Sig1 = [x.cpu().numpy() for x in ylm_moments_device(e, r=r, a=a, b=b, c=1.0, n=1.0, **ukw)]
        S1 = Sig1 - np.diag(eps)
        gm, gS = float(gmu @ m1), float(np.sum(gSig * S1))
        out["c"] = float(n) * gm + 2.0 * float(c) * float(n) * gS
        out["n"] = float(c) * gm + float(c) ** 2 * gS
    out["i"], out["p"] = gi, gp
    return float(lnl.sum()), out, lnl
