"""
``FluxIntegral``: host-side mirror of the reference class of the same name
(reference ``flux.py:23-379``) for the log-likelihood path.  Same constructor
arguments and the same public methods ``mean / cov / design_matrix / kernel``
with the same argument meaning (``i`` in degrees, ``p`` in the units of ``t``,
``u`` the limb-darkening coefficients).  All numbers come from the HIP kernels
through ``Engine``; this class only validates arguments and sequences calls.
"""
import numpy as np

from .defaults import defaults
from .engine import get_engine, make_stars
from .ops import CheckBoundsOp, Eager

__all__ = ["FluxIntegral"]


class FluxIntegral(object):
    def __init__(
        self,
        mean_ylm,
        cov_ylm,
        udeg=defaults["udeg"],
        marginalize_over_inclination=defaults["marginalize_over_inclination"],
        covpts=defaults["covpts"],
        ydeg=defaults["ydeg"],
        **kwargs
    ):
        self._udeg = int(udeg)
        self._ydeg = int(ydeg)
        self._nylm = (self._ydeg + 1) ** 2
        self._marginalize_over_inclination = bool(marginalize_over_inclination)
        self._covpts = int(covpts)
        self._engine = get_engine(self._ydeg, self._udeg, kwargs.get("device"))
        # the moments may already live on the device (upstream_device.py): no round trip
        self._dev_moments = None
        if hasattr(mean_ylm, "is_cuda") and hasattr(cov_ylm, "is_cuda"):
            self._dev_moments = (mean_ylm.contiguous(), cov_ylm.contiguous())
            self._mean_ylm = self._cov_ylm = None
        else:
            self._mean_ylm = np.ascontiguousarray(np.asarray(mean_ylm, dtype=np.float64).reshape(-1))
            self._cov_ylm = np.ascontiguousarray(np.asarray(cov_ylm, dtype=np.float64))
        self._check_i = CheckBoundsOp(name="i", lower=0, upper=90.0 + 1e-4)
        self._check_p = CheckBoundsOp(name="p", lower=0, upper=np.inf)
        self._bind()

    # the engine keeps ONE set of Ylm moments resident; (re)bind ours before use
    def _bind(self):
        e = self._engine
        if getattr(e, "_moments_owner", None) is not self:
            if self._dev_moments is not None:
                e.set_moments_dev(*self._dev_moments)
            else:
                e.set_moments(self._mean_ylm, self._cov_ylm)
            e._moments_owner = self

    def _ingest(self, t, i, p, u):
        t = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(-1))
        i = float(np.asarray(i))
        p = float(np.asarray(p))
        # bounds as in flux.py:236-239 (i in [0, pi/2] rad, p >= 0, tol 1e-6)
        if i * np.pi / 180 < -1e-6 or i * np.pi / 180 > 0.5 * np.pi + 1e-6:
            raise ValueError("i out of bounds: %f" % i)
        self._check_p(p)
        u = np.asarray(u, dtype=np.float64).reshape(-1)[: self._udeg]
        if u.shape[0] < self._udeg:
            raise ValueError("Vector `u` has the wrong size.")
        return t, i, p, u

    def _rta1(self, u):
        return self._engine.f64(self._engine.rTA1L(u))

    def _table(self, u):
        self._bind()
        return self._engine.kernel_table(self._rta1(u), self._covpts)

    def design_matrix(self, t, i, p, u):
        t, i, p, u = self._ingest(t, i, p, u)
        self._bind()
        A = self._engine.design_matrix(t[None, :], make_stars(1, period=p, inc_deg=i), self._rta1(u))
        return Eager(A[0].cpu().numpy())

    def mean(self, t, i, p, u):
        t, i, p, u = self._ingest(t, i, p, u)
        return Eager(self._flux_mean(t, i, p, u) * np.ones_like(t))

    def _flux_mean(self, t, i, p, u):
        self._bind()
        if self._marginalize_over_inclination:
            _, mv = self._table(u)
            return float(mv[0, 0].item())
        _, mean, _ = self._engine.cov_conditional(
            t[None, :], make_stars(1, period=p, inc_deg=i), self._rta1(u), normalized=False)
        return float(mean[0].item())

    def cov(self, t, i, p, u):
        t, i, p, u = self._ingest(t, i, p, u)
        self._bind()
        e = self._engine
        stars = make_stars(1, period=p, inc_deg=i)
        if self._marginalize_over_inclination:
            tab, mv = self._table(u)
            cov, _ = e.cov_marginal(t[None, :], stars, self._covpts, tab, mv, normalized=False)
        else:
            cov, _, _ = e.cov_conditional(t[None, :], stars, self._rta1(u), normalized=False)
        return Eager(cov[0].cpu().numpy())

    def kernel(self, t, i, p, u):
        """flux.py:352-365."""
        t, i, p, u = self._ingest(t, i, p, u)
        self._bind()
        e = self._engine
        if self._marginalize_over_inclination:
            rta1 = self._rta1(u)
            # W (flux.py:199-209) and the second moment at the requested lags
            idx = _lib_tables(self._ydeg)
            rho = rta1[0][e.dev(idx["m0"].astype(np.int64))]
            lof = e.dev(idx["l_of"].astype(np.int64))
            _, Wnp = _hostconst_W(self._ydeg)
            W = e.f64(Wnp) * rho[lof][:, None] * rho[lof][None, :]
            _, Ez = e.polar_moments()
            theta = 2 * np.pi * np.mod(t / p, 1.0)
            mom2 = e.special_tensordotRz(W, Ez, theta)
            _, mv = self._table(u)
            return Eager((mom2 - mv[0, 0] ** 2).cpu().numpy())
        return Eager(np.asarray(self.cov(t, i, p, u))[0])


def _lib_tables(ydeg):
    from . import _lib

    return _lib.index_tables(ydeg)


def _hostconst_W(ydeg):
    from . import hostconst

    return hostconst.marginal_constants(ydeg)
