"""
``StarryProcess``: source-compatible front end of the reference class
(reference ``sp.py:38-1396``) restricted to the log-likelihood hot path:

    sp = StarryProcess(r=..., a=..., b=..., c=..., n=..., ydeg=15, ...)
    sp.mean(t, i, p, u); sp.cov(t, i, p, u)
    sp.log_likelihood(t, flux, data_cov, i, p, u, baseline_mean, baseline_var)

Keyword names, defaults, argument meaning and error behaviour follow the
reference (``sp.py:39-51, 643-703, 1052-1188``).  Results are eager NumPy values
with a no-op ``.eval()`` so scripts written for the lazy Theano graph run
unchanged.  Every number on the path mean / cov / log_likelihood is produced on
the GPU by ``libsp_hip.so``.

The Ylm moments (mu_y, Sigma_y) come from the hyperparameters through the host
module ``upstream`` (reference sp.py:257-266); they can also be injected with
``mean_ylm= / cov_ylm=`` (what the tests and ``bench.py`` do with the golden
moments).

Addition over the reference: ``log_likelihood_ensemble`` evaluates many stars,
each with its own period / inclination / limb darkening / noise, in one batched
device call -- the calibrate-style use case the reference describes but leaves
unimplemented (joss/paper.md:160-172).
"""
import numpy as np

from .defaults import defaults
from .engine import get_engine, make_stars
from .flux import FluxIntegral
from .ops import AlphaBetaOp, CheckBoundsOp, Eager
from .temporal import kernel_id

__all__ = ["StarryProcess", "StarryProcessSum"]


def _neg_inf_if_nan(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(np.isnan(x), -np.inf, x)


class StarryProcess(object):
    def __init__(
        self,
        r=defaults["r"],
        dr=defaults["dr"],
        c=defaults["c"],
        n=defaults["n"],
        tau=defaults["tau"],
        temporal_kernel=defaults["temporal_kernel"],
        marginalize_over_inclination=defaults["marginalize_over_inclination"],
        normalized=defaults["normalized"],
        covpts=defaults["covpts"],
        **kwargs
    ):
        mu = kwargs.pop("mu", None)
        sigma = kwargs.pop("sigma", None)
        mean_ylm = kwargs.pop("mean_ylm", None)
        cov_ylm = kwargs.pop("cov_ylm", None)
        if mu is None and sigma is None:
            a = kwargs.pop("a", defaults["a"])
            b = kwargs.pop("b", defaults["b"])
        elif (kwargs.get("a", None) is None and kwargs.get("b", None) is None) and (
            mu is not None and sigma is not None
        ):
            from .upstream import gauss2beta

            a, b = gauss2beta(mu, sigma)
        else:
            raise ValueError("Must provide either `a` and `b` *or* `mu` and `sigma`.")

        if tau is None:
            self._tau = 0.0
            self._time_variable = False
            self._temporal = None
        else:
            self._tau = float(CheckBoundsOp(name="tau", lower=0, upper=np.inf)(tau))
            self._time_variable = True
            self._temporal = kernel_id(temporal_kernel)

        self._ydeg = int(kwargs.get("ydeg", defaults["ydeg"]))
        assert self._ydeg >= 5, "Degree of map must be >= 5."
        self._udeg = int(kwargs.get("udeg", defaults["udeg"]))
        assert self._udeg >= 0, "Degree of limb darkening must be >= 0."
        self._nylm = (self._ydeg + 1) ** 2
        self._covpts = int(covpts)
        self._kwargs = kwargs
        self._normalized = bool(normalized)
        self._normN = int(kwargs.get("normalization_order", defaults["normalization_order"]))
        self._normzmax = float(kwargs.get("normalization_zmax", defaults["normalization_zmax"]))
        self._get_alpha_beta = AlphaBetaOp(self._normN)
        self._marginalize_over_inclination = bool(marginalize_over_inclination)
        self._r, self._dr, self._a, self._b, self._c, self._n = r, dr, a, b, c, n

        self._engine = get_engine(self._ydeg, self._udeg, kwargs.get("device"))
        dev_moments = None
        from_hyper = mean_ylm is None or cov_ylm is None
        if from_hyper:
            # upstream="reference" (default): the reference's algorithm on the host, comparable
            # digit by digit on the same host; upstream="device": the same integrals by exact
            # quadrature of rotations on the GPU (upstream_device.py), ~50x faster and free of
            # the reference's rounding noise in the high degrees
            how = kwargs.get("upstream", "reference")
            ukw = {k: v for k, v in kwargs.items() if k not in ("ydeg", "upstream")}
            if how == "device":
                from .upstream_device import ylm_moments_device

                dev_moments = ylm_moments_device(self._engine, r=r, dr=dr, a=a, b=b, c=c, n=n, **ukw)
            elif how == "reference":
                from .upstream import ylm_moments

                mean_ylm, cov_ylm = ylm_moments(r=r, dr=dr, a=a, b=b, c=c, n=n, ydeg=self._ydeg, **ukw)
            else:
                raise ValueError("upstream must be 'reference' or 'device'")
        self._dev_moments = dev_moments
        self._from_hyper = from_hyper      # built from (r, dr, a, b, c, n), not from explicit moments
        if dev_moments is None:
            self._host_moments = (np.asarray(mean_ylm, dtype=np.float64).reshape(-1),
                                  np.asarray(cov_ylm, dtype=np.float64))
            if self._host_moments[0].shape != (self._nylm,) or self._host_moments[1].shape != (self._nylm, self._nylm):
                raise ValueError("mean_ylm / cov_ylm have the wrong shape for ydeg=%d" % self._ydeg)
        else:
            self._host_moments = None      # copied back only if somebody asks (properties below)

        self._flux = FluxIntegral(
            dev_moments[0] if dev_moments is not None else self._host_moments[0],
            dev_moments[1] if dev_moments is not None else self._host_moments[1],
            udeg=self._udeg,
            marginalize_over_inclination=self._marginalize_over_inclination,
            covpts=self._covpts,
            ydeg=self._ydeg,
            device=kwargs.get("device"),
        )
        self._z = None

    def _moments_np(self):
        if self._host_moments is None:
            self._host_moments = tuple(x.cpu().numpy() for x in self._dev_moments)
        return self._host_moments

    _mean_ylm = property(lambda self: self._moments_np()[0])
    _cov_ylm = property(lambda self: self._moments_np()[1])

    # -- hyperparameters (read-only views, sp.py:286-367) -------------------------
    a = property(lambda self: self._a)
    b = property(lambda self: self._b)
    r = property(lambda self: self._r)
    dr = property(lambda self: self._dr)
    c = property(lambda self: self._c)
    n = property(lambda self: self._n)
    tau = property(lambda self: self._tau)
    ydeg = property(lambda self: self._ydeg)
    udeg = property(lambda self: self._udeg)
    normalized = property(lambda self: self._normalized)
    covpts = property(lambda self: self._covpts)
    marginalize_over_inclination = property(lambda self: self._marginalize_over_inclination)
    mean_ylm = property(lambda self: Eager(self._mean_ylm))
    cov_ylm = property(lambda self: Eager(self._cov_ylm))

    # -- sums of processes (sp.py:1190-1197, 1335-1400) -------------------------------
    def __add__(self, other):
        return StarryProcessSum(self, other)

    def __radd__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self          # so that sum([sp1, sp2, ...]) works
        return self.__add__(other)

    def log_jac(self):
        """Log |Jacobian| of the (a, b) -> (mu, sigma) transform (sp.py:1004-1050)."""
        from .upstream import log_jac

        return Eager(np.float64(log_jac(self._a, self._b, **self._kwargs)))

    # -- mean / cov (sp.py:643-703) --------------------------------------------------
    def mean(self, t, i=defaults["i"], p=defaults["p"], u=defaults["u"][: defaults["udeg"]]):
        if self._normalized:
            return Eager(np.zeros_like(np.asarray(t, dtype=np.float64).reshape(-1)))
        return self._flux.mean(t, i, p, u)

    def _device_cov(self, t, i, p, u):
        f = self._flux
        t, i, p, u = f._ingest(t, i, p, u)
        f._bind()
        e = self._engine
        stars = make_stars(1, period=p, inc_deg=i, tau=self._tau)
        if self._marginalize_over_inclination:
            tab, mv = f._table(u)
            cov, z = e.cov_marginal(t[None, :], stars, self._covpts, tab, mv, temporal=self._temporal,
                                    normalized=self._normalized, norm_order=self._normN)
            fmean = mv[0, 0]
        else:
            cov, mean, z = e.cov_conditional(t[None, :], stars, f._rta1(u), temporal=self._temporal,
                                             normalized=self._normalized, norm_order=self._normN)
            fmean = mean[0]
        if self._normalized:
            self._z = float(z[0].item())
        return t, cov[0], fmean

    def cov(self, t, i=defaults["i"], p=defaults["p"], u=defaults["u"][: defaults["udeg"]]):
        _, cov, _ = self._device_cov(t, i, p, u)
        return Eager(cov.cpu().numpy())

    # -- log likelihood (sp.py:1052-1188) ------------------------------------------------
    def log_likelihood(
        self,
        t,
        flux,
        data_cov,
        i=defaults["i"],
        p=defaults["p"],
        u=defaults["u"][: defaults["udeg"]],
        baseline_mean=defaults["baseline_mean"],
        baseline_var=defaults["baseline_var"],
    ):
        f = self._flux
        t, i, p, u = f._ingest(t, i, p, u)
        K = t.shape[0]
        flux = np.asarray(flux, dtype=np.float64)
        F = flux.reshape(1, K) if flux.ndim == 1 else flux.reshape(-1, K)
        data_cov = np.asarray(data_cov, dtype=np.float64)
        bmean = np.asarray(baseline_mean, dtype=np.float64)
        bvar = np.asarray(baseline_var, dtype=np.float64)
        simple = data_cov.ndim <= 1 and bmean.ndim == 0 and bvar.ndim == 0
        e = self._engine
        f._bind()
        if simple:
            stars = make_stars(1, period=p, inc_deg=i, tau=self._tau, baseline_var=float(bvar),
                               baseline_mean=float(bmean),
                               data_var=float(data_cov) if data_cov.ndim == 0 else 0.0)
            diag = e.f64(data_cov.reshape(1, K)) if data_cov.ndim == 1 else None
            rta1 = f._rta1(u)
            tab = mv = None
            if self._marginalize_over_inclination:
                tab, mv = f._table(u)
            out, status = e.lnlike_ensemble(
                e.f64(t[None, :]), e.f64(F[None, :, :]), e.stars_to_device(stars), diag=diag,
                conditional=not self._marginalize_over_inclination, covpts=self._covpts, tab=tab,
                meanvar=mv, rta1=rta1, temporal=self._temporal, normalized=self._normalized,
                norm_order=self._normN, zmax=self._normzmax)
            return Eager(_neg_inf_if_nan(out.cpu().numpy())[0])
        # general data / baseline covariances: assemble on the device, add the
        # extra terms with tensor ops, then the batched factorisation
        tt, cov, fmean = self._device_cov(t, i, p, u)
        C = cov.clone()
        if data_cov.ndim == 0:
            C.diagonal().add_(float(data_cov))
        elif data_cov.ndim == 1:
            C.diagonal().add_(e.f64(data_cov))
        else:
            C += e.f64(data_cov)
        C += e.f64(bvar) if bvar.ndim else float(bvar)
        gp_mean = 0.0 if self._normalized else fmean
        resid = e.f64(F) - (gp_mean + (e.f64(bmean) if bmean.ndim else float(bmean)))
        out, status = e.cholesky_lnlike(C[None, :, :], resid[None, :, :])
        val = _neg_inf_if_nan(out.cpu().numpy())[0]
        if self._normalized and self._z is not None and self._z > self._normzmax:
            val = -np.inf
        return Eager(val)

    def log_likelihood_samples(
        self,
        t,
        flux,
        data_cov,
        samples,
        i=defaults["i"],
        p=defaults["p"],
        u=defaults["u"][: defaults["udeg"]],
        baseline_mean=defaults["baseline_mean"],
        baseline_var=defaults["baseline_var"],
        depth=6,
        out_of_bounds="raise",
    ):
        """``log_likelihood(t, flux, data_cov, ...)`` of THIS process's settings (degree, normalisation, lag grid,
        temporal kernel) at many hyperparameter vectors: samples (ns, 5) = rows of (r, a, b, c, n) -> (ns,) values, each
        what ``StarryProcess(r=r, a=a, b=b, c=c, n=n, <same settings>, upstream="device").log_likelihood(...)``
        returns.  What a sampler does with the reference one call at a time (sp.py:1052-1062 driven by
        calibrate/sample.py:95-107) is here ONE batched device step per 64 samples (calibrate.SampleBatches) --
        marginalised, normalised processes with one spot radius and scalar or per-cadence data variance; anything
        else is evaluated sample by sample.  ``out_of_bounds="inf"``: samples outside the reference's parameter bounds
        (a ValueError there and, by default, here) get -inf and are not evaluated."""
        from .calibrate import MAX_STREAMS_SAMPLES, SampleBatches, clamp_depth
        from .engine import engine_slots

        f = self._flux
        t, i, p, u = f._ingest(t, i, p, u)
        K = t.shape[0]
        samples = np.atleast_2d(np.asarray(samples, dtype=np.float64))
        if samples.shape[1] != 5:
            raise ValueError("samples must be (ns, 5): r, a, b, c, n")
        if out_of_bounds == "inf":
            from .engine import samples_in_bounds

            ok = samples_in_bounds(samples)
            if not ok.all():
                out = np.full(samples.shape[0], -np.inf)
                if ok.any():
                    out[ok] = np.asarray(self.log_likelihood_samples(t, flux, data_cov, samples[ok], i=i, p=p, u=u,
                                                                     baseline_mean=baseline_mean, baseline_var=baseline_var,
                                                                     depth=depth))
                return Eager(out)
        elif out_of_bounds != "raise":
            raise ValueError("out_of_bounds must be 'raise' or 'inf'")
        flux = np.asarray(flux, dtype=np.float64)
        F = flux.reshape(1, K) if flux.ndim == 1 else flux.reshape(-1, K)
        data_cov = np.asarray(data_cov, dtype=np.float64)
        bmean, bvar = np.asarray(baseline_mean, dtype=np.float64), np.asarray(baseline_var, dtype=np.float64)
        batched = (self._marginalize_over_inclination and self._normalized and self._dr is None and K >= 2
                   and data_cov.ndim <= 1 and bmean.ndim == 0 and bvar.ndim == 0)
        if not batched:
            kw = dict(self._kwargs)
            kw.update(dr=self._dr, tau=self._tau if self._time_variable else None, temporal_kernel=self._temporal or "matern32",
                      marginalize_over_inclination=self._marginalize_over_inclination, normalized=self._normalized,
                      covpts=self._covpts, upstream="device")
            return Eager(np.array([float(StarryProcess(r=r, a=a, b=b, c=c, n=n, **kw).log_likelihood(
                t, flux, data_cov, i=i, p=p, u=u, baseline_mean=baseline_mean, baseline_var=baseline_var))
                for r, a, b, c, n in samples]))
        key = (t.tobytes(), F.tobytes(), data_cov.tobytes(), float(p), tuple(np.asarray(u, dtype=float).reshape(-1)),
               float(bmean), float(bvar), int(depth))
        cache = self.__dict__.get("_sample_batches")
        if cache is None or cache[0] != key:
            # (the data set is planned once and kept: a sampler calls this with the same data every iteration)
            slots = engine_slots(self._ydeg, self._udeg, self._kwargs.get("device"),
                                 clamp_depth(depth, limit=MAX_STREAMS_SAMPLES))
            e0 = slots[0][0]
            stars = make_stars(1, period=p, inc_deg=i, tau=self._tau, baseline_var=float(bvar), baseline_mean=float(bmean),
                               data_var=float(data_cov) if data_cov.ndim == 0 else 0.0)
            ukw = {k: self._kwargs[k] for k in ("epsy", "epsy15", "spts", "eps4", "smoothing", "sfac", "abmin",
                                                "log_alpha_max", "log_beta_max") if k in self._kwargs}
            sb = SampleBatches(slots, e0.f64(t[None, :]), e0.f64(F[None, :, :]), stars,
                               e0.f64(e0.rTA1L(np.asarray(u, dtype=np.float64))), self._covpts,
                               diag_dev=e0.f64(data_cov.reshape(1, K)) if data_cov.ndim == 1 else None,
                               temporal=self._temporal, norm_order=self._normN, zmax=self._normzmax, upstream_kwargs=ukw)
            cache = self._sample_batches = (key, sb)
        out = cache[1](samples)
        import torch

        torch.cuda.synchronize(out.device)
        return Eager(_neg_inf_if_nan(out[:, 0].cpu().numpy()))

    def log_likelihood_grad(
        self,
        t,
        flux,
        data_cov,
        i=defaults["i"],
        p=defaults["p"],
        u=defaults["u"][: defaults["udeg"]],
        baseline_mean=defaults["baseline_mean"],
        baseline_var=defaults["baseline_var"],
    ):
        """(lnL, grads): the log-likelihood of ONE light curve (scalar or per-point data variance, scalar
        baseline terms) and its gradient -- what ``theano.grad(sp.log_likelihood(...), [r, a, b, c, n, p, ...])``
        is in the reference (grad.py: one reverse sweep through the library's reverse-mode kernels).  A process
        built from hyperparameters returns d/d(r, dr, a, b, c, n) and d/dp, d/dtau, d/di (conditional branch);
        one built from explicit moments returns d/d(mean_ylm, cov_ylm) in their place."""
        from .grad import hyper_gradient, log_likelihood_with_grad

        kw = dict(i=float(np.asarray(i)), p=float(np.asarray(p)), u=np.asarray(u, dtype=np.float64),
                  tau=self._tau if self._time_variable else None,
                  temporal_kernel=self._temporal or "matern32", baseline_mean=float(baseline_mean),
                  baseline_var=float(baseline_var),
                  marginalize_over_inclination=self._marginalize_over_inclination, normalized=self._normalized,
                  covpts=self._covpts, ydeg=self._ydeg, udeg=self._udeg, norm_order=self._normN,
                  zmax=self._normzmax, device=self._kwargs.get("device"))
        if self._from_hyper:
            # (whichever upstream built the moments: the chain rule through the hyperparameters runs on the
            #  device quadrature, whose moments are the same integrals -- upstream_device.py)
            ukw = {k: self._kwargs[k] for k in ("epsy", "epsy15", "spts", "eps4", "smoothing", "sfac", "cutoff",
                                                "abmin", "log_alpha_max", "log_beta_max") if k in self._kwargs}
            lnl, g = hyper_gradient(t, flux, data_cov, r=self._r, dr=self._dr, a=self._a, b=self._b, c=self._c,
                                    n=self._n, upstream_kwargs=ukw,
                                    moments0=None if self._dev_moments is not None else self._host_moments, **kw)
        else:
            lnl, g = log_likelihood_with_grad(self._mean_ylm, self._cov_ylm, t, flux, data_cov, **kw)
        return Eager(np.float64(lnl)), g

    # -- prior samples (sp.py:489-516, 729-765, 1237-1282) -------------------------------
    # The random numbers are NumPy's: the reference's Theano RandomStream cannot be
    # reproduced, so these methods are pinned by their moments, not by sample values.
    def _rng(self, seed):
        return np.random.RandomState(self._kwargs.get("seed", 0) if seed is None else seed)

    @property
    def cho_cov_ylm(self):
        """Lower Cholesky factor of Sigma_y (sp.py:436-441), factored on the device."""
        L, info = self._engine.cho_factor(self._cov_ylm)
        L = L.cpu().numpy()
        return Eager(np.full_like(L, np.nan) if int(info.reshape(-1)[0].item()) else L)

    def sample_ylm(self, t=None, nsamples=1, seed=None):
        """Samples of the spherical-harmonic coefficients from the prior, shape
        (nsamples, nylm) (sp.py:503-507).  The time-variable form (``t`` given) needs the
        reference's SampleYlmTemporalOp, which is outside this package's scope."""
        if t is not None:
            raise NotImplementedError("time-variable Ylm samples are not implemented")
        u = self._rng(seed).randn(self._nylm, int(nsamples))
        return Eager((self._mean_ylm[:, None] + np.array(self.cho_cov_ylm) @ u).T)

    def sample(self, t, i=defaults["i"], p=defaults["p"], u=defaults["u"][: defaults["udeg"]],
               nsamples=1, eps=defaults["eps"], seed=None):
        """Light curves drawn from the prior, shape (nsamples, ntimes) (sp.py:729-765):
        mean + L z with L the device Cholesky factor of cov(t) + eps I."""
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        cov = np.array(self.cov(t, i, p, u))
        cov[np.diag_indices_from(cov)] += eps
        L, info = self._engine.cho_factor(cov)
        L = L.cpu().numpy()
        if int(info.reshape(-1)[0].item()):
            L = np.full_like(L, np.nan)
        U = self._rng(seed).randn(t.shape[0], int(nsamples))
        return Eager((np.array(self.mean(t, i, p, u))[:, None] + L @ U).T)

    def flux(self, y, t, i=defaults["i"], p=defaults["p"], u=defaults["u"][: defaults["udeg"]]):
        """Light curves of given spherical-harmonic vectors y (nsamples, nylm)
        (sp.py:1237-1282), through the device design matrix."""
        if self._time_variable:
            raise NotImplementedError("time-variable maps are not implemented")
        y = np.atleast_2d(np.asarray(y, dtype=np.float64))
        A = np.array(self._flux.design_matrix(t, i, p, u))      # (ntimes, nylm)
        flux = (A @ y.T).T
        if self._normalized:
            flux = (1.0 + flux) / np.mean(1.0 + flux, axis=-1).reshape(-1, 1) - 1.0
        return Eager(flux)

    # -- conditioning on data (sp.py:767-1002) ---------------------------------------------
    def predict(
        self,
        t,
        flux,
        data_cov,
        t_sample=None,
        i=defaults["i"],
        p=defaults["p"],
        u=defaults["u"][: defaults["udeg"]],
        baseline_mean=defaults["baseline_mean"],
        baseline_var=defaults["baseline_var"],
    ):
        """Mean and covariance of the light curve distribution conditioned on the observed
        flux (sp.py:767-903).  As in the reference: not implemented for normalized
        processes.  The three covariance blocks come from ONE device assembly on the
        concatenated times [t_sample, t]; the conditioning is one factorisation with the
        cross covariance riding along as extra rows (``sp_gp_condition``)."""
        if self._normalized:
            raise NotImplementedError("Method not implemented when the flux is normalized.")
        e = self._engine
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        K = t.shape[0]
        if t_sample is None:
            tall, Ks = t, K
        else:
            ts = np.asarray(t_sample, dtype=np.float64).reshape(-1)
            Ks = ts.shape[0]
            tall = np.concatenate([ts, t])
        _, cov, fmean = self._device_cov(tall, i, p, u)
        mean = float(fmean)
        if t_sample is None:
            Ktt, Kst, Kss = cov.clone(), cov.clone(), cov.clone()
        else:
            Kss, Kst, Ktt = cov[:Ks, :Ks].clone(), cov[:Ks, Ks:].clone(), cov[Ks:, Ks:].clone()
        data_cov = np.asarray(data_cov, dtype=np.float64)
        if data_cov.ndim == 0:
            Ktt.diagonal().add_(float(data_cov))
        elif data_cov.ndim == 1:
            Ktt.diagonal().add_(e.f64(data_cov))
        else:
            Ktt += e.f64(data_cov)
        bvar = np.asarray(baseline_var, dtype=np.float64)
        bv = e.f64(bvar) if bvar.ndim else float(bvar)
        Ktt += bv
        Kss += bv
        Kst += bv
        y = e.f64(np.asarray(flux, dtype=np.float64).reshape(-1) - np.asarray(baseline_mean, dtype=np.float64))
        mu, Kpost, info = e.gp_condition(Ktt, Kst, Kss, y - mean)
        mu = mu.cpu().numpy() + mean
        Kpost = Kpost.cpu().numpy()
        if int(info.item()):
            mu, Kpost = np.full_like(mu, np.nan), np.full_like(Kpost, np.nan)
        return Eager(mu), Eager(Kpost)

    def sample_conditional(self, t, flux, data_cov, t_sample=None, i=defaults["i"], p=defaults["p"],
                           u=defaults["u"][: defaults["udeg"]], baseline_mean=defaults["baseline_mean"],
                           baseline_var=defaults["baseline_var"], nsamples=1, eps=1e-12, seed=None):
        """Samples from the conditional distribution (sp.py:905-1002): predict, then
        mean + L z with L the device Cholesky factor of the posterior covariance.  The
        random numbers are NumPy's (the reference's Theano stream cannot be reproduced)."""
        mu, Kpost = self.predict(t, flux, data_cov, t_sample, i, p, u, baseline_mean, baseline_var)
        Kpost = np.array(Kpost)
        Kpost[np.diag_indices_from(Kpost)] += eps
        L, info = self._engine.cho_factor(Kpost)
        L = L.cpu().numpy()
        if int(info.reshape(-1)[0].item()):
            L = np.full_like(L, np.nan)
        z = np.random.RandomState(seed).randn(Kpost.shape[0], int(nsamples))
        return Eager(np.array(mu)[None, :] + (L @ z).T)

    def log_likelihood_ensemble(self, t, flux, data_cov, i=None, p=None, u=None,
                                baseline_mean=0.0, baseline_var=0.0):
        """Per-star log-likelihoods of S independent stars in one device call.

        t: (K,) or (S, K); flux: (S, K); data_cov: scalar, (S,) or (S, K);
        i, p: scalars or (S,); u: (udeg,) shared or (S, udeg).

        Ragged ensembles: ``t`` and ``flux`` may be lists of S 1-D arrays of different
        lengths (and ``data_cov`` a list of per-cadence variance vectors): the light
        curves are padded to the longest and each star is evaluated on its own cadences
        only (``sp_star.nobs``)."""
        e = self._engine
        f = self._flux
        nobs = 0
        if isinstance(flux, (list, tuple)) and len(flux) == 0:
            return Eager(np.empty(0))
        if isinstance(flux, (list, tuple)):
            S = len(flux)
            lens = np.array([np.size(x) for x in flux], dtype=np.int32)
            K = int(lens.max())
            if not isinstance(t, (list, tuple)) or len(t) != S or any(np.size(a) != n for a, n in zip(t, lens)):
                raise ValueError("ragged ensembles need one time array per light curve, of the same length")
            tp = np.empty((S, K))
            fp = np.zeros((S, K))
            for s_ in range(S):
                n_ = int(lens[s_])
                tp[s_, :n_] = np.asarray(t[s_], dtype=np.float64).reshape(-1)
                tp[s_, n_:] = tp[s_, n_ - 1]
                fp[s_, :n_] = np.asarray(flux[s_], dtype=np.float64).reshape(-1)
            if isinstance(data_cov, (list, tuple)) and np.ndim(data_cov[0]) == 1:
                dp = np.ones((S, K))
                for s_ in range(S):
                    dp[s_, : lens[s_]] = np.asarray(data_cov[s_], dtype=np.float64)
                data_cov = dp
            t, flux, nobs = tp, fp, lens
        flux = np.asarray(flux, dtype=np.float64)
        S, K = flux.shape
        t = np.asarray(t, dtype=np.float64)
        t = np.broadcast_to(t, (S, K)) if t.ndim == 1 else t
        p = np.broadcast_to(np.asarray(defaults["p"] if p is None else p, dtype=np.float64), (S,))
        i = np.broadcast_to(np.asarray(defaults["i"] if i is None else i, dtype=np.float64), (S,))
        if np.any(p < -1e-6):
            raise ValueError("p out of bounds")
        if np.any(i * np.pi / 180 < -1e-6) or np.any(i * np.pi / 180 > 0.5 * np.pi + 1e-6):
            raise ValueError("i out of bounds")
        u = np.asarray(defaults["u"][: self._udeg] if u is None else u, dtype=np.float64)
        if u.ndim == 1:
            utab, table = u[None, : self._udeg], np.zeros(S, dtype=np.int32)
        else:
            utab, table = np.unique(u[:, : self._udeg], axis=0, return_inverse=True)
            table = table.astype(np.int32).reshape(-1)
        data_cov = np.asarray(data_cov, dtype=np.float64)
        diag = None
        dvar = 0.0
        if data_cov.ndim == 2:
            diag = e.f64(np.ascontiguousarray(data_cov))
        else:
            dvar = np.broadcast_to(data_cov, (S,))
        stars = make_stars(S, period=p, inc_deg=i, tau=self._tau,
                           baseline_var=np.broadcast_to(np.asarray(baseline_var, float), (S,)),
                           baseline_mean=np.broadcast_to(np.asarray(baseline_mean, float), (S,)),
                           data_var=dvar, table=table, nobs=nobs)
        f._bind()
        rta1 = e.f64(e.rTA1L(utab))
        tab = mv = None
        if self._marginalize_over_inclination:
            tab, mv = e.kernel_table(rta1, self._covpts)
        out, status = e.lnlike_ensemble(
            e.f64(np.ascontiguousarray(t)), e.f64(flux[:, None, :]), e.stars_to_device(stars), diag=diag,
            conditional=not self._marginalize_over_inclination, covpts=self._covpts, tab=tab,
            meanvar=mv, rta1=rta1, temporal=self._temporal, normalized=self._normalized,
            norm_order=self._normN, zmax=self._normzmax)
        return Eager(_neg_inf_if_nan(out.cpu().numpy()))


class StarryProcessSum(StarryProcess):
    """Sum of independent processes (several spot populations on one star): the moments of
    the spherical-harmonic vectors add (sp.py:1335-1400); everything downstream -- flux mean
    and covariance, log-likelihood, prediction -- is the base class on the summed moments."""

    def __init__(self, first, second):
        if not isinstance(second, StarryProcess):
            raise AssertionError("Can only add instances of `StarryProcess` to each other.")
        for name, what in (("_ydeg", "ydeg"), ("_udeg", "udeg"), ("_normalized", "normalized"),
                           ("_marginalize_over_inclination", "marginalize_over_inclination"),
                           ("_covpts", "covpts")):
            assert getattr(first, name) == getattr(second, name), "Mismatch in `%s`." % what
        assert not first._time_variable and not second._time_variable, (
            "Sums of `StarryProcess` instances not implemented for time-variable surfaces.")
        kwargs = dict(first._kwargs)
        kwargs.pop("upstream", None)
        StarryProcess.__init__(
            self,
            mean_ylm=first._mean_ylm + second._mean_ylm,
            cov_ylm=first._cov_ylm + second._cov_ylm,
            marginalize_over_inclination=first._marginalize_over_inclination,
            normalized=first._normalized,
            covpts=first._covpts,
            **kwargs,
        )
        # hyperparameters are those of the children, not of the sum
        self._r = self._dr = self._a = self._b = self._c = self._n = None
        self._children = []
        for child in (first, second):
            self._children += getattr(child, "_children", [child])

    def log_jac(self):
        raise NotImplementedError("the latitude Jacobian is defined per child process")
