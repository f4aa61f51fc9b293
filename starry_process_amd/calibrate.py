"""
Ensemble log-probability callable, the counterpart of the reference's
``calibrate.get_log_prob`` (calibrate/log_prob.py:7-106) --
what the dynesty / emcee drivers of the reference actually call.

    log_prob = get_log_prob(t, flux, ferr=1e-3, p=1.0)
    log_prob(r, a, b, c, n)                       # same positional order as the reference

The reference compiles a Theano function of the free scalars; here the returned
callable evaluates eagerly: hyperparameters -> (mu_y, Sigma_y) on the host
(``upstream.py``), then one batched device call for all light curves, which share
one covariance and ride along as extra rows of the same factorisation (the
multi-right-hand-side fast path of sp.py:1162-1171).  Argument order of the
callable (log_prob.py:93-102):  [flux,] r, a, b, c, n [, m] [, v] [, i].

``get_log_prob_ensemble`` is the per-star generalisation (own period /
inclination / limb darkening / noise per light curve, one covariance each),
sharded over the ranks of a ``torch.distributed`` job when one is initialised.
"""
import numpy as np

from .sp import StarryProcess

__all__ = ["get_log_prob", "get_log_prob_ensemble"]


def get_log_prob(
    t,
    flux=None,
    ferr=1.0e-3,
    p=1.0,
    ydeg=15,
    baseline_log_var=0.0,
    baseline_mean=0.0,
    apply_jac=True,
    normalized=True,
    marginalize_over_inclination=True,
    u=[0.0, 0.0],
    device=None,
    upstream="reference",
):
    """``upstream``: "reference" (the reference's moment algorithm on the host, ~25-75 ms per
    call) or "device" (the same integrals by quadrature of rotations on the GPU, < 1 ms;
    see upstream_device.py for how the two compare)."""
    t = np.asarray(t, dtype=np.float64).reshape(-1)
    K = len(t)
    free_flux = flux is None
    fixed_flux = None if free_flux else np.atleast_2d(np.asarray(flux, dtype=np.float64))

    def log_prob(*args):
        args = list(args)
        fl = np.atleast_2d(np.asarray(args.pop(0), dtype=np.float64)) if free_flux else fixed_flux
        r, a, b, c, n = (float(x) for x in args[:5])
        rest = args[5:]
        m = float(rest.pop(0)) if baseline_mean is None else float(baseline_mean)
        v = float(rest.pop(0)) if baseline_log_var is None else float(baseline_log_var)
        i = float(rest.pop(0)) if not marginalize_over_inclination else 60.0
        if rest:
            raise TypeError("too many arguments")
        sp = StarryProcess(
            ydeg=ydeg, r=r, a=a, b=b, c=c, n=n, normalized=normalized,
            marginalize_over_inclination=marginalize_over_inclination, covpts=K - 1,
            # the reference callable has no z > zmax guard (log_prob.py:53-91)
            normalization_zmax=np.inf, device=device, upstream=upstream,
        )
        ll = float(sp.log_likelihood(t, fl, ferr ** 2, i=i, p=p, u=u, baseline_mean=m,
                                     baseline_var=10.0 ** v))
        if np.isnan(ll):
            ll = -np.inf
        return ll + float(sp.log_jac()) if apply_jac else ll

    return log_prob


def get_log_prob_ensemble(
    t,
    flux,
    ferr=1.0e-3,
    p=1.0,
    i=None,
    u=None,
    ydeg=15,
    baseline_log_var=0.0,
    baseline_mean=0.0,
    apply_jac=True,
    normalized=True,
    marginalize_over_inclination=True,
    covpts=None,
    device=None,
    upstream="reference",
):
    """log_prob(r, a, b, c, n) = sum over stars of per-star log-likelihoods (+ log_jac),
    each star with its own period / inclination / limb darkening / noise:
    t (K,) or (S, K); flux (S, K); ferr, p, i scalars or (S,); u (udeg,) or (S, udeg);
    or t and flux lists of S arrays of different lengths (ragged ensemble).
    Under an initialised torch.distributed job the stars are sharded over the ranks
    (one RCCL all-gather of S doubles per call); every rank returns the same value."""
    from . import ensemble

    if not isinstance(flux, (list, tuple)):     # (lists: light curves of different lengths)
        flux = np.asarray(flux, dtype=np.float64)
    S = len(flux)
    ferr2 = np.broadcast_to(np.asarray(ferr, dtype=np.float64) ** 2, (S,))

    def log_prob(r, a, b, c, n):
        kw = {} if covpts is None else {"covpts": covpts}
        sp = StarryProcess(ydeg=ydeg, r=float(r), a=float(a), b=float(b), c=float(c), n=float(n),
                           normalized=normalized,
                           marginalize_over_inclination=marginalize_over_inclination,
                           device=device, upstream=upstream, **kw)
        lnl = ensemble.sharded_log_likelihood(sp, t, flux, ferr2, i=i, p=p, u=u,
                                              baseline_mean=baseline_mean,
                                              baseline_var=10.0 ** baseline_log_var)
        ll = float(np.sum(lnl))
        if np.isnan(ll):
            ll = -np.inf
        return ll + float(sp.log_jac()) if apply_jac else ll

    return log_prob
