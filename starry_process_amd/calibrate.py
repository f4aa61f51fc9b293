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

``EnsembleLogProb`` is the same quantity for MANY hyperparameter samples at once
(the positions of all walkers / live points of an iteration -- emcee's
``vectorize=True``): the data stay on the GPU, the moments come from the device
upstream, and ``depth`` samples are kept in flight on separate streams.
"""
import numpy as np

from .sp import StarryProcess

__all__ = ["get_log_prob", "get_log_prob_ensemble", "EnsembleLogProb", "SampleBatches", "MAX_STREAMS"]

# Independent evaluations in flight on one GPU.  Four is where the throughput peaks; a fifth stream LOSES 10-25 %
# (108k against 120k evaluations/s at cfg3's shape, bench.py; EnsembleLogProb 0.584 -> 0.818 ms per sample with
# four likelihood streams + the upstream's own, tools/attic/elp_modes.py, round 5) -- next to GPU_MAX_HW_QUEUES
# (engine._want_hw_queues).  Callers' ``depth`` is clamped to it, with one warning.
MAX_STREAMS = 4
# ... except for steps that carry their upstream (SampleBatches: a third of a step's time on a stream is the samples'
# moments and tables, light kernels that leave the GPU to the other streams): 88-92k evaluations/s of one K = 1000
# light curve with four streams, 94-98k with five, 95-99k with six (round 6, one box, bench.bench_samples).
MAX_STREAMS_SAMPLES = 6
_warned_depth = [False]


def clamp_depth(depth, extra_streams=0, limit=None):
    """``depth`` likelihood streams + ``extra_streams`` others, held to MAX_STREAMS (``limit``) concurrent streams."""
    import warnings

    depth = max(1, int(depth))
    allowed = max(1, (MAX_STREAMS if limit is None else int(limit)) - int(extra_streams))
    if depth > allowed:
        if not _warned_depth[0]:
            warnings.warn("starry_process_amd: depth=%d (+%d) exceeds %d concurrent streams, beyond which the GPU's "
                          "throughput DROPS by 10-25 %% (measured); using depth=%d" % (depth, extra_streams, MAX_STREAMS,
                                                                                      allowed), RuntimeWarning, stacklevel=3)
            _warned_depth[0] = True
        depth = allowed
    return depth


class SampleBatches(object):
    """log-likelihoods of MANY hyperparameter samples for ONE planned data set, ``group`` samples per library call:

        lnl[b, s] = log_likelihood of star s under sample b = (r, a, b, c, n),        samples (ns, 5) -> (ns, S)

    The systems of a call are (sample, star) pairs, ``group`` x S of them (about 64: what fills the GPU): the samples'
    polar moments (sp_polar_moments_samples), their kernel tables (sp_kernel_table_samples) and ONE planned likelihood
    call on a replicated data plan (sp_plan_replicate) whose stars carry the table of their sample.  A single light
    curve -- how the reference is called, sp.py:1052-1062 driven by calibrate/sample.py:95-107 -- then runs at the
    rate of a 64-star ensemble instead of one latency-bound step per sample.  Marginal, normalised branch, one spot
    radius (dr = None); consecutive groups go to the slots' streams in turn."""

    def __init__(self, slots, t_dev, flux_dev, stars, rta1_dev, covpts, diag_dev=None, temporal=None, group=None,
                 norm_order=20, zmax=0.023, upstream_kwargs=None, plan=None):
        import torch

        from .engine import stars_for_samples

        self._slots = slots
        e0 = slots[0][0]
        self.S, self.K = int(t_dev.shape[0]), int(t_dev.shape[1])
        self.M = int(flux_dev.shape[1])
        self.group = max(1, int(group) if group else -(-64 // self.S))
        self._ntab = int(rta1_dev.shape[0])
        self._rta1, self._covpts = rta1_dev, int(covpts)
        self._norm_order, self._zmax = int(norm_order), float(zmax)
        self._ukw = dict(upstream_kwargs or {})
        n = self.group * self.S
        stars_d = e0.stars_to_device(stars)
        self._base_plan = plan if plan is not None else e0.plan_data(t_dev, flux_dev, stars_d, diag=diag_dev,
                                                                      covpts=self._covpts, temporal=temporal)
        self._plan = e0.replicate_plan(self._base_plan, self.group)
        self._stars = e0.stars_to_device(stars_for_samples(stars, self.group, self._ntab))
        self._buf = []
        for e, _ in slots:
            self._buf.append(dict(ws=e.workspace(n, self.K, self.M), ez=e.empty(self.group, e.N),
                                  Ez=e.empty(self.group, e.N, e.N),
                                  tab=e.empty(self.group * self._ntab, 5, self._covpts + 4),
                                  mv=e.empty(self.group * self._ntab, 2)))
            e.set_size_basis(**self._ukw)
        torch.cuda.synchronize(e0.device)

    def __call__(self, samples, out=None):
        """samples (ns, 5) -> device tensor (ns, S); nothing is synchronised: the caller does, once."""
        import torch

        samples = np.atleast_2d(np.asarray(samples, dtype=np.float64))
        ns, g, S = samples.shape[0], self.group, self.S
        ngroups = -(-ns // g)
        e0 = self._slots[0][0]
        raw = e0.empty(ngroups, g * S)
        if ns < ngroups * g:          # (the last group is filled up with its own last sample; those values are dropped)
            samples = np.vstack([samples, np.repeat(samples[-1:], ngroups * g - ns, axis=0)])
        cur = torch.cuda.current_stream(e0.device)
        start = torch.cuda.Event()
        start.record(cur)
        # (the streams' first groups start STAGGERED, each behind the upstream of the one before: started together, six
        #  streams of equal steps stay in lockstep -- all in their light upstream phase at once, then all in their
        #  factorisations at once -- and the call runs at 71k evaluations/s instead of 90-99k, bimodally from run to run)
        stagger = None
        # (at most QUEUED groups of a stream wait behind the one that runs: a host that runs hundreds of steps ahead of
        #  the GPU ends up blocked inside the runtime's launch path -- 0.8 instead of 0.16 ms of host time per step, 52-58k
        #  evaluations/s instead of 95-100k, bimodally -- so it waits HERE, for a group of three steps back)
        QUEUED = 3
        pending = [[] for _ in self._slots]
        for gi in range(ngroups):
            k = gi % len(self._slots)
            (e, stream), b = self._slots[k], self._buf[k]
            if len(pending[k]) >= QUEUED:
                pending[k].pop(0).synchronize()
            with torch.cuda.stream(stream):
                if gi < len(self._slots):
                    stream.wait_event(start)
                    if stagger is not None:
                        stream.wait_event(stagger)
                e.polar_moments_samples(samples[gi * g:(gi + 1) * g], ez=b["ez"], Ez=b["Ez"], **self._ukw)
                e.kernel_table_samples(b["ez"], b["Ez"], self._rta1, self._covpts, tab=b["tab"], meanvar=b["mv"])
                if gi + 1 < min(ngroups, len(self._slots)):
                    stagger = torch.cuda.Event()
                    stagger.record(stream)
                e.lnlike_ensemble_planned(self._plan, None, None, self._stars, b["tab"], b["mv"],
                                          norm_order=self._norm_order, zmax=self._zmax, out=raw[gi], workspace=b["ws"])
                if ngroups > QUEUED * len(self._slots):
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    pending[k].append(ev)
        for k in range(min(ngroups, len(self._slots))):
            done = torch.cuda.Event()
            done.record(self._slots[k][1])
            cur.wait_event(done)
        return raw.view(ngroups * g, S)[:ns]


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


class EnsembleLogProb(object):
    """log_prob for a batch of hyperparameter samples, same value per sample as
    ``get_log_prob_ensemble(..., upstream="device")``.

        lp = EnsembleLogProb(t, flux, ferr=1e-3, p=periods)          # data -> GPU, once
        values = lp(samples)                                          # samples (n, 5): r, a, b, c, n

    Per sample: moments by quadrature on the device (upstream_device.py) -> kernel table ->
    one batched likelihood call for this rank's stars; nothing is copied back or synchronised
    until every sample is enqueued, and sample k runs on stream k mod ``depth`` with its own
    library handle and workspace (engine.engine_slots), so that the latency-bound phases of
    one sample overlap the throughput-bound phases of its neighbours.  The moments of ALL samples
    are produced on one more stream with a handle of its own, ahead of the likelihood streams
    (an event per sample): the quadrature is a chain of a dozen small kernels, 0.3-0.4 ms of
    latency per sample that a likelihood stream would otherwise sit through with its share of
    the GPU idle: 0.726 -> 0.69 ms per sample with three likelihood streams + this one (a fifth stream in
    flight loses more than it hides: 0.83 -- the same cliff bench.py sees at five steps in flight; round 5, one box,
    tools/attic/elp_modes.py: 0.584 as shipped, 0.603 with the moments on the three likelihood streams themselves
    (upstream_stream=False), 0.699 on four of them, 0.818 with four + the upstream's own).  Under an initialised
    ``torch.distributed`` job the stars are sharded over the ranks and the per-sample sums are
    combined with ONE all-reduce for the whole batch."""

    def __init__(self, t, flux, ferr=1.0e-3, p=1.0, i=None, u=None, ydeg=15, baseline_log_var=0.0,
                 baseline_mean=0.0, apply_jac=True, normalized=True,
                 marginalize_over_inclination=True, covpts=None, device=None, depth=3, upstream_stream=True,
                 batch_samples=True, out_of_bounds="raise"):
        import torch
        import torch.distributed as dist

        from . import ensemble
        from .defaults import defaults
        from .engine import engine_slots, make_stars

        flux = np.asarray(flux, dtype=np.float64)
        if flux.ndim != 2:
            raise ValueError("flux must be (S, K); ragged ensembles: get_log_prob_ensemble")
        S, K = flux.shape
        self._dist = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if self._dist else (0, 1)
        lo, hi = ensemble.shard_bounds(S, rank, world)
        self.S, self.K, self._n_local = S, K, hi - lo
        udeg = defaults["udeg"]
        t = np.asarray(t, dtype=np.float64)
        t = np.broadcast_to(t, (S, K)) if t.ndim == 1 else t
        per = lambda x, d: np.broadcast_to(np.asarray(d if x is None else x, dtype=np.float64), (S,))[lo:hi]
        pp, ii = per(p, defaults["p"]), per(i, defaults["i"])
        if np.any(pp < -1e-6):
            raise ValueError("p out of bounds")
        if np.any(ii * np.pi / 180 < -1e-6) or np.any(ii * np.pi / 180 > 0.5 * np.pi + 1e-6):
            raise ValueError("i out of bounds")
        uu = np.asarray(defaults["u"][:udeg] if u is None else u, dtype=np.float64)
        if uu.ndim == 1:
            utab, table = uu[None, :udeg], np.zeros(hi - lo, dtype=np.int32)
        else:
            utab, table = np.unique(uu[lo:hi, :udeg], axis=0, return_inverse=True)
            table = table.astype(np.int32).reshape(-1)
        stars = make_stars(hi - lo, period=pp, inc_deg=ii, tau=0.0,
                           baseline_var=np.full(hi - lo, 10.0 ** baseline_log_var),
                           baseline_mean=per(baseline_mean, 0.0),
                           data_var=per(np.asarray(ferr, dtype=np.float64) ** 2, 1.0), table=table)
        # (depth likelihood streams + the upstream's own: never more than MAX_STREAMS concurrent streams)
        depth = clamp_depth(depth, 1)
        slots = engine_slots(ydeg, udeg, device, max(2, depth + 1))
        self._slots, self._up = slots[:-1], slots[-1]          # likelihood slots; the upstream's own engine + stream
        self._upstream_stream = bool(upstream_stream)          # False: a sample's moments on its likelihood stream
        e0 = self._slots[0][0]
        self._t = e0.f64(np.ascontiguousarray(t[lo:hi]))
        self._flux = e0.f64(np.ascontiguousarray(flux[lo:hi, None, :]))
        self._stars = e0.stars_to_device(stars)
        self._rta1 = e0.f64(e0.rTA1L(utab))
        self._ws = [e.workspace(max(hi - lo, 1), K, 1) for e, _ in self._slots]
        self._kw = dict(conditional=not marginalize_over_inclination, normalized=bool(normalized),
                        covpts=defaults["covpts"] if covpts is None else int(covpts))
        self._marg = bool(marginalize_over_inclination)
        self._ydeg, self._apply_jac = int(ydeg), bool(apply_jac)
        # The data are fixed from here on (calibrate/log_prob.py:7-55 fixes them the same way): what depends on them
        # alone -- phases, the kernel table's weights in the covariance's sum, the sums of the flux -- is taken once
        # (sp_plan_data), and every sample goes through the planned call: no pass over the K^2 entries of every
        # star's covariance before its factorisation.  One plan, read-only, shared by the slots.
        if out_of_bounds not in ("raise", "inf"):
            raise ValueError("out_of_bounds must be 'raise' or 'inf'")
        # "inf": a sample outside the reference's bounds (which raise ValueError, ops/exceptions.py:30-48) is answered
        # with -inf and not evaluated -- what a sampler's walkers need when they step out of the prior box
        self._oob = out_of_bounds
        self._plan = self._batch = None
        if self._marg and normalized and hi - lo > 0 and K >= 2:
            from ._lib import SPError

            try:
                self._plan = e0.plan_data(self._t, self._flux, self._stars, covpts=self._kw["covpts"], workspace=self._ws[0])
            except SPError:
                # (a shape the planned step does not serve -- e.g. a lag grid beyond its LDS budget, covpts > ~4 700:
                #  the unplanned call has the fallbacks)
                self._plan = None
        # Samples go through SampleBatches: packed ceil(64 / S) to a library call when there are fewer than 64 stars (one
        # light curve above all) -- the GPU sees 64 systems per step whatever S is --, one sample per call otherwise; either
        # way the sample's moments come from sp_polar_moments_samples, with no host arithmetic (the per-sample upstream
        # spends 0.5 ms of NumPy on the size integral and the Gauss-Jacobi rule whenever r or (a, b) change).
        if self._plan is not None and batch_samples:
            more = engine_slots(ydeg, udeg, device, 2) if len(self._slots) + 1 + 2 <= MAX_STREAMS_SAMPLES else []
            self._batch = SampleBatches(self._slots + [self._up] + more, self._t, self._flux, stars, self._rta1,
                                        self._kw["covpts"], plan=self._plan, zmax=0.023)
        torch.cuda.synchronize(e0.device)

    def __call__(self, samples):
        import torch
        import torch.distributed as dist

        from .upstream import log_jac
        from .upstream_device import ylm_moments_device

        samples = np.atleast_2d(np.asarray(samples, dtype=np.float64))
        if samples.shape[1] != 5:
            raise ValueError("samples must be (n, 5): r, a, b, c, n")
        if self._oob == "inf":
            from .engine import samples_in_bounds

            ok = samples_in_bounds(samples)
            if not ok.all():
                out = np.full(samples.shape[0], -np.inf)
                if ok.any():
                    out[ok] = self(samples[ok])
                return out
        ns, nl = samples.shape[0], self._n_local
        e0 = self._slots[0][0]
        outs = e0.empty(ns, max(nl, 1))
        torch.cuda.synchronize(e0.device)
        if nl and self._batch is not None:
            outs = self._batch(samples)
        elif nl:
            eu, su = self._up
            keep = []                                   # (the moments stay alive until the batch is done)
            for k, (r, a, b, c, n) in enumerate(samples):
                e, stream = self._slots[k % len(self._slots)]
                if self._upstream_stream:
                    with torch.cuda.stream(su):
                        mean, cov = ylm_moments_device(eu, r=r, a=a, b=b, c=c, n=n)
                        ready = torch.cuda.Event()
                        ready.record(su)
                    keep.append((mean, cov, ready))
                with torch.cuda.stream(stream):
                    if self._upstream_stream:
                        stream.wait_event(ready)
                    else:
                        mean, cov = ylm_moments_device(e, r=r, a=a, b=b, c=c, n=n)
                        keep.append((mean, cov))
                    e.set_moments_dev(mean, cov)
                    tab = mv = None
                    if self._marg:
                        tab, mv = e.kernel_table(self._rta1, self._kw["covpts"])
                    if self._plan is not None:
                        e.lnlike_ensemble_planned(self._plan, self._t, self._flux, self._stars, tab, mv, out=outs[k],
                                                  workspace=self._ws[k % len(self._slots)])
                    else:
                        e.lnlike_ensemble(self._t, self._flux, self._stars, tab=tab, meanvar=mv,
                                          rta1=self._rta1, out=outs[k], workspace=self._ws[k % len(self._slots)],
                                          **self._kw)
        torch.cuda.synchronize(e0.device)
        vals = outs[:, :nl]
        vals = torch.where(torch.isnan(vals), torch.full_like(vals, -float("inf")), vals)
        total = vals.sum(dim=1)
        if self._dist:
            # -inf + finite = -inf survives the sum; a NaN cannot appear (no +inf terms)
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        total = total.cpu().numpy()
        if self._apply_jac:
            from .upstream import log_jac_samples

            total = total + log_jac_samples(samples[:, 1], samples[:, 2])
        return total
