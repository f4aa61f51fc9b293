"""
One ``Engine`` per GPU: owns the ``sp_handle`` and turns torch CUDA tensors into
the raw device pointers the C ABI takes.  PyTorch is plumbing here (HBM
buffers, streams, torch.distributed); every number is produced by the HIP
kernels in ``csrc/``.
"""
import ctypes
import os

import numpy as np

from . import _lib
from . import hostconst
from ._lib import STAR_DTYPE, TEMPORAL, SPError, c_void_p, check, hptr

__all__ = ["Engine", "DataPlan", "get_engine", "make_stars", "stars_for_samples", "sample_parameters", "samples_in_bounds"]


def _torch():
    import torch

    return torch


def make_stars(S, period=1.0, inc_deg=60.0, tau=0.0, baseline_var=0.0,
               baseline_mean=0.0, data_var=0.0, table=0, nobs=0):
    """Structured host array of ``sp_star`` (inclination converted to radians,
    flux.py:236-238).  ``nobs``: valid cadences per star for ragged ensembles (0 = all)."""
    st = np.zeros(S, dtype=STAR_DTYPE)
    st["period"] = period
    st["inc"] = np.asarray(inc_deg, dtype=float) * (np.pi / 180)
    st["tau"] = tau
    st["baseline_var"] = baseline_var
    st["baseline_mean"] = baseline_mean
    st["data_var"] = data_var
    st["table"] = table
    st["nobs"] = nobs
    return st


def stars_for_samples(stars, B, ntab):
    """The sp_star array of a batch of B hyperparameter samples x S stars (sample-major: system b S + s): the S stars
    repeated B times with table = b ntab + table_s, the kernel table of sample b for the star's flux operator
    (sp_kernel_table_samples' numbering)."""
    stars = np.ascontiguousarray(stars)
    assert stars.dtype == STAR_DTYPE
    out = np.tile(stars, int(B))
    out["table"] = (np.repeat(np.arange(int(B), dtype=np.int64), stars.shape[0]) * int(ntab) + out["table"]).astype(np.int32)
    return out


def samples_in_bounds(samples, tol=1e-6):
    """Boolean mask of the rows of samples [B, 5] = (r [degrees], a, b, c, n) inside the reference's bounds (r in [0, 90],
    a, b in [0, 1], n >= 0, everything finite; size.py:68, latitude.py:176-197, contrast.py:21-33 through CheckBoundsOp's
    tolerance): what ``sample_parameters`` raises ValueError for.  A sampler's walkers leave the box; the log-probability
    callables can answer -inf for such rows instead of raising (``out_of_bounds="inf"``)."""
    sm = np.atleast_2d(np.asarray(samples, dtype=np.float64))
    r, a, b, n = sm[:, 0] * (np.pi / 180), sm[:, 1], sm[:, 2], sm[:, 4]
    ok = np.all(np.isfinite(sm), axis=1)
    ok &= (r >= -tol) & (r <= 0.5 * np.pi + tol) & (a >= -tol) & (a <= 1 + tol) & (b >= -tol) & (b <= 1 + tol) & (n >= -tol)
    return ok


def sample_parameters(samples, **kw):
    """samples [B, 5] = (r [degrees], a, b, c, n) -> [B, 5] = (r [radians], alpha, beta, c, n), what
    sp_polar_moments_samples takes: the reference's bounds (size.py:68, latitude.py:176-197, contrast.py:21-33 through
    CheckBoundsOp: ValueError outside, tolerance 1e-6) and its (a, b) -> (alpha, beta) map, for the whole batch at once
    (NumPy; one sample at a time ``upstream.ab_to_alphabeta`` does the same)."""
    from .defaults import defaults

    sm = np.array(np.atleast_2d(np.asarray(samples, dtype=np.float64)), dtype=np.float64)
    if sm.ndim != 2 or sm.shape[1] != 5:
        raise ValueError("samples must be (B, 5): r, a, b, c, n")
    r, a, b, n = sm[:, 0] * (np.pi / 180), sm[:, 1], sm[:, 2], sm[:, 4]
    from .ops import CheckBoundsOp

    for name, v, lo, hi in (("r", r, 0.0, 0.5 * np.pi), ("a", a, 0.0, 1.0), ("b", b, 0.0, 1.0), ("n", n, 0.0, np.inf)):
        CheckBoundsOp(name=name, lower=lo, upper=hi)(v)
    if not np.all(np.isfinite(sm)):
        raise ValueError("samples must be finite")
    abmin = kw.get("abmin", defaults["abmin"])
    lam = kw.get("log_alpha_max", defaults["log_alpha_max"])
    lbm = kw.get("log_beta_max", defaults["log_beta_max"])
    a, b = np.maximum(a, abmin), np.maximum(b, abmin)
    out = np.empty_like(sm)
    out[:, 0] = np.clip(r, 0.0, None)
    out[:, 1] = np.exp(a * lam)
    out[:, 2] = np.exp(np.log(0.5) + b * (lbm - np.log(0.5)))
    out[:, 3] = sm[:, 3]
    out[:, 4] = np.clip(n, 0.0, None)
    return np.ascontiguousarray(out)


_STAGE_BYTES, _STAGE_SLOTS = 1 << 16, 16   # pinned staging ring of Engine.dev (small uploads)


class Engine(object):
    def __init__(self, ydeg=15, udeg=2, device=0):
        torch = _torch()
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise SPError("no MI355X visible to PyTorch: the hot path has no CPU fallback")
        self.ydeg, self.udeg, self.device_index = int(ydeg), int(udeg), int(device)
        self.device = torch.device("cuda", self.device_index)
        self.N = (self.ydeg + 1) ** 2
        self.NWIG = ((self.ydeg + 1) * (2 * self.ydeg + 1) * (2 * self.ydeg + 3)) // 3
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)  # make sure the context exists
        h = c_void_p()
        check(L.sp_create(self.ydeg, self.udeg, self.device_index, ctypes.byref(h)))
        self._h = h
        self._L = L
        wnp, Wnp = hostconst.marginal_constants(self.ydeg)
        wnp = np.ascontiguousarray(wnp)
        Wnp = np.ascontiguousarray(Wnp)
        check(L.sp_set_marginal_constants(self._h, hptr(wnp), hptr(Wnp)))
        self._moments_id = None
        self._moments_owner = None
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.sp_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------
    def _stream(self):
        return c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def dev(self, a, dtype=None):
        """Host array / tensor -> contiguous tensor on this GPU."""
        torch = _torch()
        if isinstance(a, torch.Tensor):
            # (already here, of the right type and packed: the usual case inside a chain of ops --
            #  three no-op tensor calls cost 30 us of host time, 0.3 ms per upstream evaluation)
            if a.device == self.device and (dtype is None or a.dtype == dtype) and a.is_contiguous():
                return a
            t = a.to(self.device)
            if dtype is not None:
                t = t.to(dtype)
            return t.contiguous()
        a = np.ascontiguousarray(a)
        if dtype is not None and 0 < a.nbytes <= _STAGE_BYTES and a.dtype == np.float64 and dtype == torch.float64:
            return self._upload_small(a)
        t = torch.from_numpy(a).to(self.device)
        if dtype is not None:
            t = t.to(dtype)
        return t.contiguous()

    def _upload_small(self, a):
        """Small fp64 host array -> device through a ring of pinned staging buffers: the copy is
        enqueued on the current stream and the call returns (a pageable source makes the runtime
        stage and wait: 50 us per upload, four uploads per upstream evaluation).  A slot is reused
        only after the copy that last read it has completed (its event)."""
        torch = _torch()
        ring = self.__dict__.get("_stage_ring")
        if ring is None:
            ring = self._stage_ring = {"buf": [torch.empty(_STAGE_BYTES // 8, dtype=torch.float64).pin_memory()
                                               for _ in range(_STAGE_SLOTS)],
                                       "ev": [None] * _STAGE_SLOTS, "next": 0}
        k = ring["next"]
        ring["next"] = (k + 1) % _STAGE_SLOTS
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()
        n = a.size
        src = ring["buf"][k][:n]
        src.numpy()[...] = a.reshape(-1)
        out = torch.empty(a.shape, dtype=torch.float64, device=self.device)
        out.view(-1).copy_(src, non_blocking=True)
        ev = ring["ev"][k]
        if ev is None:
            ev = ring["ev"][k] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return out

    def f64(self, a):
        return self.dev(a, _torch().float64)

    def empty(self, *shape):
        torch = _torch()
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    def stars_to_device(self, stars):
        torch = _torch()
        stars = np.ascontiguousarray(stars)
        assert stars.dtype == STAR_DTYPE
        raw = torch.from_numpy(stars.view(np.uint8).reshape(-1).copy())
        return raw.to(self.device)

    @staticmethod
    def _p(t):
        return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)

    def synchronize(self):
        check(self._L.sp_stream_synchronize(self._h, self._stream()))

    # -- ops ------------------------------------------------------
    def Rx(self, thetas, deriv=True):
        thetas = np.ascontiguousarray(np.atleast_1d(np.asarray(thetas, dtype=np.float64)))
        n = thetas.shape[0]
        R = self.empty(n, self.NWIG)
        dR = self.empty(n, self.NWIG) if deriv else None
        check(self._L.sp_Rx(self._h, hptr(thetas), n, self._p(R), self._p(dR), self._stream()))
        return R, dR

    def dotRx(self, M, Rpacked):
        """M [rows, N] or [B, rows, N]; Rpacked [NWIG] or [B, NWIG]."""
        M = self.f64(M)
        Rpacked = self.f64(Rpacked)
        batched = M.dim() == 3
        Mb = M if batched else M.unsqueeze(0)
        B, rows, N = Mb.shape
        assert N == self.N
        strideR = self.NWIG if Rpacked.dim() == 2 else 0
        out = self.empty(B, rows, N)
        check(self._L.sp_dotRx(self._h, self._p(Mb), rows * N, N, 1, rows,
                               self._p(Rpacked), strideR, self._p(out), B, self._stream()))
        return out if batched else out[0]

    def tensordotRz(self, M, theta):
        M = self.f64(M)
        theta = self.f64(theta).reshape(-1)
        K = theta.shape[0]
        assert M.shape == (K, self.N)
        f = self.empty(K, self.N)
        check(self._L.sp_tensordotRz(self._h, self._p(M), self._p(theta), K, self._p(f), self._stream()))
        return f

    def special_tensordotRz(self, T, M, theta):
        T = self.f64(T)
        M = self.f64(M)
        theta = self.f64(theta).reshape(-1)
        assert T.shape == (self.N, self.N) and M.shape == (self.N, self.N)
        K = theta.shape[0]
        f = self.empty(K)
        check(self._L.sp_special_tensordotRz(self._h, self._p(T), self._p(M), self._p(theta), K, self._p(f), self._stream()))
        return f

    def tensordotRz_rev(self, M, theta, bf):
        """Reverse mode of tensordotRz: (bM [K, N], btheta [K])."""
        M, bf = self.f64(M), self.f64(bf)
        theta = self.f64(theta).reshape(-1)
        K = theta.shape[0]
        assert M.shape == (K, self.N) and bf.shape == (K, self.N)
        bM, bth = self.empty(K, self.N), self.empty(K)
        check(self._L.sp_tensordotRz_rev(self._h, self._p(M), self._p(theta), K, self._p(bf),
                                         self._p(bM), self._p(bth), self._stream()))
        return bM, bth

    def special_tensordotRz_rev(self, T, M, theta, bf):
        """Reverse mode of special_tensordotRz: (bM [N, N], btheta [K])."""
        T, M = self.f64(T), self.f64(M)
        theta, bf = self.f64(theta).reshape(-1), self.f64(bf).reshape(-1)
        K = theta.shape[0]
        assert T.shape == (self.N, self.N) and M.shape == (self.N, self.N) and bf.shape == (K,)
        bM, bth = self.empty(self.N, self.N), self.empty(K)
        check(self._L.sp_special_tensordotRz_rev(self._h, self._p(T), self._p(M), self._p(theta), K,
                                                 self._p(bf), self._p(bM), self._p(bth), self._stream()))
        return bM, bth

    def rTA1L_rev(self, u, bf):
        """Reverse mode of rTA1L: bu [udeg] (host)."""
        u = np.ascontiguousarray(np.asarray(u, dtype=np.float64).reshape(-1)[: self.udeg])
        bf = np.ascontiguousarray(np.asarray(bf, dtype=np.float64).reshape(-1))
        if u.shape[0] != self.udeg or bf.shape[0] != self.N:
            raise ValueError("Vector `u` or `bf` has the wrong size.")
        bu = np.empty(self.udeg)
        check(self._L.sp_rTA1L_rev(self._h, hptr(u), hptr(bf), hptr(bu)))
        return bu

    def rTA1(self):
        out = np.empty(self.N)
        check(self._L.sp_rTA1(self._h, hptr(out)))
        return out

    def rTA1L(self, u):
        """u: (udeg,) or (nsets, udeg) -> (nsets, N); udeg = 0 gives rTA1."""
        if self.udeg == 0:
            return self.rTA1()[None, :].copy()
        u = np.asarray(u, dtype=np.float64)
        u = np.ascontiguousarray(u.reshape(-1, u.shape[-1])[:, : self.udeg])
        if u.shape[1] != self.udeg:
            raise ValueError("Vector `u` has the wrong size.")
        n = u.shape[0]
        out = np.empty((n, self.N))
        check(self._L.sp_rTA1L(self._h, hptr(u), n, hptr(out)))
        return out

    # -- moments / kernel table ------------------------------------------------
    def set_moments(self, mean_ylm, cov_ylm):
        mean_ylm = np.ascontiguousarray(np.asarray(mean_ylm, dtype=np.float64).reshape(-1))
        cov_ylm = np.ascontiguousarray(np.asarray(cov_ylm, dtype=np.float64))
        assert mean_ylm.shape == (self.N,) and cov_ylm.shape == (self.N, self.N)
        # whoever bound its moments before (flux.FluxIntegral._bind) no longer owns the resident set
        self._moments_owner = None
        check(self._L.sp_set_ylm_moments(self._h, hptr(mean_ylm), hptr(cov_ylm)))

    def set_moments_dev(self, mean_ylm, cov_ylm):
        """Same with the moments already on the device (asynchronous)."""
        assert mean_ylm.is_cuda and cov_ylm.is_cuda
        self._moments_owner = None
        check(self._L.sp_set_ylm_moments_dev(self._h, self._p(mean_ylm), self._p(cov_ylm), self._stream()))

    PROF_KINDS = {"syrk": 0, "chain": 2, "panels": 4, "panel_launch": 5}

    def profile_begin(self, max_launches, kinds=("syrk",)):
        """Bracket the factorisation's launches of the given kinds with HIP events on their stream
        ("syrk" trailing updates, "chain" / "panel_launch" every panel launch under its own pair,
        "panels" the panel launches of a super-panel under ONE pair of events)."""
        mask = 0
        for k in kinds:
            mask |= 1 << self.PROF_KINDS[k]
        check(self._L.sp_profile_begin_kinds(self._h, int(max_launches), mask))

    def profile_kind(self, kind, padded=False):
        """(launches, summed milliseconds, summed algorithmic flops) of one kind; ends the profile.  padded: a fourth
        value, the flops on the padded system the launches execute (sp_profile_kind_ex)."""
        n = ctypes.c_long()
        ms = ctypes.c_double()
        fl = ctypes.c_double()
        flp = ctypes.c_double()
        check(self._L.sp_profile_kind_ex(self._h, self.PROF_KINDS[kind], ctypes.byref(n), ctypes.byref(ms),
                                         ctypes.byref(fl), ctypes.byref(flp)))
        return (n.value, ms.value, fl.value, flp.value) if padded else (n.value, ms.value, fl.value)

    def profile_end(self):
        n = ctypes.c_long()
        ms = ctypes.c_double()
        fl = ctypes.c_double()
        check(self._L.sp_profile_end(self._h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
        return n.value, ms.value, fl.value

    def polar_moments(self):
        ez = np.empty(self.N)
        Ez = np.empty((self.N, self.N))
        check(self._L.sp_get_polar_moments(self._h, hptr(ez), hptr(Ez)))
        return ez, Ez

    def kernel_table(self, rta1, covpts):
        """rta1 [ntab, N] (host or device) -> tab [ntab, 5, covpts+4], meanvar [ntab, 2]."""
        rta1 = self.f64(np.atleast_2d(rta1) if not hasattr(rta1, "dim") else rta1)
        ntab = rta1.shape[0]
        _, xp = hostconst.lag_grid(int(covpts))
        xp = np.ascontiguousarray(xp)
        assert xp.shape[0] == covpts + 4
        tab = self.empty(ntab, 5, covpts + 4)
        mv = self.empty(ntab, 2)
        check(self._L.sp_kernel_table(self._h, self._p(rta1), ntab, int(covpts), hptr(xp),
                                      self._p(tab), self._p(mv), self._stream()))
        return tab, mv

    # -- hyperparameter samples in batches (round 6) -------------------------------
    def set_size_basis(self, **kw):
        """Hands the spot profile's basis (size.py:9-47; upstream._spot_basis) to the library once per engine
        (sp_set_size_basis): what sp_polar_moments_samples integrates the sigmoid profile against."""
        from .defaults import defaults
        from .upstream import _spot_basis

        skw = {k: kw[k] for k in ("spts", "eps4", "smoothing") if k in kw}
        sfac = float(kw.get("sfac", 300))
        key = (tuple(sorted(skw.items())), sfac)
        if self.__dict__.get("_size_basis_key") == key:
            return
        theta, Bp, _ = _spot_basis(self.ydeg, **skw)
        theta, Bp = np.ascontiguousarray(theta, dtype=np.float64), np.ascontiguousarray(Bp, dtype=np.float64)
        assert Bp.shape == (self.ydeg + 1, theta.shape[0])
        check(self._L.sp_set_size_basis(self._h, hptr(theta), hptr(Bp), int(theta.shape[0]), sfac))
        self._size_basis_key = key

    def polar_moments_samples(self, samples, ez=None, Ez=None, **kw):
        """samples [B, 5] = (r [degrees], a, b, c, n) per row, the argument order of the reference's log-probability
        (calibrate/log_prob.py:93-102) -> (ez [B, N], Ez [B, N, N]) device tensors: the polar-frame moments of B
        hyperparameter samples in one library call (sp_polar_moments_samples).  Bounds are the reference's
        (ValueError before anything is launched)."""
        from .defaults import defaults

        sm = sample_parameters(samples, **kw)
        B = sm.shape[0]
        self.set_size_basis(**kw)
        if ez is None:
            ez = self.empty(B, self.N)
        if Ez is None:
            Ez = self.empty(B, self.N, self.N)
        assert tuple(ez.shape) == (B, self.N) and tuple(Ez.shape) == (B, self.N, self.N)
        check(self._L.sp_polar_moments_samples(
            self._h, B, hptr(sm), float(kw.get("epsy", defaults["epsy"])), float(kw.get("epsy15", defaults["epsy15"])),
            self._p(ez), self._p(Ez), self._stream()))
        return ez, Ez

    def kernel_table_samples(self, ez, Ez, rta1, covpts, tab=None, meanvar=None):
        """ez [B, N], Ez [B, N, N], rta1 [ntab, N] (device) -> tab [B ntab, 5, covpts + 4], meanvar [B ntab, 2]: table
        b ntab + i belongs to sample b and flux operator i (sp_kernel_table_samples)."""
        B = ez.shape[0]
        ntab = rta1.shape[0]
        xp = self.__dict__.setdefault("_xp_cache", {}).get(int(covpts))
        if xp is None:
            xp = self._xp_cache[int(covpts)] = np.ascontiguousarray(hostconst.lag_grid(int(covpts))[1])
        if tab is None:
            tab = self.empty(B * ntab, 5, covpts + 4)
        if meanvar is None:
            meanvar = self.empty(B * ntab, 2)
        check(self._L.sp_kernel_table_samples(self._h, B, self._p(ez), self._p(Ez), self._p(rta1), ntab, int(covpts),
                                              hptr(xp), self._p(tab), self._p(meanvar), self._stream()))
        return tab, meanvar

    # -- covariances -----------------------------------------------------------
    def cov_marginal(self, t, stars, covpts, tab, meanvar, temporal=None,
                     normalized=True, norm_order=20):
        t = self.f64(t)
        S, K = t.shape
        sd = self.stars_to_device(stars)
        cov = self.empty(S, K, K)
        z = self.empty(S)
        check(self._L.sp_cov_marginal_batched(
            self._h, S, K, self._p(t), self._p(sd), int(covpts), self._p(tab),
            self._p(meanvar), TEMPORAL[temporal], int(bool(normalized)), int(norm_order),
            self._p(cov), K, K * K, self._p(z), self._stream()))
        return cov, z

    def design_matrix(self, t, stars, rta1):
        t = self.f64(t)
        S, K = t.shape
        sd = self.stars_to_device(stars)
        rta1 = self.f64(rta1)
        A = self.empty(S, K, self.N)
        check(self._L.sp_design_matrix(self._h, S, K, self._p(t), self._p(sd),
                                       self._p(rta1), self._p(A), self._stream()))
        return A

    def cov_conditional(self, t, stars, rta1, temporal=None, normalized=True,
                        norm_order=20):
        t = self.f64(t)
        S, K = t.shape
        sd = self.stars_to_device(stars)
        rta1 = self.f64(rta1)
        cov = self.empty(S, K, K)
        mean = self.empty(S)
        z = self.empty(S)
        check(self._L.sp_cov_conditional_batched(
            self._h, S, K, self._p(t), self._p(sd), self._p(rta1), TEMPORAL[temporal],
            int(bool(normalized)), int(norm_order), self._p(cov), K, K * K,
            self._p(mean), self._p(z), self._stream()))
        return cov, mean, z

    # -- linear algebra ----------------------------------------------------------
    def cho_factor(self, A):
        """Lower Cholesky factor(s); A [K, K] or [B, K, K] (not modified)."""
        torch = _torch()
        A = self.f64(A).clone()
        Ab = A if A.dim() == 3 else A.unsqueeze(0)
        B, K, _ = Ab.shape
        info = torch.zeros(B, dtype=torch.int32, device=self.device)
        check(self._L.sp_cho_factor(self._h, self._p(Ab), K, K, K * K, B, self._p(info), self._stream()))
        return (Ab if A.dim() == 3 else Ab[0]), info

    def spd_inverse(self, C, workspace=None, full=True):
        """(C^-1, log det C, info) of symmetric positive definite C [K, K] or [B, K, K] by the factorisation's own
        machinery (sp_spd_inverse_batched: the identity rides through the blocked Cholesky, C^-1 = L^-T L^-1 on
        the matrix cores).  full=False returns the library's raw output: [B, Kr, Kr] (Kr = K rounded up to 64)
        with the LOWER 64 x 64 tiles valid -- what the reverse sweep of the likelihood reads (grad.py)."""
        torch = _torch()
        C = self.f64(C)
        Cb = (C if C.dim() == 3 else C.unsqueeze(0)).contiguous()
        B, K, _ = Cb.shape
        Kr = (K + 63) // 64 * 64
        nbytes = int(self._L.sp_spd_inverse_workspace_bytes(self._h, B, K))
        ws = workspace
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        out = torch.zeros(B, Kr, Kr, dtype=torch.float64, device=self.device)
        logdet = self.empty(B)
        info = torch.zeros(B, dtype=torch.int32, device=self.device)
        check(self._L.sp_spd_inverse_batched(self._h, B, K, self._p(Cb), K, K * K, self._p(out), self._p(logdet),
                                             self._p(info), self._p(ws), self._stream()))
        if not full:
            return out, logdet, info
        low = torch.tril(out[:, :K, :K])
        inv = low + torch.tril(low, -1).transpose(1, 2)
        return (inv if C.dim() == 3 else inv[0]), (logdet if C.dim() == 3 else logdet[0]), info

    def cho_solve(self, L, b):
        """(L L^T)^-1 b; L [K, K] or [B, K, K]; b [K], [K, M] or [B, K, M]."""
        L = self.f64(L)
        b = self.f64(b).clone()
        Lb = L if L.dim() == 3 else L.unsqueeze(0)
        B, K, _ = Lb.shape
        shape = b.shape
        bb = b.reshape(B, K, -1).contiguous()
        nrhs = bb.shape[2]
        check(self._L.sp_cho_solve(self._h, self._p(Lb), K, K, K * K, self._p(bb), nrhs, B, self._stream()))
        return bb.reshape(shape)

    def set_lazy_cov(self, on):
        """Covariance tiles formed at first touch by the factorisation (default on; effective under
        the deferred normalisation, without a temporal kernel)."""
        check(self._L.sp_set_lazy_cov(self._h, int(bool(on))))

    def set_defer_norm(self, on):
        """True (default): normalised likelihoods assemble the raw covariance once and apply the
        normalisation's rank-2 part to the result; False: separate row-sum pass (sp_set_defer_norm).
        Invalidates the cached workspace size."""
        check(self._L.sp_set_defer_norm(self._h, int(bool(on))))
        self._ws = None

    def tri_solve(self, L, b, trans=False):
        """L^-1 b (trans False) or L^-T b (trans True); shapes as in cho_solve."""
        L = self.f64(L)
        b = self.f64(b).clone()
        Lb = L if L.dim() == 3 else L.unsqueeze(0)
        B, K, _ = Lb.shape
        shape = b.shape
        bb = b.reshape(B, K, -1).contiguous()
        check(self._L.sp_tri_solve(self._h, self._p(Lb), K, K, K * K, self._p(bb), bb.shape[2], B,
                                   int(bool(trans)), self._stream()))
        return bb.reshape(shape)

    def solve_rev(self, L, c, c_bar, trans=False):
        """Reverse mode of c = A^-1 b, A = L or L^T: returns (A_bar, b_bar)."""
        L = self.f64(L)
        Lb = (L if L.dim() == 3 else L.unsqueeze(0)).contiguous()
        B, K, _ = Lb.shape
        c = self.f64(c)
        shape = c.shape
        cb = c.reshape(B, K, -1).contiguous()
        gb = self.f64(c_bar).reshape(B, K, -1).contiguous()
        nrhs = cb.shape[2]
        Abar = self.empty(B, K, K)
        bbar = self.empty(B, K, nrhs)
        check(self._L.sp_solve_rev(self._h, self._p(Lb), K, K, K * K, self._p(cb), self._p(gb), nrhs,
                                   B, int(bool(trans)), self._p(Abar), self._p(bbar), self._stream()))
        return (Abar if L.dim() == 3 else Abar[0]), bbar.reshape(shape)

    def cholesky_rev(self, L, L_bar):
        """Reverse mode of L = cholesky(C): C_bar from L and L_bar ([K, K] or [B, K, K])."""
        L = self.f64(L)
        Lb = (L if L.dim() == 3 else L.unsqueeze(0)).contiguous()
        B, K, _ = Lb.shape
        gb = self.f64(L_bar).reshape(B, K, K).contiguous()
        out = self.empty(B, K, K)
        check(self._L.sp_cholesky_rev(self._h, self._p(Lb), K, K, K * K, self._p(gb), B, self._p(out),
                                      self._stream()))
        return out if L.dim() == 3 else out[0]

    def gemm_nt(self, A, B, C=None, alpha=1.0, lower_only=False):
        """alpha A B^T (+ C): A [M, K], B [N, K] device tensors -> [M, N] (in place on C if given)."""
        A, B = self.f64(A).contiguous(), self.f64(B).contiguous()
        M, K = A.shape
        N = B.shape[0]
        assert B.shape[1] == K
        beta = 0 if C is None else 1
        out = self.empty(M, N) if C is None else C
        check(self._L.sp_gemm_nt(self._h, self._p(A), K, 0, self._p(B), K, 0, self._p(out), N, 0, M, N,
                                 K, float(alpha), beta, int(bool(lower_only)), 1, self._stream()))
        return out

    def gemm_nt_batched(self, A, B, C, alpha=1.0, beta=0, lower_only=False):
        """C[b] = beta C[b] + alpha A[b] B[b]^T for contiguous device tensors A [b, M, K], B [b, N, K],
        C [b, M, N] (in place)."""
        b, M, K = A.shape
        N = B.shape[1]
        assert B.shape == (b, N, K) and C.shape == (b, M, N)
        assert A.is_contiguous() and B.is_contiguous() and C.is_contiguous()
        check(self._L.sp_gemm_nt(self._h, self._p(A), K, M * K, self._p(B), K, N * K, self._p(C), N, M * N,
                                 M, N, K, float(alpha), int(beta), int(bool(lower_only)), b, self._stream()))
        return C

    def gp_condition(self, Ktt, Kst, Kss, r):
        """mu = K_st K_tt^-1 r and the posterior covariance K_ss - K_st K_tt^-1 K_st^T
        (device tensors; Kss is not modified).  Returns (mu, Kpost, info)."""
        torch = _torch()
        Ktt, Kst, r = self.f64(Ktt).contiguous(), self.f64(Kst).contiguous(), self.f64(r).contiguous()
        Kpost = self.f64(Kss).clone().contiguous()
        Ks, K = Kst.shape
        mu = self.empty(Ks)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(self._L.sp_gp_condition(self._h, K, Ks, self._p(Ktt), self._p(Kst), self._p(Kpost),
                                      self._p(r), self._p(mu), self._p(info), self._stream()))
        return mu, Kpost, info

    # -- fused likelihood ----------------------------------------------------------
    def workspace(self, S, K, M):
        torch = _torch()
        nbytes = self._L.sp_lnlike_workspace_bytes(self._h, S, K, M)
        if nbytes < 0:
            check(int(nbytes))
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def lnlike_ensemble(self, t, flux, stars_dev, diag=None, conditional=False,
                        covpts=300, tab=None, meanvar=None, rta1=None, temporal=None,
                        normalized=True, norm_order=20, zmax=0.023, out=None,
                        status=None, workspace=None):
        """All arguments already on the device (torch tensors); t [S,K],
        flux [S,M,K], stars_dev from stars_to_device().  Returns (lnlike, status)."""
        torch = _torch()
        S, K = t.shape
        M = flux.shape[1]
        ws = workspace if workspace is not None else self.workspace(S, K, M)
        if out is None:
            out = self.empty(S)
        if status is None:
            status = torch.zeros(S, dtype=torch.int32, device=self.device)
        check(self._L.sp_lnlike_ensemble(
            self._h, S, K, M, self._p(t), self._p(flux), self._p(diag), self._p(stars_dev),
            int(bool(conditional)), int(covpts), self._p(tab), self._p(meanvar),
            self._p(rta1), TEMPORAL[temporal], int(bool(normalized)), int(norm_order),
            float(zmax), self._p(ws), self._p(out), self._p(status), self._stream()))
        return out, status

    def plan_data(self, t, flux, stars_dev, diag=None, covpts=300, temporal=None, workspace=None):
        """What depends on the data alone, once per data set (sp_plan_data): phases, the weights of the kernel
        table in the covariance's sum, sums of the flux and of the variances.  t [S,K], flux [S,M,K] device
        tensors, stars_dev from stars_to_device().  The plan fixes the stars' period, nobs and tau; it may be
        shared by every engine of this GPU.  Returns a ``DataPlan``."""
        S, K = t.shape
        M = flux.shape[1]
        ws = workspace if workspace is not None else self.workspace(S, K, M)
        p = c_void_p()
        check(self._L.sp_plan_data(self._h, S, K, M, self._p(t), self._p(flux), self._p(diag), self._p(stars_dev),
                                   int(covpts), TEMPORAL[temporal], self._p(ws), self._stream(), ctypes.byref(p)))
        # (the plan records the data pointers: the tensors must outlive it)
        return DataPlan(self._L, p, S, K, M, int(covpts), temporal, diag is not None, keep=(t, flux, diag))

    def replicate_plan(self, plan, B):
        """B copies of a planned data set as one batch of B S systems (sp_plan_replicate): system b S + s is star s under
        hyperparameter sample b.  The replica owns its data; evaluate it with ``lnlike_ensemble_planned(plan, None, None,
        stars_for_samples(...), ...)``."""
        p = c_void_p()
        check(self._L.sp_plan_replicate(self._h, plan.ptr, int(B), self._stream(), ctypes.byref(p)))
        return DataPlan(self._L, p, plan.S * int(B), plan.K, plan.M, plan.covpts, plan.temporal, plan.has_diag)

    def lnlike_ensemble_planned(self, plan, t, flux, stars_dev, tab, meanvar, diag=None, norm_order=20,
                                zmax=0.023, out=None, status=None, workspace=None):
        """``lnlike_ensemble(conditional=False, normalized=True)`` on planned data (sp_lnlike_ensemble_planned):
        the same values to rounding, without the per-sample pass over the covariance's entries."""
        torch = _torch()
        S, K, M = plan.S, plan.K, plan.M
        # (t = flux = diag = None: the plan's own arrays -- a replica's copies, or the tensors of plan time)
        assert (t is None and flux is None and diag is None) or (tuple(t.shape) == (S, K) and tuple(flux.shape) == (S, M, K))
        ws = workspace if workspace is not None else self.workspace(S, K, M)
        if out is None:
            out = self.empty(S)
        if status is None:
            status = torch.zeros(S, dtype=torch.int32, device=self.device)
        check(self._L.sp_lnlike_ensemble_planned(
            self._h, plan.ptr, self._p(t), self._p(flux), self._p(diag), self._p(stars_dev), self._p(tab),
            self._p(meanvar), int(norm_order), float(zmax), self._p(ws), self._p(out), self._p(status),
            self._stream()))
        return out, status

    def lnlike_grad_marginal(self, t, flux, stars_dev, tab, meanvar, diag=None, covpts=300, temporal=None,
                             normalized=True, norm_order=20, zmax=0.023, workspace=None):
        """Device half of the ensemble gradient (sp_lnlike_grad_marginal_multi): t [S, K], flux [S, K] or [S, M, K]
        (M light curves per star on one covariance) -> (lnlike [S], ybar [S, covpts + 4], meanbar [S], status [S]):
        the log-likelihoods (summed over a star's light curves) and their derivatives with respect to each star's
        kernel table and flux mean (grad.py chains them to the hyperparameters)."""
        torch = _torch()
        S, K = t.shape
        flux = flux.reshape(S, -1, K)
        M = flux.shape[1]
        nbytes = int(self._L.sp_lnlike_grad_workspace_bytes_multi(self._h, S, K, M, int(covpts)))
        ws = workspace
        if ws is None or ws.numel() < nbytes:
            ws = self._grad_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        out, ybar, mbar = self.empty(S), self.empty(S, covpts + 4), self.empty(S)
        status = torch.zeros(S, dtype=torch.int32, device=self.device)
        check(self._L.sp_lnlike_grad_marginal_multi(
            self._h, S, K, M, self._p(t), self._p(flux), self._p(diag), self._p(stars_dev), int(covpts), self._p(tab),
            self._p(meanvar), TEMPORAL[temporal], int(bool(normalized)), int(norm_order), float(zmax), self._p(ws),
            self._p(out), self._p(ybar), self._p(mbar), self._p(status), self._stream()))
        return out, ybar, mbar, status

    def grad_workspace(self, S, K, covpts, M=1):
        torch = _torch()
        nbytes = int(self._L.sp_lnlike_grad_workspace_bytes_multi(self._h, S, K, int(M), int(covpts)))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def cholesky_lnlike(self, cov, resid):
        """cov [S,K,K] (noise included), resid [S,M,K] -> (lnlike [S], status [S])."""
        torch = _torch()
        cov = self.f64(cov)
        resid = self.f64(resid)
        S, K, _ = cov.shape
        M = resid.shape[1]
        ws = self.workspace(S, K, M)
        out = self.empty(S)
        status = torch.zeros(S, dtype=torch.int32, device=self.device)
        check(self._L.sp_cholesky_lnlike_batched(self._h, S, K, M, self._p(cov), self._p(resid),
                                                 self._p(ws), self._p(out), self._p(status), self._stream()))
        return out, status


class DataPlan(object):
    """Owner of an ``sp_plan`` (include/starry_process_amd.h: sp_plan_data)."""

    def __init__(self, L, ptr, S, K, M, covpts, temporal, has_diag, keep=None):
        self._L, self.ptr = L, ptr
        self.S, self.K, self.M, self.covpts, self.temporal, self.has_diag = S, K, M, covpts, temporal, has_diag
        self._keep = keep

    def wbar(self):
        """[S, covpts + 4] host copy of the table's weights in the covariance's sum."""
        out = np.empty((self.S, self.covpts + 4))
        check(self._L.sp_plan_get_wbar(self.ptr, hptr(out)))
        return out

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._L.sp_plan_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


_engines = {}


def _want_hw_queues(streams):
    """Several independent evaluations in flight on separate HIP streams need a hardware queue each: the
    runtime maps streams onto 4 queues by default, a fifth stream shares a queue with another and the two
    serialise (measured: four steps in flight plus the copy stream LOSE 10 % with 4 queues and gain 3 % with
    8).  ``GPU_MAX_HW_QUEUES`` is read when the HIP runtime initialises: it is set here -- where the streams
    are asked for, not at package import -- if the process has not touched the GPU yet; if it has, the
    setting cannot take effect any more and a warning says what to export.  A value set by the caller wins."""
    import os
    import warnings

    # (the runtime's default is 4 queues, and the process's other streams -- torch's own, the copies -- want theirs:
    #  EnsembleLogProb's 3 + 1 streams ran at 0.77 ms per sample with 4 queues, 0.67 with 8)
    if streams <= 2 or "GPU_MAX_HW_QUEUES" in os.environ:
        return
    torch = _torch()
    if torch.cuda.is_initialized():
        warnings.warn("starry_process_amd: %d streams in flight but the HIP runtime is already initialised with "
                      "its default of 4 hardware queues; export GPU_MAX_HW_QUEUES=8 before the first GPU call "
                      "(streams that share a queue serialise)" % streams, RuntimeWarning, stacklevel=3)
    else:
        os.environ["GPU_MAX_HW_QUEUES"] = "8"


def engine_slots(ydeg=15, udeg=2, device=None, depth=3):
    """``depth`` independent (Engine, torch.cuda.Stream) pairs on one GPU, for keeping several
    INDEPENDENT evaluations in flight (the walkers / live points a sampler evaluates per
    iteration): run evaluation i inside ``with torch.cuda.stream(stream_i)`` on ``engine_i``,
    each with its own workspace and outputs.  One evaluation alone leaves most of the GPU idle
    during its latency-bound phases (the chain of diagonal blocks); with three in flight those
    overlap the neighbours' assembly and trailing updates: 0.95 -> 0.66 ms per 64-star step
    (bench.py).  FOUR is where it peaks: a fifth stream in flight loses 10-25 % (108k against 120k
    evaluations/s at cfg3's shape; calibrate.MAX_STREAMS, to which the callers' ``depth`` is clamped -- this function
    gives what it is asked for, bench.py measures the cliff with it).  A handle is not re-entrant, hence one per slot (fresh handles; the
    process-wide engine of ``get_engine`` is left as it is, and is what depth = 1 returns)."""
    torch = _torch()
    depth = max(1, int(depth))
    _want_hw_queues(depth)
    first = get_engine(ydeg, udeg, device)
    if depth == 1:
        return [(first, torch.cuda.Stream(device=first.device))]
    out = []
    for k in range(depth):
        e = Engine(first.ydeg, first.udeg, first.device_index)
        out.append((e, torch.cuda.Stream(device=e.device)))
    return out


def get_engine(ydeg=15, udeg=2, device=None):
    """Process-wide engine cache keyed by (ydeg, udeg, device)."""
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    key = (int(ydeg), int(udeg), int(device))
    if key not in _engines:
        _engines[key] = Engine(*key)
    return _engines[key]
