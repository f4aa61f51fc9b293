"""
GPU parity of the fused log-likelihood path -- assembly, blocked factorisation (sp_panel.hip panel
kernel, trailing updates), reduction -- at the BASELINE sizes and on awkward ones:

  * cfg3's full 64-star batch (ydeg 15, K 1000) and cfg5's 32-star share of a GPU (ydeg 20, K 3000,
    Matern-3/2, u = [0.4, 0.2]): EVERY star against the value the executed reference gave for it
    (tests/golden/lnlike_full.npz, make_golden.py gen_lnlike_full), and a star's
    value independent of the batch it is evaluated in, bit for bit;
  * a sweep of sizes with partial last blocks, residual rows in the last pivot block's rows or in
    blocks of their own, single blocks, batches of 1 ... 64 stars (items dealt to XCDs unevenly),
    128-row pair items and the look-ahead on or off: against the CPU oracle's factorisation;
  * covariance tiles formed at first touch against the materialised assembly (identical bits);
  * the deferred normalisation against the direct form; failure semantics; three steps in flight.

North-star bar: fp64 log_likelihood within 1e-8 relative of the reference.
"""
import numpy as np
import pytest

from conftest import golden
from starry_process_amd.synthetic import synthetic_star

pytestmark = pytest.mark.gpu
TOL = 1e-8


def make_engine(L):
    from starry_process_amd.engine import Engine

    e = Engine(L, 2, 0)
    mom = golden("moments_L%d" % L)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    return e


@pytest.fixture(scope="module")
def engines():
    cache = {}

    def get(L):
        if L not in cache:
            cache[L] = make_engine(L)
        return cache[L]

    return get


def lnl(e, K, idx, tspan=4.0, u=(0.0, 0.0), tau=None, M=1, data_var=None):
    from starry_process_amd.engine import make_stars

    sts = [synthetic_star(int(s), K, tspan) for s in idx]
    S = len(sts)
    t = np.array([st["t"] for st in sts])
    if M == 1:
        flux = np.array([st["flux"] for st in sts])[:, None, :]
    else:   # M light curves sharing the star's covariance: the star's own curve plus shifted copies
        flux = np.array([[np.roll(st["flux"], 7 * m) * (1.0 + 0.01 * m) for m in range(M)] for st in sts])
    stars = make_stars(S, period=[st["p"] for st in sts], tau=tau or 0.0,
                       data_var=data_var if data_var is not None else 1e-6)
    rta1 = e.f64(e.rTA1L(u))
    tab, mv = e.kernel_table(rta1, 300)
    out, status = e.lnlike_ensemble(e.f64(t), e.f64(flux), e.stars_to_device(stars), covpts=300, tab=tab,
                                    meanvar=mv, temporal="matern32" if tau else None, normalized=True)
    return out.cpu().numpy(), status.cpu().numpy()


def oracle_lnl(K, idx, M=1):
    """The CPU oracle (NumPy / SciPy restatement of sp.py:1052-1188) on the same stars."""
    from oracle.sp_oracle import OracleProcess

    mom = golden("moments_L15")
    op = OracleProcess(mom["default_mean_ylm"], mom["default_cov_ylm"], ydeg=15)
    vals = []
    for s in idx:
        st = synthetic_star(int(s), K)
        if M == 1:
            flux = st["flux"]
        else:
            flux = np.array([np.roll(st["flux"], 7 * m) * (1.0 + 0.01 * m) for m in range(M)])
        vals.append(op.log_likelihood(st["t"], flux, 1e-6, p=st["p"]))
    return np.array(vals)


def test_cfg3_full_batch_every_star(engines):
    """64 stars, ydeg 15, K 1000 -- bench.py's workload: every star within 1e-8 of the reference's own
    value, and a star's value independent of the batch (bit for bit)."""
    ref = golden("lnlike_full")["cfg3_L15_K1000"]
    e = engines(15)
    v, st = lnl(e, 1000, range(64))
    assert not st.any() and np.all(np.isfinite(v))
    assert np.max(np.abs(v / ref - 1)) < TOL
    v8, _ = lnl(e, 1000, range(8))
    assert np.array_equal(v8, v[:8])
    v1, _ = lnl(e, 1000, [63])
    assert v1[0] == v[63]
    v5, _ = lnl(e, 1000, range(20, 25))       # (fewer than 8 stars: a matrix's tiles spread over the XCDs)
    assert np.array_equal(v5, v[20:25])


def test_cfg5_share_every_star(engines):
    """cfg5's share of one GPU: 32 stars, ydeg 20, K 3000, Matern-3/2 (tau 3), u = [0.4, 0.2]."""
    ref = golden("lnlike_full")["cfg5_L20_K3000"]
    e = engines(20)
    v, st = lnl(e, 3000, range(32), tspan=30.0, u=(0.4, 0.2), tau=3.0)
    assert not st.any() and np.all(np.isfinite(v))
    assert np.max(np.abs(v / ref - 1)) < TOL
    v3, _ = lnl(e, 3000, range(3), tspan=30.0, u=(0.4, 0.2), tau=3.0)
    assert np.array_equal(v3, v[:3])


@pytest.mark.parametrize("K,M", [(40, 1), (63, 1), (64, 1), (65, 1), (127, 1), (128, 1), (129, 1), (200, 1),
                                 (257, 3), (448, 1), (511, 2), (513, 1), (960, 70), (1000, 1), (1023, 1),
                                 (1024, 1), (1100, 5), (1345, 1)])
def test_awkward_sizes_against_the_oracle(engines, K, M):
    """Partial last blocks, residual rows in the last pivot block's rows or in blocks of their own
    (K + M beyond the last pivot block), a single pivot block, one or several super-panels."""
    S = 5 if K > 600 else 9
    idx = range(3, 3 + S)
    v, st = lnl(engines(15), K, idx, M=M)
    assert not st.any() and np.all(np.isfinite(v))
    nref = 2 if K > 600 else 4              # (the oracle takes seconds per star at K = 1000)
    ref = oracle_lnl(K, list(idx)[:nref], M=M)
    # (with M > 1 shifted copies a star's value can be a small difference of large terms: relative to the
    #  batch's size)
    scale = np.maximum(np.abs(ref), np.abs(ref).max())
    assert np.max(np.abs(v[:nref] - ref) / scale) < TOL


@pytest.mark.parametrize("S", [1, 3, 8, 13, 16, 64, 100, 128, 136])
def test_batch_sizes_and_pair_items(engines, S):
    """Items are dealt to the XCDs in contiguous runs (whole stars when 8 divides the batch: those launches are
    laid out by CU, chain items first); batches that 8 does not divide use the star-major deal, with 128-row
    pair items where 64-row ones would not fit the CUs in one round (S = 100 at K = 700: the first panels).
    S = 8 / 16 / 64 / 128: one, two, eight and sixteen stars per XCD in the layout by CU; 136: seventeen, beyond it.
    A star's value does not depend on any of that."""
    e = engines(15)
    a, st = lnl(e, 700, range(S))
    b, _ = lnl(e, 700, range(S))
    assert not st.any() and np.array_equal(a, b)
    one, _ = lnl(e, 700, [S - 1])
    assert one[0] == a[S - 1]
    ref = oracle_lnl(700, [0])
    assert abs(a[0] / ref[0] - 1) < TOL


def test_look_ahead_and_super_panel_width_do_not_change_the_answer(monkeypatch):
    """SP_PANEL_LA = 0 (no look-ahead items) and other super-panel widths: the same factor to rounding."""
    res = {}
    for la, sup in (("1", "0"), ("0", "0"), ("1", "4"), ("1", "16"), ("1", "3")):
        monkeypatch.setenv("SP_SUPER", sup)
        e = make_engine(15)
        e._L.sp_debug_set_look_ahead(e._h, int(la))
        v, st = lnl(e, 1000, range(10, 18))
        assert not st.any()
        res[(la, sup)] = v
    ref = res[("1", "0")]
    for k, v in res.items():
        assert np.max(np.abs(v / ref - 1)) < 1e-10, k


@pytest.mark.parametrize("K,M,S", [(1000, 1, 64), (1000, 3, 16), (700, 1, 8), (1345, 1, 24), (960, 70, 8), (200, 1, 40)])
def test_layout_by_cu_and_fused_reduction_do_not_change_a_bit(K, M, S):
    """The panel launches laid out by CU (chain items first, sleepers, the other items on the other CUs) against
    the star-major deal with pair items, and the reduction in the last launch's tail against the kernel of its
    own: where work runs and who reduces, never what is computed."""
    res = []
    for layout, fuse in ((1, 1), (0, 1), (1, 0), (0, 0)):
        e = make_engine(15)
        e._L.sp_debug_set_panel_layout(e._h, layout, fuse)
        v, st = lnl(e, K, range(5, 5 + S), M=M)
        assert not st.any() and np.all(np.isfinite(v))
        res.append(v)
    for v in res[1:]:
        assert np.array_equal(res[0], v)


@pytest.mark.parametrize("K,M,kw", [
    (200, 1, {}), (127, 1, {}), (128, 1, {}), (129, 1, {}), (513, 1, {}), (960, 70, {}), (1000, 1, {}),
    (1345, 1, {}), (1100, 5, {}), (1000, 1, dict(tau=2.5)), (1000, 1, dict(u=(0.4, 0.2))),
    (2100, 1, {}), (2176, 2, {}),       # (first trailing update on 128 x 64 tiles: blocks formed per wavefront)
])
def test_tiles_formed_at_first_touch_are_the_assembled_ones(K, M, kw):
    """sp_set_lazy_cov: the assembly leaves the tiles below the
    diagonal to the kernel that touches them first (same spline code, coefficients gathered from a
    packed copy of the star's table): the log-likelihoods are IDENTICAL to those of the
    materialised assembly -- sizes with partial blocks, residual rows in tiles of their own, a
    temporal kernel, limb darkening."""
    res = []
    for lazy in (1, 0):
        e = make_engine(15)
        e.set_lazy_cov(lazy)
        v, st = lnl(e, K, range(2, 8), M=M, **kw)
        assert not st.any() and np.all(np.isfinite(v))
        res.append(v)
    assert np.array_equal(res[0], res[1])


def test_first_touch_ragged_and_failures():
    """Ragged light curves (entries beyond a star's cadences are zero, formed or assembled) and a
    matrix that is not positive definite, with tiles formed at first touch."""
    from starry_process_amd.engine import make_stars

    res = []
    for lazy in (1, 0):
        e = make_engine(15)
        e.set_lazy_cov(lazy)
        K, S = 640, 6
        sts = [synthetic_star(30 + s, K) for s in range(S)]
        t = np.array([st["t"] for st in sts])
        flux = np.array([st["flux"] for st in sts])[:, None, :]
        dv = np.full(S, 1e-6)
        dv[4] = -1.0
        stars = make_stars(S, period=[st["p"] for st in sts], data_var=dv, nobs=[640, 639, 500, 130, 65, 3])
        tab, mv = e.kernel_table(e.f64(e.rTA1L([0.0, 0.0])), 300)
        out, status = e.lnlike_ensemble(e.f64(t), e.f64(flux), e.stars_to_device(stars), covpts=300, tab=tab,
                                        meanvar=mv, normalized=True)
        res.append((out.cpu().numpy(), status.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][0][4] == -np.inf and (res[0][1][4] & 1)
    assert np.all(np.isfinite(np.delete(res[0][0], 4)))


def test_failure_semantics(engines):
    """A covariance that is not positive definite gives -inf and the NOT_PD bit for THAT star only
    (math.py:82-91, sp.py:1186-1188)."""
    e = engines(15)
    for K in (700, 1000, 64):
        dv = np.full(6, 1e-6)
        dv[2] = -1.0
        v, st = lnl(e, K, range(6), data_var=dv)
        assert v[2] == -np.inf and (st[2] & 1), K
        ok = np.arange(6) != 2
        assert np.all(np.isfinite(v[ok])) and not st[ok].any(), K
        good, _ = lnl(e, K, [0, 1, 3, 4, 5])
        assert np.array_equal(good, v[ok]), K


def test_three_steps_in_flight_K1000():
    """bench.py's default configuration: three independent 64-star, K = 1000 steps on three
    (handle, stream) pairs, enqueued before anything is synchronised, repeated; every step gives
    exactly the bits of the same step run alone on its handle."""
    import torch
    from starry_process_amd.engine import engine_slots, make_stars

    S, K = 64, 1000
    mom = golden("moments_L15")
    slots = engine_slots(15, 2, None, 3)
    e0 = slots[0][0]
    sts = [synthetic_star(s, K) for s in range(S)]
    t_d = e0.f64(np.array([s["t"] for s in sts]))
    f_d = e0.f64(np.array([s["flux"] for s in sts])[:, None, :])
    stars = e0.stars_to_device(make_stars(S, period=[s["p"] for s in sts], data_var=1e-6))
    rta1 = e0.f64(e0.rTA1L([0.0, 0.0]))
    names = ["default", "hilat", "spread"]
    wss = [e.workspace(S, K, 1) for e, _ in slots]

    def evaluate(e, name, out, ws):
        e.set_moments_dev(e.f64(mom[name + "_mean_ylm"]), e.f64(mom[name + "_cov_ylm"]))
        tab, mv = e.kernel_table(rta1, 300)
        e.lnlike_ensemble(t_d, f_d, stars, covpts=300, tab=tab, meanvar=mv, normalized=True, out=out,
                          workspace=ws)

    for e, _ in slots:
        e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])   # (lag grid, first-use set-up)
        e.kernel_table(rta1, 300)
    torch.cuda.synchronize()
    alone = []
    for (e, stream), name, ws in zip(slots, names, wss):
        out = e.empty(S)
        with torch.cuda.stream(stream):
            evaluate(e, name, out, ws)
        torch.cuda.synchronize()
        alone.append(out.clone())
    outs = [e.empty(S) for e, _ in slots]
    for rep in range(4):
        for (e, stream), name, out, ws in zip(slots, names, outs, wss):
            with torch.cuda.stream(stream):
                evaluate(e, name, out, ws)
    torch.cuda.synchronize()
    for a, b in zip(alone, outs):
        assert torch.equal(a, b)
        assert bool(torch.isfinite(a).all())
    assert len({float(a[0]) for a in alone}) == 3


def test_deferred_normalisation_matches_direct():
    """sp_set_defer_norm: raw assembly + the normalisation applied to the factored result against
    the separately normalised matrix, over the input variants of the path: ragged light curves,
    per-cadence variances, baseline variance / mean, several light curves per star, the temporal
    kernel, the conditional branch, a z > zmax star and a matrix that is not positive definite."""
    from starry_process_amd.engine import Engine, make_stars

    mom = golden("moments_L15")
    e = {}
    for on in (0, 1):
        e[on] = Engine(15, 2, 0)
        e[on].set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
        e[on].set_defer_norm(on)
    rng = np.random.RandomState(3)
    K, S = 333, 7
    sts = [synthetic_star(20 + s, K, 6.0) for s in range(S)]
    t = np.array([st["t"] for st in sts])
    cases = {
        "plain": dict(),
        "baseline": dict(baseline_var=3e-5, baseline_mean=2e-3),
        "temporal": dict(tau=2.5, temporal="matern32"),
        # (the hot assembly separates the Matern factor only for cadences in order and strips of moderate span)
        "temporal_out_of_order": dict(tau=2.5, temporal="matern32", swap=True),
        "temporal_short_tau": dict(tau=1.0e-3, temporal="matern32"),
        "temporal_ragged": dict(tau=2.5, temporal="matern32", nobs=[K, K - 1, 200, 65, 64, 63, 2]),
        "ragged": dict(nobs=[K, K - 1, 200, 65, 64, 63, 2]),
        "multi": dict(M=4),
        "vector_variance": dict(diag=1e-6 * (1.0 + rng.rand(S, K))),
        "conditional": dict(conditional=True),
        "not_pd": dict(data_var=[1e-6, -1.0, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6]),
    }
    for name, kw in cases.items():
        M = kw.get("M", 1)
        flux = np.array([[np.roll(st["flux"], 5 * m) for m in range(M)] for st in sts])
        tt = t.copy()
        if kw.get("swap"):
            tt[:, [10, 11]] = tt[:, [11, 10]]
            tt[2, [200, 100]] = tt[2, [100, 200]]
        res = {}
        for on in (0, 1):
            eng = e[on]
            stars = make_stars(S, period=[st["p"] for st in sts], inc_deg=[st["i"] for st in sts],
                               tau=kw.get("tau", 0.0), data_var=kw.get("data_var", 1e-6),
                               baseline_var=kw.get("baseline_var", 0.0),
                               baseline_mean=kw.get("baseline_mean", 0.0), nobs=kw.get("nobs", 0))
            rta1 = eng.f64(eng.rTA1L([0.3, 0.1]))
            tab, mv = eng.kernel_table(rta1, 300)
            out, status = eng.lnlike_ensemble(
                eng.f64(tt), eng.f64(flux), eng.stars_to_device(stars),
                diag=None if "diag" not in kw else eng.f64(kw["diag"]),
                conditional=kw.get("conditional", False), covpts=300, tab=tab, meanvar=mv, rta1=rta1,
                temporal=kw.get("temporal"), normalized=True)
            res[on] = (out.cpu().numpy(), status.cpu().numpy())
        (a, sa), (b, sb) = res[0], res[1]
        assert np.array_equal(sa, sb), name
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)), name
        if name == "not_pd":
            assert not fin[1] and (sa[1] & 1) and fin[[0, 2, 3, 4, 5, 6]].all()
        else:
            assert fin.all(), name
        scale = np.maximum(np.abs(a[fin]), np.abs(a[fin]).max())
        assert np.max(np.abs(a[fin] - b[fin]) / scale) < 1e-10, (name, a, b)
    # z > zmax: -inf with the ZMAX bit in both modes
    g = golden("lnlike")
    st = synthetic_star(0, 100)
    for on in (0, 1):
        e[on].set_moments(g["zmax_guard_mean_ylm"], g["zmax_guard_cov_ylm"])
        tab, mv = e[on].kernel_table(e[on].f64(e[on].rTA1L([0.0, 0.0])), 300)
        out, status = e[on].lnlike_ensemble(e[on].f64(st["t"][None, :]), e[on].f64(st["flux"][None, None, :]),
                                            e[on].stars_to_device(make_stars(1, period=1.0, data_var=1e-6)),
                                            tab=tab, meanvar=mv)
        assert out.cpu().numpy()[0] == -np.inf and (status.cpu().numpy()[0] & 2)


@pytest.mark.parametrize("K,M", [(1000, 1), (1345, 1), (960, 70), (700, 1)])
def test_trailing_update_diagonal_tiles_on_their_lower_blocks_are_the_same_bits(K, M):
    """The diagonal tiles of the symmetric trailing update multiply only their ten blocks on and below the diagonal,
    re-dealt over the four wavefronts (csrc/sp_mm.h, SymDeal): same slices, same k-order per block as the plain loop
    -- IDENTICAL log-likelihoods, also for the inverse's factorisation (an identity riding along)."""
    from starry_process_amd import _lib

    L = _lib.lib()
    res, inv = [], []
    try:
        for on in (1, 0):
            assert L.sp_debug_set_syrk_symdiag(on) == 0
            e = make_engine(15)
            v, st = lnl(e, K, range(4, 12), M=M)
            assert not st.any() and np.all(np.isfinite(v))
            res.append(v)
            if M == 1 and K == 700:
                rng = np.random.RandomState(5)
                A = rng.randn(K, K)
                C = A @ A.T / K + np.eye(K)
                Ci, ld, info = e.spd_inverse(C)
                inv.append((Ci.cpu().numpy(), float(ld)))
    finally:
        L.sp_debug_set_syrk_symdiag(-1)
    assert np.array_equal(res[0], res[1])
    if inv:
        assert np.array_equal(inv[0][0], inv[1][0]) and inv[0][1] == inv[1][1]


@pytest.mark.parametrize("K,kw", [(2100, {}), (2112, dict(tau=2.5)), (3000, dict(tau=3.0, u=(0.4, 0.2)))])
def test_large_trailing_updates_on_128_row_tiles_are_the_same_bits(K, kw):
    """Remainders of 17 blocks and more take the trailing update's 128 x 64 tiles (csrc/sp_gemm.hip, syrk128_kernel;
    odd and even block counts, the pivot block's workgroup, blocks above the diagonal computed and dropped): same
    k-order per entry as the 64 x 64 tiles, hence IDENTICAL log-likelihoods (the golden L20 / K = 3000 values pin them
    against the reference: tests/test_gpu_golden*.py)."""
    from starry_process_amd import _lib

    L = _lib.lib()
    res = []
    try:
        for frm in (0, 17):
            assert L.sp_debug_set_syrk128_from(frm) == 0
            e = make_engine(15)
            v, st = lnl(e, K, range(3), **kw)
            assert not st.any() and np.all(np.isfinite(v))
            res.append(v)
    finally:
        L.sp_debug_set_syrk128_from(-1)
    assert np.array_equal(res[0], res[1])
