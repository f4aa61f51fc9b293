"""
GPU parity tests, op by op: the HIP kernels (through the C ABI) against the
oracle and against the golden vectors from the executed reference.

Bars: integer tables bit-exact; Rx bit-exact (same IEEE operation order, host
libm cos/sin); everything else fp64 within a stated tolerance that reflects
summation order only.
"""
import numpy as np
import pytest

from conftest import golden
from oracle import sp_oracle as orc

pytestmark = pytest.mark.gpu

LS = [5, 15, 20]


@pytest.fixture(scope="module")
def engines():
    from starry_process_amd.engine import get_engine

    return {L: get_engine(L, 2) for L in LS}


def relerr(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("L", LS)
def test_Rx_bit_exact(engines, L):
    g = golden("ops_L%d" % L)
    R, dR = engines[L].Rx(g["Rx_theta"])
    assert np.array_equal(host(R), g["Rx_R"])
    assert np.array_equal(host(dR), g["Rx_dR"])
    R2, none = engines[L].Rx(g["Rx_theta"][:2], deriv=False)
    assert none is None and np.array_equal(host(R2), g["Rx_R"][:2])


@pytest.mark.parametrize("L", LS)
def test_dotRx(engines, L):
    e = engines[L]
    N = (L + 1) ** 2
    rng = np.random.RandomState(L)
    M = rng.randn(37, N)
    Rp = orc.Rx(L, 0.83)[0]
    out = host(e.dotRx(M, Rp))
    assert relerr(out, orc.dotRx(L, M, Rp)) < 1e-14
    # batched, one rotation per batch entry
    Mb = rng.randn(3, 5, N)
    Rb = np.array([orc.Rx(L, th)[0] for th in (0.1, -1.0, 2.0)])
    outb = host(e.dotRx(Mb, Rb))
    for b in range(3):
        assert relerr(outb[b], orc.dotRx(L, Mb[b], Rb[b])) < 1e-14


@pytest.mark.parametrize("L", LS)
def test_tensordotRz_and_special(engines, L):
    e = engines[L]
    g = golden("ops_L%d" % L)
    N = (L + 1) ** 2
    rng = np.random.RandomState(int(g["tdRz_seed"]))
    M = rng.randn(50, N)
    th = rng.uniform(-7, 7, 50)
    f = host(e.tensordotRz(M, th))
    # device sincos vs glibc: <= 1-2 ulp on cos/sin, amplified by the recurrence
    assert np.max(np.abs(f - g["tdRz_f"])) < 1e-13 * np.max(np.abs(g["tdRz_f"]))
    T = rng.randn(N, N)
    MM = rng.randn(N, N)
    fs = host(e.special_tensordotRz(T, MM, th))
    assert relerr(fs, g["sptd_f"]) < 5e-13
    assert relerr(fs, orc.special_tensordotRz(L, T, MM, th)) < 5e-13


@pytest.mark.parametrize("L", LS)
def test_rTA1(engines, L):
    e = engines[L]
    g = golden("ops_L%d" % L)
    assert np.array_equal(e.rTA1(), g["rTA1"])
    tol = 3e-10 if L == 20 else 4e-12
    assert relerr(e.rTA1L(g["rTA1L_u"]), g["rTA1L"]) < tol


@pytest.mark.parametrize("L", LS)
def test_moments_and_kernel_table(engines, L):
    e = engines[L]
    g = golden("moments_L%d" % L)
    names = ["default", "hilat", "spread"] if L == 15 else ["default"]
    for name in names:
        e.set_moments(g[name + "_mean_ylm"], g[name + "_cov_ylm"])
        ez, Ez = e.polar_moments()
        if name == "default":
            assert relerr(ez, g["default_ez"]) < 1e-14
            assert relerr(Ez, g["default_Ez"]) < 1e-13
        us = np.array([[0.0, 0.0], [0.4, 0.2]])
        rta1 = e.rTA1L(us)
        tab, mv = e.kernel_table(rta1, 300)
        tab, mv = host(tab), host(mv)
        for k, utag in enumerate(("u0", "u1")):
            pre = "%s_%s_" % (name, utag)
            assert abs(mv[k, 0] - g[pre + "mean"]) < 1e-13 * abs(g[pre + "mean"])
            assert abs(mv[k, 1] - g[pre + "var"]) < 1e-10 * abs(g[pre + "var"])
            scale = np.max(np.abs(g[pre + "yp"]))
            assert np.max(np.abs(tab[k, 0] - g[pre + "yp"])) < 4e-12 * scale
            for r, key in enumerate(("a0", "a1", "a2", "a3")):
                assert np.max(np.abs(tab[k, 1 + r, :301] - g[pre + key])) < 8e-12 * scale


def _case_kwargs(g, tag):
    return dict(i=float(g[tag + "_i"]), p=float(g[tag + "_p"]), u=g[tag + "_u"])


MARG = {
    "marg_raw": dict(norm=False),
    "marg_norm": dict(norm=True),
    "marg_mat32": dict(norm=True, tau=2.5, temporal="matern32"),
    "marg_expsq": dict(norm=False, tau=1.5, temporal="expsquared"),
    "marg_cp64": dict(norm=True, covpts=63),
}
COND = {"cond_raw": dict(norm=False), "cond_norm": dict(norm=True)}


@pytest.mark.parametrize("L", LS)
@pytest.mark.parametrize("tag", sorted(MARG))
def test_cov_marginal(engines, L, tag):
    from starry_process_amd.engine import make_stars

    e = engines[L]
    g = golden("cov_L%d" % L)
    mom = golden("moments_L%d" % L)
    c = MARG[tag]
    kw = _case_kwargs(g, tag)
    covpts = c.get("covpts", 300)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    tab, mv = e.kernel_table(e.rTA1L(kw["u"]), covpts)
    t = g["t"]
    stars = make_stars(1, period=kw["p"], tau=c.get("tau", 0.0))
    cov, z = e.cov_marginal(t[None, :], stars, covpts, tab, mv,
                            temporal=c.get("temporal"), normalized=c["norm"])
    cov = host(cov)[0]
    ref = g[tag + "_cov"]
    assert np.max(np.abs(cov - ref)) < 5e-11 * np.max(np.abs(ref))
    if c["norm"]:
        assert abs(host(z)[0] - g[tag + "_z"]) < 1e-10 * abs(g[tag + "_z"])


@pytest.mark.parametrize("L", LS)
def test_spline_index_bit_exact(engines, L):
    """The int64 interpolation index is integer work: bit-exact, checked through
    a covariance whose table encodes the index itself (a0 = index, a1..a3 = 0)."""
    import torch
    from starry_process_amd.engine import make_stars

    e = engines[L]
    g = golden("cov_L%d" % L)
    mom = golden("moments_L%d" % L)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    for tag, covpts in (("marg_raw", 300), ("marg_cp64", 63)):
        p = float(g[tag + "_p"])
        tab, mv = e.kernel_table(e.rTA1L([0.0, 0.0]), covpts)
        np_ = covpts + 4
        fake = torch.zeros_like(tab)
        fake[0, 1, :] = torch.arange(np_, dtype=torch.float64, device=fake.device)
        stars = make_stars(1, period=p)
        cov, _ = e.cov_marginal(g["t"][None, :], stars, covpts, fake, mv, normalized=False)
        inds = np.rint(host(cov)[0]).astype("int64").reshape(-1)
        assert np.array_equal(inds, g[tag + "_inds"])


@pytest.mark.parametrize("L", LS)
@pytest.mark.parametrize("tag", sorted(COND))
def test_cov_conditional(engines, L, tag):
    from starry_process_amd.engine import make_stars

    e = engines[L]
    g = golden("cov_L%d" % L)
    mom = golden("moments_L%d" % L)
    c = COND[tag]
    kw = _case_kwargs(g, tag)
    e.set_moments(mom["default_mean_ylm"], mom["default_cov_ylm"])
    rta1 = e.rTA1L(kw["u"])
    stars = make_stars(1, period=kw["p"], inc_deg=kw["i"])
    t = g["t"]
    A = host(e.design_matrix(t[None, :], stars, rta1))[0]
    tolA = 3e-10 if L == 20 else 4e-12
    assert relerr(A, g[tag + "_A"]) < tolA
    cov, mean, z = e.cov_conditional(t[None, :], stars, rta1, normalized=c["norm"])
    cov = host(cov)[0]
    ref = g[tag + "_cov"]
    tol = 1e-8 if L == 20 else 5e-11
    assert np.max(np.abs(cov - ref)) < tol * np.max(np.abs(ref))
    assert abs(host(mean)[0] - g[tag + "_fluxmean"]) < 1e-11 * abs(g[tag + "_fluxmean"])
    if c["norm"]:
        assert abs(host(z)[0] - g[tag + "_z"]) < 1e-9 * abs(g[tag + "_z"])


@pytest.mark.parametrize("K", [1, 7, 64, 100, 129, 300])
def test_cho_factor_solve(engines, K):
    import scipy.linalg

    e = engines[5]
    rng = np.random.RandomState(K)
    B = 3
    X = rng.randn(B, K, K + 5)
    A = X @ X.transpose(0, 2, 1) + 1e-3 * np.eye(K)
    Lg, info = e.cho_factor(A)
    Lg = host(Lg)
    assert not host(info).any()
    for b in range(B):
        Lr = scipy.linalg.cholesky(A[b], lower=True)
        assert relerr(Lg[b], Lr) < 1e-12
        assert np.all(np.triu(Lg[b], 1) == 0)
    rhs = rng.randn(B, K, 4)
    x = host(e.cho_solve(Lg, rhs))
    for b in range(B):
        xr = scipy.linalg.cho_solve((scipy.linalg.cholesky(A[b], lower=True), True), rhs[b])
        assert relerr(x[b], xr) < 1e-9
    # single matrix, vector rhs
    x1 = host(e.cho_solve(Lg[0], rhs[0, :, 0]))
    assert relerr(x1, x[0][:, 0]) < 1e-14


def test_cho_factor_not_pd_gives_nan(engines):
    e = engines[5]
    A = np.eye(70)
    A[40, 40] = -1.0
    Lg, info = e.cho_factor(np.stack([A, np.eye(70)]))
    Lg = host(Lg)
    assert host(info)[0] == 1 and host(info)[1] == 0
    assert np.all(np.isnan(Lg[0]))          # math.py:88-91
    assert np.array_equal(Lg[1], np.eye(70))


def test_gemm_layout_asymmetric(engines):
    """A = I with an ASYMMETRIC second operand catches a transposed fp64 MFMA
    accumulator map; goes through the conditional covariance GEMMs."""
    from starry_process_amd.engine import make_stars

    e = engines[5]
    N = 36
    rng = np.random.RandomState(3)
    S = rng.randn(N, N)
    S = S @ S.T + np.eye(N)
    mu = rng.randn(N)
    e.set_moments(mu, S)
    t = np.linspace(0.0, 3.0, 70)
    stars = make_stars(1, period=1.3, inc_deg=50.0)
    rta1 = e.rTA1L([0.1, 0.2])
    A = host(e.design_matrix(t[None, :], stars, rta1))[0]
    cov, mean, _ = e.cov_conditional(t[None, :], stars, rta1, normalized=False)
    ref = A @ S @ A.T
    assert relerr(host(cov)[0], ref) < 1e-13
    assert abs(host(mean)[0] - (A @ mu)[0]) < 1e-13 * abs((A @ mu)[0])


def test_allgather_lnlike_single_rank_communicator():
    """sp_allgather_lnlike through a real RCCL communicator of one rank:
    the library resolves ncclAllGather in the running process (torch's RCCL)."""
    import ctypes
    import os
    import torch
    from starry_process_amd import _lib
    from starry_process_amd.engine import get_engine

    e = get_engine(5, 2, 0)
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)
    uid = (ctypes.c_char * 128)()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()

    class UID(ctypes.Structure):
        _fields_ = [("b", ctypes.c_char * 128)]

    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
    u = UID()
    ctypes.memmove(ctypes.byref(u), uid, 128)
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, u, 0) == 0
    try:
        x = torch.arange(7, dtype=torch.float64, device="cuda") * 1.5 - 2.0
        y = torch.full((7,), float("nan"), dtype=torch.float64, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().sp_allgather_lnlike(e._h, comm, x.data_ptr(), 7, y.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        assert _lib.lib().sp_allgather_lnlike(e._h, None, x.data_ptr(), 7, y.data_ptr(), st) == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("L", [5, 15, 20])
def test_reverse_mode_ops(engines, L):
    """tensordotRz_rev, special_tensordotRz_rev, rTA1L_rev against the
    reference's own reverse-mode kernels run through the ops' grad wiring
    (tests/golden/make_golden.py: gen_rev), and against finite differences of the forward ops."""
    import torch
    from starry_process_amd.ops import tensordotRzOp, special_tensordotRzOp, rTA1LOp

    g = golden("rev_L%d" % L)
    N = (L + 1) ** 2
    rng = np.random.RandomState(int(g["seed"]))
    K = int(g["K"])
    M, th, bf = rng.randn(K, N), rng.uniform(-7, 7, K), rng.randn(K, N)
    Tm, Mm, bfs, bfu = rng.randn(N, N), rng.randn(N, N), rng.randn(K), rng.randn(N)
    op = tensordotRzOp(ydeg=L, udeg=2)
    bM, bth = (np.array(x) for x in op.grad([M, th], [bf]))
    assert np.abs(bM - g["td_bM"]).max() < 1e-13 * np.abs(g["td_bM"]).max()
    assert np.abs(bth - g["td_btheta"]).max() < 1e-12 * np.abs(g["td_btheta"]).max()
    # finite difference in theta of <bf, f>
    eps = 1e-6
    fp = np.array(op(M, th + eps))
    fm = np.array(op(M, th - eps))
    fd = ((fp - fm) * bf).sum(1) / (2 * eps)
    assert np.abs(fd - bth).max() < 1e-6 * np.abs(bth).max()
    sop = special_tensordotRzOp(ydeg=L, udeg=2)
    zT, bMs, bths = (np.array(x) for x in sop.grad([Tm, Mm, th], [bfs]))
    assert not zT.any()
    assert np.abs(bMs - g["sp_bM"]).max() < 1e-12 * np.abs(g["sp_bM"]).max()
    assert np.abs(bths - g["sp_btheta"]).max() < 1e-12 * np.abs(g["sp_btheta"]).max()
    fd = (np.array(sop(Tm, Mm, th + eps)) - np.array(sop(Tm, Mm, th - eps))) * bfs / (2 * eps)
    assert np.abs(fd - bths).max() < 1e-6 * np.abs(bths).max()
    lop = rTA1LOp(ydeg=L, udeg=2)
    for u, bu_ref in zip(g["ld_u"], g["ld_bu"]):
        bu = np.array(lop.grad([u], [bfu])[0])
        # rTA1L itself carries 1e-12 (L<=15) .. 1e-10 (L=20) of summation noise
        assert np.abs(bu - bu_ref).max() < (3e-10 if L <= 15 else 3e-8) * np.abs(bu_ref).max()
