"""
The device upstream's quadrature over the WHOLE latitude prior box of the reference
(latitude.py:176-197: a, b in [0, 1] -> alpha <= exp(5), beta <= exp(10)).

CPU half (no GPU): the Gauss-Jacobi rule of the library (sp_gauss_jacobi, Golub-Welsch) is finite,
ordered and equal to an independent LAPACK evaluation everywhere in the box -- round 3 used
scipy.special.roots_jacobi, whose normalisation overflows for b >= 0.737 --, integrates the Beta
law's moments exactly, and the method itself (oracle.ylm_moments_quadrature: the same rotations on
the CPU) reproduces the reference's (mu_y, Sigma_y) at the 18 grid points of
tests/golden/upstream_grid.npz within the per-degree bounds of tests/test_gpu_upstream_device.py.
"""
import warnings

import numpy as np
import pytest

from conftest import golden
from starry_process_amd import upstream
from starry_process_amd.upstream_device import gauss_jacobi, quadrature_nodes


def grid_cov_errors(S, g, i, j):
    """(scale, max error rows l <= 4, rows l <= 8, (l >= 9)^2 block) against the grid fixture."""
    top = g["cov_top"][i, j]
    scale = np.abs(top).max()
    d = np.abs(S[:81] - top)
    il = np.tril_indices(256 - 81)
    low = np.abs(S[81:, 81:][il] - g["cov_low_f32"][i, j].astype(np.float64)).max()
    return scale, d[:25].max(), d.max(), low


def test_gauss_jacobi_over_the_prior_box():
    from oracle.sp_oracle import gauss_jacobi as gj_oracle

    with warnings.catch_warnings():
        warnings.simplefilter("error")                  # no overflow / invalid anywhere
        for a in np.linspace(0.0, 1.0, 11):
            for b in np.linspace(0.0, 1.0, 100):        # the b scan of tests/test_lnlike.py:84-88
                alpha, beta = upstream.ab_to_alphabeta(a, b)
                for n in (7, 17, 22):                   # ydeg + 2 for ydeg = 5, 15, 20
                    t, w = gauss_jacobi(n, beta - 1.0, alpha - 1.0)
                    assert np.all(np.isfinite(t)) and np.all(np.isfinite(w))
                    assert np.all(np.diff(t) > 0) and t[0] > -1 and t[-1] < 1
                    assert np.all(w > 0) and abs(w.sum() - 1) < 1e-14
                    t2, w2 = gj_oracle(n, beta - 1.0, alpha - 1.0)
                    assert np.abs(t - t2).max() < 1e-11 * (t2[-1] - t2[0])
                    assert np.abs(w - w2).max() < 1e-11
                phi, wphi, lam = quadrature_nodes(15, alpha, beta)
                assert np.all(np.isfinite(phi)) and np.all(np.isfinite(wphi)) and abs(wphi.sum() - 1) < 1e-14


@pytest.mark.parametrize("a,b", [(0.4, 0.27), (0.0, 0.0), (1.0, 1.0), (0.5, 0.74), (0.5, 0.9), (0.0, 1.0), (1.0, 0.0)])
def test_gauss_jacobi_integrates_beta_moments(a, b):
    """E[x^k], x ~ Beta(alpha, beta), k < 2n: prod_j (alpha + j) / (alpha + beta + j)."""
    alpha, beta = upstream.ab_to_alphabeta(a, b)
    n = 17
    t, w = gauss_jacobi(n, beta - 1.0, alpha - 1.0)
    x = 0.5 * (1.0 + t)
    m = 1.0
    for k in range(2 * n):
        assert abs(np.sum(w * x ** k) / m - 1) < 1e-10, k
        m *= (alpha + k) / (alpha + beta + k)


def test_gauss_jacobi_rejects_bad_arguments():
    from starry_process_amd._lib import SPError

    for n, a, b in ((0, 0.0, 0.0), (5, -1.0, 0.0), (5, 0.0, -2.0), (5, np.nan, 0.0), (5, np.inf, 0.0)):
        with pytest.raises(SPError):
            gauss_jacobi(n, a, b)
    t, w = gauss_jacobi(1, 2.0, 0.5)                    # one node: the mean of the law
    assert abs(t[0] - (0.5 - 2.0) / (2.0 + 0.5 + 2.0)) < 1e-15 and w[0] == 1.0


def test_quadrature_method_matches_reference_over_the_grid():
    from oracle import sp_oracle as orc

    g = golden("upstream_grid")
    s1, _ = upstream.size_moments(20.0, None, 15)
    for i, a in enumerate(g["a"]):
        for j, b in enumerate(g["b"]):
            alpha, beta = upstream.ab_to_alphabeta(a, b)
            mu, S = orc.ylm_moments_quadrature(s1, s1[None, :], alpha, beta, 0.1, 10.0, 15)
            assert np.all(np.isfinite(mu)) and np.all(np.isfinite(S))
            mr = g["mean_ylm"][i, j]
            assert np.abs(mu - mr).max() < 1e-9 * np.abs(mr).max(), (a, b)
            scale, e4, e8, elow = grid_cov_errors(S, g, i, j)
            assert e4 < 1e-7 * scale and e8 < 1e-5 * scale and elow < 5e-2 * scale, (a, b)


def extended_box_errors(mu, S, k):
    """Per degree l = 0 .. 15, against the extended-precision arbiter at corner k of the prior box
    (tests/golden/upstream_extended_box.npz, tools/upstream_extended.py box: 80-bit arithmetic, 50-digit nodes):
    (error of (mu, S), error of the REFERENCE's own moments at that point), each max |.| over the degree's rows
    relative to max |Sigma_y| (mu: to max |mu_y|).  The reference's moments are the grid fixture's: rows l <= 8 in
    full, the (l >= 9)^2 block in float32 -- its errors there are floored at the storage's 6e-8 of an entry."""
    g, x = golden("upstream_grid"), golden("upstream_extended_box")
    a, b = float(x["a"][k]), float(x["b"][k])
    i, j = list(g["a"]).index(a), int(np.argmin(np.abs(g["b"] - b)))
    assert abs(float(g["b"][j]) - b) < 1e-12
    N = 256
    il = np.tril_indices(N)
    Se = np.zeros((N, N))
    Se[il] = x["cov_ylm_lower"][k]
    Se = Se + np.tril(Se, -1).T
    scale = np.abs(Se).max()
    mu_e = x["mean_ylm"][k]
    emu = (np.abs(mu - mu_e).max() / np.abs(mu_e).max(), np.abs(g["mean_ylm"][i, j] - mu_e).max() / np.abs(mu_e).max())
    top = g["cov_top"][i, j]
    il2 = np.tril_indices(N - 81)
    rows = il2[0] + 81
    low_ref = np.abs(g["cov_low_f32"][i, j].astype(np.float64) - Se[81:, 81:][il2])
    low_own = np.abs(S[81:, 81:][il2] - Se[81:, 81:][il2])
    own, ref = np.zeros(16), np.zeros(16)
    for l in range(16):
        blk = slice(l * l, (l + 1) ** 2)
        own[l] = np.abs(S[blk, :] - Se[blk, :]).max() / scale
        if l <= 8:
            ref[l] = np.abs(top[blk, :] - Se[blk, :]).max() / scale
        else:
            m = (rows >= l * l) & (rows < (l + 1) ** 2)
            ref[l] = low_ref[m].max() / scale
            own[l] = max(own[l], low_own[m].max() / scale)
    return (a, b), emu, own, ref


@pytest.mark.parametrize("k", range(5))
def test_quadrature_against_extended_precision_at_the_corners_of_the_box(k):
    """where the device upstream and the reference differ most -- the corners of the latitude prior
    box and b = 0.9 -- the quadrature of rotations (here its CPU statement) is right to rounding in EVERY degree and
    closer to the extended-precision value than the reference's algorithm is (whose error reaches 7.6e-3 of
    max |Sigma_y| at a = 1, b = 0): the 1.6e-5 between the two log-likelihoods over the box is the reference's."""
    from oracle import sp_oracle as orc

    x = golden("upstream_extended_box")
    assert np.all(x["rule_convergence"] < 1e-15)
    s1, _ = upstream.size_moments(20.0, None, 15)
    alpha, beta = upstream.ab_to_alphabeta(float(x["a"][k]), float(x["b"][k]))
    mu, S = orc.ylm_moments_quadrature(s1, s1[None, :], alpha, beta, 0.1, 10.0, 15)
    ab, emu, own, ref = extended_box_errors(mu, S, k)
    assert emu[0] < 1e-14 and emu[0] <= max(emu[1], 1e-14), (ab, emu)
    assert np.all(own < 1e-13), (ab, own)
    assert np.all(own <= np.maximum(ref, 1e-13)), (ab, own, ref)
    if ab in ((0.0, 0.0), (1.0, 0.0)):
        assert ref[15] > 1e-3                       # (what the reference's top degree is worth there)


def test_assembly_chunks_partition_the_tiles_by_cost():
    """The hot assembly kernel's chunks (csrc/sp_assemble.hip, asm_chunks): every tile in exactly one chunk, in order,
    and no chunk heavier than the mean by more than one tile's weight -- a function of the shape alone (host code)."""
    from starry_process_amd import _lib

    L = _lib.lib()
    for ntr in (1, 2, 3, 9, 16, 17, 47, 64):
        ntiles = ntr * (ntr + 1) // 2
        w = np.array([38 if a == ntr - 1 else (12 if (b == 0 or a == b) else 10)
                      for b in range(ntr) for a in range(b, ntr)])
        for nchunk in sorted({1, 2, max(1, -(-ntiles // 17)), max(1, -(-ntiles // 8)), min(511, ntiles)}):
            st = np.zeros(nchunk + 1, dtype=np.int32)
            assert L.sp_debug_asm_chunks(ntr, nchunk, _lib.hptr(st)) == 0
            assert st[0] == 0 and st[-1] == ntiles and np.all(np.diff(st) >= 0)
            cost = np.array([w[st[c]:st[c + 1]].sum() for c in range(nchunk)])
            assert cost.sum() == w.sum()
            assert cost.max() <= w.sum() / nchunk + 38
    bad = np.zeros(4, dtype=np.int32)
    assert L.sp_debug_asm_chunks(0, 1, _lib.hptr(bad)) != 0
    assert L.sp_debug_asm_chunks(16, 600, _lib.hptr(bad)) != 0


def _gj_mp(n, a, b):
    """The rule at 60 digits: eigen-decomposition of the Jacobi matrix in mpmath."""
    import mpmath as mp

    a, b = mp.mpf(a), mp.mpf(b)
    J, ab = mp.zeros(n, n), a + b
    J[0, 0] = (b - a) / (ab + 2)
    for k in range(1, n):
        s = 2 * k + ab
        J[k, k] = (b - a) * (b + a) / (s * (s + 2))
        num = 4 * (1 + a) * (1 + b) / (s * s * (s + 1)) if k == 1 else \
            4 * k * (k + a) * (k + b) * (k + ab) / (s * s * (s + 1) * (s - 1))
        J[k - 1, k] = J[k, k - 1] = mp.sqrt(num)
    E, Q = mp.eigsy(J)
    idx = sorted(range(n), key=lambda i: E[i])
    return [E[i] for i in idx], [Q[0, i] ** 2 for i in idx]


@pytest.mark.parametrize("a,b", [(0.4, 0.27), (0.0, 0.0), (1.0, 1.0), (0.5, 0.9), (0.0, 1.0), (1.0, 0.0)])
def test_gauss_jacobi_derivatives_against_60_digit_differences(a, b):
    """sp_gauss_jacobi_grad (first-order perturbation of the Jacobi matrix's eigenproblem) against central
    differences of the rule evaluated at 60 digits (step 1e-20: no rounding, no truncation to speak of), at the
    default and at the corners of the prior box.  These derivatives are what makes the device upstream's gradient
    exact (the reference: analytic d/d alpha, d/d beta, ops/include/latitude.h:21-173)."""
    mp = pytest.importorskip("mpmath")
    from starry_process_amd.upstream_device import gauss_jacobi_grad

    mp.mp.dps = 60
    alpha, beta = upstream.ab_to_alphabeta(a, b)
    n, h = 17, mp.mpf(10) ** -20
    t, w, t_a, w_a, t_b, w_b = gauss_jacobi_grad(n, beta - 1.0, alpha - 1.0)
    t0, w0 = gauss_jacobi(n, beta - 1.0, alpha - 1.0)
    assert np.array_equal(t, t0) and np.array_equal(w, w0)
    for dw in (w_a, w_b):                       # the weights keep summing to 1
        assert abs(dw.sum()) < 1e-10 * np.abs(dw).max() + 1e-15
    ja, jb = mp.mpf(beta) - 1, mp.mpf(alpha) - 1
    for which, (dt, dw) in enumerate(((t_a, w_a), (t_b, w_b))):
        hi = _gj_mp(n, ja + h, jb) if which == 0 else _gj_mp(n, ja, jb + h)
        lo = _gj_mp(n, ja - h, jb) if which == 0 else _gj_mp(n, ja, jb - h)
        ft = np.array([float((x - y) / (2 * h)) for x, y in zip(hi[0], lo[0])])
        fw = np.array([float((x - y) / (2 * h)) for x, y in zip(hi[1], lo[1])])
        assert np.abs(dt - ft).max() < 1e-9 * np.abs(ft).max(), (which, np.abs(dt - ft).max() / np.abs(ft).max())
        assert np.abs(dw - fw).max() < 1e-6 * np.abs(fw).max(), (which, np.abs(dw - fw).max() / np.abs(fw).max())
