"""
Reverse mode of the triangular solves and of the Cholesky factorisation on the device
(sp_tri_solve, sp_solve_rev, sp_cholesky_rev behind starry_process_amd.math, SURVEY 8f
next #3) against the reference's Solve.L_op run eagerly (tests/golden/linalg_rev.npz), the
oracle's restatement of the Theano Cholesky L_op, and the closed form of the likelihood
gradient.
"""
import numpy as np
import pytest

from conftest import golden
from oracle import sp_oracle as orc

pytestmark = pytest.mark.gpu


def _spd(rng, K):
    B = rng.randn(K, K)
    return B.dot(B.T) + K * np.eye(K)


@pytest.mark.parametrize("K", [100, 200, 1000])
def test_cho_factor_forward_error(K):
    """The blocked factor of a WELL-conditioned matrix (condition ~5) against LAPACK's to 5e-14 of its largest entry
    (rounding alone: ~1e-15).  The rows below a pivot block are products with the block's explicit inverse
    (sp_diag.h: inverse_block_row): the comparisons with the oracle (1e-8 of a log-likelihood) would not see an
    inverse that is wrong in its tenth digit."""
    from starry_process_amd.math import cho_factor

    rng = np.random.RandomState(K)
    C = _spd(rng, K)
    L = np.array(cho_factor(C))
    ref = np.linalg.cholesky(C)
    assert np.abs(np.tril(L) - ref).max() < 5e-14 * np.abs(ref).max()


def test_solve_and_L_op_match_reference():
    from starry_process_amd.math import Solve, cho_factor, cho_solve

    g = golden("linalg_rev")
    L = np.array(cho_factor(g["C"]))
    assert np.abs(L - g["L"]).max() < 1e-13 * np.abs(g["L"]).max()
    for tag, struct, Amat in (("lower", "lower_triangular", g["L"]),
                              ("upper", "upper_triangular", g["L"].T.copy())):
        op = Solve(A_structure=struct, lower=(tag == "lower"))
        for rhs in ("vec", "mat"):
            k = "%s_%s_" % (tag, rhs)
            c = np.array(op(Amat, g[k + "b"]))
            assert c.shape == g[k + "c"].shape
            assert np.abs(c - g[k + "c"]).max() < 1e-13 * np.abs(g[k + "c"]).max()
            A_bar, b_bar = (np.array(x) for x in op.L_op([Amat, g[k + "b"]], [g[k + "c"]], [g[k + "c_bar"]]))
            assert np.abs(A_bar - g[k + "A_bar"]).max() < 1e-13 * np.abs(g[k + "A_bar"]).max()
            assert np.abs(b_bar - g[k + "b_bar"]).max() < 1e-13 * np.abs(g[k + "b_bar"]).max()
            tri = np.tril if tag == "lower" else np.triu
            assert np.array_equal(A_bar, tri(A_bar))
    b = g["lower_mat_b"]
    x = np.array(cho_solve(g["L"], b))
    assert np.abs(x - np.linalg.solve(g["C"], b)).max() < 1e-13


@pytest.mark.parametrize("K", [1, 5, 64, 100, 257])
def test_cholesky_L_op_matches_oracle(K):
    from starry_process_amd.math import Cholesky, cho_factor

    rng = np.random.RandomState(K)
    C = _spd(rng, K)
    L = np.array(cho_factor(C))
    L_bar = np.tril(rng.randn(K, K))
    C_bar = np.array(cho_factor.L_op([C], [L], [L_bar])[0])
    ref = orc.cholesky_L_op(orc.cho_factor(C), L_bar)
    assert np.abs(C_bar - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.array_equal(C_bar, np.tril(C_bar))
    # entries of L_bar above the diagonal do not matter (L has none)
    C_bar2 = np.array(cho_factor.L_op([C], [L], [L_bar + np.triu(rng.randn(K, K), 1)])[0])
    assert np.abs(C_bar2 - ref).max() < 1e-12 * np.abs(ref).max()
    # on_error semantics (math.py:75-91)
    bad = C.copy()
    bad[0, 0] = -1.0
    Lbad = np.array(cho_factor(bad))
    assert np.isnan(Lbad).all()
    assert np.isnan(np.array(cho_factor.L_op([bad], [Lbad], [L_bar])[0])).all()
    with pytest.raises(np.linalg.LinAlgError):
        Cholesky(on_error="raise")(bad)


def test_batched_device_tensors():
    import torch
    from starry_process_amd.engine import get_engine

    e = get_engine(15, 2)
    rng = np.random.RandomState(8)
    B, K, M = 3, 70, 2
    C = np.stack([_spd(rng, K) for _ in range(B)])
    L, info = e.cho_factor(C)
    assert int(info.abs().sum()) == 0
    Lh = L.cpu().numpy()
    c = rng.randn(B, K, M)
    c_bar = rng.randn(B, K, M)
    L_bar = np.tril(rng.randn(B, K, K))
    for trans in (False, True):
        A_bar, b_bar = e.solve_rev(L, c, c_bar, trans=trans)
        assert isinstance(A_bar, torch.Tensor) and A_bar.is_cuda
        for s in range(B):
            Amat = Lh[s].T if trans else Lh[s]
            rA, rb = orc.solve_L_op(Amat, None, c[s], c_bar[s], not trans)
            assert np.abs(A_bar[s].cpu().numpy() - rA).max() < 1e-12 * np.abs(rA).max()
            assert np.abs(b_bar[s].cpu().numpy() - rb).max() < 1e-12 * np.abs(rb).max()
    C_bar = e.cholesky_rev(L, L_bar).cpu().numpy()
    for s in range(B):
        ref = orc.cholesky_L_op(Lh[s], L_bar[s])
        assert np.abs(C_bar[s] - ref).max() < 1e-12 * np.abs(ref).max()


def test_likelihood_gradient_closed_form():
    """d lnL / d C through the device L_ops = 1/2 (a a^T - C^-1), a = C^-1 r, at K = 300."""
    from starry_process_amd.math import Solve, cho_factor

    rng = np.random.RandomState(5)
    K = 300
    C = _spd(rng, K)
    r = rng.randn(K)
    lo = Solve("lower_triangular", lower=True)
    up = Solve("upper_triangular", lower=False)
    L = np.array(cho_factor(C))
    y = np.array(lo(L, r))
    x = np.array(up(L.T.copy(), y))
    U_bar, y_bar = (np.array(v) for v in up.L_op([L.T.copy(), y], [x], [-0.5 * r]))
    L_bar1, r_bar = (np.array(v) for v in lo.L_op([L, r], [y], [y_bar]))
    L_bar = L_bar1 + U_bar.T - np.diag(1.0 / np.diag(L))
    C_bar = np.array(cho_factor.L_op([C], [L], [L_bar])[0])
    Ci = np.linalg.inv(C)
    expect = 0.5 * (np.outer(x, x) - Ci)
    expect = np.tril(expect + expect.T - np.diag(np.diag(expect)))
    assert np.abs(C_bar - expect).max() < 1e-11 * np.abs(expect).max()
    # the half of d lnL / d r = -C^-1 r that flows through the solves (the other half is the
    # explicit r in r . x)
    assert np.abs(r_bar + 0.5 * x).max() < 1e-12 * np.abs(x).max()


@pytest.mark.parametrize("K,B", [(64, 3), (100, 2), (129, 1), (500, 4), (1000, 8)])
def test_spd_inverse_batched(K, B):
    """sp_spd_inverse_batched (the identity riding through the blocked factorisation, C^-1 = L^-T L^-1) against
    NumPy: inverse to 1e-10 of its norm, log-determinant to 1e-12 relative; a matrix that is not positive
    definite is flagged and its log-determinant NaN."""
    from starry_process_amd.engine import get_engine

    e = get_engine(5, 2)
    rng = np.random.RandomState(K + B)
    A = rng.randn(B, K, K)
    C = A @ A.transpose(0, 2, 1) / K + 0.5 * np.eye(K)[None]
    inv, logdet, info = e.spd_inverse(C)
    inv, logdet, info = inv.cpu().numpy(), logdet.cpu().numpy(), info.cpu().numpy()
    assert not info.any()
    ref = np.linalg.inv(C)
    assert np.abs(inv - ref).max() < 1e-10 * np.abs(ref).max()
    assert np.array_equal(inv, inv.transpose(0, 2, 1))
    sign, ld = np.linalg.slogdet(C)
    assert np.all(sign > 0) and np.abs(logdet / ld - 1).max() < 1e-12
    bad = C.copy()
    bad[0, K // 2, K // 2] = -1.0
    _, logdet_b, info_b = e.spd_inverse(bad)
    assert info_b.cpu().numpy()[0] != 0 and np.isnan(logdet_b.cpu().numpy()[0])
    assert not info_b.cpu().numpy()[1:].any()
