"""
Oracle restatement of the reverse mode of Solve / Cholesky:
  * Solve.L_op against the reference's own L_op executed eagerly (tests/golden/linalg_rev.npz,
    made by make_golden.py: gen_linalg_rev);
  * Cholesky.L_op (inherited by the reference from Theano / Aesara, absent here) against
    finite differences, the way the reference's tests pin gradients (verify_grad,
    tests/test_lnlike.py:100-136).
"""
import numpy as np

from conftest import golden
from oracle import sp_oracle as orc


def test_solve_L_op_matches_reference():
    g = golden("linalg_rev")
    L = g["L"]
    for tag, lower, Amat in (("lower", True, L), ("upper", False, L.T.copy())):
        for rhs in ("vec", "mat"):
            k = "%s_%s_" % (tag, rhs)
            c = orc._solve_tri(Amat, g[k + "b"], lower)
            assert np.abs(c - g[k + "c"]).max() < 1e-14 * np.abs(g[k + "c"]).max()
            A_bar, b_bar = orc.solve_L_op(Amat, g[k + "b"], g[k + "c"], g[k + "c_bar"], lower)
            assert np.abs(A_bar - g[k + "A_bar"]).max() < 1e-14 * np.abs(g[k + "A_bar"]).max()
            assert np.abs(b_bar - g[k + "b_bar"]).max() < 1e-14 * np.abs(g[k + "b_bar"]).max()
            tri = np.tril if lower else np.triu
            assert np.array_equal(A_bar, tri(A_bar))


def test_cholesky_L_op_finite_differences():
    rng = np.random.RandomState(3)
    K = 14
    B = rng.randn(K, K)
    C = B.dot(B.T) + K * np.eye(K)
    L = orc.cho_factor(C)
    L_bar = np.tril(rng.randn(K, K))
    C_bar = orc.cholesky_L_op(L, L_bar)
    assert np.array_equal(C_bar, np.tril(C_bar))
    eps = 1e-6
    fd = np.zeros((K, K))
    for i in range(K):
        for j in range(i + 1):
            E = np.zeros((K, K))
            E[i, j] = E[j, i] = 1.0
            fd[i, j] = (np.sum(L_bar * orc.cho_factor(C + eps * E))
                        - np.sum(L_bar * orc.cho_factor(C - eps * E))) / (2 * eps)
    assert np.abs(fd - C_bar).max() < 1e-7 * np.abs(C_bar).max()
    # a factor that failed (all NaN, math.py:88-91) gives an all-NaN gradient
    assert np.isnan(orc.cholesky_L_op(np.full((K, K), np.nan), L_bar)).all()


def test_lnlike_gradient_through_the_L_ops():
    """Chain the two L_ops through lnL = -1/2 r^T C^-1 r - sum log L_ii (sp.py:1157-1188):
    the result must be the textbook 1/2 (a a^T - C^-1), a = C^-1 r."""
    rng = np.random.RandomState(4)
    K = 20
    B = rng.randn(K, K)
    C = B.dot(B.T) + K * np.eye(K)
    r = rng.randn(K)
    L = orc.cho_factor(C)
    y = orc._solve_tri(L, r, True)
    x = orc._solve_tri(L.T, y, False)
    # lnL = -1/2 r.x - sum log diag L;  x_bar = -r/2
    U_bar, y_bar = orc.solve_L_op(L.T, y, x, -0.5 * r, False)
    L_bar1, _ = orc.solve_L_op(L, r, y, y_bar, True)
    L_bar = L_bar1 + U_bar.T - np.diag(1.0 / np.diag(L))
    C_bar = orc.cholesky_L_op(L, L_bar)
    full = C_bar + np.tril(C_bar, -1).T            # gradient w.r.t. a symmetric perturbation pair
    Ci = np.linalg.inv(C)
    expect = 0.5 * (np.outer(x, x) - Ci)
    expect = expect + expect.T - np.diag(np.diag(expect))
    assert np.abs(np.tril(full) - np.tril(expect)).max() < 1e-12
