import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import golden
from starry_process_amd import upstream
orig = upstream.matrix_sqrt
def traced(Q, neig=None, mindiff=1e-15):
    U = orig(Q, neig, mindiff)
    w = np.linalg.eigvalsh(Q)
    print("matrix_sqrt N=%d neig=%s  |Q|max=%.3e  recon err=%.3e  eig[top]=%.3e eig[-neig]=%.3e nsmall(<1e-15)=%d nneg=%d sym=%.2e" % (
        Q.shape[0], neig, np.abs(Q).max(), np.abs(U @ U.T - Q).max(), w[-1], w[-(neig or Q.shape[0])], (np.abs(w) < 1e-15).sum(), (w < 0).sum(), np.abs(Q-Q.T).max()))
    return U
upstream.matrix_sqrt = traced
upstream._cache.clear()
mom = golden("moments_L15")
mu, S = upstream.ylm_moments(ydeg=15)
print("mean diff", np.abs(mu-mom["default_mean_ylm"]).max()/np.abs(mu).max(), "cov diff", np.abs(S-mom["default_cov_ylm"]).max()/np.abs(S).max())
np.show_config()
R = mom["default_cov_ylm"]
d = np.abs(S - R)
i, j = np.unravel_index(d.argmax(), d.shape)
print("max|S| %.3e max|R| %.3e  maxdiff %.3e at (%d,%d): S=%.6e R=%.6e" % (np.abs(S).max(), np.abs(R).max(), d.max(), i, j, S[i, j], R[i, j]))
print("diag diff max", np.abs(np.diag(S) - np.diag(R)).max(), " offdiag", np.abs(d - np.diag(np.diag(d))).max())
import scipy
print("numpy", np.__version__, "scipy", scipy.__version__)
e, E = upstream.size_moments(20.0, None, 15)
print("size q sum %.17e" % e.sum())
al, be = upstream.ab_to_alphabeta(0.4, 0.27)
q, Q = upstream.latitude_integrals(15, al, be)
print("lat q sum %.17e Q sum %.17e" % (q.sum(), Q.sum()))
ql, Ql = upstream._longitude_integrals(15)
print("lon q sum %.17e Q sum %.17e" % (ql.sum(), Ql.sum()))
from starry_process_amd.hostconst import wigner_poly
print("R sums %.17e %.17e" % (sum(r.sum() for r in wigner_poly(15, 0, 1, 0, -1)), sum(np.abs(r).sum() for r in wigner_poly(15, 1, 0, 1, 0))))
for l in range(16):
    blk = slice(l * l, (l + 1) ** 2)
    print("l=%2d  max|dS| in rows of degree l: %.3e   (max|S| there %.3e)" % (l, d[blk, :].max(), np.abs(R[blk, :]).max()))
