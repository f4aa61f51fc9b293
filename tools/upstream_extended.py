#!/usr/bin/env python
"""
Extended-precision arbiter for the upstream moments: which of
{the reference's algorithm, the quadrature of rotations} gives the integrals the reference DEFINES?

    python tools/upstream_extended.py            # writes tests/golden/upstream_extended.npz

The defining expectations (latitude.py:199-212, longitude.py:19-24, contrast.py:18-33)

    mu_y    = pi c n  E[ Ry(lambda) Rx(phi) s ]
    Sigma_y = (pi c)^2 n ( E[ (Ry Rx s)(Ry Rx s)^T ] - E[.] E[.]^T ) + diag(eps)

are evaluated here in 80-bit extended precision (x87 long double, 64-bit mantissa, eps = 1.1e-19):
  * Gauss-Jacobi nodes and weights from mpmath at 50 digits (Golub-Welsch on the Jacobi matrix),
    refined rule (2 x the nodes the double-precision versions use: the rule is exact either way);
  * the rotations by the oracle's C restatement of the reference's Wigner recursion
    (oracle/sp_oracle.c: rotar / dlmn, tensordotRz, dotRx) recompiled with every `double` turned
    into `long double` -- the double version is pinned bit for bit against the reference;
  * all sums in long double.
The spot-size vectors (size.py) are the double-precision ones in all three evaluations: the
question is what the latitude / longitude averaging does to them.

Runs in the build container (needs gcc and mpmath; neither the GPU nor /root/reference).  The
fixture holds inputs (hyperparameters) and outputs (mu, Sigma rounded to double) only.
"""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

HYPER = {
    "default": dict(r=20.0, a=0.40, b=0.27, c=0.1, n=10.0),
    "hilat": dict(r=15.0, a=0.62, b=0.11, c=0.2, n=5.0),
    "spread": dict(r=25.0, dr=5.0, a=0.3, b=0.5, c=0.05, n=20.0),
}

DRIVER = r"""
/* mu, Sigma by quadrature of rotations, everything in long double */
#include <stdlib.h>
#include <string.h>
void orc_Rx(int ydeg, long double theta, long double *R, long double *Rp);
void orc_tensordotRz(int ydeg, const long double *M, const long double *theta, int K, long double *f);
void orc_dotRx(int ydeg, const long double *M, int rows, const long double *Rpk, long double *out);
int orc_nwig(int l);
void quad_ld(int ydeg, int nvec, const long double *vecs, int nphi, const long double *phi,
             const long double *wphi, int nlam, const long double *lam, long double *mom1,
             long double *mom2) {
  const int N = (ydeg + 1) * (ydeg + 1), nw = orc_nwig(ydeg);
  long double *R = malloc(sizeof(long double) * nw), *Rd = malloc(sizeof(long double) * nw);
  long double *Rp = malloc(sizeof(long double) * nw), *Rm = malloc(sizeof(long double) * nw);
  long double *V = malloc(sizeof(long double) * nvec * N), *U = malloc(sizeof(long double) * nvec * N);
  long double *Z = malloc(sizeof(long double) * nvec * N), *A = malloc(sizeof(long double) * nvec * N);
  long double *th = malloc(sizeof(long double) * nvec);
  const long double hp = 1.57079632679489661923132169163975144L;
  orc_Rx(ydeg, hp, Rp, Rd);
  orc_Rx(ydeg, -hp, Rm, Rd);
  memset(mom1, 0, sizeof(long double) * N);
  memset(mom2, 0, sizeof(long double) * N * N);
  for (int k = 0; k < nphi; ++k) {
    orc_Rx(ydeg, phi[k], R, Rd);
    orc_dotRx(ydeg, vecs, nvec, R, V);
    orc_dotRx(ydeg, V, nvec, Rp, U);
    for (int q = 0; q < nlam; ++q) {
      for (int j = 0; j < nvec; ++j) th[j] = lam[q];
      orc_tensordotRz(ydeg, U, th, nvec, Z);
      orc_dotRx(ydeg, Z, nvec, Rm, A);
      const long double w = wphi[k] / nlam;
      for (int i = 0; i < N; ++i) mom1[i] += w * A[i];
      for (int j = (nvec > 1 ? 1 : 0); j < nvec; ++j) {
        const long double *a = A + (size_t)j * N;
        for (int i = 0; i < N; ++i) {
          const long double wa = w * a[i];
          long double *row = mom2 + (size_t)i * N;
          for (int c = 0; c < N; ++c) row[c] += wa * a[c];
        }
      }
    }
  }
  free(R); free(Rd); free(Rp); free(Rm); free(V); free(U); free(Z); free(A); free(th);
}
"""


def build_ld():
    src = open(os.path.join(ROOT, "oracle", "sp_oracle.c")).read()
    src = re.sub(r"\bdouble\b", "long double", src)
    for f in ("sqrt", "cos", "sin", "fabs", "floor"):
        src = re.sub(r"\b%s\(" % f, "%sl(" % f, src)
    src = re.sub(r"(\d\.\d+(?:e[-+]?\d+)?)\b(?![lLfF])", r"\1L", src)   # literals in long double
    tmp = tempfile.mkdtemp(prefix="sp_ld_")
    open(os.path.join(tmp, "orc_ld.c"), "w").write(src)
    open(os.path.join(tmp, "drv_ld.c"), "w").write(DRIVER)
    so = os.path.join(tmp, "liborc_ld.so")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp",
                           "-o", so, os.path.join(tmp, "orc_ld.c"), os.path.join(tmp, "drv_ld.c"), "-lm"])
    return ctypes.CDLL(so)


def gauss_jacobi_mp(n, a, b, dps=50):
    """Nodes t in (-1, 1) and weights (sum 1) for the weight (1 - t)^a (1 + t)^b, mpmath."""
    import mpmath as mp

    mp.mp.dps = dps
    a, b = mp.mpf(a), mp.mpf(b)
    J = mp.zeros(n, n)
    for k in range(n):
        J[k, k] = (b - a) / (a + b + 2) if k == 0 else (b * b - a * a) / ((2 * k + a + b) * (2 * k + a + b + 2))
    for k in range(1, n):
        if k == 1:
            e = 2 / (2 + a + b) * mp.sqrt((1 + a) * (1 + b) / (3 + a + b))
        else:
            e = 2 / (2 * k + a + b) * mp.sqrt(k * (k + a) * (k + b) * (k + a + b)
                                             / ((2 * k + a + b + 1) * (2 * k + a + b - 1)))
        J[k, k - 1] = J[k - 1, k] = e
    E, Q = mp.eigsy(J)
    w = [Q[0, i] ** 2 for i in range(n)]
    tot = sum(w)
    return [E[i] for i in range(n)], [wi / tot for wi in w]


def ld(x):
    import mpmath as mp

    return np.longdouble(mp.nstr(x, 25)) if not isinstance(x, (float, int)) else np.longdouble(x)


def moments_extended(lib, hp, ydeg=15, refine=2):
    import mpmath as mp

    from starry_process_amd.upstream import ab_to_alphabeta, size_moments

    N = (ydeg + 1) ** 2
    s1, eigS = size_moments(hp["r"], hp.get("dr"), ydeg)
    dr = hp.get("dr")
    cols = eigS.T[np.abs(eigS).sum(axis=0) > 0.0] if dr is not None else s1[None, :]
    vecs = cols if dr is None else np.vstack([s1[None, :], cols])
    if dr is None:
        vecs = np.vstack([s1[None, :], s1[None, :]])       # row 0: first moment; rows 1..: second-moment factor
    alpha, beta = ab_to_alphabeta(hp["a"], hp["b"])
    nq = refine * (ydeg + 2)
    t, w = gauss_jacobi_mp(nq, mp.mpf(float(beta)) - 1, mp.mpf(float(alpha)) - 1)
    phis, wph = [], []
    for ti, wi in zip(t, w):
        ph = mp.acos((1 + ti) / 2)
        phis += [ph, -ph]
        wph += [wi / 2, wi / 2]
    nl = refine * (2 * ydeg + 3)
    lams = [2 * mp.pi * q / nl for q in range(nl)]
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    phi_a = np.array([ld(x) for x in phis], dtype=np.longdouble)
    w_a = np.array([ld(x) for x in wph], dtype=np.longdouble)
    lam_a = np.array([ld(x) for x in lams], dtype=np.longdouble)
    v_a = np.ascontiguousarray(vecs.astype(np.longdouble))
    m1 = np.zeros(N, dtype=np.longdouble)
    m2 = np.zeros((N, N), dtype=np.longdouble)
    lib.quad_ld(ctypes.c_int(ydeg), ctypes.c_int(v_a.shape[0]), P(v_a), ctypes.c_int(len(phis)), P(phi_a), P(w_a),
                ctypes.c_int(nl), P(lam_a), P(m1), P(m2))
    c, n = np.longdouble(hp["c"]), np.longdouble(hp["n"])
    pi = np.longdouble(mp.nstr(mp.pi, 25))
    mean = pi * c * n * m1
    cov = (pi * c) ** 2 * n * (m2 - np.outer(m1, m1))
    lamd = np.ones(N, dtype=np.longdouble) * np.longdouble(1e-12)
    lamd[15 ** 2:] = np.longdouble(1e-9)
    cov = cov + np.diag(lamd)
    return mean, cov


def per_degree(D, scale, ydeg=15):
    out = []
    for l in range(ydeg + 1):
        blk = slice(l * l, (l + 1) ** 2)
        out.append(float(np.max(np.abs(D[blk, :])) / scale))
    return np.array(out)


def main():
    from oracle import sp_oracle as orc
    from starry_process_amd.upstream import ab_to_alphabeta, size_moments

    lib = build_ld()
    ref = np.load(os.path.join(ROOT, "tests", "golden", "moments_L15.npz"))
    out = {}
    for name, hp in HYPER.items():
        mean, cov = moments_extended(lib, hp)
        mean2, cov2 = moments_extended(lib, hp, refine=3)
        conv = float(np.max(np.abs(cov2 - cov)) / np.max(np.abs(cov)))
        mu_e, Sig_e = mean.astype(np.float64), cov.astype(np.float64)
        out[name + "_mean_ylm"], out[name + "_cov_ylm"] = mu_e, Sig_e
        out[name + "_rule_convergence"] = conv
        # the double-precision quadrature (oracle: same nodes and rotations as the device version)
        s1, eigS = size_moments(hp["r"], hp.get("dr"), 15)
        cols = eigS.T[np.abs(eigS).sum(axis=0) > 0.0] if hp.get("dr") is not None else s1[None, :]
        alpha, beta = ab_to_alphabeta(hp["a"], hp["b"])
        mu_q, Sig_q = orc.ylm_moments_quadrature(s1, cols, alpha, beta, hp["c"], hp["n"], 15)
        mu_r, Sig_r = ref[name + "_mean_ylm"], ref[name + "_cov_ylm"]
        scale = np.max(np.abs(Sig_e))
        dq, dr_ = per_degree(Sig_q - Sig_e, scale), per_degree(Sig_r - Sig_e, scale)
        out[name + "_quadrature_vs_extended_by_degree"] = dq
        out[name + "_reference_vs_extended_by_degree"] = dr_
        print("== %s: rule refinement 2x -> 3x changes Sigma by %.1e; max|mu| diff: quadrature %.1e, reference %.1e"
              % (name, conv, np.max(np.abs(mu_q - mu_e)) / np.max(np.abs(mu_e)),
                 np.max(np.abs(mu_r - mu_e)) / np.max(np.abs(mu_e))))
        print("   l : |Sigma_quadrature - Sigma_ext| / max|Sigma|   |Sigma_reference - Sigma_ext| / max|Sigma|")
        for l in range(16):
            print("  %2d :   %9.2e                                     %9.2e" % (l, dq[l], dr_[l]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "upstream_extended.npz"), **out)


# the corners of the reference's latitude prior box a, b in [0, 1] (latitude.py:176-197: alpha, beta up to exp(10)) and
# the point of the b scan where round 3 returned NaN -- where the device upstream and the reference differ most
# (VERDICT r04 item 4); r, c, n as in tests/golden/upstream_grid.npz
BOX = [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (0.5, 0.9)]


def main_box():
    """python tools/upstream_extended.py box   ->  tests/golden/upstream_extended_box.npz
    (a, b, mean_ylm [5, N], the lower triangle of cov_ylm [5, N (N + 1) / 2] rounded to double, the change under a
    threefold rule)"""
    lib = build_ld()
    N = 256
    il = np.tril_indices(N)
    means, covs, convs = [], [], []
    for a, b in BOX:
        hp = dict(r=20.0, a=a, b=b, c=0.1, n=10.0)
        mean, cov = moments_extended(lib, hp)
        mean3, cov3 = moments_extended(lib, hp, refine=3)
        conv = float(np.max(np.abs(cov3 - cov)) / np.max(np.abs(cov)))
        print("a = %.2f b = %.2f: rule refinement 2x -> 3x changes Sigma by %.1e, mu by %.1e" % (
            a, b, conv, float(np.max(np.abs(mean3 - mean)) / np.max(np.abs(mean)))))
        means.append(mean.astype(np.float64))
        covs.append(cov.astype(np.float64)[il])
        convs.append(conv)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "upstream_extended_box.npz"),
                        a=np.array([p[0] for p in BOX]), b=np.array([p[1] for p in BOX]), mean_ylm=np.array(means),
                        cov_ylm_lower=np.array(covs), rule_convergence=np.array(convs))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "box":
        main_box()
    else:
        main()
