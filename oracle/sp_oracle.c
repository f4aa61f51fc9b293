/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the parts of the reference's
 * log-likelihood hot path that the reference implements in C++
 * (starry_process/ops/include/wigner.h).  It exists to CHECK the HIP product
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); nothing in
 * starry_process_amd/ may include, link, import or call it.
 *
 * Parity status: PINNED.  Every function here is checked in
 * tests/test_oracle_golden.py against vectors produced by the reference's own
 * C++ compiled from its own headers (oracle/_ref, recipe oracle/Makefile) and
 * committed under tests/golden/.
 *
 * Built with -ffp-contract=off: the reference is compiled for baseline x86-64
 * (no FMA), see reference ops/base_op.py:81-90.
 *
 * Layout conventions (reference ops/include/utils.h:33-38, wigner.h:22-30):
 *   - everything fp64, matrices row-major;
 *   - Ylm flat index n(l,m) = l*l + l + m;
 *   - packed Wigner array: block l (a (2l+1)x(2l+1) row-major matrix) starts at
 *     nwig(l-1) and the array has nwig(ydeg) entries.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* wigner.h:22-24 */
int orc_nwig(int l) { return ((l + 1) * (2 * l + 1) * (2 * l + 3)) / 3; }
/* wigner.h:30 */
int orc_nwigl(int l) { return (2 * l + 1) * (2 * l + 1); }

/*
 * Integer layout tables.  All outputs have N=(ydeg+1)^2
 * entries unless noted.
 *   l_of[n], m_of[n] : degree/order of flat index n
 *   mirror[n]        : n(l,-m) = l*l + 2*l - j with j = n - l*l  (wigner.h:336)
 *   m0[l]            : n(l,0) = l*l + l, ydeg+1 entries            (flux.py:200)
 *   blk[l]           : nwig(l-1), ydeg+2 entries (blk[ydeg+1] = NWIG)
 */
void orc_index_tables(int ydeg, int *l_of, int *m_of, int *mirror, int *m0,
                      int *blk) {
  int n = 0;
  for (int l = 0; l <= ydeg; ++l) {
    for (int m = -l; m <= l; ++m) {
      l_of[n] = l;
      m_of[n] = m;
      mirror[n] = l * l + 2 * l - (n - l * l);
      ++n;
    }
    m0[l] = l * l + l;
    blk[l] = l == 0 ? 0 : orc_nwig(l - 1);
  }
  blk[ydeg + 1] = orc_nwig(ydeg);
}

/*
 * The integer cos/sin(k*pi/2)-type factors the real-Wigner assembly uses for a
 * rotation about x (wigner.h:232-270).  For mp = 1..ydeg: cosmal[mp],
 * sinmal[mp], sgn[mp]; for m = 1..ydeg: cosmga[m], sinmga[m].  Entry 0 of each
 * array is unused (set to 0).  Arrays have ydeg+1 entries.
 */
void orc_wigner_int_tables(int ydeg, int *cosmal, int *sinmal, int *sgn,
                           int *cosmga, int *sinmga) {
  int ca = 0, sa = -1, sg = -1;
  cosmal[0] = sinmal[0] = sgn[0] = cosmga[0] = sinmga[0] = 0;
  for (int mp = 1; mp <= ydeg; ++mp) {
    cosmal[mp] = ca;
    sinmal[mp] = sa;
    sgn[mp] = sg;
    sg = -sg;
    int t = sa;
    sa = -ca;
    ca = t;
  }
  int cg = 0, sgm = 1;
  for (int m = 1; m <= ydeg; ++m) {
    cosmga[m] = cg;
    sinmga[m] = sgm;
    int t = -sgm;
    sgm = cg;
    cg = t;
  }
}

/* ------------------------------------------------------------------------ */
/* Wigner matrices about x: rotar + dlmn (wigner.h:36-139, 145-276)          */
/* ------------------------------------------------------------------------ */

/* element (r, c) of the degree-l block that starts at `b` */
#define EL(b, l, r, c) ((b)[(r) * (2 * (l) + 1) + (c)])

/* One step l-2, l-1 -> l of the complex d-matrix recursion and its derivative
 * (Alvarez Collado et al. eqs 19-21; wigner.h:36-139). */
static void dl_step(int l, double c2, double s2, const double *A2,
                    const double *A2p, const double *A1, const double *A1p,
                    double *A, double *Ap) {
  const int top = 2 * l;
  double tg;
  if (fabs(s2) < 1.0e-14) /* SP_WIGNER_TOL, constants.h:69-71 */
    tg = s2;
  else
    tg = (1.0 - c2) / s2;

  /* last row (m' = l): corners, then the recurrence towards smaller m */
  const double a11 = EL(A1, l - 1, 2 * l - 2, 2 * l - 2);
  const double a11p = EL(A1p, l - 1, 2 * l - 2, 2 * l - 2);
  const double a10 = EL(A1, l - 1, 2 * l - 2, 0);
  const double a10p = EL(A1p, l - 1, 2 * l - 2, 0);
  EL(A, l, top, top) = 0.5 * a11 * (1.0 + c2);
  EL(Ap, l, top, top) = 0.5 * (a11p * (1.0 + c2) - a11 * s2);
  EL(A, l, top, 0) = 0.5 * a10 * (1.0 - c2);
  EL(Ap, l, top, 0) = 0.5 * (a10p * (1.0 - c2) + a10 * s2);
  for (int m = l - 1; m >= 1 - l; --m) {
    const double rt = sqrt((double)(l + m + 1) / (l - m));
    const double nxt = EL(A, l, top, m + 1 + l);
    const double nxtp = EL(Ap, l, top, m + 1 + l);
    EL(A, l, top, m + l) = -tg * rt * nxt;
    EL(Ap, l, top, m + l) = -rt * (nxt / (1.0 + c2) + tg * nxtp);
  }

  /* rows m' = l-1 .. 0, columns shrinking by one at each end per row */
  const int al = l, al1 = l - 1, tal1 = 2 * l - 1;
  const double ali = 1.0 / al1;
  const double cosaux = c2 * al * al1;
  int lo = 1 - l, hi = l - 1;
  for (int mp = l - 1; mp >= 0; --mp) {
    const int laux = l + mp, lbux = l - mp;
    const double aux = ali / sqrt((double)(laux * lbux));
    const double cux = sqrt((double)((laux - 1) * (lbux - 1))) * al;
    for (int m = hi; m >= lo; --m) {
      const int lauz = l + m, lbuz = l - m;
      const double auz = 1.0 / sqrt((double)(lauz * lbuz));
      const double fact = aux * auz;
      const double p1 = EL(A1, l - 1, mp + l - 1, m + l - 1);
      const double p1p = EL(A1p, l - 1, mp + l - 1, m + l - 1);
      const double cm = cosaux - (double)(m * mp);
      double term = tal1 * cm * p1;
      double termp = tal1 * (-s2 * al * al1 * p1 + cm * p1p);
      if (lbuz != 1 && lbux != 1) {
        const double cuz = sqrt((double)((lauz - 1) * (lbuz - 1)));
        term = term - EL(A2, l - 2, mp + l - 2, m + l - 2) * cux * cuz;
        termp = termp - EL(A2p, l - 2, mp + l - 2, m + l - 2) * cux * cuz;
      }
      EL(A, l, mp + l, m + l) = fact * term;
      EL(Ap, l, mp + l, m + l) = fact * termp;
    }
    ++lo;
    --hi;
  }

  /* reflection: (-1)^(m-m') d[m,m'] = d[m',m]  (wigner.h:113-125) */
  int sign = 1;
  lo = -l;
  hi = l - 1;
  for (int m = l; m > 0; --m) {
    for (int mp = lo; mp <= hi; ++mp) {
      EL(A, l, mp + l, m + l) = sign * EL(A, l, m + l, mp + l);
      EL(Ap, l, mp + l, m + l) = sign * EL(Ap, l, m + l, mp + l);
      sign = -sign;
    }
    ++lo;
    --hi;
  }
  /* inversion: (-1)^(m-m') d[-m',-m] = d[m',m]  (wigner.h:127-138) */
  lo = -l;
  hi = lo;
  for (int m = l - 1; m > -(l + 1); --m) {
    sign = -1;
    for (int mp = hi; mp >= lo; --mp) {
      EL(A, l, mp + l, m + l) = sign * EL(A, l, -mp + l, -m + l);
      EL(Ap, l, mp + l, m + l) = sign * EL(Ap, l, -mp + l, -m + l);
      sign = -sign;
    }
    ++hi;
  }
}

/* Packed real rotation matrices R^l_x(theta), l = 0..ydeg, and d/dtheta
 * (wigner.h:145-284; Op signature ops/wigner/Rx.py:8-43). ydeg >= 1. */
void orc_Rx(int ydeg, double theta, double *R, double *Rp) {
  const int nw = orc_nwig(ydeg);
  double *D = (double *)calloc((size_t)nw, sizeof(double));
  double *Dp = (double *)calloc((size_t)nw, sizeof(double));
  const double r2 = sqrt(2.0);
  const double c2 = cos(theta), s2 = sin(theta);
  const double c2p = -s2, s2p = c2;

  /* degree 0 and 1 written out (wigner.h:162-204) */
  D[0] = 1.0;
  Dp[0] = 0.0;
  D[9] = 0.5 * (1.0 + c2);
  Dp[9] = 0.5 * c2p;
  D[8] = -s2 / r2;
  Dp[8] = -s2p / r2;
  D[7] = 0.5 * (1.0 - c2);
  Dp[7] = -0.5 * c2p;
  D[6] = -D[8];
  Dp[6] = -Dp[8];
  D[5] = D[9] - D[7];
  Dp[5] = Dp[9] - Dp[7];
  D[4] = D[8];
  Dp[4] = Dp[8];
  D[3] = D[7];
  Dp[3] = Dp[7];
  D[2] = D[6];
  Dp[2] = Dp[6];
  D[1] = D[9];
  Dp[1] = Dp[9];

  R[0] = 1.0;
  Rp[0] = 0.0;
  R[1] = D[9] - D[7];
  Rp[1] = Dp[9] - Dp[7];
  R[2] = -r2 * D[6];
  Rp[2] = -r2 * Dp[6];
  R[3] = 0;
  Rp[3] = 0;
  R[4] = -r2 * D[8];
  Rp[4] = -r2 * Dp[8];
  R[5] = D[5];
  Rp[5] = Dp[5];
  R[6] = R[7] = R[8] = 0;
  Rp[6] = Rp[7] = Rp[8] = 0;
  R[9] = D[9] + D[7];
  Rp[9] = Dp[9] + Dp[7];

  for (int l = 2; l <= ydeg; ++l) {
    const double *A2 = D + orc_nwig(l - 3), *A2p = Dp + orc_nwig(l - 3);
    const double *A1 = D + orc_nwig(l - 2), *A1p = Dp + orc_nwig(l - 2);
    double *A = D + orc_nwig(l - 1), *Ap = Dp + orc_nwig(l - 1);
    if (l == 2) { /* nwig(-1) = 0 */
      A2 = D;
      A2p = Dp;
    }
    dl_step(l, c2, s2, A2, A2p, A1, A1p, A, Ap);

    /* complex -> real (wigner.h:225-271) */
    double *Q = R + orc_nwig(l - 1), *Qp = Rp + orc_nwig(l - 1);
    EL(Q, l, l, l) = EL(A, l, l, l);
    EL(Qp, l, l, l) = EL(Ap, l, l, l);
    int cosmal = 0, sinmal = -1, sign = -1;
    for (int mp = 1; mp <= l; ++mp) {
      int cosmga = 0, sinmga = 1;
      EL(Q, l, mp + l, l) = r2 * EL(A, l, l, mp + l) * cosmal;
      EL(Qp, l, mp + l, l) = r2 * EL(Ap, l, l, mp + l) * cosmal;
      EL(Q, l, -mp + l, l) = r2 * EL(A, l, l, mp + l) * sinmal;
      EL(Qp, l, -mp + l, l) = r2 * EL(Ap, l, l, mp + l) * sinmal;
      for (int m = 1; m <= l; ++m) {
        const double d1 = EL(A, l, -mp + l, -m + l);
        const double d1p = EL(Ap, l, -mp + l, -m + l);
        const double d2 = sign * EL(A, l, mp + l, -m + l);
        const double d2p = sign * EL(Ap, l, mp + l, -m + l);
        const int cosag = cosmal * cosmga - sinmal * sinmga;
        const int cosagm = cosmal * cosmga + sinmal * sinmga;
        const int sinag = sinmal * cosmga + cosmal * sinmga;
        const int sinagm = sinmal * cosmga - cosmal * sinmga;
        EL(Q, l, l, m + l) = r2 * EL(A, l, m + l, l) * cosmga;
        EL(Qp, l, l, m + l) = r2 * EL(Ap, l, m + l, l) * cosmga;
        EL(Q, l, l, -m + l) = -r2 * EL(A, l, m + l, l) * sinmga;
        EL(Qp, l, l, -m + l) = -r2 * EL(Ap, l, m + l, l) * sinmga;
        EL(Q, l, mp + l, m + l) = d1 * cosag + d2 * cosagm;
        EL(Qp, l, mp + l, m + l) = d1p * cosag + d2p * cosagm;
        EL(Q, l, mp + l, -m + l) = -d1 * sinag + d2 * sinagm;
        EL(Qp, l, mp + l, -m + l) = -d1p * sinag + d2p * sinagm;
        EL(Q, l, -mp + l, m + l) = d1 * sinag + d2 * sinagm;
        EL(Qp, l, -mp + l, m + l) = d1p * sinag + d2p * sinagm;
        EL(Q, l, -mp + l, -m + l) = d1 * cosag - d2 * cosagm;
        EL(Qp, l, -mp + l, -m + l) = d1p * cosag - d2p * cosagm;
        const int t = -sinmga;
        sinmga = cosmga;
        cosmga = t;
      }
      sign = -sign;
      const int t = sinmal;
      sinmal = -cosmal;
      cosmal = t;
    }
  }
  free(D);
  free(Dp);
}

/* cos(n*theta_k), sin(n*theta_k), n = 0..ydeg, by the reference's Chebyshev
 * recurrence (wigner.h:305-316). Output arrays are K x (ydeg+1), row-major. */
static void cheb_trig(int ydeg, const double *theta, int K, double *cn,
                      double *sn) {
  const int W = ydeg + 1;
  for (int k = 0; k < K; ++k) {
    double *c = cn + (size_t)k * W, *s = sn + (size_t)k * W;
    c[0] = 1.0;
    s[0] = 0.0;
    if (ydeg >= 1) {
      c[1] = cos(theta[k]);
      s[1] = sin(theta[k]);
    }
    for (int n = 2; n <= ydeg; ++n) {
      c[n] = 2.0 * c[n - 1] * c[1] - c[n - 2];
      s[n] = 2.0 * s[n - 1] * c[1] - s[n - 2];
    }
  }
}

/* f = M . Rz(theta), M and f are K x N (wigner.h:289-339). */
void orc_tensordotRz(int ydeg, const double *M, const double *theta, int K,
                     double *f) {
  const int N = (ydeg + 1) * (ydeg + 1), W = ydeg + 1;
  double *cn = (double *)malloc(sizeof(double) * (size_t)K * W);
  double *sn = (double *)malloc(sizeof(double) * (size_t)K * W);
  cheb_trig(ydeg, theta, K, cn, sn);
  for (int k = 0; k < K; ++k) {
    const double *Mk = M + (size_t)k * N;
    double *fk = f + (size_t)k * N;
    for (int l = 0; l <= ydeg; ++l) {
      for (int j = 0; j < 2 * l + 1; ++j) {
        const int m = j - l;
        const double cm = cn[(size_t)k * W + (m < 0 ? -m : m)];
        const double sm = m < 0 ? -sn[(size_t)k * W - m] : sn[(size_t)k * W + m];
        fk[l * l + j] = Mk[l * l + j] * cm + Mk[l * l + 2 * l - j] * sm;
      }
    }
  }
}

/* f_k = sum_j [cosmt . (T o M) + sinmt . TM2]_{kj}, as coded in the reference:
 * two (K x N)(N x N) products then a row sum (wigner.h:409-459).  T and M are
 * N x N.  Summation order: for each output (k, j) accumulate over n ascending,
 * then sum over j ascending. */
void orc_special_tensordotRz(int ydeg, const double *T, const double *M,
                             const double *theta, int K, double *f) {
  const int N = (ydeg + 1) * (ydeg + 1), W = ydeg + 1;
  double *cn = (double *)malloc(sizeof(double) * (size_t)K * W);
  double *sn = (double *)malloc(sizeof(double) * (size_t)K * W);
  double *TM1 = (double *)malloc(sizeof(double) * (size_t)N * N);
  double *TM2 = (double *)malloc(sizeof(double) * (size_t)N * N);
  int *mo = (int *)malloc(sizeof(int) * (size_t)N);
  double *row = (double *)malloc(sizeof(double) * (size_t)N);
  cheb_trig(ydeg, theta, K, cn, sn);
  for (int l = 0, n = 0; l <= ydeg; ++l)
    for (int m = -l; m <= l; ++m, ++n) mo[n] = m;
  for (int r = 0; r < N; ++r)
    for (int l = 0; l <= ydeg; ++l)
      for (int m = -l; m <= l; ++m) {
        const int c = l * l + l + m, cm = l * l + l - m;
        TM1[(size_t)r * N + c] = T[(size_t)r * N + c] * M[(size_t)r * N + c];
        TM2[(size_t)r * N + c] = T[(size_t)r * N + c] * M[(size_t)r * N + cm];
      }
  for (int k = 0; k < K; ++k) {
    memset(row, 0, sizeof(double) * (size_t)N);
    for (int n = 0; n < N; ++n) {
      const int m = mo[n];
      const double cm = cn[(size_t)k * W + (m < 0 ? -m : m)];
      const double sm = m < 0 ? -sn[(size_t)k * W - m] : sn[(size_t)k * W + m];
      const double *t1 = TM1 + (size_t)n * N, *t2 = TM2 + (size_t)n * N;
      for (int j = 0; j < N; ++j) row[j] += cm * t1[j] + sm * t2[j];
    }
    double acc = 0.0;
    for (int j = 0; j < N; ++j) acc += row[j];
    f[k] = acc;
  }
  free(cn);
  free(sn);
  free(TM1);
  free(TM2);
  free(mo);
  free(row);
}

/* M . blockdiag(R^l): rows x N times the packed Wigner array (flux.py:74-86). */
void orc_dotRx(int ydeg, const double *M, int rows, const double *Rpk,
               double *f) {
  const int N = (ydeg + 1) * (ydeg + 1);
  for (int r = 0; r < rows; ++r) {
    const double *Mr = M + (size_t)r * N;
    double *fr = f + (size_t)r * N;
    for (int l = 0; l <= ydeg; ++l) {
      const int w = 2 * l + 1;
      const double *B = Rpk + (l == 0 ? 0 : orc_nwig(l - 1));
      for (int c = 0; c < w; ++c) {
        double acc = 0.0;
        for (int i = 0; i < w; ++i) acc += Mr[l * l + i] * B[i * w + c];
        fr[l * l + c] = acc;
      }
    }
  }
}
