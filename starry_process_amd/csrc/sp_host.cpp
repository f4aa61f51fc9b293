// Host-side constants of the flux operator: index tables,
// the phase-curve solution vector and change of basis (row a5), the
// limb-darkening basis, and the AlphaBeta series (row a15).
//
// The reference computes these in C++ too (ops/include/flux.h, compile-time
// SP__LMAX / SP__UMAX); here the degrees are run-time values held by the
// handle and the polynomial algebra is done on dense coefficient vectors.
//
// Polynomial basis: index n(l,m) = l*l + l + m stands for the monomial
//   x^(mu/2) y^(nu/2)            if nu = l + m is even,
//   x^((mu-1)/2) y^((nu-1)/2) z  if it is odd,          mu = l - m,
// with z^2 = 1 - x^2 - y^2 folded back into the basis (flux.h:206-236).
#include <cmath>
#include <cstring>

#include "sp_internal.h"

namespace {

inline int nidx(int l, int m) { return l * l + l + m; }

// acc += v * (monomial n(l1,m1) * monomial n(l2,m2))   (flux.h:209-236)
inline void mono_mul_acc(double *acc, int l1, int m1, int l2, int m2, double v) {
  const bool z1 = ((l1 + m1) & 1) != 0, z2 = ((l2 + m2) & 1) != 0;
  const int l = l1 + l2, m = m1 + m2;
  if (z1 && z2) {
    acc[nidx(l - 2, m)] += v;
    acc[nidx(l, m - 2)] += -v;
    acc[nidx(l, m + 2)] += -v;
  } else {
    acc[nidx(l, m)] += v;
  }
}

// out = z * p, p of degree <= deg, out of degree deg+1  (flux.h:74-96)
void times_z(int deg, const double *p, double *out, int nout) {
  for (int i = 0; i < nout; ++i) out[i] = 0.0;
  for (int l = 0; l <= deg; ++l)
    for (int m = -l; m <= l; ++m) {
      const double v = p[nidx(l, m)];
      const int lz = l + 1, nz = lz * lz + lz + m;
      if ((l + m) & 1) {
        out[nz - 4 * lz + 2] += v;
        out[nz - 2] -= v;
        out[nz + 2] -= v;
      } else {
        out[nz] += v;
      }
    }
}

// phase-curve solution vector r^T for degree `deg` (flux.h:22-68)
void solution_vector(int deg, std::vector<double> &rT) {
  rT.assign((size_t)(deg + 1) * (deg + 1), 0.0);
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: l, m = 0 mod 4;  pass 1: l, m = 2 mod 4
    double amp0 = pass == 0 ? M_PI : 0.5 * M_PI;
    double lfac1 = pass == 0 ? 1.0 : 0.5;
    double lfac2 = pass == 0 ? 2.0 / 3.0 : 4.0 / 15.0;
    for (int l = 2 * pass; l <= deg; l += 4) {
      double amp = amp0;
      for (int m = 2 * pass; m <= l; m += 4) {
        const int mu = l - m, nu = l + m;
        rT[nidx(l, m)] = amp * lfac1;
        rT[nidx(l, -m)] = amp * lfac1;
        if (l < deg) {
          rT[nidx(l + 1, m)] = amp * lfac2;
          rT[nidx(l + 1, -m)] = amp * lfac2;
        }
        amp *= (nu + 2.0) / (mu - 2.0);
      }
      lfac1 /= (l / 2 + 2) * (l / 2 + 3);
      lfac2 /= (l / 2 + 2.5) * (l / 2 + 3.5);
      if (pass == 0)
        amp0 *= 0.0625 * (l + 2) * (l + 2);
      else
        amp0 *= 0.0625 * l * (l + 4);
    }
  }
}

// Dense A1 (Ylm -> polynomial basis), degree `deg`, row-major Nd x Nd
// (flux.h:98-279).
void change_of_basis(int deg, std::vector<double> &A1) {
  const int Nd = (deg + 1) * (deg + 1);
  A1.assign((size_t)Nd * Nd, 0.0);
  const double norm = 2.0 / std::sqrt(M_PI);

  // amplitudes per Ylm (flux.h:189-203)
  std::vector<double> amp(Nd, 0.0);
  for (int l = 0; l <= deg; ++l) {
    amp[nidx(l, 0)] = std::sqrt((double)(2 * (2 * l + 1)));
    for (int m = 1; m <= l; ++m) {
      amp[nidx(l, m)] =
          -amp[nidx(l, m - 1)] / std::sqrt((double)((l + m) * (l - m + 1)));
      amp[nidx(l, -m)] = amp[nidx(l, m)];
    }
    amp[nidx(l, 0)] *= std::sqrt(0.5);
  }
  for (int i = 0; i < Nd; ++i) amp[i] /= (2 * std::sqrt(M_PI));

  // z-polynomials P[l][m], m >= 0, as dense vectors (flux.h:102-139)
  std::vector<std::vector<double>> P((size_t)Nd);
  std::vector<double> zp(Nd);
  double term = 1.0, fac = 1.0;
  for (int m = 0; m <= deg; ++m) {
    std::vector<double> &pmm = P[nidx(m, m)];
    pmm.assign(Nd, 0.0);
    pmm[0] = fac;
    for (int l = m + 1; l <= deg; ++l) {
      std::vector<double> &cur = P[nidx(l, m)];
      cur.assign(Nd, 0.0);
      times_z(deg - 1, P[nidx(l - 1, m)].data(), zp.data(), Nd);
      for (int i = 0; i < Nd; ++i) cur[i] = (2 * l - 1) * zp[i] / (l - m);
      if (l > m + 1) {
        const std::vector<double> &p2 = P[nidx(l - 2, m)];
        for (int i = 0; i < Nd; ++i) cur[i] -= (l + m - 1) * p2[i] / (l - m);
      }
    }
    fac *= -term;
    term += 2;
  }

  // (x, y) terms of each Ylm (flux.h:158-183): lists of (l', m', value)
  struct Term {
    int l, m;
    double v;
  };
  std::vector<std::vector<Term>> XY((size_t)Nd);
  for (int m = 0; m <= deg; ++m) {
    double t1 = 1.0, t2 = m;
    for (int j = 0; j <= m; j += 2) {
      if (j > 0) {
        t1 *= -(m - j + 1.0) * (m - j + 2.0) / (j * (j - 1.0));
        t2 *= -(m - j) * (m - j + 1.0) / (j * (j + 1.0));
      }
      for (int l = m; l <= deg; ++l) {
        XY[nidx(l, m)].push_back({m, 2 * j - m, t1});
        if (j < m) XY[nidx(l, -m)].push_back({m, 2 * (j + 1) - m, t2});
      }
    }
  }

  // column by column: product of the z part and the (x, y) part
  std::vector<double> col(Nd);
  for (int l = 0; l <= deg; ++l)
    for (int m = -l; m <= l; ++m) {
      const int c = nidx(l, m);
      const std::vector<double> &pz = P[nidx(l, m < 0 ? -m : m)];
      for (int i = 0; i < Nd; ++i) col[i] = 0.0;
      for (int lz = 0; lz <= deg; ++lz)
        for (int mz = -lz; mz <= lz; ++mz) {
          const double vz = pz[nidx(lz, mz)];
          if (vz == 0) continue;
          for (const Term &t : XY[c])
            if (lz + t.l <= deg)
              mono_mul_acc(col.data(), lz, mz, t.l, t.m, vz * t.v * norm * amp[c]);
        }
      for (int r = 0; r < Nd; ++r) A1[(size_t)r * Nd + c] = col[r];
    }
}

}  // namespace

void sp_build_index_tables(int ydeg, int32_t *l_of, int32_t *m_of,
                           int32_t *mirror, int32_t *m0, int32_t *blk) {
  for (int l = 0; l <= ydeg; ++l) {
    for (int m = -l; m <= l; ++m) {
      const int n = nidx(l, m);
      l_of[n] = l;
      m_of[n] = m;
      mirror[n] = nidx(l, -m);
    }
    m0[l] = nidx(l, 0);
    blk[l] = l == 0 ? 0 : sp_nwig_of(l - 1);
  }
  blk[ydeg + 1] = sp_nwig_of(ydeg);
}

void sp_build_flux_constants(int ydeg, int udeg, std::vector<double> &rT,
                             std::vector<double> &A1, std::vector<double> &U1,
                             std::vector<double> &rta1) {
  const int LU = ydeg + udeg;
  const int NLU = (LU + 1) * (LU + 1), N = (ydeg + 1) * (ydeg + 1);
  solution_vector(LU, rT);
  change_of_basis(LU, A1);

  // rTA1 at degree ydeg (flux.h:302-309): the leading N x N block of A1 and
  // the leading N entries of rT do not depend on the total degree
  rta1.assign(N, 0.0);
  for (int c = 0; c < N; ++c) {
    double s = 0.0;
    for (int r = 0; r < N; ++r) {
      const double a = A1[(size_t)r * NLU + c];
      if (a != 0.0) s += rT[r] * a;
    }
    rta1[c] = s;
  }

  // limb-darkening basis U1 (flux.h:332-409)
  const int nu1 = (udeg + 1) * (udeg + 1);
  U1.assign((size_t)nu1 * (udeg + 1), 0.0);
  if (udeg == 0) return;
  const int W = LU + 1;
  std::vector<double> LT((size_t)W * W, 0.0), YT((size_t)W * W, 0.0);
  for (int l = 0; l < W; ++l) {
    double lck = 1.0;
    for (int k = 0; k <= l; ++k) {
      LT[(size_t)k * W + l] = ((k + 1) % 2 == 0) ? lck : -lck;
      lck *= (l - k) / (k + 1.0);
    }
  }
  for (int par = 0; par < 2; ++par) {
    double twol = par == 0 ? 1.0 : 2.0, lfac = 1.0, fac0 = par == 0 ? 1.0 : 0.5;
    for (int l = par; l < W; l += 2) {
      const double a = twol * std::sqrt((2 * l + 1) / (4 * M_PI)) / lfac;
      double lck = par == 0 ? 1.0 : (double)l, fac = fac0;
      for (int k = par; k <= l; k += 2) {
        YT[(size_t)k * W + l] = a * lck * fac;
        fac *= (k + l + 1.0) / (k - l + 1.0);
        lck *= (l - k) * (l - k - 1) / ((k + 1.0) * (k + 2.0));
      }
      fac0 *= par == 0 ? -0.25 * (l + 1) * (l + 1) : -0.25 * (l + 2) * l;
      lfac *= (l + 1.0) * (l + 2.0);
      twol *= 4.0;
    }
  }
  // YT is upper triangular: U0 = YT^{-1} LT by back substitution, then / norm
  const double norm = 2.0 / std::sqrt(M_PI);
  std::vector<double> U0((size_t)W * W, 0.0);
  for (int c = 0; c < W; ++c)
    for (int r = W - 1; r >= 0; --r) {
      double s = LT[(size_t)r * W + c];
      for (int k = r + 1; k < W; ++k) s -= YT[(size_t)r * W + k] * U0[(size_t)k * W + c];
      U0[(size_t)r * W + c] = s / YT[(size_t)r * W + r];
    }
  for (double &v : U0) v /= norm;
  // U1 = (A1 . X . U0)[:nu1, :udeg+1], X(l(l+1), l) = 1
  for (int r = 0; r < nu1; ++r)
    for (int c = 0; c <= udeg; ++c) {
      double s = 0.0;
      for (int l = 0; l < W; ++l)
        s += A1[(size_t)r * NLU + l * (l + 1)] * U0[(size_t)l * W + c];
      U1[(size_t)r * (udeg + 1) + c] = s;
    }
}

// v -> rT . L(p) . A1 for a limb-darkening polynomial p in the (udeg + 1)^2 basis; LINEAR
// in p (flux.h:415-441 computeLp, 500-523)
static void rTA1L_linear(const sp_handle *h, const double *p, double *out) {
  const int ydeg = h->ydeg, udeg = h->udeg, N = h->N;
  const int LU = ydeg + udeg, NLU = (LU + 1) * (LU + 1);
  // v = rT . Lp, column n1 of Lp being (Ylm-basis monomial n1) x p
  std::vector<double> v(N, 0.0), col(NLU);
  for (int l1 = 0; l1 <= ydeg; ++l1)
    for (int m1 = -l1; m1 <= l1; ++m1) {
      const int n1 = nidx(l1, m1);
      for (int i = 0; i < NLU; ++i) col[i] = 0.0;
      for (int l2 = 0; l2 <= udeg; ++l2)
        for (int m2 = -l2; m2 <= l2; ++m2)
          mono_mul_acc(col.data(), l1, m1, l2, m2, p[nidx(l2, m2)]);
      double s = 0.0;
      for (int r = 0; r < NLU; ++r) s += h->rT[r] * col[r];
      v[n1] = s;
    }
  for (int c = 0; c < N; ++c) {
    double s = 0.0;
    for (int r = 0; r < N; ++r) s += v[r] * h->A1[(size_t)r * NLU + c];
    out[c] = s;
  }
}

// limb-darkening polynomial p = U1 . [-1, u] and rT . p (flux.h:506-512)
static double ld_polynomial(const sp_handle *h, const double *u, double *p) {
  const int udeg = h->udeg, nu1 = (udeg + 1) * (udeg + 1);
  double dotp = 0.0;
  for (int r = 0; r < nu1; ++r) {
    double s = h->U1[(size_t)r * (udeg + 1)] * -1.0;
    for (int c = 1; c <= udeg; ++c) s += h->U1[(size_t)r * (udeg + 1) + c] * u[c - 1];
    p[r] = s;
    dotp += h->rT[r] * s;
  }
  return dotp;
}

// rTA1L(u) (flux.h:500-523)
void sp_host_rTA1L(const sp_handle *h, const double *u, double *out) {
  const int udeg = h->udeg, N = h->N;
  const int nu1 = (udeg + 1) * (udeg + 1);
  if (udeg == 0) {
    for (int i = 0; i < N; ++i) out[i] = h->rta1[i];
    return;
  }
  // normalised to pi / (rT . p)
  double p[(SP_MAX_UDEG + 1) * (SP_MAX_UDEG + 1)];
  const double dotp = ld_polynomial(h, u, p);
  const double scale = (1.0 / dotp) * M_PI;
  for (int r = 0; r < nu1; ++r) p[r] *= scale;
  rTA1L_linear(h, p, out);
}

// Reverse mode of rTA1L (flux.h:529-557): bu = (DDp bf)^T DpDu, with DDp the (constant)
// Jacobian of the linear map above and DpDu the derivative of the normalised polynomial.
void sp_host_rTA1L_rev(const sp_handle *h, const double *u, const double *bf, double *bu) {
  const int udeg = h->udeg, N = h->N;
  const int nu1 = (udeg + 1) * (udeg + 1);
  if (udeg == 0) return;
  double p[(SP_MAX_UDEG + 1) * (SP_MAX_UDEG + 1)], bp[(SP_MAX_UDEG + 1) * (SP_MAX_UDEG + 1)];
  const double dotp = ld_polynomial(h, u, p);
  const double norm = 1.0 / dotp;
  for (int r = 0; r < nu1; ++r) p[r] *= norm * M_PI;
  // bp = DDp . bf : row i of DDp is the image of the i-th unit polynomial
  std::vector<double> e(nu1), row(N);
  for (int i = 0; i < nu1; ++i) {
    for (int r = 0; r < nu1; ++r) e[r] = r == i ? 1.0 : 0.0;
    rTA1L_linear(h, e.data(), row.data());
    double s = 0.0;
    for (int n = 0; n < N; ++n) s += row[n] * bf[n];
    bp[i] = s;
  }
  // DpDu = pi norm U1 - p (rT . U1) norm   (columns 1..udeg <-> u)
  for (int c = 1; c <= udeg; ++c) {
    double rTU = 0.0;
    for (int r = 0; r < nu1; ++r) rTU += h->rT[r] * h->U1[(size_t)r * (udeg + 1) + c];
    double s = 0.0;
    for (int r = 0; r < nu1; ++r)
      s += bp[r] * (M_PI * norm * h->U1[(size_t)r * (udeg + 1) + c] - p[r] * rTU * norm);
    bu[c - 1] = s;
  }
}

extern "C" {

int sp_index_tables(int ydeg, int32_t *l_of, int32_t *m_of, int32_t *mirror,
                    int32_t *m0, int32_t *blk) {
  if (ydeg < 0 || ydeg > SP_MAX_YDEG || !l_of || !m_of || !mirror || !m0 || !blk)
    return SP_ERR_INVALID;
  sp_build_index_tables(ydeg, l_of, m_of, mirror, m0, blk);
  return SP_OK;
}

// cos/sin of -(k pi/2) ("alpha" = -pi/2 per step) and of +(k pi/2) ("gamma"),
// as exact integers from k mod 4, and the alternating sign (wigner.h:232-270).
int sp_wigner_int_tables(int ydeg, int32_t *cosmal, int32_t *sinmal,
                         int32_t *sgn, int32_t *cosmga, int32_t *sinmga) {
  if (ydeg < 0 || ydeg > SP_MAX_YDEG || !cosmal || !sinmal || !sgn || !cosmga ||
      !sinmga)
    return SP_ERR_INVALID;
  static const int c4[4] = {1, 0, -1, 0}, s4[4] = {0, 1, 0, -1};
  cosmal[0] = sinmal[0] = sgn[0] = cosmga[0] = sinmga[0] = 0;
  for (int k = 1; k <= ydeg; ++k) {
    cosmal[k] = c4[k & 3];
    sinmal[k] = -s4[k & 3];
    cosmga[k] = c4[k & 3];
    sinmga[k] = s4[k & 3];
    sgn[k] = (k & 1) ? -1 : 1;
  }
  return SP_OK;
}

// ops/norm/norm.py:26-44
int sp_alpha_beta(double z, int order, double *alpha, double *beta,
                  double *dalpha_dz, double *dbeta_dz) {
  if (order < 0) return SP_ERR_INVALID;
  double fac = 1.0, a = 0.0, b = 0.0, da = 0.0, db = 0.0, df = 0.0;
  for (int n = 0; n <= order; ++n) {
    da += df;
    db += 2 * n * df;
    df = (2 * n + 3) * (df * z + fac);
    a += fac;
    b += 2 * n * fac;
    fac *= z * (2 * n + 3);
  }
  if (alpha) *alpha = a;
  if (beta) *beta = b;
  if (dalpha_dz) *dalpha_dz = da;
  if (dbeta_dz) *dbeta_dz = db;
  return SP_OK;
}

int sp_version(void) { return 100; }

const char *sp_strerror(int status) {
  switch (status) {
    case SP_OK: return "ok";
    case SP_ERR_INVALID: return "invalid argument";
    case SP_ERR_HIP: return "HIP runtime error";
    case SP_ERR_NO_DEVICE: return "no usable gfx950 device";
    case SP_ERR_STATE: return "constants or Ylm moments not set";
    case SP_ERR_ALLOC: return "allocation failed";
    case SP_ERR_COMM: return "RCCL not loaded in this process, or the collective failed";
    default: return "unknown status";
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Upstream of the hot path: latitude moment integrals of
// the Beta-distributed spot latitude (reference ops/include/latitude.h:21-173,
// ops/include/special.h:172-232), values only -- the reference also carries
// d/dalpha, d/dbeta for its reverse-mode Ops, which are out of scope here.
// ---------------------------------------------------------------------------
namespace {

// Gauss 2F1 by its power series.  The reference stops when the value AND its
// two parameter derivatives have converged (special.h:187-189); the derivative
// terms are carried along only to stop at the same term.
double hyp2f1_series(double a, double b, double c, double z) {
  double term = a * b * z / c;
  double dtermdb = a * z / c, dtermdc = -term / c;
  double value = 1.0 + term, dfdb = dtermdb, dfdc = dtermdc;
  int n = 1;
  while ((std::fabs(term / value) > 1e-15 || std::fabs(dtermdb / dfdb) > 1e-13 ||
          std::fabs(dtermdc / dfdc) > 1e-13) &&
         n < 500) {
    a += 1;
    b += 1;
    c += 1;
    n += 1;
    const double fac1 = a * z / c / n;
    const double fac2 = fac1 * b;
    const double fac3 = -fac2 / c;
    dtermdb *= fac2;
    dtermdb += fac1 * term;
    dtermdc *= fac2;
    dtermdc += fac3 * term;
    term *= fac2;
    value += term;
    dfdb += dtermdb;
    dfdc += dtermdc;
  }
  return value;
}

}  // namespace

extern "C" int sp_latitude_integrals(int ydeg, double alpha, double beta, double *q,
                                     double *Q) {
  if (ydeg < 0 || ydeg > SP_MAX_YDEG || !q || !Q) return SP_ERR_INVALID;
  alpha = alpha > 0.0 ? alpha : 0.0;  // ops/latitude/latitude.cc:47-48
  beta = beta > 0.0 ? beta : 0.0;
  const int n = 4 * ydeg + 1, N = (ydeg + 1) * (ydeg + 1);
  std::vector<double> B(n), F(n), term((size_t)n * n, 0.0);
  B[0] = 1.0;
  for (int k = 1; k < n; ++k) {
    const double c1 = 1.0 / (alpha + beta + k - 1.0);
    B[k] = (alpha + k - 1.0) * c1 * B[k - 1];
  }
  const double ab = alpha + beta;
  F[0] = std::sqrt(2.0) * hyp2f1_series(-0.5, beta, ab, 0.5);
  if (n > 1) F[1] = std::sqrt(2.0) * hyp2f1_series(-0.5, beta, ab + 1.0, 0.5);
  for (int k = 2; k < n; ++k) {
    const double c1 = (ab + k - 1.0) / ((alpha + k - 1.0) * (ab + k - 0.5));
    const double c2 = c1 * (ab + k - 2.0);
    const double c3 = c1 * (1.5 - beta);
    F[k] = c2 * F[k - 2] + c3 * F[k - 1];
  }
  for (int k = 0; k < n; ++k) F[k] = F[k] * B[k];
  for (int i = 0; i < n; ++i) {
    const double *func = (i % 2 == 0) ? B.data() : F.data();
    const int i2 = (i % 2 == 0) ? i / 2 : (i - 1) / 2;
    for (int j = 0; j < n; j += 2) {
      const int j2 = j / 2;
      double fac1 = 1.0, acc = 0.0;
      for (int k1 = 0; k1 < i2 + 1; ++k1) {
        double fac2 = fac1;
        for (int k2 = 0; k2 < j2 + 1; ++k2) {
          acc += fac2 * func[k1 + k2];
          fac2 *= (k2 - j2) / (k2 + 1.0);
        }
        fac1 *= (i2 - k1) / (k1 + 1.0);
      }
      term[(size_t)i * n + j] = acc;
    }
  }
  int n1 = 0;
  double inv_two_l1 = 1.0;
  for (int l1 = 0; l1 <= ydeg; ++l1) {
    for (int m1 = -l1; m1 <= l1; ++m1) {
      const int j1 = m1 + l1, i1 = l1 - m1;
      q[n1] = term[(size_t)j1 * n + i1] * inv_two_l1;
      int n2 = 0;
      double inv = inv_two_l1;
      for (int l2 = 0; l2 <= ydeg; ++l2) {
        for (int m2 = -l2; m2 <= l2; ++m2) {
          const int j2 = m2 + l2, i2 = l2 - m2;
          Q[(size_t)n1 * N + n2] = term[(size_t)(j1 + j2) * n + (i1 + i2)] * inv;
          ++n2;
        }
        inv *= 0.5;
      }
      ++n1;
    }
    inv_two_l1 *= 0.5;
  }
  return SP_OK;
}

// ---------------------------------------------------------------------------
// Gauss-Jacobi rule for the weight (1 - t)^a (1 + t)^b on (-1, 1), a, b > -1,
// weights normalised to sum 1 (the rule of an expectation under a Beta law).
// Golub-Welsch: the nodes are the eigenvalues of the symmetric tridiagonal
// Jacobi matrix of the orthonormal recurrence, the weights the squared first
// components of its normalised eigenvectors -- the zeroth moment
// 2^(a+b+1) B(a+1, b+1), which overflows / underflows for the shape
// parameters at the edge of the reference's prior box (latitude.py:176-197:
// beta up to exp(10)), never appears.  Implicit-shift QL on (d, e) that carries
// only the first row of the eigenvector matrix.
// ---------------------------------------------------------------------------
extern "C" int sp_gauss_jacobi(int n, double a, double b, double *nodes, double *weights) {
  if (n < 1 || n > 4096 || !(a > -1.0) || !(b > -1.0) || !nodes || !weights ||
      !std::isfinite(a) || !std::isfinite(b))
    return SP_ERR_INVALID;
  std::vector<double> d(n), e(n, 0.0), z(n, 0.0);
  const double ab = a + b;
  d[0] = (b - a) / (ab + 2.0);
  for (int k = 1; k < n; ++k) {
    const double s = 2.0 * k + ab;
    d[k] = (b - a) * (b + a) / (s * (s + 2.0));
    // off-diagonal k-1 <-> k; for k = 1 the factor (k + a + b) / (2k + a + b - 1) is 1
    const double num = (k == 1) ? 4.0 * (1.0 + a) * (1.0 + b) / ((s * s) * (s + 1.0))
                                : 4.0 * k * (k + a) * (k + b) * (k + ab) / ((s * s) * (s + 1.0) * (s - 1.0));
    e[k - 1] = std::sqrt(num);
  }
  z[0] = 1.0;
  for (int l = 0; l < n; ++l) {
    for (int iter = 0;; ++iter) {
      int m = l;
      for (; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m == l) break;
      if (iter == 200) return SP_ERR_INVALID;
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = std::hypot(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      for (; i >= l; --i) {
        double f = s * e[i];
        const double bb = c * e[i];
        r = std::hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) {
          d[i + 1] -= p;
          e[m] = 0.0;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * bb;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - bb;
        f = z[i + 1];
        z[i + 1] = s * z[i] + c * f;
        z[i] = c * z[i] - s * f;
      }
      if (r == 0.0 && i >= l) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return d[x] < d[y]; });
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += z[i] * z[i];
  for (int i = 0; i < n; ++i) {
    nodes[i] = d[order[i]];
    weights[i] = z[order[i]] * z[order[i]] / sum;
  }
  return SP_OK;
}


// ---------------------------------------------------------------------------
// The same rule WITH its derivatives with respect to the two exponents: what the reference gets from the
// analytic d/d alpha, d/d beta of its latitude integrals (ops/include/latitude.h:21-173) the quadrature of
// rotations gets from here -- an n-point rule is exact for the polynomials it integrates WHATEVER (a, b), so
//   d/da sum_k w_k F(t_k)  =  sum_k dw_k/da F(t_k) + w_k F'(t_k) dt_k/da
// is the exact derivative of the expectation.  First-order perturbation of the symmetric eigenproblem J = V L V^T:
//   dt_i = v_i^T dJ v_i,      dv_i = sum_{j != i} v_j (v_j^T dJ v_i) / (t_i - t_j),      dw_i = 2 v_i[0] dv_i[0],
// with the eigenvectors rebuilt from the nodes by the orthonormal three-term recurrence (v_i[k] = sqrt(w_i) p_k(t_i))
// and dJ/da, dJ/db from the closed forms of the recurrence coefficients (carried as value + two derivatives).
// ---------------------------------------------------------------------------
namespace {
struct D2 {          // value and its derivatives with respect to a and b
  double v, a, b;
};
inline D2 operator+(D2 x, D2 y) { return {x.v + y.v, x.a + y.a, x.b + y.b}; }
inline D2 operator-(D2 x, D2 y) { return {x.v - y.v, x.a - y.a, x.b - y.b}; }
inline D2 operator*(D2 x, D2 y) { return {x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b}; }
inline D2 operator/(D2 x, D2 y) {
  const double q = x.v / y.v;
  return {q, (x.a - q * y.a) / y.v, (x.b - q * y.b) / y.v};
}
inline D2 konst(double c) { return {c, 0.0, 0.0}; }
inline D2 d2sqrt(D2 x) {
  const double r = std::sqrt(x.v);
  return {r, 0.5 * x.a / r, 0.5 * x.b / r};
}
}  // namespace

extern "C" int sp_gauss_jacobi_grad(int n, double a, double b, double *nodes, double *weights,
                                    double *dnodes_da, double *dweights_da, double *dnodes_db,
                                    double *dweights_db) {
  if (!dnodes_da || !dweights_da || !dnodes_db || !dweights_db) return SP_ERR_INVALID;
  int rc = sp_gauss_jacobi(n, a, b, nodes, weights);
  if (rc) return rc;
  const D2 A{a, 1.0, 0.0}, B{b, 0.0, 1.0}, AB = A + B;
  std::vector<D2> d(n), e(n > 1 ? n - 1 : 0);
  d[0] = (B - A) / (AB + konst(2.0));
  for (int k = 1; k < n; ++k) {
    const D2 s = konst(2.0 * k) + AB, kk = konst((double)k);
    d[k] = (B - A) * (B + A) / (s * (s + konst(2.0)));
    const D2 num = (k == 1) ? konst(4.0) * (konst(1.0) + A) * (konst(1.0) + B) / (s * s * (s + konst(1.0)))
                            : konst(4.0) * kk * (kk + A) * (kk + B) * (kk + AB) /
                                  (s * s * (s + konst(1.0)) * (s - konst(1.0)));
    e[k - 1] = d2sqrt(num);
  }
  // eigenvectors, column i = node i: forward recurrence, normalised
  std::vector<double> V((size_t)n * n);
  for (int i = 0; i < n; ++i) {
    double *v = &V[(size_t)i * n];
    const double x = nodes[i];
    v[0] = 1.0;
    if (n > 1) v[1] = (x - d[0].v) * v[0] / e[0].v;
    for (int k = 1; k + 1 < n; ++k) v[k + 1] = ((x - d[k].v) * v[k] - e[k - 1].v * v[k - 1]) / e[k].v;
    double nn = 0.0;
    for (int k = 0; k < n; ++k) nn += v[k] * v[k];
    nn = 1.0 / std::sqrt(nn);
    for (int k = 0; k < n; ++k) v[k] *= nn;
  }
  for (int which = 0; which < 2; ++which) {
    double *dt = which ? dnodes_db : dnodes_da, *dw = which ? dweights_db : dweights_da;
    auto dd = [&](int k) { return which ? d[k].b : d[k].a; };
    auto de = [&](int k) { return which ? e[k].b : e[k].a; };
    auto form = [&](const double *vj, const double *vi) {      // v_j^T dJ v_i
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc += dd(k) * vj[k] * vi[k];
      for (int k = 0; k + 1 < n; ++k) acc += de(k) * (vj[k] * vi[k + 1] + vj[k + 1] * vi[k]);
      return acc;
    };
    for (int i = 0; i < n; ++i) {
      const double *vi = &V[(size_t)i * n];
      dt[i] = form(vi, vi);
      double dv0 = 0.0;
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const double *vj = &V[(size_t)j * n];
        dv0 += vj[0] * form(vj, vi) / (nodes[i] - nodes[j]);
      }
      dw[i] = 2.0 * vi[0] * dv0;
    }
  }
  return SP_OK;
}
