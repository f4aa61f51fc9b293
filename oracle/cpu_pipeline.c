/* TEST / BENCH INFRASTRUCTURE -- never linked into or called by the product path.
 *
 * CPU restatement in plain C of the per-star part of the reference's marginal likelihood path
 * ("our C CPU restatement of the identical pipeline, OpenMP over stars, one
 * star per core"), used by bench.py's cpu_baseline leg and checked against the NumPy oracle and
 * the golden vectors in tests/test_cpu_pipeline.py:
 *
 *   theta = 2 pi mod(t / p, 1)                                      flux.py:262
 *   cov_ij = cubic(|theta_i - theta_j|) on the lag grid             flux.py:256-276
 *   normalisation with the AlphaBeta series                          sp.py:705-727, ops/norm/norm.py:26-44
 *   C = cov + data_var I                                             sp.py:1135-1151
 *   L = potrf(C); y = L^-1 r; lnlike                                 math.py:75-100, sp.py:1154-1188
 *
 * LAPACK's dpotrf / dtrtrs are passed in as function pointers (bench.py takes them from the
 * SciPy that runs the oracle: scipy.linalg.cython_lapack), so this file links against nothing.
 * The kernel table (xp, a0..a3), the flux mean and variance are per hyperparameter sample and
 * come from the oracle's Python (oracle/sp_oracle.py: kernel_table).                         */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef void (*potrf_fn)(char *uplo, int *n, double *a, int *lda, int *info);
typedef void (*trtrs_fn)(char *uplo, char *trans, char *diag, int *n, int *nrhs, double *a, int *lda,
                         double *b, int *ldb, int *info);

static void alpha_beta(double z, int N, double *alpha, double *beta) {
  double fac = 1.0, a = 0.0, b = 0.0;
  for (int n = 0; n <= N; ++n) {
    a += fac;
    b += 2 * n * fac;
    fac *= z * (2 * n + 3);
  }
  *alpha = a;
  *beta = b;
}

/* one star; work: K*K doubles.  Returns the log-likelihood (-inf on failure). */
static double star_lnlike(int K, const double *t, const double *flux, double period, double data_var,
                          int covpts, const double *xp, const double *a0, const double *a1,
                          const double *a2, const double *a3, double mean, double var, int normalized,
                          int norm_order, double zmax, potrf_fn potrf, trtrs_fn trtrs, double *C,
                          double *theta, double *q, double *r) {
  const double dx = 2.0 * M_PI / covpts;
  for (int i = 0; i < K; ++i) {
    double u = t[i] / period;
    u = u - floor(u);                     /* np.mod(x, 1) for finite x */
    theta[i] = 2.0 * M_PI * u;
  }
  double total = 0.0;
  if (K == 1) {
    C[0] = var;
    total = var;
    q[0] = var;
  } else {
    for (int i = 0; i < K; ++i) {
      double rs = 0.0;
      double *row = C + (size_t)i * K;
      for (int j = 0; j < K; ++j) {
        const double x = fabs(theta[i] - theta[j]);
        const long idx = (long)floor(x / dx);
        const double x0 = (x - xp[idx + 1]) / dx;
        const double v = a0[idx] + a1[idx] * x0 + a2[idx] * x0 * x0 + a3[idx] * x0 * x0 * x0;
        row[j] = v;
        rs += v;
      }
      q[i] = rs;
      total += rs;
    }
  }
  double z = 0.0;
  if (normalized) {
    const double mu = 1.0 + mean, m = total / ((double)K * K);
    z = m / (mu * mu);
    double alpha, beta;
    alpha_beta(z, norm_order, &alpha, &beta);
    for (int i = 0; i < K; ++i) q[i] /= (K * m);
    const double c1 = alpha / (mu * mu), cp = z * (alpha + beta), cq = z * alpha;
    for (int i = 0; i < K; ++i) {
      double *row = C + (size_t)i * K;
      const double pi = 1.0 - q[i];
      for (int j = 0; j < K; ++j) row[j] = c1 * row[j] + (cp * pi * (1.0 - q[j]) - cq * q[i] * q[j]);
    }
  }
  for (int i = 0; i < K; ++i) C[(size_t)i * K + i] += data_var;
  /* C is symmetric: LAPACK's column-major "U" is this row-major lower triangle, i.e. L^T = U */
  int n = K, info = 0, one = 1;
  char U = 'U', T = 'T', N = 'N';
  potrf(&U, &n, C, &n, &info);
  if (info != 0) return -INFINITY;
  const double gp_mean = normalized ? 0.0 : mean;
  for (int i = 0; i < K; ++i) r[i] = flux[i] - gp_mean;
  trtrs(&U, &T, &N, &n, &one, C, &n, r, &n, &info);     /* U^T y = r  (U^T = L) */
  if (info != 0) return -INFINITY;
  double quad = 0.0, logdet = 0.0;
  for (int i = 0; i < K; ++i) {
    quad += r[i] * r[i];
    logdet += log(C[(size_t)i * K + i]);
  }
  double v = -0.5 * quad - logdet - 0.5 * K * log(2.0 * M_PI);
  if (normalized && z > zmax) v = -INFINITY;
  if (v != v) v = -INFINITY;
  return v;
}

/* S stars, OpenMP over stars (one star per thread at a time).  t, flux: [S][K].
 * tab: [5][covpts + 4] = xp, a0, a1, a2, a3 (a* hold covpts + 1 entries).  Returns the number of
 * threads used; seconds (wall) in *seconds.                                                  */
int sp_cpu_lnlike(int S, int K, const double *t, const double *flux, const double *period,
                  const double *data_var, int covpts, const double *tab, double mean, double var,
                  int normalized, int norm_order, double zmax, void *potrf, void *trtrs,
                  int nthreads, double *lnlike, double *seconds) {
  const int np = covpts + 4;
  const double *xp = tab, *a0 = tab + np, *a1 = tab + 2 * np, *a2 = tab + 3 * np, *a3 = tab + 4 * np;
  if (nthreads > 0) omp_set_num_threads(nthreads);
  int used = 1;
  const double w0 = omp_get_wtime();
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
    double *C = (double *)malloc(sizeof(double) * ((size_t)K * K + 3 * (size_t)K));
    double *theta = C + (size_t)K * K, *q = theta + K, *r = q + K;
#pragma omp for schedule(dynamic, 1)
    for (int s = 0; s < S; ++s)
      lnlike[s] = star_lnlike(K, t + (size_t)s * K, flux + (size_t)s * K, period[s], data_var[s], covpts,
                              xp, a0, a1, a2, a3, mean, var, normalized, norm_order, zmax,
                              (potrf_fn)potrf, (trtrs_fn)trtrs, C, theta, q, r);
    free(C);
  }
  *seconds = omp_get_wtime() - w0;
  return used;
}
