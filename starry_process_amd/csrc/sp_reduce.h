// The reduction of one star's factored system to its log-likelihood, shared by lnlike_reduce_kernel
// (sp_cholesky.hip) and by the panel kernel, whose workgroup that factors the LAST pivot block goes on
// to reduce its star when the extra rows live in that block's row tile (sp_panel.hip, SpReduceArgs).
#ifndef SP_REDUCE_H
#define SP_REDUCE_H

#include "sp_internal.h"

// lnlike = -1/2 sum_m |y_m|^2 - M sum_i log L_ii - K M / 2 log(2 pi)
// (sp.py:1157-1188).  One workgroup (256 threads) per star.
//
// coef != nullptr: deferred normalisation (sp_assemble.hip: defer_finish_kernel, or the planned step's
// assemble_planned_kernel).  The factored matrix is B = Sigma + D / c1 = L L^T (D: the data variances) and the
// true covariance is
//     C = c1 (B + d_p p p^T + d_1 1 1^T + d_q q q^T),    q = Sigma 1 / (K m),  p = 1 - q      (sp.py:705-727).
// Round 5: q never exists as a vector.  Sigma 1 = B 1 - d (d = diag(D) / c1), so every Gram entry of q under
// B^-1 follows from rows that ride anyway -- u_1 = L^-1 1 in row K + M and, for per-cadence variances only,
// u_d = L^-1 d in row K + M + 1 (a scalar variance: u_d = delta u_1) -- and from sums of the data:
//     1^T B^-1 q = (K - u_1.u_d) / (K m)
//     q^T B^-1 q = (K^2 m - sum(d) + u_d.u_d) / (K m)^2
//     r^T B^-1 q = (sum(r) - y.u_d) / (K m)
// (K: the star's valid cadences; tools/planned_identities.py checks them against the oracle).  Rounds 2-4
// carried p and q as two more rows, which took the row sums of Sigma BEFORE the factorisation: a pass over
// all K^2 entries per evaluation.  rscal_s: {K m, sum(d), delta, sum(r_0), sum(r_1), ...} (SP_RSCAL_HEAD + M
// doubles per star).  The matrix determinant lemma and the Sherman-Morrison formula, one rank at a time (the
// two non-negative terms first), give log det C and r^T C^-1 r from those Gram entries; a rank-1 step whose
// pivot 1 + d u^T B^-1 u is not positive means C is not positive definite: the same -inf the reference's
// failed Cholesky gives (math.py:82-91, sp.py:1186-1188).
typedef SpCoef RedCoef;   // (deferred form: zab = d_p, za = d_q)

// COHERENT: read past the L1 (agent-scope loads) -- for a caller whose inputs were written by ANOTHER workgroup of
// the same launch.  (The panel kernel's tail does not need it: what it reads was written by earlier launches or
// by its own workgroup, ordered by the barrier it takes first.)
template <bool COHERENT>
__device__ __forceinline__ double red_ld(const double *p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// Where the reduction reads the factored system: diag(i) = L_ii, row(m, k) = entry k of riding row m (row K + m of the
// system: the residual rows 0 .. M - 1, then L^-1 1, then L^-1 d).  RedSrcMem: the system in memory, row-major with
// leading dimension ld (every caller but the small-K kernel, which holds them in LDS arrays of its own: sp_small.hip).
template <bool COHERENT>
struct RedSrcMem {
  const double *Mx;
  long ld;
  int K;
  __device__ __forceinline__ double diag(int i) const { return red_ld<COHERENT>(Mx + (size_t)i * ld + i); }
  __device__ __forceinline__ double row(int m, int k) const { return red_ld<COHERENT>(Mx + (size_t)(K + m) * ld + k); }
};

// red: 48 doubles of LDS; s-indexed pointers are the star's own entries (null where the caller has none).
// NT: the threads that take part -- 256, the whole workgroup (barriers inside), or 64: the caller's FIRST WAVEFRONT
// alone, the other three must not call (no barrier inside, `red` unused; the small-K kernel, whose stars are at most
// 128 cadences long: four wavefronts through the logarithms, the shuffles and the scalar finish were a quarter of that
// kernel's vector instructions, tools/small_k_pmc.sh).  U: entries per thread and pass of the first loop.
template <bool COHERENT, class SRC, int NT = 256, int U = 4>
__device__ __forceinline__ void lnlike_reduce_src(
    const SRC src, int K, int M, const int32_t *info_s,
    double *__restrict__ lnlike_s, uint32_t *status_s, uint32_t *status_out_s,
    const sp_star *star_s, const RedCoef *coef_s, const double *rscal_s, int dvec, double *red, int tid,
    int notpd_in = 0) {
  const int wave = tid >> 6;
  // sums of v[0 .. 8) over the workgroup, in every thread (all eight always: constant indices
  // keep v in registers; the unused ones are zero)
  auto block_sum = [&](double (&v)[8]) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
      for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
    if (NT == 64) {
#pragma unroll
      for (int a = 0; a < 8; ++a) v[a] = __shfl(v[a], 0, 64);
      return;
    }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
      for (int a = 0; a < 8; ++a) red[wave * 8 + a] = v[a];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = (red[a] + red[8 + a]) + (red[16 + a] + red[24 + a]);
  };
  double v[8];
  for (int a = 0; a < 8; ++a) v[a] = 0.0;
  // (the first light curve's residuals, row 0, ride in the same pass as rows M and M + 1)
  const bool defer = coef_s != nullptr;
  const bool dv = defer && dvec;
  // (four rows of loads in flight at a time: the diagonal is one cache line per entry, and one
  //  entry per round trip made this loop 5 us of a 9 us reduction)
  for (int base = 0; base < K; base += NT * U) {
    double dg[U], r[U], pa[U], pb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + tid + NT * u;
      const bool ok = i < K;
      const int ii = ok ? i : 0;
      dg[u] = src.diag(ii);
      r[u] = src.row(0, ii);
      pa[u] = defer ? src.row(M, ii) : 0.0;
      pb[u] = dv ? src.row(M + 1, ii) : 0.0;
      if (!ok) {
        dg[u] = 1.0;
        r[u] = pa[u] = pb[u] = 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[0] += log(dg[u]);
      const double rr = r[u], a = pa[u], b = pb[u];
      v[1] += a * a; v[2] += a * b; v[3] += b * b;
      v[4] += rr * rr; v[5] += rr * a; v[6] += rr * b;
    }
  }
  block_sum(v);
  const double logdet = v[0];
  // (ragged ensembles: the padding rows have unit pivots and zero residuals, only the constants know the
  //  number of valid cadences)
  const int nobs = (star_s && star_s->nobs > 0 && star_s->nobs < K) ? star_s->nobs : K;
  // rank-1 steps on the 3 x 3 Gram matrix H (0 = p, 1 = 1, 2 = q): factor f_k and old column c_k
  double f[3] = {0.0, 0.0, 0.0}, col[3][3], logs = 0.0;
  bool notpd = notpd_in != 0;
  double c1 = 1.0, km = 1.0, delta = 0.0;
  if (defer) {
    const RedCoef rc = *coef_s;
    c1 = rc.c1;
    km = rscal_s[0];
    delta = rscal_s[2];
    const double sd = rscal_s[1];
    const double G11 = v[1], G1d = dv ? v[2] : delta * v[1], Gdd = dv ? v[3] : delta * delta * v[1];
    const double H1q = ((double)nobs - G1d) / km;
    const double Hqq = ((double)nobs * km - sd + Gdd) / (km * km);
    double H[3][3] = {{G11 - 2.0 * H1q + Hqq, G11 - H1q, H1q - Hqq}, {G11 - H1q, G11, H1q}, {H1q - Hqq, H1q, Hqq}};
    const double d[3] = {rc.zab, rc.d1, rc.za};
    for (int k = 0; k < 3; ++k) {
      for (int a = 0; a < 3; ++a) col[k][a] = H[a][k];
      if (d[k] == 0.0) continue;
      const double piv = 1.0 + d[k] * H[k][k];
      if (!(piv > 0.0)) notpd = true;
      logs += log(piv);
      f[k] = d[k] / piv;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) H[a][b] -= f[k] * col[k][a] * col[k][b];
    }
  }
  double quad = 0.0;
  for (int m = 0; m < M; ++m) {
    double w[8];
    for (int a = 0; a < 8; ++a) w[a] = 0.0;
    if (m == 0) {
      w[0] = v[4]; w[1] = v[5]; w[2] = v[6];
    } else {
      for (int k = tid; k < K; k += NT) {
        const double r = src.row(m, k);
        w[0] += r * r;
        if (defer) w[1] += r * src.row(M, k);
        if (dv) w[2] += r * src.row(M + 1, k);
      }
      block_sum(w);
    }
    double g = w[0], h[3] = {0.0, 0.0, 0.0};
    if (defer) {
      const double h1 = w[1], hq = (rscal_s[SP_RSCAL_HEAD + m] - (dv ? w[2] : delta * w[1])) / km;
      h[0] = h1 - hq;
      h[1] = h1;
      h[2] = hq;
    }
    for (int k = 0; k < 3; ++k) {
      if (f[k] == 0.0) continue;
      const double hk = h[k];
      g -= f[k] * hk * hk;
      for (int a = 0; a < 3; ++a) h[a] -= f[k] * hk * col[k][a];
    }
    quad += g;
  }
  if (tid == 0) {
    double val = -0.5 * quad / c1;
    val -= M * (logdet + 0.5 * nobs * log(c1) + 0.5 * logs);
    val -= 0.5 * nobs * M * 1.8378770664093453;  // log(2 pi)
    uint32_t st = status_s ? *status_s : 0u;
    const int bad = info_s ? (COHERENT ? __hip_atomic_load(info_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *info_s) : 0;
    if (bad || notpd) st |= SP_STAR_NOT_PD;
    if (val != val) st |= SP_STAR_NAN;
    if (st & (SP_STAR_NOT_PD | SP_STAR_ZMAX | SP_STAR_NAN)) val = -INFINITY;
    // (a plan that does not belong to these stars is a caller's error, not a rejected sample: NaN, so that it cannot
    //  pass for one)
    if (st & SP_STAR_STALE_PLAN) val = __builtin_nan("");
    *lnlike_s = val;
    if (status_s) *status_s = st;
    if (status_out_s) *status_out_s = st;
  }
}

template <bool COHERENT>
__device__ __forceinline__ void lnlike_reduce_body(
    const double *__restrict__ Mx, long ld, int K, int M, const int32_t *info_s,
    double *__restrict__ lnlike_s, uint32_t *status_s, uint32_t *status_out_s,
    const sp_star *star_s, const RedCoef *coef_s, const double *rscal_s, int dvec, double *red, int tid) {
  lnlike_reduce_src<COHERENT>(RedSrcMem<COHERENT>{Mx, ld, K}, K, M, info_s, lnlike_s, status_s, status_out_s, star_s, coef_s,
                              rscal_s, dvec, red, tid);
}

#endif
