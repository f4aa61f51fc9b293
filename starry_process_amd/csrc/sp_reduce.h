// The reduction of one star's factored system to its log-likelihood, shared by lnlike_reduce_kernel
// (sp_cholesky.hip) and by the panel kernel, whose workgroup that factors the LAST pivot block goes on
// to reduce its star when the extra rows live in that block's row tile (sp_panel.hip, SpReduceArgs).
#ifndef SP_REDUCE_H
#define SP_REDUCE_H

#include "sp_internal.h"

// lnlike = -1/2 sum_m |y_m|^2 - M sum_i log L_ii - K M / 2 log(2 pi)
// (sp.py:1157-1188).  One workgroup (256 threads) per star.
//
// coef != nullptr: deferred normalisation (sp_assemble.hip, defer_finish_kernel).  The factored
// matrix is B'' = Sigma + N / c1 and the true covariance is
//     C = c1 (B'' + d_p p p^T + d_1 1 1^T + d_q q q^T),
// with y_p, y_q, y_1 = L''^-1 p, q, 1 in the three rows below the residuals.  The matrix
// determinant lemma and the Sherman-Morrison formula, one rank at a time (the two non-negative
// terms first), give log det C and r^T C^-1 r from the Gram matrix of those rows and the
// residuals'; a rank-1 step whose pivot 1 + d u^T B^-1 u is not positive means C is not positive
// definite: the same -inf the reference's failed Cholesky gives (math.py:82-91, sp.py:1186-1188).
typedef SpCoef RedCoef;   // (deferred form: zab = d_p, za = d_q)

// COHERENT: read past the L1 (agent-scope loads) -- for a caller whose inputs were written by ANOTHER workgroup of
// the same launch.  (The panel kernel's tail does not need it: what it reads was written by earlier launches or
// by its own workgroup, ordered by the barrier it takes first.)
template <bool COHERENT>
__device__ __forceinline__ double red_ld(const double *p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// red: 48 doubles of LDS; s-indexed pointers are the star's own entries (null where the caller has none)
template <bool COHERENT>
__device__ __forceinline__ void lnlike_reduce_body(
    const double *__restrict__ Mx, long ld, int K, int M, const int32_t *info_s,
    double *__restrict__ lnlike_s, uint32_t *status_s, uint32_t *status_out_s,
    const sp_star *star_s, const RedCoef *coef_s, double *red, int tid) {
  const int wave = tid >> 6;
  // sums of v[0 .. 12) over the workgroup, in every thread (all twelve always: constant indices
  // keep v in registers; the unused ones are zero)
  auto block_sum = [&](double (&v)[12]) {
#pragma unroll
    for (int a = 0; a < 12; ++a)
      for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
      for (int a = 0; a < 12; ++a) red[wave * 12 + a] = v[a];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] = (red[a] + red[12 + a]) + (red[24 + a] + red[36 + a]);
  };
  double v[12];
  for (int a = 0; a < 12; ++a) v[a] = 0.0;
  const double *yp = Mx + (size_t)(K + M) * ld, *yq = yp + ld, *y1 = yq + ld;
  const double *y0 = Mx + (size_t)K * ld;       // the first light curve's residuals ride in the same pass
  const bool defer = coef_s != nullptr;
  // (four rows of loads in flight at a time: the diagonal is one cache line per entry, and one
  //  entry per round trip made this loop 5 us of a 9 us reduction)
  for (int base = 0; base < K; base += 1024) {
    double dg[4], r[4], pa[4], pb[4], pc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + tid + 256 * u;
      const bool ok = i < K;
      const int ii = ok ? i : 0;
      dg[u] = red_ld<COHERENT>(Mx + (size_t)ii * ld + ii);
      r[u] = red_ld<COHERENT>(y0 + ii);
      if (defer) {
        pa[u] = red_ld<COHERENT>(yp + ii);
        pb[u] = red_ld<COHERENT>(yq + ii);
        pc[u] = red_ld<COHERENT>(y1 + ii);
      } else {
        pa[u] = pb[u] = pc[u] = 0.0;
      }
      if (!ok) {
        dg[u] = 1.0;
        r[u] = pa[u] = pb[u] = pc[u] = 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[0] += log(dg[u]);
      const double rr = r[u], a = pa[u], b = pb[u], c = pc[u];
      v[7] += rr * rr;
      v[1] += a * a; v[2] += a * b; v[3] += a * c; v[4] += b * b; v[5] += b * c; v[6] += c * c;
      v[8] += rr * a; v[9] += rr * c; v[10] += rr * b;
    }
  }
  block_sum(v);
  const double logdet = v[0];
  // rank-1 steps on the 3 x 3 Gram matrix H (0 = p, 1 = 1, 2 = q): factor f_k and old column c_k
  double f[3] = {0.0, 0.0, 0.0}, col[3][3], logs = 0.0;
  bool notpd = false;
  double c1 = 1.0;
  if (defer) {
    const RedCoef rc = *coef_s;
    c1 = rc.c1;
    double H[3][3] = {{v[1], v[3], v[2]}, {v[3], v[6], v[5]}, {v[2], v[5], v[4]}};
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
    const double *y = Mx + (size_t)(K + m) * ld;
    double w[12];
    for (int a = 0; a < 12; ++a) w[a] = 0.0;
    if (m == 0) {
      w[0] = v[7]; w[1] = v[8]; w[2] = v[9]; w[3] = v[10];
    } else {
      for (int k = tid; k < K; k += 256) {
        const double r = red_ld<COHERENT>(y + k);
        w[0] += r * r;
        if (defer) {
          w[1] += r * red_ld<COHERENT>(yp + k);
          w[2] += r * red_ld<COHERENT>(y1 + k);
          w[3] += r * red_ld<COHERENT>(yq + k);
        }
      }
      block_sum(w);
    }
    double g = w[0], h[3] = {w[1], w[2], w[3]};
    for (int k = 0; k < 3; ++k) {
      if (f[k] == 0.0) continue;
      const double hk = h[k];
      g -= f[k] * hk * hk;
      for (int a = 0; a < 3; ++a) h[a] -= f[k] * hk * col[k][a];
    }
    quad += g;
  }
  if (tid == 0) {
    // (ragged ensembles: the padding rows have unit pivots and zero residuals, only
    //  the constants know the number of valid cadences)
    const int nobs = (star_s && star_s->nobs > 0 && star_s->nobs < K) ? star_s->nobs : K;
    double val = -0.5 * quad / c1;
    val -= M * (logdet + 0.5 * nobs * log(c1) + 0.5 * logs);
    val -= 0.5 * nobs * M * 1.8378770664093453;  // log(2 pi)
    uint32_t st = status_s ? *status_s : 0u;
    const int bad = info_s ? (COHERENT ? __hip_atomic_load(info_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *info_s) : 0;
    if (bad || notpd) st |= SP_STAR_NOT_PD;
    if (val != val) st |= SP_STAR_NAN;
    if (st & (SP_STAR_NOT_PD | SP_STAR_ZMAX | SP_STAR_NAN)) val = -INFINITY;
    *lnlike_s = val;
    if (status_s) *status_s = st;
    if (status_out_s) *status_out_s = st;
  }
}

#endif
