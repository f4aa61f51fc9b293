// Reverse sweep of the marginal-branch log-likelihood for an ENSEMBLE of stars (round 4; the counterpart of
// theano.grad(sp.log_likelihood(...), [r, a, b, c, n]) for the whole batch, tests/test_lnlike.py:100-136).
//
//   lnL_s = -1/2 r^T C^-1 r - 1/2 log det C - K/2 log 2 pi,      C = C(table_s, flux mean)      (sp.py:1129-1188)
//   d lnL / dC = G = (alpha alpha^T - C^-1) / 2,  alpha = C^-1 r
//
// C^-1 comes from sp_spd_inverse_batched (the identity riding through the blocked factorisation, sp_api.hip).
// C depends on the hyperparameters only through the star's kernel TABLE yp[covpts + 4] (the second moment on the
// lag grid, flux.py:310-320) and the scalar flux mean:  Sigma_ij = spline(|theta_i - theta_j|; yp) T_ij is LINEAR
// in yp, and the normalisation (sp.py:705-727)
//     C = c1 Sigma + s1 p p^T - s2 q q^T + D + b 1 1^T,   q = Sigma 1 / (K m),  p = 1 - q,  m = mean(Sigma),
//     z = m / mu^2,  c1 = alpha_n(z) / mu^2,  s1 = z (alpha_n + beta_n),  s2 = z alpha_n,  mu = 1 + flux mean
// adds a rank-2 term whose adjoint needs nothing but products of C^-1 with r, p, q, 1:
//     <G, dC> = <H, dSigma> + kappa_mu d mu,      H_ij = c1 G_ij + w_i + w_j
// (the scalars of grad_scalars_kernel).  The sweep therefore ends in ONE pass over the K^2 entries per star that
// scatters H_ij T_ij times the four cubic weights of entry (i, j) into the table's adjoint ybar[covpts + 4]; the
// chain from (r, a, b, c, n) to the table is five numbers long and taken by the caller (grad.py).
#include "sp_internal.h"
#include "sp_cov.h"

namespace {

__device__ __forceinline__ double wsum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// C^-1 [p, q, 1, r_0 .. r_{M-1}] per star from the LOWER tiles of C^-1 alone (the inverse comes out of the blocked
// factorisation as its lower tiles, the diagonal ones complete; rounds 4-5 mirrored them into the upper ones first -- 8 MB
// written and 8 more read per star -- and summed down the columns of the full matrix, one row per memory round trip:
// 0.12 + 0.17 ms of the 2.36 ms sweep at 64 x 1000).  Four vectors per pass (v0 = 4 * pass): vec[s][0..M+2][K] =
// C^-1 p, C^-1 q, C^-1 1, alpha_m = C^-1 r_m.
// One workgroup per lower tile (ta >= tb), grid (ntr (ntr + 1) / 2, S): the tile is read ONCE (sixteen loads per thread in
// flight) and gives both of its products,
//   down its columns: sum_r T[r][c] x[64 ta + r]  -> entries 64 tb + c of the result, slot ta;
//   along its rows:   sum_c T[r][c] x[64 tb + c]  -> entries 64 ta + r, slot tb   (through the LDS; not for a diagonal tile,
//                                                    which is complete and counted once),
// into part[s][slot][k][K]: every (block of the result, slot) has exactly one writer, and grad_matvec_reduce_kernel adds
// the slots in a fixed order.
__global__ __launch_bounds__(256) void grad_matvec_kernel(
    int K, int Kr, int M, int v0, const double *__restrict__ Cinv, const double *__restrict__ flux,
    const sp_star *__restrict__ stars, const SpCoef *__restrict__ coef, const double *__restrict__ qv,
    int normalized, double *__restrict__ part) {
  __shared__ double red[4][4][64];
  __shared__ double tile[64][65];
  __shared__ double xs[2][4][64];         // [0]: entries 64 ta + ., [1]: entries 64 tb + .
  const int s = blockIdx.y, c = threadIdx.x & 63, jq = threadIdx.x >> 6, ntr = Kr / 64;
  const int t = blockIdx.x;
  int ta = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);     // row tile (ta >= tb)
  while (ta * (ta + 1) / 2 > t) --ta;
  while ((ta + 1) * (ta + 2) / 2 <= t) ++ta;
  const int tb = t - ta * (ta + 1) / 2;
  const double *T = Cinv + (size_t)s * Kr * Kr + (size_t)(64 * ta) * Kr + 64 * tb;
  double v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = T[(size_t)(jq + 4 * u) * Kr + c];
  if (threadIdx.x < 128) {
    // entry j of the pass's four vectors (the first pass: p, q, ones and the first light curve's residuals -- all there
    // is with one light curve per star; further passes: the residuals of light curves v0 - 3 .. v0, past the last the
    // last one again, not stored); zero beyond the K cadences
    const int which = threadIdx.x >> 6, j = 64 * (which == 0 ? ta : tb) + c;
    const double shift = coef[s].gpmean + stars[s].baseline_mean;
    double x[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < K) {
      if (v0 == 0) {
        const double q = normalized ? qv[(size_t)s * K + j] : 0.0;
        x[0] = 1.0 - q;
        x[1] = q;
        x[2] = 1.0;
        x[3] = flux[(size_t)s * M * K + j] - shift;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = v0 + k - 3;
          x[k] = flux[((size_t)s * M + (m < M ? m : M - 1)) * K + j] - shift;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[which][k][c] = x[k];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) tile[jq + 4 * u][c] = v[u];
  __syncthreads();
  double *P = part + (size_t)s * ntr * 4 * K;
  // down the columns
  {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += v[u] * xs[0][k][jq + 4 * u];
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][jq][c] = a[k];
  }
  __syncthreads();
  if (jq == 0 && 64 * tb + c < K) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      P[((size_t)ta * 4 + k) * K + 64 * tb + c] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
  }
  if (ta == tb) return;
  // along the rows: thread (c, jq) is row c of the tile, columns cc = jq mod 4
  __syncthreads();
  {
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int cc = jq; cc < 64; cc += 4) {
      const double w = tile[c][cc];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += w * xs[1][k][cc];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k][jq][c] = a[k];
  }
  __syncthreads();
  if (jq == 0 && 64 * ta + c < K) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      P[((size_t)tb * 4 + k) * K + 64 * ta + c] = (red[k][0][c] + red[k][1][c]) + (red[k][2][c] + red[k][3][c]);
  }
}

// vec[s][v0 + k][i] = sum over the slots, in their order; grid (ceil(K / 256), S, 4)
__global__ __launch_bounds__(256) void grad_matvec_reduce_kernel(int K, int ntr, int NV, int v0,
                                                                 const double *__restrict__ part, double *__restrict__ vec) {
  const int s = blockIdx.y, k = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K || v0 + k >= NV) return;
  const double *P = part + (size_t)s * ntr * 4 * K + (size_t)k * K + i;
  double a = 0.0;
  for (int sl = 0; sl < ntr; ++sl) a += P[(size_t)sl * 4 * K];
  vec[((size_t)s * NV + v0 + k) * K + i] = a;
}

// one workgroup per star: the dot products, lnL, the scalar adjoints and the vector w (into vec[s][0], over C^-1 p);
// the alpha_m stay in vec[s][3 + m].  With M light curves on one covariance (sp.py:1162-1171)
//   lnL = sum_m -1/2 r_m^T C^-1 r_m - M/2 log det C - M K/2 log 2 pi,      G = (sum_m alpha_m alpha_m^T - M C^-1) / 2:
// every quadratic form in alpha below is summed over m, every term of C^-1 alone counts M times.
__global__ __launch_bounds__(256) void grad_scalars_kernel(
    int K, int Kr, int M, const double *__restrict__ Cinv, const double *__restrict__ flux,
    const sp_star *__restrict__ stars, const SpCoef *__restrict__ coef, const double *__restrict__ qv,
    const double *__restrict__ diag, const double *__restrict__ logdet, const int32_t *__restrict__ info,
    int normalized, int order, double zmax, double *__restrict__ vec, double *__restrict__ dots /* [S][M][2] */,
    double *__restrict__ lnlike, double *__restrict__ meanbar, double *__restrict__ hcoef, uint32_t *__restrict__ status) {
  __shared__ double red[5][4];
  const int s = blockIdx.x, tid = threadIdx.x, NV = M + 3;
  const sp_star st = stars[s];
  const SpCoef c = coef[s];
  const double shift = c.gpmean + st.baseline_mean;
  double *V = vec + (size_t)s * NV * K;
  const double *Ci = Cinv + (size_t)s * Kr * Kr;
  auto block_sums = [&](double (&d)[5]) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const double v = wsum(d[k]);
      if ((tid & 63) == 0) red[k][tid >> 6] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
  };
  // the matrix's own: 0 p.Cp  1 q.Cq  2 q.Cp  3 1.C1  4 tr(Cinv D)
  double mt[5] = {0, 0, 0, 0, 0};
  for (int i = tid; i < K; i += 256) {
    const double cp = V[i], cq = V[K + i], c1v = V[2 * K + i];
    const double q = normalized ? qv[(size_t)s * K + i] : 0.0, p = 1.0 - q;
    const double D = diag ? diag[(size_t)s * K + i] : st.data_var;
    mt[0] += p * cp;
    mt[1] += q * cq;
    mt[2] += q * cp;
    mt[3] += c1v;
    mt[4] += Ci[(size_t)i * Kr + i] * D;
  }
  block_sums(mt);
  // per light curve: r.a, a.p, a.q, a.1, a.D.a -- summed as the forms the adjoints need
  double R = 0.0, P2 = 0.0, Q2 = 0.0, PQ = 0.0, O2 = 0.0, O1 = 0.0, AD = 0.0;
  for (int m = 0; m < M; ++m) {
    const double *al = V + (size_t)(3 + m) * K, *f = flux + ((size_t)s * M + m) * K;
    double d[5] = {0, 0, 0, 0, 0};
    for (int i = tid; i < K; i += 256) {
      const double a = al[i];
      const double q = normalized ? qv[(size_t)s * K + i] : 0.0;
      const double D = diag ? diag[(size_t)s * K + i] : st.data_var;
      d[0] += (f[i] - shift) * a;
      d[1] += a * (1.0 - q);
      d[2] += a * q;
      d[3] += a;
      d[4] += a * a * D;
    }
    block_sums(d);
    R += d[0];
    P2 += d[1] * d[1];
    Q2 += d[2] * d[2];
    PQ += d[1] * d[2];
    O2 += d[3] * d[3];
    O1 += d[3];
    AD += d[4];
    if (tid == 0) {
      dots[((size_t)s * M + m) * 2] = d[1];
      dots[((size_t)s * M + m) * 2 + 1] = d[2];
    }
  }
  __syncthreads();       // (dots: written by thread 0, read by all below -- through memory, same workgroup)
  const double Kd = (double)K, Md = (double)M;
  // (the sweep takes every cadence as valid: a ragged star -- 0 < nobs < K -- would get a silently wrong value from
  //  the K-long products above; it gets NaN and SP_STAR_NAN instead)
  const bool ragged = st.nobs > 0 && st.nobs < K;
  const bool bad = (info && info[s] != 0) || (normalized && c.z > zmax) || ragged;
  double c1 = 1.0, wconst = 0.0, gscale_u = 0.0, gscale_v = 0.0, mbar = O1;
  if (normalized) {
    const double z = c.z, mu = c.mu, m = c.m;
    // alpha_n(z), beta_n(z) and their derivatives (ops/norm/norm.py:26-44): f_0 = 1, f_{n+1} = f_n z (2 n + 3)
    double f = 1.0, fp = 0.0, an = 0.0, bn = 0.0, dan = 0.0, dbn = 0.0;
    for (int n = 0; n <= order; ++n) {
      an += f;
      bn += 2 * n * f;
      dan += fp;
      dbn += 2 * n * fp;
      const double fn = f * z * (2 * n + 3), fpn = (2 * n + 3) * (f + z * fp);
      f = fn;
      fp = fpn;
    }
    c1 = an / (mu * mu);
    const double ab = an + bn, s1 = z * ab, s2 = z * an, b = st.baseline_var;
    const double A1 = 0.5 * (P2 - Md * mt[0]), A2 = 0.5 * (Q2 - Md * mt[1]);
    const double aSa = (R - s1 * P2 + s2 * Q2 - AD - b * O2) / c1;
    const double CiS = (Kd - s1 * mt[0] + s2 * mt[1] - mt[4] - b * mt[3]) / c1;
    const double A0 = 0.5 * (aSa - Md * CiS);
    const double uq = 0.5 * (PQ - Md * mt[2]);
    const double gq = -2.0 * (s1 * uq + s2 * A2);
    const double kz = A0 * dan / (mu * mu) + A1 * (ab + z * (dan + dbn)) - A2 * (an + z * dan);
    const double km = kz / (mu * mu) - gq / m;
    mbar = -2.0 * z * kz / mu - 2.0 * A0 * an / (mu * mu * mu);
    wconst = km / (2.0 * Kd * Kd);
    gscale_u = -2.0 * s1 / (2.0 * Kd * m);
    gscale_v = -2.0 * s2 / (2.0 * Kd * m);
  }
  // w_i = g_i / (2 K m) + kappa_m / (2 K^2),  g = -2 (s1 u + s2 v),  u = G p = (sum_m alpha_m (a_m.p) - M C^-1 p) / 2,
  // v = G q likewise
  const double *dt = dots + (size_t)s * M * 2;
  for (int i = tid; i < K; i += 256) {
    const double cp = V[i], cq = V[K + i];
    double up = 0.0, vq = 0.0;
    for (int m = 0; m < M; ++m) {
      const double al = V[(size_t)(3 + m) * K + i];
      up += al * dt[2 * m];
      vq += al * dt[2 * m + 1];
    }
    const double u = 0.5 * (up - Md * cp), v = 0.5 * (vq - Md * cq);
    V[i] = bad ? 0.0 : gscale_u * u + gscale_v * v + wconst;
  }
  if (tid == 0) {
    const double ll = -0.5 * R - 0.5 * Md * logdet[s] - 0.5 * Md * Kd * 1.8378770664093453;   // log(2 pi)
    const bool dead = bad || !(ll == ll);
    lnlike[s] = ragged ? __builtin_nan("") : (dead ? -INFINITY : ll);
    meanbar[s] = dead ? 0.0 : mbar;
    hcoef[s] = dead ? 0.0 : c1;           // (0: the scatter adds nothing for a star the likelihood rejects)
    if (status) status[s] = ((info && info[s]) ? SP_STAR_NOT_PD : 0u) | ((normalized && c.z > zmax) ? SP_STAR_ZMAX : 0u) |
                            (((!(ll == ll) && !bad) || ragged) ? SP_STAR_NAN : 0u);
  }
}

// scatter of H_ij T_ij c_m(x0_ij) into the table's adjoint: one workgroup per LOWER 64 x 64 tile (C^-1 and H are
// symmetric: the tiles below the diagonal count twice; the lower storage is read along its rows: coalesced), the
// bins in LDS (ds_add_f64), one partial table per workgroup (grad_bins_reduce_kernel adds them in a fixed order)
template <int TK, bool ONE>     // ONE: one light curve per star (M == 1: no loop over them per entry)
__global__ __launch_bounds__(256) void grad_scatter_kernel(
    int K, int Kr, int Mrt, const double *__restrict__ Cinv, const double *__restrict__ theta, const double *__restrict__ t,
    const sp_star *__restrict__ stars, int covpts, const double *__restrict__ vec, const double *__restrict__ hcoef,
    double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double bins[];   // covpts + 4
  const int M = ONE ? 1 : Mrt;
  const int s = blockIdx.y, np = covpts + 4, tid = threadIdx.x;
  for (int k = tid; k < np; k += 256) bins[k] = 0.0;
  __syncthreads();
  const int tile = blockIdx.x;
  int ta = (int)((sqrtf(8.0f * tile + 1.0f) - 1.0f) * 0.5f);     // row tile (ta >= tb)
  while (ta * (ta + 1) / 2 > tile) --ta;
  while ((ta + 1) * (ta + 2) / 2 <= tile) ++ta;
  const int tb = tile - ta * (ta + 1) / 2;
  const int i = tb * 64 + (tid & 63), rq = tid >> 6;
  const double c1 = hcoef[s];
  const sp_star st = stars[s];
  if (i < K && c1 != 0.0) {
    const double *V = vec + (size_t)s * (M + 3) * K, *Al = V + 3 * (size_t)K;     // w in V[0], alpha_m in V[3 + m]
    const double *Ci = Cinv + (size_t)s * Kr * Kr;
    const double thi = theta[(size_t)s * K + i], ai = Al[i], wi = V[i], Md = (double)M;
    const double ti = TK != SP_TEMPORAL_NONE ? t[(size_t)s * K + i] : 0.0;
    const double dx = 6.283185307179586 / covpts, inv_dx = 1.0 / dx;
    const double mult = ta > tb ? 2.0 : 1.0;
    const int jend = ta * 64 + 64 < K ? ta * 64 + 64 : K;
    for (int j = ta * 64 + rq; j < jend; j += 4) {
      double aa = ai * Al[j];
      for (int m = 1; m < M; ++m) aa += Al[(size_t)m * K + i] * Al[(size_t)m * K + j];
      const double H = mult * (c1 * 0.5 * (aa - Md * Ci[(size_t)j * Kr + i]) + wi + V[j]);
      const double T = temporal_factor(TK, ti, TK != SP_TEMPORAL_NONE ? t[(size_t)s * K + j] : 0.0, st.tau);
      // the segment of the lag and the position inside it: SplineGen's index (flux.py:262-265)
      int idx;
      double x;
      {
#pragma clang fp contract(off)
        const double lag = fabs(thi - theta[(size_t)s * K + j]);
        const double qd = lag * inv_dx;
        idx = (int)qd;
        x = qd - (double)idx;
        if (fabs(x - 0.5) > 0.5 - 1.0e-9) {
          idx = (int)floor(lag / dx);
          x = qd - (double)idx;
        }
        idx = idx < 0 ? 0 : (idx > covpts ? covpts : idx);
      }
      // value = sum_m yp[idx + m] c_m(x):  a0 = y1, a1 = -y0/3 - y1/2 + y2 - y3/6, a2 = (y0 + y2)/2 - y1,
      // a3 = ((y1 - y2) + (y3 - y0)/3)/2   (flux.py:322-330)
      const double x2 = x * x, x3 = x2 * x, HT = H * T;
      atomicAdd(&bins[idx], HT * (-x / 3.0 + 0.5 * x2 - x3 / 6.0));
      atomicAdd(&bins[idx + 1], HT * (1.0 - 0.5 * x - x2 + 0.5 * x3));
      atomicAdd(&bins[idx + 2], HT * (x + 0.5 * x2 - 0.5 * x3));
      atomicAdd(&bins[idx + 3], HT * (-x / 6.0 + x3 / 6.0));
    }
  }
  __syncthreads();
  double *P = partial + ((size_t)s * gridDim.x + blockIdx.x) * np;
  for (int k = tid; k < np; k += 256) P[k] = bins[k];
}

// ybar[s][k] = sum over the tiles' partial tables, in a fixed order: thread (k, g) adds the partials w = g, g + 4, ...,
// the four groups are added in order.  grid (ceil(np / 64), S)
__global__ __launch_bounds__(256) void grad_bins_reduce_kernel(int np, int nwg, const double *__restrict__ partial,
                                                               double *__restrict__ ybar) {
  __shared__ double red[4][64];
  const int s = blockIdx.y, k = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  double a = 0.0;
  if (k < np) {
#pragma unroll 4
    for (int w = g; w < nwg; w += 4) a += partial[((size_t)s * nwg + w) * np + k];
  }
  red[g][threadIdx.x & 63] = a;
  __syncthreads();
  if (g == 0 && k < np)
    ybar[(size_t)s * np + k] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

int sp_launch_grad_sweep(int S, int K, int Kr, int M, double *Cinv, const double *theta, const double *t,
                         const double *flux, const sp_star *stars, const void *coef, const double *qv,
                         const double *diag, const double *logdet, const int32_t *info, int covpts, int temporal,
                         int normalized, int order, double zmax, double *vec, double *dots, double *hcoef,
                         double *partial, double *lnlike, double *ybar, double *meanbar, uint32_t *status,
                         hipStream_t st) {
  const int ntr = Kr / 64, np = covpts + 4;
  // (part: the scatter's partial tables later -- [S][ntr][4][K] doubles of it here, sp_api.hip: grad_layout)
  for (int v0 = 0; v0 < M + 3; v0 += 4) {
    hipLaunchKernelGGL(grad_matvec_kernel, dim3(ntr * (ntr + 1) / 2, S), dim3(256), 0, st, K, Kr, M, v0, Cinv, flux, stars,
                       (const SpCoef *)coef, qv, normalized, partial);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(grad_matvec_reduce_kernel, dim3((K + 255) / 256, S, 4), dim3(256), 0, st, K, ntr, M + 3, v0, partial,
                       vec);
    SP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(grad_scalars_kernel, dim3(S), dim3(256), 0, st, K, Kr, M, Cinv, flux, stars, (const SpCoef *)coef,
                     qv, diag, logdet, info, normalized, order, zmax, vec, dots, lnlike, meanbar, hcoef, status);
  SP_LAUNCH_CHECK();
  const size_t lds = sizeof(double) * np;
  if (lds > 60 * 1024) return SP_ERR_INVALID;
  dim3 grid(ntr * (ntr + 1) / 2, S);
#define SP_SCATTER(TK)                                                                                               \
  do {                                                                                                               \
    if (M == 1)                                                                                                      \
      hipLaunchKernelGGL((grad_scatter_kernel<TK, true>), grid, dim3(256), lds, st, K, Kr, M, Cinv, theta, t, stars,  \
                         covpts, vec, hcoef, partial);                                                               \
    else                                                                                                             \
      hipLaunchKernelGGL((grad_scatter_kernel<TK, false>), grid, dim3(256), lds, st, K, Kr, M, Cinv, theta, t, stars, \
                         covpts, vec, hcoef, partial);                                                               \
  } while (0)
  if (temporal == SP_TEMPORAL_NONE) SP_SCATTER(SP_TEMPORAL_NONE);
  else if (temporal == SP_TEMPORAL_MATERN32) SP_SCATTER(SP_TEMPORAL_MATERN32);
  else if (temporal == SP_TEMPORAL_EXPSQUARED) SP_SCATTER(SP_TEMPORAL_EXPSQUARED);
  else return SP_ERR_INVALID;
#undef SP_SCATTER
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(grad_bins_reduce_kernel, dim3((np + 63) / 64, S), dim3(256), 0, st, np, ntr * (ntr + 1) / 2, partial, ybar);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
