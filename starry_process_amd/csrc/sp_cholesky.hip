// Batched fp64 Cholesky, triangular solves and the likelihood reduction
//  for gfx950.
//
// Right-looking blocked factorisation, panels of SP_NB = 64 columns:
//   panel_kernel   factors the 64x64 diagonal block in LDS and solves the rows
//                  below it (X L_d^T = P), 256 rows per workgroup;
//   sp_launch_gemm_nt (sp_gemm.hip) applies the trailing update C -= X X^T on
//                  the matrix cores, lower-triangle tiles only.
// The systems are padded to a multiple of 64 rows and carry the residual
// vectors as EXTRA ROWS below the matrix: factoring
//     [ C   . ]          gives          [ L   . ]
//     [ r^T . ]                         [ y^T . ] ,   y = L^-1 r,
// so r^T C^-1 r = |y|^2 falls out of the factorisation and no separate
// triangular solve is needed for the likelihood.
#include "sp_internal.h"

#define DLD 65  // padded row length of the diagonal block in LDS

namespace {

__device__ __forceinline__ double read_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Factor the leading nact columns of the 64-row block held in LDS (rows beyond
// nact only receive the triangular solve).  Executed by ONE wavefront; lane r
// keeps row r in registers and the right-looking rank-1 updates fetch L[k][c]
// from lane k with v_readlane -- no LDS traffic and no barriers inside the
// 64-column sweep.
__device__ __forceinline__ void block_potrf_wave(double *sD, int nact, int lane,
                                                 int *bad) {
  double x[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) x[c] = sD[lane * DLD + c];
  int notpd = 0;
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    if (c < nact) {
      const double piv = read_lane(x[c], c);
      if (!(piv > 0.0)) notpd = 1;
      const double d = sqrt(piv);
      x[c] = (lane == c) ? d : x[c] / d;
#pragma unroll
      for (int k = c + 1; k < 64; ++k) x[k] -= x[c] * read_lane(x[c], k);
    }
  }
#pragma unroll
  for (int c = 0; c < 64; ++c)
    if (c <= lane) sD[lane * DLD + c] = x[c];
  if (notpd) *bad = 1;
}

// grid (nchunks, S).  Every workgroup factors the diagonal block redundantly
// (cheap, and it avoids a grid-wide dependency); chunk 0 writes it back.
__global__ __launch_bounds__(256) void panel_kernel(double *__restrict__ sys,
                                                    long ld, long stride, int Kp,
                                                    int c0, int nact,
                                                    int32_t *__restrict__ info) {
  __shared__ double sD[64 * DLD];
  __shared__ int s_bad;
  double *Mx = sys + (size_t)blockIdx.y * stride;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_bad = 0;
  // stage the diagonal block (coalesced: 16 lanes x 4 doubles per row)
  {
    const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      const double *src = Mx + (size_t)(c0 + r) * ld + c0 + cj;
#pragma unroll
      for (int e = 0; e < 4; ++e) sD[r * DLD + cj + e] = src[e];
    }
  }
  __syncthreads();
  if (tid < 64) block_potrf_wave(sD, nact, lane, &s_bad);
  __syncthreads();
  if (blockIdx.x == 0) {
    if (tid == 0 && s_bad && info) info[blockIdx.y] = 1;
    const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      double *dst = Mx + (size_t)(c0 + r) * ld + c0 + cj;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cj + e <= r && cj + e < nact) dst[e] = sD[r * DLD + cj + e];
    }
  }
  // rows below the block: x L_d^T = p, one row per thread
  const int row = c0 + 64 + blockIdx.x * 256 + tid;
  if (row < Kp) {
    double *p = Mx + (size_t)row * ld + c0;
    double x[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) x[c] = p[c];
#pragma unroll
    for (int c = 0; c < 64; ++c) {
      if (c < nact) {
        double v = x[c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= x[k] * sD[c * DLD + k];
        x[c] = v / sD[c * DLD + c];
      }
    }
#pragma unroll
    for (int c = 0; c < 64; ++c)
      if (c < nact) p[c] = x[c];
  }
}

// lnlike = -1/2 sum_m |y_m|^2 - M sum_i log L_ii - K M / 2 log(2 pi)
// (sp.py:1157-1188).  One workgroup per star.
__global__ __launch_bounds__(256) void lnlike_reduce_kernel(
    const double *__restrict__ sys, long ld, long stride, int K, int M,
    const int32_t *__restrict__ info, double *__restrict__ lnlike,
    uint32_t *__restrict__ status) {
  __shared__ double red[8];
  const int s = blockIdx.x;
  const double *Mx = sys + (size_t)s * stride;
  double ld_part = 0.0, q_part = 0.0;
  for (int i = threadIdx.x; i < K; i += 256) ld_part += log(Mx[(size_t)i * ld + i]);
  for (int m = 0; m < M; ++m) {
    const double *y = Mx + (size_t)(K + m) * ld;
    for (int k = threadIdx.x; k < K; k += 256) q_part += y[k] * y[k];
  }
  for (int off = 32; off > 0; off >>= 1) {
    ld_part += __shfl_down(ld_part, off, 64);
    q_part += __shfl_down(q_part, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = ld_part;
    red[4 + (threadIdx.x >> 6)] = q_part;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double logdet = (red[0] + red[1]) + (red[2] + red[3]);
    const double quad = (red[4] + red[5]) + (red[6] + red[7]);
    double v = -0.5 * quad;
    v -= M * logdet;
    v -= 0.5 * K * M * 1.8378770664093453;  // log(2 pi)
    uint32_t st = status ? status[s] : 0u;
    if (info && info[s]) st |= SP_STAR_NOT_PD;
    if (v != v) st |= SP_STAR_NAN;
    if (st & (SP_STAR_NOT_PD | SP_STAR_ZMAX | SP_STAR_NAN)) v = -INFINITY;
    lnlike[s] = v;
    if (status) status[s] = st;
  }
}

// copy a batch of K x K matrices into zero/identity padded Kp x Kp systems
__global__ __launch_bounds__(256) void pad_in_kernel(const double *__restrict__ A,
                                                     int K, long lda, long strideA,
                                                     double *__restrict__ sys, int Kp,
                                                     long strideS, int M,
                                                     const double *__restrict__ resid) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Kp) return;
  double v = 0.0;
  if (i < K && j < K)
    v = A[(size_t)s * strideA + (size_t)i * lda + j];
  else if (i >= K && i < K + M && j < K && resid)
    v = resid[((size_t)s * M + (i - K)) * K + j];
  else if (i == j)
    v = 1.0;
  sys[(size_t)s * strideS + (size_t)i * Kp + j] = v;
}

// copy the lower factor back, zero the strict upper triangle, NaN on failure
__global__ __launch_bounds__(256) void pad_out_kernel(const double *__restrict__ sys,
                                                      int Kp, long strideS,
                                                      double *__restrict__ A, int K,
                                                      long lda, long strideA,
                                                      const int32_t *__restrict__ info) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  double v = j <= i ? sys[(size_t)s * strideS + (size_t)i * Kp + j] : 0.0;
  if (info && info[s]) v = __builtin_nan("");
  A[(size_t)s * strideA + (size_t)i * lda + j] = v;
}

// x = (L L^T)^-1 b (math.py:97-100).  grid (nrhs, batch), one right-hand side
// per workgroup; b is [K, nrhs] row-major per matrix.  Blocked substitution:
// the 64x64 diagonal blocks are solved by one wavefront with lane shuffles, the
// off-diagonal updates are one row (forward) / one column (backward) per thread.
__global__ __launch_bounds__(256) void cho_solve_kernel(
    const double *__restrict__ Lall, int K, long ldl, long strideL,
    double *__restrict__ Ball, int nrhs) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int Kr = ((K + 63) / 64) * 64;
  double *x = lds;        // Kr
  double *sL = lds + Kr;  // 64 * DLD
  const double *L = Lall + (size_t)blockIdx.y * strideL;
  double *B = Ball + (size_t)blockIdx.y * K * nrhs + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < Kr; i += 256) x[i] = i < K ? B[(size_t)i * nrhs] : 0.0;
  const int nb = Kr / 64;
  __syncthreads();
  // forward: L y = b
  for (int blk = 0; blk < nb; ++blk) {
    const int c0 = blk * 64, n = K - c0 < 64 ? K - c0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      sL[r * DLD + c] = (c0 + r < K && c <= r) ? L[(size_t)(c0 + r) * ldl + c0 + c]
                                               : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = x[c0 + lane];
      for (int c = 0; c < n; ++c) {
        const double xc = __shfl(v, c, 64) / sL[c * DLD + c];
        if (lane == c)
          v = xc;
        else if (lane > c)
          v -= sL[lane * DLD + c] * xc;
      }
      x[c0 + lane] = v;
    }
    __syncthreads();
    for (int j = c0 + 64 + tid; j < K; j += 256) {
      const double *row = L + (size_t)j * ldl + c0;
      double acc = x[j];
      for (int k = 0; k < n; ++k) acc -= row[k] * x[c0 + k];
      x[j] = acc;
    }
    __syncthreads();
  }
  // backward: L^T x = y
  for (int blk = nb - 1; blk >= 0; --blk) {
    const int c0 = blk * 64, n = K - c0 < 64 ? K - c0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      sL[r * DLD + c] = (c0 + r < K && c <= r) ? L[(size_t)(c0 + r) * ldl + c0 + c]
                                               : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = x[c0 + lane];
      for (int c = n - 1; c >= 0; --c) {
        const double xc = __shfl(v, c, 64) / sL[c * DLD + c];
        if (lane == c)
          v = xc;
        else if (lane < c)
          v -= sL[c * DLD + lane] * xc;
      }
      x[c0 + lane] = v;
    }
    __syncthreads();
    for (int j = tid; j < c0; j += 256) {
      double acc = x[j];
      for (int k = 0; k < n; ++k) acc -= L[(size_t)(c0 + k) * ldl + j] * x[c0 + k];
      x[j] = acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < K; i += 256) B[(size_t)i * nrhs] = x[i];
}

}  // namespace

// ---- launchers ---------------------------------------------------------------

// In-place factorisation of S padded systems (Kp x Kp, ld = Kp): the leading
// K x K part is factored, rows K..Kp-1 only receive the triangular solve.
int sp_launch_cholesky_systems(sp_handle *h, double *sys, int S, int K, int Kp,
                               int32_t *info, hipStream_t st) {
  const long ld = Kp, stride = (long)Kp * Kp;
  const int nsteps = (K + SP_NB - 1) / SP_NB;
  for (int j = 0; j < nsteps; ++j) {
    const int c0 = j * SP_NB;
    const int nact = K - c0 < SP_NB ? K - c0 : SP_NB;
    const int below = Kp - (c0 + SP_NB);
    int nchunks = (below + 255) / 256;
    if (nchunks < 1) nchunks = 1;
    hipLaunchKernelGGL(panel_kernel, dim3(nchunks, S), dim3(256), 0, st, sys, ld,
                       stride, Kp, c0, nact, info);
    SP_LAUNCH_CHECK();
    const int c1 = c0 + SP_NB;
    if (c1 < K) {
      const int n = Kp - c1;
      double *X = sys + (size_t)c1 * ld + c0;
      double *T = sys + (size_t)c1 * ld + c1;
      const bool timed = h && h->prof_on && h->prof_used + 2 <= h->prof_ev.size();
      if (timed) SP_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
      int rc = sp_launch_gemm_nt(X, ld, stride, X, ld, stride, T, ld, stride, n, n,
                                 SP_NB, -1.0, 1, 1, S, st);
      if (rc != SP_OK) return rc;
      if (timed) {
        SP_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
        h->prof_used += 2;
        // algorithmic work of a symmetric rank-64 update of an n x n block:
        // n (n + 1) / 2 entries x 64 multiply-adds
        h->prof_flops += (double)S * (double)n * (n + 1) * SP_NB;
        h->prof_launches += 1;
      }
    }
  }
  return SP_OK;
}

int sp_launch_lnlike_reduce(const double *sys, int S, int K, int M, int Kp,
                            const int32_t *info, double *lnlike, uint32_t *status,
                            hipStream_t st) {
  hipLaunchKernelGGL(lnlike_reduce_kernel, dim3(S), dim3(256), 0, st, sys,
                     (long)Kp, (long)Kp * Kp, K, M, info, lnlike, status);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_pad_in(const double *A, int K, long lda, long strideA, double *sys,
                     int Kp, int M, const double *resid, int S, hipStream_t st) {
  hipLaunchKernelGGL(pad_in_kernel, dim3((Kp + 255) / 256, Kp, S), dim3(256), 0,
                     st, A, K, lda, strideA, sys, Kp, (long)Kp * Kp, M, resid);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_pad_out(const double *sys, int Kp, double *A, int K, long lda,
                      long strideA, const int32_t *info, int S, hipStream_t st) {
  hipLaunchKernelGGL(pad_out_kernel, dim3((K + 255) / 256, K, S), dim3(256), 0, st,
                     sys, Kp, (long)Kp * Kp, A, K, lda, strideA, info);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_cho_solve(const double *L, int K, long ldl, long strideL, double *B,
                        int nrhs, int batch, hipStream_t st) {
  const int Kr = ((K + 63) / 64) * 64;
  const size_t lds = sizeof(double) * ((size_t)Kr + 64 * DLD);
  if (lds > 150 * 1024) return SP_ERR_INVALID;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cho_solve_kernel),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(cho_solve_kernel, dim3(nrhs, batch), dim3(256), lds, st, L, K,
                     ldl, strideL, B, nrhs);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
