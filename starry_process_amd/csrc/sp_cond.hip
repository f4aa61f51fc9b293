// Conditional branch of the flux covariance (flux.py:337-343; SURVEY 8a a13 with a14-a16 fused):
//
//     Sigma_flux = A Sigma_y A^T            A: K x N design matrix of the star (a12)
//
// as  B1 = A Sigma_y  (sp_launch_gemm_nt, 128 x 128 MM2 tiles) and then, HERE, the LOWER 64 x 64
// tiles of  B1 A^T  on the matrix cores (sp_mm.h, MM2<64, 64, 8, 6, 4>) with the assembly of the
// system in the product's epilogue: a tile leaves the accumulators multiplied by its temporal
// factor (temporal.py:8-16), masked to the star's cadences, with its row / column partial sums for
// the deferred normalisation (sp.py:705-727; sp_assemble.hip) -- or, for an un-normalised process,
// with the data variance and the baseline variance added (sp.py:1135-1151) -- straight into the
// padded system, next to the residual rows and the identity padding.
//
// Round 2 formed the whole K x K product (both triangles), wrote it (0.5 GB per 64-star step at
// K = 1000) and had assemble_kernel read it back: 40 % of the flops and the round trip are gone.
//
// Operands: A and B1 hold Kr = roundup(K, 64) rows per star (rows >= K zero), N % 16 == 0 columns.
#include "sp_internal.h"
#include "sp_tile.h"
#include "sp_cov.h"
#include "sp_mm.h"

namespace {

typedef SpCoef CondCoef;

using CondCore = MM2<64, 64, 8, 6, 4>;

// DEFER: raw tiles + partial sums (deferred normalisation); else: un-normalised process, noise added
template <bool DEFER>
__global__ __launch_bounds__(256) void cond_system_kernel(
    const double *__restrict__ B1, const double *__restrict__ A, int N, long strideAB, int K, int M,
    int Kp, const double *__restrict__ t, const sp_star *__restrict__ stars, int temporal,
    const CondCoef *__restrict__ coef, const double *__restrict__ diag,
    const double *__restrict__ flux, double *__restrict__ sys, double *__restrict__ part, int ntr,
    int batch) {
  __shared__ __attribute__((aligned(16))) double lds[CondCore::LDS_DOUBLES];
  int s, tile;
  if (!sp_xcd_decode(blockIdx.x, batch, ntr * (ntr + 1) / 2, s, tile)) return;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int i0 = 64 * ti, j0 = 64 * tj;
  const sp_star st = stars[s];
  const int nobs = star_nobs(st, K);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;

  mm_d4 acc[1][4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[0][n] = mm_d4{0.0, 0.0, 0.0, 0.0};
  if (i0 < K) {   // (then j0 < K too: a tile of the product)
    CondCore mm;
    mm.init(B1 + (size_t)s * strideAB + (size_t)i0 * N, N, A + (size_t)s * strideAB + (size_t)j0 * N, N);
    mm.prologue(lds, 0, N);
    mm.loop(lds, 0, N, acc);
  }
  // accumulator (n, r) of this lane: row i0 + 16 wave + fk + 4 r, column j0 + 16 n + fr
  const double gpmean = DEFER ? 0.0 : coef[s].gpmean;
  double tr[4], tc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + 16 * wave + fk + 4 * k, j = j0 + 16 * k + fr;
    tr[k] = (temporal != SP_TEMPORAL_NONE && i < nobs) ? t[(size_t)s * K + i] : 0.0;
    tc[k] = (temporal != SP_TEMPORAL_NONE && j < nobs) ? t[(size_t)s * K + j] : 0.0;
  }
  double rsum[4] = {0.0, 0.0, 0.0, 0.0}, csum[4] = {0.0, 0.0, 0.0, 0.0};
  double *ob = sys + (size_t)s * Kp * Kp;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 16 * wave + fk + 4 * r;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int j = j0 + 16 * n + fr;
      double val = 0.0;
      if (i < nobs && j < nobs) {
        const double rawv = acc[0][n][r] * temporal_factor(temporal, tr[r], tc[n], st.tau);
        val = rawv;
        if (DEFER) {
          rsum[r] += rawv;
          csum[n] += rawv;
        } else {
          if (i == j) val += diag ? diag[(size_t)s * K + i] : st.data_var;
          val += st.baseline_var;
        }
      } else if (i >= K && i < K + M && j < nobs) {
        val = flux[((size_t)s * M + (i - K)) * K + j] - (gpmean + st.baseline_mean);
      } else if (i == j) {
        val = 1.0;
      }
      ob[(size_t)i * Kp + j] = val;
    }
  }
  if (!DEFER) return;
  // row sums of this tile's 64 columns: over the 16 lanes that share (wave, fk)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v = rsum[r];
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    const int i = i0 + 16 * wave + fk + 4 * r;
    if (fr == 0 && i < K) part[((size_t)s * ntr + tj) * K + i] = v;
  }
  if (ti > tj) {
    // column sums = row sums of the mirror tile (tj, ti), which is never formed: over the 4 lane
    // groups of a wavefront, then over the 4 wavefronts (fixed order: deterministic)
    __syncthreads();   // (the product's last slice has been read)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      double v = csum[n];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (fk == 0) lds[wave * 64 + 16 * n + fr] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const double a = (lds[threadIdx.x] + lds[64 + threadIdx.x]) + (lds[128 + threadIdx.x] + lds[192 + threadIdx.x]);
      const int j = j0 + threadIdx.x;
      if (j < K) part[((size_t)s * ntr + ti) * K + j] = a;
    }
  }
}

}  // namespace

// The lower tiles of B1 A^T into the padded systems, assembled (see the header).  defer: part != null.
int sp_launch_cond_system(const double *B1, const double *A, int N, int Kr, int S, int K, int M, int Kp,
                          const double *t, const sp_star *stars, int temporal, const void *coef,
                          const double *diag, const double *flux, double *sys, double *part,
                          hipStream_t st) {
  if (S <= 0) return SP_OK;
  if ((N % 16) || (Kr % 64) || Kr < K || (Kp % 64) || Kp < K + M) return SP_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B1)) & 15) return SP_ERR_INVALID;
  const int ntr = Kp / 64;
  const long nblk = sp_xcd_grid(S, (long)ntr * (ntr + 1) / 2);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  if (part)
    hipLaunchKernelGGL((cond_system_kernel<true>), dim3((unsigned)nblk), dim3(256), 0, st, B1, A, N,
                       (long)Kr * N, K, M, Kp, t, stars, temporal, (const CondCoef *)coef, diag, flux, sys,
                       part, ntr, S);
  else
    hipLaunchKernelGGL((cond_system_kernel<false>), dim3((unsigned)nblk), dim3(256), 0, st, B1, A, N,
                       (long)Kr * N, K, M, Kp, t, stars, temporal, (const CondCoef *)coef, diag, flux, sys,
                       part, ntr, S);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
