// Batched fp64 "NT" matrix product on the CDNA4 matrix cores:
//
//     C[b] = beta * C[b] + alpha * A[b] . B[b]^T          (beta in {0, 1})
//
// This one kernel carries the two GEMM-shaped steps of the hot path:
//   - the trailing update of the blocked Cholesky, C -= X X^T   
//   - the conditional covariance A Sigma_y A^T as two products  
//
// Tiling: one 256-thread workgroup (4 wavefronts, one per SIMD)
// owns a 64 x 64 tile of C; wavefront w owns rows 16w..16w+15 as four
// 16 x 16 accumulators of v_mfma_f64_16x16x4_f64.  A and B^T row panels are
// staged through LDS in 32-deep slices, rows padded to 34 doubles (68 dwords =
// 4 mod 64 banks) so that the per-lane fragment reads (16 rows x 2 k per
// 32-lane half) are bank-conflict free with ds_read_b64.
//
// Workgroup -> tile mapping is XCD-aware: blockIdx.x % 8 selects the XCD
// (round-robin dispatch), and all tiles of one matrix are given to one XCD so
// that the row panels every tile re-reads stay in that XCD's 4 MiB L2.
#include "sp_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define GT 64   // tile edge
#define GK 32   // k slice
#define GLD 34  // padded LDS row length (doubles)

namespace {

__device__ __forceinline__ void stage_panel(const double *P,
                                            long ld, int row0, int nrows,
                                            int k0, int Kd, double scale,
                                            bool vec_ok, double *__restrict__ s) {
  // 64 rows x 32 k  ->  s[row][k], 4 passes of 16 rows, 16 lanes x 16 B per row
  const int t = threadIdx.x;
  const int cpair = (t & 15) * 2;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = (t >> 4) + 16 * pass;
    const int gr = row0 + r;
    d2 v = {0.0, 0.0};
    if (gr < nrows) {
      const double *src = P + (size_t)gr * ld + k0 + cpair;
      if (vec_ok && k0 + cpair + 1 < Kd) {
        v = *reinterpret_cast<const d2 *>(src);
      } else {
        if (k0 + cpair < Kd) v.x = src[0];
        if (k0 + cpair + 1 < Kd) v.y = src[1];
      }
    }
    v.x *= scale;
    v.y *= scale;
    *reinterpret_cast<d2 *>(s + r * GLD + cpair) = v;
  }
}

// (A and C may alias: the triangular solve X = P L^-T runs in place, each
//  workgroup reads its whole A row-tile before it stores the same C tile.)
__global__ __launch_bounds__(256) void gemm_nt_kernel(
    const double *A, long lda, long strideA,
    const double *__restrict__ B, long ldb, long strideB, double *C,
    long ldc, long strideC, int Mrows, int Nrows, int Kd, double alpha,
    int beta, int lower_only, int batch, int ntm, int ntn, int ntiles) {
  __shared__ __attribute__((aligned(16))) double sA[GT * GLD];
  __shared__ __attribute__((aligned(16))) double sB[GT * GLD];

  // XCD-aware decode: blocks b and b+8 share an XCD
  const int b = blockIdx.x;
  const int xcd = b & 7, slot = b >> 3;
  const int mtx = (slot / ntiles) * 8 + xcd;
  if (mtx >= batch) return;
  const int tile = slot % ntiles;
  int ti, tj;
  if (lower_only) {
    ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    tj = tile - ti * (ti + 1) / 2;
  } else {
    ti = tile / ntn;
    tj = tile % ntn;
  }
  const int row0 = ti * GT, col0 = tj * GT;
  const double *Ab = A + (size_t)mtx * strideA;
  const double *Bb = B + (size_t)mtx * strideB;
  double *Cb = C + (size_t)mtx * strideC;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;

  d4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = d4{0.0, 0.0, 0.0, 0.0};
  if (beta) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = row0 + 16 * wave + fk + 4 * r, gj = col0 + 16 * n + fr;
        if (gi < Mrows && gj < Nrows) acc[n][r] = Cb[(size_t)gi * ldc + gj];
      }
  }

  const bool vecA = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(Ab) & 15) == 0);
  const bool vecB = ((ldb & 1) == 0) && ((reinterpret_cast<uintptr_t>(Bb) & 15) == 0);

  for (int k0 = 0; k0 < Kd; k0 += GK) {
    stage_panel(Ab, lda, row0, Mrows, k0, Kd, alpha, vecA, sA);
    stage_panel(Bb, ldb, col0, Nrows, k0, Kd, 1.0, vecB, sB);
    __syncthreads();
    const double *pa = sA + (16 * wave + fr) * GLD + fk;
    const double *pb = sB + fr * GLD + fk;
#pragma unroll
    for (int kk = 0; kk < GK; kk += 4) {
      const double a = pa[kk];
      const double b0 = pb[kk];
      const double b1 = pb[16 * GLD + kk];
      const double b2 = pb[32 * GLD + kk];
      const double b3 = pb[48 * GLD + kk];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[3], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = row0 + 16 * wave + fk + 4 * r, gj = col0 + 16 * n + fr;
      if (gi < Mrows && gj < Nrows) Cb[(size_t)gi * ldc + gj] = acc[n][r];
    }
}

}  // namespace

int sp_launch_gemm_nt(const double *A, long lda, long strideA, const double *B,
                      long ldb, long strideB, double *C, long ldc, long strideC,
                      int Mrows, int Nrows, int Kd, double alpha, int beta,
                      int lower_only, int batch, hipStream_t st) {
  if (Mrows <= 0 || Nrows <= 0 || batch <= 0) return SP_OK;
  if (Kd < 0) return SP_ERR_INVALID;
  const int ntm = (Mrows + GT - 1) / GT, ntn = (Nrows + GT - 1) / GT;
  if (lower_only && ntm != ntn) return SP_ERR_INVALID;
  const int ntiles = lower_only ? ntm * (ntm + 1) / 2 : ntm * ntn;
  const long nblk = 8L * ((batch + 7) / 8) * ntiles;
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)nblk), dim3(256), 0, st, A,
                     lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows,
                     Nrows, Kd, alpha, beta, lower_only, batch, ntm, ntn, ntiles);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
