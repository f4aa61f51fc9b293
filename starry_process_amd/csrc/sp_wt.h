// Wave-tile fp64 product for gfx950: every WAVEFRONT owns a 64 x 64 tile of C as 4 x 4
// accumulators of v_mfma_f64_16x16x4_f64 and feeds them straight from global memory (L2):
// no LDS, no workgroup barrier, no staging pass.
//
//     acc[64 x 64] += A[64 x Kd] . B[64 x Kd]^T          (row-major A, B; "NT")
//
// Why this shape on this chip: the fp64 MFMA holds a SIMD's matrix pipe for 64 cycles per
// 16 x 16 x 4 block, so a wavefront needs only 8 operand loads (16 B per lane each) per 32
// MFMAs = 2048 cycles -- a quarter of the L1's request rate with all four SIMDs busy, a third of
// an XCD's L2 bandwidth with all 32 CUs busy.  What costs time in an LDS-staged fp64 kernel is
// not bandwidth but the rendezvous: four wavefronts meeting at a barrier once per slice, each
// arriving with its own DMA and read latencies (measured: 0.63-0.70 of peak however deep the
// DMA pipeline, tools/mm_bench.py).  Here each wavefront is its own software pipeline: the loads
// of step s + 1 are in flight while step s is multiplied, the compiler's counted s_waitcnt
// retires them in order, and a stalled wavefront stalls nobody else.
//
// Fragment layout.  The MFMA sums over its 4 k-entries whatever their order, as long as A and
// B agree, so lane l (row r = l & 15, k-group q = l >> 4) loads the 16 bytes
// {k0 + 2 q, k0 + 2 q + 1} of its row: the .x halves of the four k-groups feed one MFMA
// (k = k0 + {0, 2, 4, 6}), the .y halves the next (k = k0 + {1, 3, 5, 7}).  A row is read as
// 64 contiguous bytes per load instruction.
//
// Requirements (launchers fall back otherwise): Kd a multiple of 16, lda / ldb even, 16-byte
// aligned bases, full 64 x 64 tiles.
#ifndef SP_WT_H
#define SP_WT_H

#include <hip/hip_runtime.h>

typedef double wt_d4 __attribute__((ext_vector_type(4)));
typedef double wt_d2 __attribute__((ext_vector_type(2)));

struct WtStage {
  wt_d2 a[4], b[4];
};

// A, B: first row of the 64-row panels, already offset to the first k of the product.
// DIAG: A and B are the same panel (diagonal tile of a symmetric update): one set of loads,
// and only the accumulators m >= n are formed.
template <bool DIAG>
struct WT {
  const double *pa[4], *pb[4];

  __device__ __forceinline__ void init(const double *A, long lda, const double *B, long ldb) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      pa[m] = A + (size_t)(16 * m + r) * lda + 2 * q;
      pb[m] = B + (size_t)(16 * m + r) * ldb + 2 * q;
    }
  }
  __device__ __forceinline__ void load(WtStage &s, int k) const {
#pragma unroll
    for (int m = 0; m < 4; ++m) s.a[m] = *reinterpret_cast<const wt_d2 *>(pa[m] + k);
    if (!DIAG) {
#pragma unroll
      for (int m = 0; m < 4; ++m) s.b[m] = *reinterpret_cast<const wt_d2 *>(pb[m] + k);
    }
  }
  __device__ __forceinline__ void mul(const WtStage &s, wt_d4 (&acc)[4][4]) const {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (DIAG && n > m) continue;
          const double av = h ? s.a[m].y : s.a[m].x;
          const double bv = DIAG ? (h ? s.a[n].y : s.a[n].x) : (h ? s.b[n].y : s.b[n].x);
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m][n], 0, 0, 0);
        }
  }
  // acc += A[:, 0 : kd] . B[:, 0 : kd]^T, kd a multiple of 16
  __device__ __forceinline__ void run(int kd, wt_d4 (&acc)[4][4]) const {
    if (kd <= 0) return;
    WtStage s0, s1;
    load(s0, 0);
    load(s1, 8);
    for (int k = 0; k < kd; k += 16) {
      mul(s0, acc);
      if (k + 16 < kd) load(s0, k + 16);
      mul(s1, acc);
      if (k + 24 < kd) load(s1, k + 24);
    }
  }
};

// accumulator element (m, n)[r] of a wavefront's tile is C[16 m + (lane >> 4) + 4 r][16 n + (lane & 15)]
__device__ __forceinline__ int wt_row(int m, int r) { return 16 * m + ((threadIdx.x & 63) >> 4) + 4 * r; }
__device__ __forceinline__ int wt_col(int n) { return 16 * n + (threadIdx.x & 15); }

#endif
