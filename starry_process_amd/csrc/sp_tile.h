// Workgroup-level building blocks of the blocked Cholesky (one 256-thread
// workgroup = 4 wavefronts, 64 x 64 tiles) shared by sp_gemm.hip and
// sp_cholesky.hip:
//   tile_mac         acc -= A B^T for one full 64 x 64 tile, operands staged
//                    through LDS in 32-deep slices (the inner loop of
//                    gemm_nt_kernel without bounds checks);
//   tile_trsm_store  X = T L_d^-T for a tile held in accumulators, by the
//                    four-lanes-per-row substitution of trsm_quad_kernel;
//   superpanel_factor  the whole 64 w x 64 w diagonal block of a super-panel
//                    (w <= 4 panels) by ONE workgroup: per panel the diagonal
//                    tile update, diag_block, and the solves of the tiles below
//                    it inside the block.
// All of them use one LDS region of SP_TILE_LDS_DOUBLES doubles, one after the
// other (operand slices / tile image / L^T image).
#ifndef SP_TILE_H
#define SP_TILE_H

#include "sp_diag.h"

#define TBK 32                 // depth of one LDS operand slice
#define TLDW (TBK + 1)         // its padded row (odd: conflict-free ds_read2_b64 fragments)
#define SP_TILE_LDS_DOUBLES (SP_DIAG_LDS_DOUBLES > 2 * 64 * TLDW ? SP_DIAG_LDS_DOUBLES : 2 * 64 * TLDW)

// per-star scratch of the factorisation: up to 4 L_d^T images + one counter line
#define SP_LT_IMG 4096
#define SP_LT_STRIDE (4 * SP_LT_IMG + 8)

struct TileRegs {
  d2v v[4];
};

// 64 rows x 32 columns starting at P (row stride ld), 16 lanes x 16 B per row
__device__ __forceinline__ void tile_stage_load(const double *P, long ld, int k0, TileRegs &R) {
  const int t = threadIdx.x;
  const double *src = P + (size_t)(t >> 4) * ld + k0 + (t & 15) * 2;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass)
    R.v[pass] = *reinterpret_cast<const d2v *>(src + (size_t)(16 * pass) * ld);
}
__device__ __forceinline__ void tile_stage_store(const TileRegs &R, double *s) {
  const int t = threadIdx.x;
  double *dst = s + (t >> 4) * TLDW + (t & 15) * 2;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    dst[16 * pass * TLDW] = R.v[pass].x;
    dst[16 * pass * TLDW + 1] = R.v[pass].y;
  }
}

// acc (wave w: rows 16 w .., four 16 x 16 column blocks) -= A[64 x Kd] . B[64 x Kd]^T
// A, B: row-major with strides lda / ldb, 16-byte aligned rows, Kd a multiple of 32.
// Ends with a barrier (the LDS region is free again).
__device__ __forceinline__ void tile_mac(d4 (&acc)[4], const double *A, long lda,
                                         const double *B, long ldb, int Kd, double *smem) {
  double *sA = smem, *sB = smem + 64 * TLDW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  TileRegs ra, rb;
  tile_stage_load(A, lda, 0, ra);
  tile_stage_load(B, ldb, 0, rb);
  for (int k0 = 0; k0 < Kd; k0 += TBK) {
    tile_stage_store(ra, sA);
    tile_stage_store(rb, sB);
    __syncthreads();
    if (k0 + TBK < Kd) {
      tile_stage_load(A, lda, k0 + TBK, ra);
      tile_stage_load(B, ldb, k0 + TBK, rb);
    }
    const double *pa = sA + (16 * wave + fr) * TLDW + fk;
    const double *pb = sB + fr * TLDW + fk;
#pragma unroll
    for (int kk = 0; kk < TBK; kk += 4) {
      const double a = -pa[kk];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[kk], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[16 * TLDW + kk], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[32 * TLDW + kk], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[48 * TLDW + kk], acc[3], 0, 0, 0);
    }
    __syncthreads();
  }
}

// accumulators <-> a full 64 x 64 tile in global memory (row stride ld)
__device__ __forceinline__ void tile_load(d4 (&acc)[4], const double *C, long ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[n][r] = C[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * n + fr];
}

// ---- substitution (shared with trsm_quad_kernel) ------------------------------
struct TrsmRow {
  d2v v[8];
};

template <int K>
__device__ __forceinline__ void trsm_fetch(TrsmRow &r, const double *sLT, int q) {
  if (K < 64) {
    const double *row = sLT + (K < 64 ? K : 0) * 64 + 2 * q;
#pragma unroll
    for (int i = (K >> 3); i < 8; ++i) r.v[i] = *reinterpret_cast<const d2v *>(row + 8 * i);
  }
}

template <int K>
struct TrsmStep {
  static __device__ __forceinline__ void run(double (&x)[16], const TrsmRow &cur,
                                             const TrsmRow &nxt, const double *sLT, int q) {
    constexpr int IK = K >> 3, QK = (K >> 1) & 3, REG = 2 * IK + (K & 1);
    constexpr int CTRL = QK * 0x55;  // quad_perm:[QK, QK, QK, QK]
    TrsmRow nn;
    trsm_fetch<K + 2>(nn, sLT, q);
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x[REG]), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x[REG]), CTRL, 0xf, 0xf, false);
    const double xk = __hiloint2double(hi, lo);
    // (cur.v holds 0 for the columns <= K, so x[REG] itself is left alone)
#pragma unroll
    for (int i = IK; i < 8; ++i) {
      x[2 * i] = fma(-xk, cur.v[i].x, x[2 * i]);
      x[2 * i + 1] = fma(-xk, cur.v[i].y, x[2 * i + 1]);
    }
    TrsmStep<K + 1>::run(x, nxt, nn, sLT, q);
  }
};
template <>
struct TrsmStep<64> {
  static __device__ __forceinline__ void run(double (&)[16], const TrsmRow &, const TrsmRow &,
                                             const double *, int) {}
};

// the L_d^T image (diag_block's `lt`) from global memory into registers / LDS
struct LtRegs {
  d2v v[8];
};
__device__ __forceinline__ void lt_load(LtRegs &R, const double *LT) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
    R.v[i] = *reinterpret_cast<const d2v *>(LT + 2 * (threadIdx.x + 256 * i));
}
// sLT[64 * 64] gets the image with a zero diagonal, sRd[64] the diagonal (1 / L_kk)
__device__ __forceinline__ void lt_store(const LtRegs &R, double *sLT, double *sRd) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = 2 * (threadIdx.x + 256 * i);
    const int k = e >> 6, c = e & 63;
    d2v v = R.v[i];
    const bool d0 = c == k, d1 = c + 1 == k;
    if (d0 || d1) sRd[k] = d0 ? v.x : v.y;
    v.x = d0 ? 0.0 : v.x;
    v.y = d1 ? 0.0 : v.y;
    *reinterpret_cast<d2v *>(sLT + e) = v;
  }
}

// rows of a 64-row tile, quad layout: thread t holds row t >> 2, columns 8 i + 2 (t & 3) + {0, 1}
__device__ __forceinline__ void quad_solve_store(double (&x)[16], const double *sLT,
                                                 const double *sRd, double *Xrow, bool valid) {
  const int q = threadIdx.x & 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const d2v rd = *reinterpret_cast<const d2v *>(sRd + 8 * i + 2 * q);
    x[2 * i] *= rd.x;
    x[2 * i + 1] *= rd.y;
  }
  TrsmRow r0, r1;
  trsm_fetch<0>(r0, sLT, q);
  trsm_fetch<1>(r1, sLT, q);
  TrsmStep<0>::run(x, r0, r1, sLT, q);
  if (valid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      d2v v;
      v.x = x[2 * i];
      v.y = x[2 * i + 1];
      *reinterpret_cast<d2v *>(Xrow + 8 * i) = v;
    }
  }
}

// X = T L_d^-T for the tile in `acc`; X goes to the 64 x 64 tile at Xg (row stride
// ld).  `lt` = the L_d^T image already in registers (load it before the product that
// fills acc to hide its latency).  Uses smem[0 .. 64*66) + [4096 .. 4160).
// Begins with the region free, ends with a barrier (region free, X visible to the
// workgroup).
__device__ __forceinline__ void tile_trsm_store(const d4 (&acc)[4], const LtRegs &lt,
                                                double *Xg, long ld, double *smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  // accumulator layout -> row-major image -> quad layout
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) smem[(16 * wave + fk + 4 * r) * BLD + 16 * n + fr] = acc[n][r];
  __syncthreads();
  double x[16];
  {
    const double *row = smem + (tid >> 2) * BLD + 2 * (tid & 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const d2v v = *reinterpret_cast<const d2v *>(row + 8 * i);
      x[2 * i] = v.x;
      x[2 * i + 1] = v.y;
    }
  }
  __syncthreads();
  lt_store(lt, smem, smem + 4096);
  __syncthreads();
  quad_solve_store(x, smem, smem + 4096, Xg + (size_t)(tid >> 2) * ld + 2 * (tid & 3), true);
  __syncthreads();
}

// The 64 wp x 64 wp diagonal block of a super-panel (first column cS, wp <= 4
// panels, K = order of the matrix proper) of ONE system, by one workgroup.
// On entry the block has received every update from the columns left of cS.
// Writes L in place, the L_d^T images to LT[q * SP_LT_IMG], sets *info on a
// non-positive pivot.  Left-looking inside the block: a tile is brought up to
// date by the panels of this super-panel just before it is factored / solved.
__device__ __forceinline__ void superpanel_factor(double *Mx, long ld, int cS, int wp, int K,
                                                  double *LT, int32_t *info, double *smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  int notpd = 0;
  for (int q = 0; q < wp; ++q) {
    const int c0 = cS + 64 * q;
    const int nact = K - c0 < 64 ? K - c0 : 64;
    double *Tqq = Mx + (size_t)c0 * ld + c0;
    d4 acc[4];
    tile_load(acc, Tqq, ld);
    if (q > 0) tile_mac(acc, Mx + (size_t)c0 * ld + cS, ld, Mx + (size_t)c0 * ld + cS, ld, 64 * q, smem);
    if (nact < 64) {
      // rows >= nact (right-hand sides / padding) keep their updated values in
      // global memory for the solve below
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int li = 16 * wave + fk + 4 * r;
          if (li >= nact) Tqq[(size_t)li * ld + 16 * n + fr] = acc[n][r];
        }
    }
    // diagonal tile: identity outside the active part, zero strict upper part
    double *sD = smem, *sRd = smem + 64 * BLD;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int li = 16 * wave + fk + 4 * r, lj = 16 * n + fr;
        double v = (li < nact && lj < nact) ? acc[n][r] : (li == lj ? 1.0 : 0.0);
        if (lj > li) v = 0.0;
        sD[li * BLD + lj] = v;
      }
    __syncthreads();
    double *LTq = LT + (size_t)q * SP_LT_IMG;
    notpd |= diag_block(sD, sRd, LTq);
    {
      const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = ri + 16 * pass;
        double *dst = Tqq + (size_t)r * ld + cj;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cj + e <= r && r < nact) dst[e] = sD[r * BLD + cj + e];
      }
    }
    __syncthreads();   // L_d^T image visible to the workgroup, LDS region free
    const bool tail = nact < 64;        // rows nact..63 of this tile still need the solve
    if (q + 1 < wp || tail) {
      LtRegs lt;
      lt_load(lt, LTq);
      if (tail) {
        double x[16];
        const int lrow = nact + (tid >> 2);
        const bool valid = lrow < 64;
        double *prow = Tqq + (size_t)(valid ? lrow : nact) * ld + 2 * (tid & 3);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const d2v v = *reinterpret_cast<const d2v *>(prow + 8 * i);
          x[2 * i] = v.x;
          x[2 * i + 1] = v.y;
        }
        lt_store(lt, smem, smem + 4096);
        __syncthreads();
        quad_solve_store(x, smem, smem + 4096, prow, valid);
        __syncthreads();
      }
      for (int t = q + 1; t < wp; ++t) {
        double *Ttq = Mx + (size_t)(cS + 64 * t) * ld + c0;
        tile_load(acc, Ttq, ld);
        if (q > 0)
          tile_mac(acc, Mx + (size_t)(cS + 64 * t) * ld + cS, ld, Mx + (size_t)c0 * ld + cS, ld,
                   64 * q, smem);
        tile_trsm_store(acc, lt, Ttq, ld, smem);
      }
    }
  }
  if (notpd && info) *info = 1;
}

#endif
