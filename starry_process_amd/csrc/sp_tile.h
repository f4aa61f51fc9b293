// Forward substitution against a 64 x 64 diagonal block with four lanes per row
// (used by trsm_quad_kernel, sp_cholesky.hip): the L_d^T image written by
// diag_block (sp_diag.h) is staged in LDS; lane q of a quad holds the columns
// 8 i + 2 q + {0, 1}; step k broadcasts x_k inside the quad (DPP quad_perm) and
// updates the remaining columns.
#ifndef SP_TILE_H
#define SP_TILE_H

#include "sp_diag.h"

// XCD-aware work decode shared by the tile kernels.  Work items are (matrix, tile) pairs,
// matrix-major.  The hardware hands consecutive workgroups to the 8 XCDs round-robin, so
// XCD x = blockIdx.x % 8 is given the CONTIGUOUS items [x c, (x + 1) c), c = ceil(items / 8):
// whole matrices when the batch is a multiple of 8 (every tile of a star re-reads that star's
// row panels from one 4 MiB L2), runs of neighbouring tiles of one matrix when it is not -- a
// single large matrix is spread over all 8 XCDs instead of one.  Launch 8 c workgroups.
__device__ __forceinline__ bool sp_xcd_decode(int b, int batch, int ntiles, int &mtx, int &tile) {
  const int items = batch * ntiles;
  const int c = (items + 7) >> 3;
  const int slot = b >> 3;
  const int item = (b & 7) * c + slot;
  if (slot >= c || item >= items) return false;
  mtx = item / ntiles;
  tile = item - mtx * ntiles;
  return true;
}
static inline long sp_xcd_grid(long batch, long ntiles) { return 8L * ((batch * ntiles + 7) / 8); }


#define SP_TILE_LDS_DOUBLES SP_DIAG_LDS_DOUBLES

// per-star scratch of the factorisation: L_d^T images of 64 x 64 doubles, `lts` doubles apart
// from star to star (sp_lt_stride, sp_internal.h).  The one-launch-per-panel mode ping-pongs
// between the first two (it writes the next panel's image while the workgroups of the current
// launch still read this one); the recursive driver keeps the image of every block.
#define SP_LT_IMG 4096

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
// (tid: callers inside long loops pass a laundered copy of threadIdx.x so that the index
//  arithmetic is redone per use instead of being hoisted and kept in registers)
__device__ __forceinline__ void lt_load(LtRegs &R, const double *LT, int tid = threadIdx.x) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // row k of the image is zero left of the diagonal (lt[k][c] = L_ck / L_cc, c >= k): those
    // pairs are not fetched -- the image is a third of a solve workgroup's traffic
    const int e = 2 * (tid + 256 * i), k = e >> 6, c = e & 63;
    R.v[i] = (c + 1 >= k) ? *reinterpret_cast<const d2v *>(LT + e) : d2v{0.0, 0.0};
  }
}
// sLT[64 * 64] gets the image with a zero diagonal, sRd[64] the diagonal (1 / L_kk)
__device__ __forceinline__ void lt_store(const LtRegs &R, double *sLT, double *sRd, int tid = threadIdx.x) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = 2 * (tid + 256 * i);
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
                                                 const double *sRd, double *Xrow, bool valid,
                                                 int tid = threadIdx.x) {
  const int q = tid & 3;
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

// ---- operand of the panel solve as a BLOCK substitution on the matrix cores (SP_PANEL_MFMA_SOLVE 2) ----
// W (64 x 64, row-major): the 16 x 16 blocks of L_d below the diagonal as they are, and in place
// of each diagonal block L_cc its inverse M_c.  A solve X = T L_d^-T is then, block column by
// block column,  X_c = (T_c - sum_{k<c} X_k L_ck^T) M_c^T : 40 MFMAs per wavefront, no inverse of
// the whole block to form -- only the four 16 x 16 leaves are inverted, each by one wavefront
// (lane = row, four columns per 16-lane group, x_k handed round by DPP row broadcasts).
// after diag_block: sD holds L (rows of BLD doubles, zero above the diagonal), sRd 1 / L_cc.
// All 256 threads; reads LDS only, writes W.
__device__ __forceinline__ void diag_solve_operand(const double *sD, const double *sRd,
                                                   double *__restrict__ W, int tid = threadIdx.x) {
  for (int e = tid; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    if ((r >> 4) > (c >> 4)) W[e] = sD[r * BLD + c];
  }
  const int wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4, o = 16 * wave;
  double Lrow[16];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const d2v v = *reinterpret_cast<const d2v *>(sD + (o + i) * BLD + o + k);
    Lrow[k] = v.x;
    Lrow[k + 1] = v.y;
  }
  const double rd = sRd[o + i];
  double s[4], res[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    s[t] = (i == 4 * t + g) ? 1.0 : 0.0;
    res[t] = 0.0;
  }
  LeafInvStep<0>::run(s, res, Lrow, rd, i);
  double *dst = W + (size_t)(o + i) * 64 + o + g;
#pragma unroll
  for (int t = 0; t < 4; ++t) dst[4 * t] = res[t];
}

#endif
