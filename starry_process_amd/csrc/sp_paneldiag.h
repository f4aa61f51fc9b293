// The "D item" of the blocked Cholesky factorisation (math.py:75-91): ONE workgroup
// factors the 64 x 64 pivot block j of a star and leaves what the panel kernel's solves need.
//
//   tile (j, j), complete (the panel kernels keep the coming diagonal tiles up to date eagerly)
//     -> LDS, identity-padded beyond the `nact` active columns
//     -> diag_block (sp_diag.h): L_d in the lower triangle, L_d^-1 formed in the shadow of the
//        factorisation in the upper one
//     -> L_d back to the system; L_d^-1 to the star's image slot IN THE FRAGMENT ORDER of the
//        solve X = T L_d^-T on the matrix cores (sp_panel.hip): ten 16 x 16 blocks (nb >= kb), each
//        256 doubles in two planes of 128 (k steps s = 0, 1 and s = 2, 3), lane-major
//            img[blk * 256 + (s >> 1) * 128 + 2 lane + (s & 1)] = Linv[16 nb + PI(lane & 15)][16 kb + 4 (lane >> 4) + s]
//        so that a lane fetches a fragment with two 16-byte reads, consecutive lanes 16 bytes apart
//        (conflict-free from the LDS copy the panel kernel's workgroups make of it).  PI(i) = 4 (i mod 4) + i div 4: the row permutation that makes an MFMA
//        accumulator hold four CONSECUTIVE columns per lane (see sp_panel.hip).
//     -> partial last block: the rows of the tile below the active ones (residual rows, padding)
//        carry every update already and are solved here, X = A21 L11^-T.
//
// Called by the panel kernel (its D role, sp_panel.hip) and by the workgroup of tile (0, 0) of the
// symmetric trailing update (sp_gemm.hip), which would otherwise have nothing to do: the block is
// factored beside the products of the launch, never behind them.
#ifndef SP_PANELDIAG_H
#define SP_PANELDIAG_H

#include "sp_tile.h"

#define SP_IMG_DOUBLES 2560   // ten fragment-ordered 16 x 16 blocks

// block index of (nb, kb), nb >= kb: rows of blocks in DESCENDING nb (the order the solve walks them)
__device__ __forceinline__ constexpr int sp_img_block(int nb, int kb) {
  return (nb == 3 ? 0 : nb == 2 ? 4 : nb == 1 ? 7 : 9) + kb;
}
__device__ __forceinline__ constexpr int sp_pi16(int i) { return 4 * (i & 3) + (i >> 2); }

// what a launch needs to factor a pivot block beside its own work (mm_nt_kernel, sp_gemm.hip)
struct DiagFuse {
  double *sys;       // systems (null: nothing to factor)
  long ld, stride;
  int j, nact;       // pivot block, its active columns
  double *img;       // per-star scratch (lts doubles apart); slot j mod 3 receives the image (sp_tile.h)
  long lts;
  int32_t *info;
  int tri0 = -1;     // >= 0: an identity rides along from this row on (sp_spd_inverse_batched): row tile j + ti of the
  int s0 = 0;        // update holds nothing left of column block (64 (j + ti) - tri0) / 64; s0: the product's first block
};

// lds: SP_DIAG_LDS_DOUBLES doubles, free for the whole call.  All 256 threads.
//
// panel_diag_load: tile (j, j) from memory into sD, identity-padded; panel_diag_core: everything
// after that (a caller that holds the tile in registers fills sD itself, sp_panel.hip).
__device__ __forceinline__ void panel_diag_load(const double *D, long ld, int nact, double *sD, int tid) {
  const int c = (tid & 15) * 4, r0 = tid >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = r0 + 16 * pass;
    const d2v a = *reinterpret_cast<const d2v *>(D + (size_t)r * ld + c);
    const d2v b = *reinterpret_cast<const d2v *>(D + (size_t)r * ld + c + 2);
    double v[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cc = c + e;
      double x = (r < nact && cc < nact) ? v[e] : (r == cc ? 1.0 : 0.0);
      if (cc > r) x = 0.0;
      v[e] = x;
    }
    *reinterpret_cast<d2v *>(sD + r * BLD + c) = d2v{v[0], v[1]};
    *reinterpret_cast<d2v *>(sD + r * BLD + c + 2) = d2v{v[2], v[3]};
  }
}

// sD holds the padded block (a barrier is taken here before it is read)
__device__ __forceinline__ void panel_diag_core(double *D, long ld, int nact, double *__restrict__ img,
                                                int32_t *info_star, double *lds, int tid,
                                                long long *dbg = nullptr, int nlive = 64) {
  double *sD = lds, *sRd = lds + 64 * BLD;
  __syncthreads();
  if (dbg && tid == 0) dbg[0] = wall_clock64();
  const int notpd = diag_block(sD, sRd, tid);
  if (dbg && tid == 0) dbg[1] = wall_clock64();
  if (notpd && info_star) *info_star = 1;
  // (diag_block ends on a barrier behind the last block row of the inverse: sD / sRd are final)
  // the image, fragment order
  for (int e = tid; e < 640; e += 256) {
    const int blk = e >> 6, lane = e & 63, fr = lane & 15, fk = lane >> 4;
    const int nb = blk < 4 ? 3 : (blk < 7 ? 2 : (blk < 9 ? 1 : 0));
    const int kb = blk - (nb == 3 ? 0 : nb == 2 ? 4 : nb == 1 ? 7 : 9);
    const int n = 16 * nb + sp_pi16(fr), k0 = 16 * kb + 4 * fk;
    double v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = k0 + s;
      v[s] = k < n ? sD[k * BLD + n] : (k == n ? sRd[k] : 0.0);   // Linv[n][k]
    }
    double *dst = img + (size_t)blk * 256 + 2 * lane;
    *reinterpret_cast<d2v *>(dst) = d2v{v[0], v[1]};
    *reinterpret_cast<d2v *>(dst + 128) = d2v{v[2], v[3]};
  }
  // L_d back to the system: the active rows, whole 64-column rows (zero above the diagonal)
  {
    const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      if (r < nact) {
        double *dst = D + (size_t)r * ld + cj;
        const d2v a = *reinterpret_cast<const d2v *>(sD + r * BLD + cj);
        const d2v b = *reinterpret_cast<const d2v *>(sD + r * BLD + cj + 2);
        *reinterpret_cast<d2v *>(dst) = d2v{cj <= r ? a.x : 0.0, cj + 1 <= r ? a.y : 0.0};
        *reinterpret_cast<d2v *>(dst + 2) = d2v{cj + 2 <= r ? b.x : 0.0, cj + 3 <= r ? b.y : 0.0};
      }
    }
  }
  if (dbg && tid == 0) dbg[2] = wall_clock64();
  if (nact < 64) {
    // rows nact .. 63 of the tile against the block just factored: X[r][n] = sum_{k <= n} A[r][k] Linv[n][k]
    // (columns >= nact: identity padding, they stay).  A21 takes the place of the padding rows of sD.
    // (nlive: rows of the tile that carry data -- the rest is identity padding and solves to itself)
    const int nrow = (nlive < 64 ? (nlive > nact ? nlive : nact) : 64) - nact;
    __syncthreads();
    for (int e = tid; e < nrow * 64; e += 256) {
      const int r = nact + (e >> 6), c = e & 63;
      if (c < nact) sD[r * BLD + c] = D[(size_t)r * ld + c];
    }
    __syncthreads();
    for (int e = tid; e < nrow * 64; e += 256) {
      const int r = nact + (e >> 6), n = e & 63;
      if (n >= nact) continue;
      double acc = sD[r * BLD + n] * sRd[n];
      for (int k = 0; k < n; ++k) acc = fma(sD[r * BLD + k], sD[k * BLD + n], acc);
      D[(size_t)r * ld + n] = acc;
    }
  }
}

__device__ __forceinline__ void panel_diag_item(double *M, long ld, int j, int nact,
                                                double *__restrict__ img, int32_t *info_star,
                                                double *lds, int tid, long long *dbg = nullptr) {
  double *D = M + (size_t)(64 * j) * ld + 64 * j;
  panel_diag_load(D, ld, nact, lds, tid);
  panel_diag_core(D, ld, nact, img, info_star, lds, tid, dbg);
}

#endif
