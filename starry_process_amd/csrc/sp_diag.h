// Device routine shared by sp_cholesky.hip (stand-alone diagonal-block kernel)
// and sp_gemm.hip (fused into the tile-(0,0) workgroup of a block-column update).
//
// diag_block(): Cholesky factor L of a 64 x 64 SPD block held in LDS, by one
// 256-thread workgroup, and the operands of the panel solve that follows
// (trsm_quad_kernel, sp_cholesky.hip): L^T and the reciprocal diagonal.
//
// The block is cut in four block columns of 16; wavefront w owns block column w.
//   factor_panel(w): first the 16 x 16 diagonal leaf with lane = row (the four
//     16-lane DPP rows carry the same copy): column c is scaled by 1 / sqrt(pivot)
//     in every lane at once and the rank-1 step a_j -= l * l_j takes l_j from lane
//     j by a DPP row broadcast (v_mov_b64_dpp row_newbcast) -- no LDS traffic, no
//     scalar registers, no barrier inside the leaf.  Then the rows below the leaf,
//     one per lane, by substitution against the leaf.  fp64 VALU and fp64 MFMA
//     have the same peak on gfx950, so nothing is lost by leaving the matrix
//     cores here; the leaf is a latency chain of ~130 cycles per column.
//   update: block column w is brought up to date by wavefront w itself as soon
//     as a panel k < w is published (left-looking, on the MFMA from LDS), so the
//     owner of the next panel starts factoring while the others still update.
//   One workgroup barrier per panel.
// Critical path ~ 4 x (16 columns x ~130 cycles + ~1 us of update / LDS turn).
//
// LDS: sD[64 * BLD] (block in, L out in the lower part) + sRd[64] (1 / L_cc)
// + 16 x 16 scaled leaf for the substitution.
// `lt` (global, 64 x 64 row-major) receives  lt[k][c] = L[c][k] / L[c][c] for
// c > k, 1 / L[k][k] for c == k, 0 for c < k.
#ifndef SP_DIAG_H
#define SP_DIAG_H

#include <hip/hip_runtime.h>

#define BLD 66   // LDS row of the 64x64 block: even (16-B aligned rows), 132 dwords = 4 mod 64 banks
#define SP_DIAG_LDS_DOUBLES (64 * BLD + 64 + 256)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double read_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to ~1 ulp: hardware seed (v_rsq_f64) + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
  double e = fma(-p * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-p * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  return r;
}

// a-operand / NT b-operand fragment: M[row0 + (lane & 15)][col0 + 4 s + (lane >> 4)]
__device__ __forceinline__ double frag_rowmajor(const double *M, int ldm, int row0, int col0,
                                                int s, int lane) {
  return M[(row0 + (lane & 15)) * ldm + col0 + 4 * s + (lane >> 4)];
}
// accumulator <-> memory, C/D map of the fp64 MFMA: col = lane & 15, row = (lane >> 4) + 4 reg
__device__ __forceinline__ d4 acc_load(const double *M, int ldm, int row0, int col0, int lane) {
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = M[(row0 + (lane >> 4) + 4 * r) * ldm + col0 + (lane & 15)];
  return v;
}
__device__ __forceinline__ void acc_store(double *M, int ldm, int row0, int col0, int lane, d4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) M[(row0 + (lane >> 4) + 4 * r) * ldm + col0 + (lane & 15)] = v[r];
}

// lane j of each 16-lane DPP row, broadcast to the whole row (DPP64 row_newbcast)
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, true);
}

// rank-1 step of leaf column C on the columns J..15 of every row
template <int C, int J>
struct LeafUpd {
  static __device__ __forceinline__ void run(double (&a)[16], double l) {
    a[J] = fma(-l, row_bcast<J>(l), a[J]);
    LeafUpd<C, J + 1>::run(a, l);
  }
};
template <int C>
struct LeafUpd<C, 16> {
  static __device__ __forceinline__ void run(double (&)[16], double) {}
};

// columns C..15 of the 16 x 16 leaf; lane (row of 16) i holds row i, all four
// DPP rows of the wavefront carry the same copy.  r = 1 / sqrt(pivot C).
template <int C>
struct LeafCol {
  static __device__ __forceinline__ void run(double (&a)[16], double r, double &rd_mine,
                                             int &notpd, int i) {
    const double l = a[C] * r;   // column C of L (rows < C: unused)
    a[C] = l;
    rd_mine = (i == C) ? r : rd_mine;
    // the next pivot first, so that its rsqrt overlaps the rest of this column
    a[C + 1] = fma(-l, row_bcast<C + 1>(l), a[C + 1]);
    const double p = row_bcast<C + 1>(a[C + 1]);
    if (!(p > 0.0)) notpd = 1;
    const double rn = rsqrt_nr(p);
    LeafUpd<C, C + 2>::run(a, l);
    LeafCol<C + 1>::run(a, rn, rd_mine, notpd, i);
  }
};
template <>
struct LeafCol<15> {
  static __device__ __forceinline__ void run(double (&a)[16], double r, double &rd_mine, int &,
                                             int i) {
    a[15] = a[15] * r;
    rd_mine = (i == 15) ? r : rd_mine;
  }
};

// Block column `kb` (columns o = 16 kb .. o + 15, rows o .. 63) by ONE wavefront.
// Stage 1: the 16 x 16 diagonal leaf.  Lane i (mod 16) holds row i in 16
//   registers; a column's entries reach the other rows through DPP row
//   broadcasts (no LDS, no scalar registers, no barrier).
// Stage 2: the rows below the leaf, one per lane, by substitution against the
//   leaf (pre-scaled:  lt[k][c] = L_ck / L_cc  read from LDS at a uniform address).
// Returns 1 if a pivot was not positive.
__device__ __forceinline__ int factor_panel(double *sD, double *sRd, double *sLt, int kb,
                                            int lane, long long *ts = nullptr) {
  const int o = 16 * kb, i = lane & 15;
  int notpd = 0;
  if (ts) ts[0] = clock64();
  {
    double *prow = sD + (o + i) * BLD + o;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
      const d2v v = *reinterpret_cast<const d2v *>(prow + j);
      a[j] = v.x;
      a[j + 1] = v.y;
    }
    const double p = row_bcast<0>(a[0]);
    if (!(p > 0.0)) notpd = 1;
    double rd_mine = 0.0;
    if (ts) ts[1] = clock64();
    LeafCol<0>::run(a, rsqrt_nr(p), rd_mine, notpd, i);
    if (ts) ts[2] = clock64();
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        d2v v;
        v.x = j > i ? 0.0 : a[j];          // strict upper part of the leaf
        v.y = j + 1 > i ? 0.0 : a[j + 1];
        *reinterpret_cast<d2v *>(prow + j) = v;
      }
      sRd[o + i] = rd_mine;
#pragma unroll
      for (int k = 0; k < 16; ++k) sLt[k * 16 + i] = k < i ? a[k] * rd_mine : 0.0;
    }
  }
  if (ts) ts[3] = clock64();
  const int nbelow = 48 - o;
  if (nbelow > 0) {
    double *prow = sD + (o + 16 + (lane < nbelow ? lane : nbelow - 1)) * BLD + o;
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
      const d2v v = *reinterpret_cast<const d2v *>(prow + j);
      const d2v rd = *reinterpret_cast<const d2v *>(sRd + o + j);
      x[j] = v.x * rd.x;
      x[j + 1] = v.y * rd.y;
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) {
#pragma unroll
      for (int c = (k + 1) & ~1; c < 16; c += 2) {
        const d2v lt = *reinterpret_cast<const d2v *>(sLt + k * 16 + c);
        x[c] = fma(-x[k], lt.x, x[c]);
        x[c + 1] = fma(-x[k], lt.y, x[c + 1]);
      }
    }
    if (lane < nbelow) {
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        d2v v;
        v.x = x[j];
        v.y = x[j + 1];
        *reinterpret_cast<d2v *>(prow + j) = v;
      }
    }
  }
  if (ts) ts[4] = clock64();
  return notpd;
}

// All 256 threads of the workgroup call this with the block already in sD
// (lower triangle valid, upper part mirrored or zero; see callers for the
// identity padding of a partial block) and a barrier behind the stores.
// Returns 1 in every thread of wavefronts that saw a non-positive pivot
// (callers OR it through global memory).
__device__ __forceinline__ int diag_block(double *sD, double *sRd, double *__restrict__ lt,
                                          long long *dbg = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int notpd = 0;
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    if (wave == kb) {
      long long ts[5];
      notpd |= factor_panel(sD, sRd, sRd + 64, kb, lane, dbg ? ts : nullptr);
      if (dbg && lane == 0)
        for (int q = 0; q < 5; ++q) dbg[8 * kb + q] = ts[q];
    }
    __syncthreads();
    if (dbg && wave == (kb < 3 ? kb + 1 : 3) && lane == 0) dbg[8 * kb + 5] = clock64();
    if (wave > kb) {
      // A_{ib,w} -= L_{ib,kb} L_{w,kb}^T for the row blocks ib >= w of my block column
      const int o = 16 * kb, w = wave;
      double bfrag[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bfrag[s] = frag_rowmajor(sD, BLD, 16 * w, o, s, lane);
      for (int ib = w; ib < 4; ++ib) {
        d4 acc = acc_load(sD, BLD, 16 * ib, 16 * w, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag_rowmajor(sD, BLD, 16 * ib, o, s, lane),
                                                     bfrag[s], acc, 0, 0, 0);
        acc_store(sD, BLD, 16 * ib, 16 * w, lane, acc);
      }
    }
    if (dbg && wave == (kb < 3 ? kb + 1 : 3) && lane == 0) dbg[8 * kb + 6] = clock64();
  }
  // operands of the panel solve: L^T with the reciprocal diagonal
  for (int e = tid; e < 4096; e += 256) {
    const int k = e >> 6, c = e & 63;
    lt[e] = c > k ? sD[c * BLD + k] * sRd[c] : (c == k ? sRd[k] : 0.0);
  }
  return notpd;
}

#endif
