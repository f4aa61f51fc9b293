// Device routine shared by sp_cholesky.hip (stand-alone diagonal-block kernel)
// and sp_gemm.hip (fused into the tile-(0,0) workgroup of a block-column update).
//
// diag_block(): Cholesky factor L of a 64 x 64 SPD block held in LDS, by one
// 256-thread workgroup, and the operands of the panel solve that follows
// (trsm_quad_kernel, sp_cholesky.hip): L^T and the reciprocal diagonal.
//
// The block is cut in four block columns of 16; wavefront w owns block column w.
//   factor_panel(w): the 16 columns x (64 - 16 w) rows of block column w in a
//     ROW-PER-LANE layout (lane = row, 16 registers = the row's entries).
//     Column c: pivot from lane c (v_readlane), l = a_c / sqrt(pivot) in every
//     lane at once, then a_j -= l * l_j with l_j = v_readlane(l, j) as a scalar
//     operand.  No LDS traffic, no barrier inside the 16 columns, and the rows
//     below the 16 x 16 diagonal leaf are factored in the same sweep (no leaf
//     inverse, no separate triangular solve).  fp64 VALU and fp64 MFMA have the
//     same peak on gfx950, so nothing is lost by leaving the matrix cores here.
//   update: block column w is brought up to date by wavefront w itself as soon
//     as a panel k < w is published (left-looking, on the MFMA from LDS), so the
//     owner of the next panel starts factoring while the others still update.
//   One workgroup barrier per panel.
// Critical path ~ 4 x (16 columns x ~130 cycles + ~1 us of update / LDS turn).
//
// LDS: sD[64 * BLD] (block in, L out in the lower part) + sRd[64] (1 / L_cc).
// `lt` (global, 64 x 64 row-major) receives  lt[k][c] = L[c][k] / L[c][c] for
// c > k, 1 / L[k][k] for c == k, 0 for c < k.
#ifndef SP_DIAG_H
#define SP_DIAG_H

#include <hip/hip_runtime.h>

#define BLD 66   // LDS row of the 64x64 block: even (16-B aligned rows), 132 dwords = 4 mod 64 banks
#define SP_DIAG_LDS_DOUBLES (64 * BLD + 64)

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

// Block column `kb` (columns o = 16 kb .. o + 15, rows o .. 63) by ONE wavefront.
// Returns 1 if a pivot was not positive.
__device__ __forceinline__ int factor_panel(double *sD, double *sRd, int kb, int lane) {
  const int o = 16 * kb;
  const int nrow = 64 - o;
  const int row = o + (lane < nrow ? lane : nrow - 1);   // idle lanes shadow the last row
  double *prow = sD + row * BLD + o;
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; j += 2) {
    const d2v v = *reinterpret_cast<const d2v *>(prow + j);
    a[j] = v.x;
    a[j + 1] = v.y;
  }
  int notpd = 0;
  double p = read_lane(a[0], 0);
  if (!(p > 0.0)) notpd = 1;
  double r = rsqrt_nr(p);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const double l = a[c] * r;   // column c of L, every row at once (rows < c: unused)
    a[c] = l;
    if (lane == c) sRd[o + c] = r;
    if (c < 15) {
      // the next pivot first, so that its rsqrt overlaps the rest of this column
      const double s1 = read_lane(l, c + 1);
      a[c + 1] = fma(-l, s1, a[c + 1]);
      p = read_lane(a[c + 1], c + 1);
      if (!(p > 0.0)) notpd = 1;
      r = rsqrt_nr(p);
#pragma unroll
      for (int j = c + 2; j < 16; ++j) {
        const double s = read_lane(l, j);
        a[j] = fma(-l, s, a[j]);
      }
    }
  }
  if (lane < nrow) {
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
      d2v v;
      v.x = (lane < 16 && j > lane) ? 0.0 : a[j];          // strict upper part of the leaf
      v.y = (lane < 16 && j + 1 > lane) ? 0.0 : a[j + 1];
      *reinterpret_cast<d2v *>(prow + j) = v;
    }
  }
  return notpd;
}

// All 256 threads of the workgroup call this with the block already in sD
// (lower triangle valid, upper part mirrored or zero; see callers for the
// identity padding of a partial block) and a barrier behind the stores.
// Returns 1 in every thread of wavefronts that saw a non-positive pivot
// (callers OR it through global memory).
__device__ __forceinline__ int diag_block(double *sD, double *sRd, double *__restrict__ lt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int notpd = 0;
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    if (wave == kb) notpd |= factor_panel(sD, sRd, kb, lane);
    __syncthreads();
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
  }
  // operands of the panel solve: L^T with the reciprocal diagonal
  for (int e = tid; e < 4096; e += 256) {
    const int k = e >> 6, c = e & 63;
    lt[e] = c > k ? sD[c * BLD + k] * sRd[c] : (c == k ? sRd[k] : 0.0);
  }
  return notpd;
}

#endif
