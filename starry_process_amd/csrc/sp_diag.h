// Device routine shared by sp_cholesky.hip (stand-alone diagonal-block kernel)
// and sp_gemm.hip (fused into the tile-(0,0) workgroup of a block-column update).
//
// diag_block(): Cholesky factor L of a 64 x 64 SPD block held in LDS and its
// inverse, by one 256-thread workgroup.  4 x 4 sub-blocks of 16 x 16:
//   leaf kb (wavefront kb): the 16 x 16 leaf sits in ONE accumulator tile of
//     v_mfma_f64_16x16x4_f64 (lane (fk, fr) holds rows fk + 4 q, column fr).  By
//     symmetry row c = column c, and row c is held by the 16 lanes of group
//     fk = c & 3 in register q = c >> 2 -- exactly the operand slot k = fk of the
//     MFMA.  So the rank-1 step  A <- A - l l^T  (l = column c of L) needs no lane
//     traffic: lanes of that group pass l, all others pass 0.  The leaf inverse
//     rides along: L = L_0 .. L_15 with L_c = I + (l_c - e_c) e_c^T, hence
//     Y <- Y - u_c (e_c^T Y), u_c = (l_c - e_c) / l_cc, Y_0 = I  ends at Y = L^-1.
//   sub-diagonal blocks  L_ik = A_ik Y_kk^T  and trailing sub-blocks
//     A_ij -= L_ik L_jk^T  on MFMA with LDS operands, spread over the waves;
//   off-diagonal blocks of L^-1:  X_ij = -Y_ii sum_{k=j}^{i-1} L_ik X_kj, block
//     column j on wavefront j, whose own leaf inverse Y_jj is still in its
//     accumulator registers (the C/D register map of the fp64 MFMA IS the
//     k-major operand map, so an accumulator tile feeds the next MFMA directly);
//     finished blocks X_ij are parked in the unused UPPER blocks (j, i) of the
//     LDS tile.
// LDS: sD[64 * BLD] (block in, L out in the lower part) + sY[3][16 * YLD] (leaf
// inverses).  The full inverse is written to `inv` (global, 64 x 64 row-major).
#ifndef SP_DIAG_H
#define SP_DIAG_H

#include <hip/hip_runtime.h>

#define BLD 66   // LDS row of the 64x64 block: 132 dwords = 4 mod 64 banks
#define YLD 16   // LDS row of a 16x16 leaf inverse
// three leaf-inverse slots: Y_0 is only needed while block column 0 is solved,
// its slot is reused for Y_3.  64*66 + 3*256 doubles = 39,936 B <= 40 KiB, so the
// fused GEMM variant keeps 4 workgroups per CU like the plain one.
#define SP_DIAG_LDS_DOUBLES (64 * BLD + 3 * 16 * YLD)
#define SP_YSLOT(kb) ((kb) == 3 ? 0 : (kb))

typedef double d4 __attribute__((ext_vector_type(4)));

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
// NN b-operand fragment: M[row0 + 4 s + (lane >> 4)][col0 + (lane & 15)]
__device__ __forceinline__ double frag_kmajor(const double *M, int ldm, int row0, int col0,
                                              int s, int lane) {
  return M[(row0 + 4 * s + (lane >> 4)) * ldm + col0 + (lane & 15)];
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

// All 256 threads of the workgroup call this with the block already in sD
// (lower triangle valid; see callers for the identity padding of a partial
// block) and a barrier behind the stores.  Returns 1 in every thread of
// wavefronts that saw a non-positive pivot (callers OR it through LDS/global).
__device__ __forceinline__ int diag_block(double *sD, double *sY, double *__restrict__ inv) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  int notpd = 0;
  d4 Ymine = {0.0, 0.0, 0.0, 0.0};  // wavefront kb keeps its leaf inverse Y_kk
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    const int o = 16 * kb;
    double *sYk = sY + SP_YSLOT(kb) * 16 * YLD;
    if (wave == kb) {
      d4 Am, Ym;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = fk + 4 * q;
        Am[q] = fr <= row ? sD[(o + row) * BLD + o + fr] : sD[(o + fr) * BLD + o + row];
        Ym[q] = row == fr ? 1.0 : 0.0;
      }
      // Pivots run one column ahead of the MFMA chain: r_c = 1/sqrt(p_c) is
      // ready before column c is applied, so the critical path per column is
      // MFMA -> one multiply -> MFMA instead of MFMA -> readlane -> rsqrt -> MFMA.
      // p_{c+1} = a_{c+1,c+1} - (a_{c+1,c} r_c)^2 with both a's read from the
      // tile as it stands BEFORE update c (i.e. after update c-1).
      double r_cur, r_nxt;
      {
        const double p0 = read_lane(Am[0], 0);
        if (!(p0 > 0.0)) notpd = 1;
        r_cur = rsqrt_nr(p0);
        const double a10 = read_lane(Am[0], 16 * 1 + 0);   // row 1: group 1, reg 0
        const double a11 = read_lane(Am[0], 16 * 1 + 1);
        const double l10 = a10 * r_cur;
        const double p1 = fma(-l10, l10, a11);
        if (!(p1 > 0.0)) notpd = 1;
        r_nxt = rsqrt_nr(p1);
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int g = c & 3, q = c >> 2;
        const double r = r_cur;
        const bool act = (fk == g) && (fr >= c);
        const double arow = Am[q];
        const double t0 = arow * r;
        const double l = act ? t0 : 0.0;              // l_{fr,c}; fr == c: sqrt(piv)
        const double lr = l * r;                       // 0 outside the active lanes
        const double u = (act && fr == c) ? 1.0 - r : lr;
        const double yq = Ym[q];
        const double yrow = (fk == g) ? yq : 0.0;      // row c of Y
        if (act) sD[(o + fr) * BLD + o + c] = l;
        Am = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, Am, 0, 0, 0);
        Ym = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, yrow, Ym, 0, 0, 0);
        r_cur = r_nxt;
        if (c + 2 < 16) {
          // pivot of column c+2 from the tile after update c, plus the (not yet
          // applied) contribution of column c+1
          const int R = c + 2, gR = R & 3, qR = R >> 2;
          const double a21 = read_lane(Am[qR], 16 * gR + c + 1);
          const double a22 = read_lane(Am[qR], 16 * gR + c + 2);
          const double l21 = a21 * r_nxt;
          const double p2 = fma(-l21, l21, a22);
          if (!(p2 > 0.0)) notpd = 1;
          r_nxt = rsqrt_nr(p2);
        }
      }
      acc_store(sYk, YLD, 0, 0, lane, Ym);
      acc_store(inv, 64, o, o, lane, Ym);
      Ymine = Ym;
    }
    __syncthreads();
    // sub-diagonal blocks of this block column: L_ik = A_ik . Y_kk^T
    if (wave > kb) {
      const int ib = wave;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag_rowmajor(sD, BLD, 16 * ib, o, s, lane),
                                                   frag_rowmajor(sYk, YLD, 0, 0, s, lane), acc,
                                                   0, 0, 0);
      acc_store(sD, BLD, 16 * ib, o, lane, acc);
    }
    __syncthreads();
    // trailing sub-blocks: A_ij -= L_ik L_jk^T, kb < jb <= ib
    {
      int qn = 0;
      for (int ib = kb + 1; ib < 4; ++ib)
        for (int jb = kb + 1; jb <= ib; ++jb, ++qn) {
          if ((qn & 3) != wave) continue;
          d4 acc = acc_load(sD, BLD, 16 * ib, 16 * jb, lane);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(
                -frag_rowmajor(sD, BLD, 16 * ib, o, s, lane),
                frag_rowmajor(sD, BLD, 16 * jb, o, s, lane), acc, 0, 0, 0);
          acc_store(sD, BLD, 16 * ib, 16 * jb, lane, acc);
        }
    }
    __syncthreads();
  }
  // off-diagonal blocks of L^-1.  Block column j on wavefront j; X_jj = Ymine.
  // Finished X_ij (i > j) is parked in the free upper block (j, i) of sD.
#pragma unroll 1
  for (int i = 1; i < 4; ++i) {
    if (wave < i) {
      const int j = wave;
      d4 t = {0.0, 0.0, 0.0, 0.0};
      // k = j: X_jj straight from the accumulator registers (k-major operand map)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(frag_rowmajor(sD, BLD, 16 * i, 16 * j, s, lane),
                                                 Ymine[s], t, 0, 0, 0);
      for (int k = j + 1; k < i; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          t = __builtin_amdgcn_mfma_f64_16x16x4f64(
              frag_rowmajor(sD, BLD, 16 * i, 16 * k, s, lane),
              frag_kmajor(sD, BLD, 16 * j, 16 * k, s, lane), t, 0, 0, 0);
      d4 xacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(
            -frag_rowmajor(sY + SP_YSLOT(i) * 16 * YLD, YLD, 0, 0, s, lane), t[s], xacc, 0, 0,
            0);
      acc_store(sD, BLD, 16 * j, 16 * i, lane, xacc);   // park X_ij in block (j, i)
      acc_store(inv, 64, 16 * i, 16 * j, lane, xacc);
    }
    __syncthreads();
  }
  // zero the strict upper blocks of the inverse (it is lower triangular)
  for (int e = tid; e < 6 * 256; e += 256) {
    const int b = e >> 8, w = e & 255;
    const int bi = b < 3 ? 0 : (b < 5 ? 1 : 2);
    const int bj = b < 3 ? b + 1 : (b < 5 ? b - 1 : 3);
    inv[(16 * bi + (w >> 4)) * 64 + 16 * bj + (w & 15)] = 0.0;
  }
  return notpd;
}

#endif
