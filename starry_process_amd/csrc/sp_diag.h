// diag_block(): Cholesky factor L of a 64 x 64 SPD block held in LDS, by one 256-thread workgroup,
// and its inverse L^-1, formed in the shadow of the factorisation (callers: panel_diag_core,
// sp_paneldiag.h, which writes both out; the small-K kernel, sp_small.hip).
//
// The block is cut in four block columns of 16; wavefront w owns block column w.
//   factor_panel(w): the 16 x 16 diagonal leaf with lane = row (the four 16-lane DPP rows carry
//     the same copy): column c is scaled by 1 / sqrt(pivot) in every lane at once and the rank-1
//     step a_j -= l * l_j takes l_j from lane j by a DPP row broadcast folded into the multiply-add
//     (v_fmac_f64_dpp row_newbcast) -- no LDS traffic, no scalar registers, no barrier inside the
//     leaf.  Then, still in registers, the leaf's INVERSE M = L_leaf^-1 by substitution on the
//     identity with the same instruction (LeafInvStep: the rows pre-scaled to a unit diagonal, one
//     multiply-add per step and column, four columns per lane): ~40 instructions.
//   rows below the leaf: X = A M^T on the matrix cores, one 16-row block per wavefront (below_mfma;
//     rounds 2-5: substitution, fifteen dependent steps each behind an LDS read, 1 700 cycles per
//     block column).
//   update: the tiles right of a published panel are updated on the MFMA from LDS; the owner of
//     the next panel takes the one tile its leaf needs and starts factoring while the three other
//     wavefronts share the rest -- and form the finished block row of the inverse from M.
//   Two workgroup barriers per panel.
//
// LDS: sD[64 * BLD] (block in; L out in the lower part, L^-T in the upper) + sRd[64] (1 / L_cc,
// which is also the diagonal of the inverse) + the current leaf's 16 x 16 inverse.
#ifndef SP_DIAG_H
#define SP_DIAG_H

#include <hip/hip_runtime.h>

#define BLD 66   // LDS row of the 64x64 block: even (16-B aligned rows), 132 dwords = 4 mod 64 banks
#define SP_DIAG_LDS_DOUBLES (64 * BLD + 64 + 256)   // block, reciprocal diagonal, the current leaf's inverse

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double read_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to ~1 ulp: hardware seed (v_rsq_f64, measured 5e-8 relative) and ONE third-order
// step, r (1 - e)^(-1/2) = r (1 + e/2 + 3 e^2 / 8 + O(e^3)) with e = 1 - p r^2: the same
// 1.4e-16 as two Newton steps (measured over 1e6 arguments), four dependent operations
// instead of six on the pivot-to-pivot chain of the leaf.
__device__ __forceinline__ double rsqrt_nr(double p) {
  const double r = __builtin_amdgcn_rsq(p);
  const double e = fma(-p * r, r, 1.0);
  return fma(r * e, fma(0.375, e, 0.5), r);
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

// a += bcast_J(b) * (-c): the row broadcast folded into the multiply-add (v_fmac_f64 is a VOP2 on gfx90a+ and
// takes a DPP64 row_newbcast on its first source): ONE instruction per updated entry where a v_mov_b64_dpp and
// a v_fma_f64 were two -- the leaf is bound by the instructions it issues, not by its pivot chain.  Same value
// bit for bit (a product's sign is exact).  The caller keeps two wait states between the VALU write of b and
// this read of it through DPP (s_nop 1: inline assembly is opaque to the hazard recogniser).
template <int J, int WAIT = 0>      // WAIT: idle wait states in front (s_nop WAIT - 1), 0 = none
__device__ __forceinline__ void fmac_bcast(double &a, double b, double c) {
  static_assert(WAIT >= 0 && WAIT <= 8, "s_nop takes up to 8 wait states");
  if (WAIT > 0)
    asm volatile("s_nop %4\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(a)
                 : "v"(b), "v"(c), "n"(J), "n"(WAIT > 0 ? WAIT - 1 : 0));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(a)
                 : "v"(b), "v"(c), "n"(J));
}

// rank-1 step of leaf column C on the columns J..15 of every row
template <int C, int J>
struct LeafUpd {
  static __device__ __forceinline__ void run(double (&a)[16], double l) {
    fmac_bcast<J, (J == C + 2 ? 2 : 0)>(a[J], l, l);     // (the first of a column waits out the hazard, wherever it is scheduled)
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

// M = L_cc^-1 by substitution on the identity, lane = row, FOUR columns per lane (columns 4 t + g for the lanes of
// 16-lane group g: s[t] ends as M[i][4 t + g], the A fragment (step t) of an MFMA whose A operand is M).  The rows are
// pre-scaled (Ls[k] = L_ik / L_ii, zero for k >= i), so the system has a unit diagonal: lane K's s IS x_K when step K
// comes, and the step is ONE instruction per column -- s_i -= Ls_iK * s_K with the row broadcast folded into the
// multiply-add (fmac_bcast, as in the leaf); lanes i <= K are left alone by the zeros of Ls, no select, no multiply on
// the chain.  Column 4 t + g is zero above its diagonal: chain t starts at step 4 t.  (Rounds 2-5: multiply by 1 / L_KK,
// select, two v_mov_dpp and a multiply-add per step and column -- 6 instructions where this has one; the block row's
// wavefronts held the next leaf's barrier up for ~2 000 cycles per block column, and the last block row, which nothing
// hides, was 2 us of the pivot block's 12.6.)
template <int K>
struct LeafInvStep {
  static __device__ __forceinline__ void run(double (&s)[4], const double (&Ls)[16]) {
    constexpr int NT = K / 4 + 1;      // chains that have started
    // Two wait states between a VALU write of a register and its read through DPP -- and inline assembly is opaque to
    // the compiler's hazard recogniser: it scheduled the select that INITIALISES a column (s[3]) directly in front of
    // that column's first step, whose asm carried no idle states of its own because the chains' own spacing did not
    // need them.  The inverse was then wrong in its TENTH digit in some instantiations: every comparison with the
    // oracle passed, only the 1e-12 comparison of the two samplers (tests/test_gpu_facade.py) did not.  So EVERY
    // instruction here brings at least two idle states (more while few chains are in turn: ~130 cycles per block row,
    // of ~2 000 saved), and tools/check_dpp_hazard.py (tests/test_host.py) reads the compiled kernels for a write of a
    // DPP source less than two wait states ahead of any v_fmac_f64_dpp.
    constexpr int W = NT == 1 ? 8 : (NT == 2 ? 4 : (NT == 3 ? 3 : 2)), W0 = W;
    fmac_bcast<K, W0>(s[0], s[0], Ls[K]);
    if (NT > 1) fmac_bcast<K, W>(s[1], s[1], Ls[K]);
    if (NT > 2) fmac_bcast<K, W>(s[2], s[2], Ls[K]);
    if (NT > 3) fmac_bcast<K, W>(s[3], s[3], Ls[K]);
    LeafInvStep<K + 1>::run(s, Ls);
  }
};
template <>
struct LeafInvStep<15> {
  static __device__ __forceinline__ void run(double (&)[4], const double (&)[16]) {}
};

// Block column `kb` (columns o = 16 kb .. o + 15): the leaf and its inverse, by ONE wavefront.
// Stage 1: the 16 x 16 diagonal leaf.  Lane i (mod 16) holds row i in 16 registers; a column's entries reach the
//   other rows through DPP row broadcasts (no LDS, no scalar registers, no barrier).  Rows to sD, 1 / L_ii to sRd.
// Stage 2: M = L_leaf^-1 from the rows still in registers (LeafInvStep), to sM ([column][row], what below_mfma and
//   inverse_block_row load as their fragment: M[i][4 t + g] for lane (i, g)); INV: its strictly lower part also
//   transposed into sD's upper triangle, where the block's inverse lives.
// Returns 1 if a pivot was not positive.
template <bool INV>
__device__ __forceinline__ int factor_panel(double *sD, double *sRd, double *sM, int kb,
                                            int lane, long long *ts = nullptr) {
  const int o = 16 * kb, i = lane & 15, g = lane >> 4;
  int notpd = 0;
  if (ts) ts[0] = clock64();
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
  }
  // (the last leaf of a block without an inverse: nothing below it, nobody wants M)
  if (!INV && kb == 3) return notpd;
  // the rows scaled to a unit diagonal, in place (on and right of the diagonal: zero)
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = k >= i ? 0.0 : a[k] * rd_mine;
  double res[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) res[t] = (i == 4 * t + g) ? rd_mine : 0.0;
  LeafInvStep<0>::run(res, a);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = 4 * t + g;
    sM[col * 16 + i] = res[t];
    if (INV && i > col) sD[(o + col) * BLD + o + i] = res[t];
  }
  if (ts) ts[3] = clock64();
  if (ts) ts[4] = clock64();
  return notpd;
}

// the leaf's inverse as every wavefront's fragment: res[t] = M[i][4 t + g] for lane (i, g) -- the B operand of
// below_mfma (M^T[k = 4 t + g][n = i]) and the A operand of inverse_block_row
__device__ __forceinline__ void load_leaf_inverse(const double *sM, int lane, double (&res)[4]) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) res[t] = sM[(4 * t + g) * 16 + i];
}

// The rows below the 16 x 16 leaf of block column kb, X = A L_leaf^-T = A M^T, on the matrix cores: the 16-row
// block ib = wavefront index (ib > kb; wavefronts up to kb have none), in place.
__device__ __forceinline__ void below_mfma(double *sD, const double (&res)[4], int kb, int wave, int lane) {
  if (wave <= kb) return;
  const int o = 16 * kb;
  double af[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) af[st] = frag_rowmajor(sD, BLD, 16 * wave, o, st, lane);
  d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int st = 0; st < 4; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[st], res[st], acc, 0, 0, 0);
  acc_store(sD, BLD, 16 * wave, o, lane, acc);
}

// ---- inverse of the block, formed in the shadow of the factorisation ---------------------------------
// Linv = L^-1 is built block row by block row (16 x 16 blocks) while the NEXT leaf is being factored
// by its wavefront and the others would wait:  Linv_cc = M_c = L_cc^-1 (factor_panel),
// Linv_cj = -M_c sum_{k=j}^{c-1} L_ck Linv_kj (matrix cores; the partial sum comes out of the MFMA in
// exactly the layout of the next B operand, so nothing goes through LDS in between).  Storage: the
// UPPER triangle of sD, which the factorisation never touches: sD[r][c] = Linv[c][r] for r < c; the
// diagonal of Linv is sRd.
// block Linv_cj (j < c) of block row c of the inverse by ONE wavefront, from the leaf's inverse M_c (res: its fragment,
// load_leaf_inverse)
__device__ __forceinline__ void inverse_block_row(double *sD, const double *sRd, int c, int j, int lane,
                                                  const double (&res)[4]) {
  if (j < 0) return;
  const int i = lane & 15, g = lane >> 4, o = 16 * c;
  d4 S = d4{0.0, 0.0, 0.0, 0.0};
  for (int k = j; k < c; ++k) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const double a = frag_rowmajor(sD, BLD, o, 16 * k, st, lane);          // L_ck[i][4 st + g]
      const int m = 4 * st + g;
      double b = sD[(16 * j + i) * BLD + 16 * k + m];                        // Linv_kj[m][n = i]
      if (k == j) b = m > i ? b : (m == i ? sRd[16 * j + i] : 0.0);          // the diagonal block: M_j
      S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, S, 0, 0, 0);
    }
  }
  d4 R = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int st = 0; st < 4; ++st) R = __builtin_amdgcn_mfma_f64_16x16x4f64(-res[st], S[st], R, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) sD[(16 * j + i) * BLD + o + g + 4 * r] = R[r];   // Linv_cj[g + 4 r][n = i]
}

// All 256 threads of the workgroup call this with the block already in sD (lower triangle valid,
// zero above it; see panel_diag_load for the identity padding of a partial block) and a barrier
// behind the stores.  On return (behind a barrier) sD holds L below the diagonal, L^-T above it
// (sD[r][c] = Linv[c][r] for r < c) and sRd the reciprocal diagonal.  Returns 1 in every thread of
// wavefronts that saw a non-positive pivot (callers OR it through global memory).
// INV = false: L alone -- the upper triangle stays zero, sRd is still the reciprocal diagonal (the small-K kernel's
// blocks whose inverse nobody multiplies with: forming it was half of this function's vector instructions).
template <bool INV = true>
__device__ __forceinline__ int diag_block(double *sD, double *sRd, int tid_in = threadIdx.x,
                                          long long *dbg = nullptr) {
  int notpd = 0;
  double *sM = sRd + 64;
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    // (a laundered copy of the thread index per block column: the lane-derived predicates of the leaf and the
    //  inverse -- (i == C), (k >= i), ... some 60 of them -- are loop invariants, and hoisted out of this loop they
    //  are 120 scalar registers held across it: 100 of them spilled to vector lanes and reloaded inside the loops of
    //  the critical chain.  Recomputing a compare costs one instruction.)
    int tid = tid_in;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    if (wave == kb) {
      long long ts[5];
      notpd |= factor_panel<INV>(sD, sRd, sM, kb, lane, dbg ? ts : nullptr);
      if (dbg && lane == 0)
        for (int q = 0; q < 5; ++q) dbg[8 * kb + q] = ts[q];
    }
    __syncthreads();
    // (every wavefront takes the leaf's inverse NOW: the next owner overwrites sM behind the next barrier, while the
    //  others may still be forming this block row of the inverse)
    double res[4];
    load_leaf_inverse(sM, lane, res);
    if (kb < 3) {
      below_mfma(sD, res, kb, wave, lane);
      __syncthreads();
    }
    if (dbg && wave == (kb < 3 ? kb + 1 : 3) && lane == 0) dbg[8 * kb + 5] = clock64();
    if (kb < 3) {
      // trailing update inside the block: A_{ib,w} -= L_{ib,kb} L_{w,kb}^T for the tiles
      // ib >= w > kb.  The owner of the next panel takes only the tile its leaf needs and goes
      // on to factor; the other tiles are dealt to the three other wavefronts (the barrier
      // after the next leaf is ahead of their first reader).
      const int o = 16 * kb;
      auto upd_tile = [&](int ib, int w) {
        double bfrag[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bfrag[s] = frag_rowmajor(sD, BLD, 16 * w, o, s, lane);
        d4 acc = acc_load(sD, BLD, 16 * ib, 16 * w, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag_rowmajor(sD, BLD, 16 * ib, o, s, lane),
                                                     bfrag[s], acc, 0, 0, 0);
        acc_store(sD, BLD, 16 * ib, 16 * w, lane, acc);
      };
      if (wave == kb + 1) {
        upd_tile(kb + 1, kb + 1);
      } else {
        const int hw = wave < kb + 1 ? wave : wave - 1;   // 0..2
        int t = 0;
        for (int w = kb + 1; w < 4; ++w)
          for (int ib = w; ib < 4; ++ib) {
            if (ib == kb + 1 && w == kb + 1) continue;
            if (t % 3 == hw) upd_tile(ib, w);
            ++t;
          }
        // ... and block row kb of the inverse, while the next leaf is being factored
        if (INV) inverse_block_row(sD, sRd, kb, (hw < kb) ? hw : -1, lane, res);
      }
    } else if (INV) {
      // the last block row of the inverse (nothing left to hide it behind)
      inverse_block_row(sD, sRd, 3, wave < 3 ? wave : -1, lane, res);
      __syncthreads();
    }
    if (dbg && wave == (kb < 3 ? kb + 1 : 3) && lane == 0) dbg[8 * kb + 6] = clock64();
  }
  return notpd;    // (INV = false: the barrier behind the last leaf was the last one)
}

#endif
