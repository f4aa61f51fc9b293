// Global -> registers -> LDS staging of a 64-row x BK operand slice (shared by the register-staged
// product loops of sp_gemm.hip and sp_panel.hip): the loads of slice k + 1 are in flight while slice
// k feeds the MFMAs.  BK/2 lanes x 16 B per row, 256 / (BK/2) rows per pass, 64 / that passes.
// LDS rows of BK + 1 doubles: hipcc fuses the per-k-step fragment reads into ds_read2_b64, which is
// banked mod 32 dwords in 16-lane groups -- an ODD row length puts the 16 rows of a group on 16
// distinct bank pairs (an even row length of BK + 2 cost 42 % extra LDS cycles, profiles/r01_*).
#ifndef SP_STAGE_H
#define SP_STAGE_H

#include <hip/hip_runtime.h>

typedef double sp_d2 __attribute__((ext_vector_type(2)));

#define SP_GT 64   // tile edge

template <int BK>
struct PanelRegs {
  sp_d2 v[SP_GT / (256 / (BK / 2))];
};

// Thread -> (row of a pass, first column of its pair).  Rounds 1-4: 16 consecutive lanes = ONE row's 16 pairs, whose
// LDS stores (ds_write_b64 / ds_write2_b64: served in groups of 16 consecutive lanes, banks of 4 B mod 32) land
// 4 dwords apart -- eight banks twice over, a 2-way conflict on every store (`SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE` = 0.216 for the panel kernel, profiles/r04_pmc_sq.txt: 8 conflicting stores against 40
// conflict-free fragment reads per slice).  Now a group of 16 lanes holds 8 pairs of TWO neighbouring rows: the odd
// row length (BK + 1 doubles = 66 dwords = 2 mod 32) shifts the second row's banks by two -- 16 distinct bank pairs,
// conflict-free (host model: tools/lds_bank_model.py).  The global side is unchanged in what it touches: 8 lanes x
// 16 B = one 128-byte line of a row, the row's other line by the lanes 16 further on.
template <int BK>
__device__ __forceinline__ void stage_map(int t, int &row, int &cpair) {
  constexpr int LPR = BK / 2;
#ifdef SP_STAGE_OLDMAP
  if (false) {
#else
  if (BK == 32) {
#endif
    const int l = t & 31;
    row = 2 * (t >> 5) + ((l >> 3) & 1);
    cpair = 2 * ((l & 7) + 8 * (l >> 4));
  } else {
    row = t / LPR;
    cpair = (t % LPR) * 2;
  }
}

template <int BK>
__device__ __forceinline__ void stage_load(const double *P, long ld, int row0, int nrows,
                                           int k0, int Kd, bool vec_ok, PanelRegs<BK> &R) {
  constexpr int LPR = BK / 2, RPP = 256 / LPR;
  int r0, cpair;
  stage_map<BK>(threadIdx.x, r0, cpair);
#pragma unroll
  for (int pass = 0; pass < SP_GT / RPP; ++pass) {
    const int gr = row0 + r0 + RPP * pass;
    sp_d2 v = {0.0, 0.0};
    if (gr < nrows) {
      const double *src = P + (size_t)gr * ld + k0 + cpair;
      if (vec_ok && k0 + cpair + 1 < Kd) {
        v = *reinterpret_cast<const sp_d2 *>(src);
      } else {
        if (k0 + cpair < Kd) v.x = src[0];
        if (k0 + cpair + 1 < Kd) v.y = src[1];
      }
    }
    R.v[pass] = v;
  }
}

// the same without bounds checks: full 64-row tiles, BK | Kd, 16-byte aligned rows
// (tid: a caller inside a long-lived loop passes a laundered copy of threadIdx.x)
template <int BK>
__device__ __forceinline__ void stage_load_fast(const double *P, long ld, int row0, int k0,
                                                PanelRegs<BK> &R, int tid = threadIdx.x) {
  constexpr int LPR = BK / 2, RPP = 256 / LPR;
  int r0, cpair;
  stage_map<BK>(tid, r0, cpair);
  const double *src = P + (size_t)(row0 + r0) * ld + k0 + cpair;
#pragma unroll
  for (int pass = 0; pass < SP_GT / RPP; ++pass)
    R.v[pass] = *reinterpret_cast<const sp_d2 *>(src + (size_t)(RPP * pass) * ld);
}

template <int BK>
__device__ __forceinline__ void stage_store(const PanelRegs<BK> &R, double scale,
                                            double *__restrict__ s, int tid = threadIdx.x) {
  constexpr int LDW = BK + 1, LPR = BK / 2, RPP = 256 / LPR;
  int r0, cpair;
  stage_map<BK>(tid, r0, cpair);
#pragma unroll
  for (int pass = 0; pass < SP_GT / RPP; ++pass) {
    const int r = r0 + RPP * pass;
    s[r * LDW + cpair] = R.v[pass].x * scale;
    s[r * LDW + cpair + 1] = R.v[pass].y * scale;
  }
}

#endif
