// Covariance entries of the marginal path, shared by the assembly (sp_assemble.hip) and by the
// kernels of the factorisation that form tiles at their first touch instead of reading them
// (LazyCov below; sp_gemm.hip): phase-lag spline lookup (flux.py:262-272), temporal kernels
// (temporal.py:8-16), valid-cadence masks.
#ifndef SP_COV_H
#define SP_COV_H

#include "sp_internal.h"

// valid cadences of a star: sp_star.nobs when 0 < nobs < K (ragged ensembles), else K
__device__ __forceinline__ int star_nobs(const sp_star &st, int K) {
  return (st.nobs > 0 && st.nobs < K) ? st.nobs : K;
}

__device__ __forceinline__ double temporal_factor(int kind, double ti, double tj,
                                                  double tau) {
#pragma clang fp contract(off)
  if (kind == SP_TEMPORAL_NONE) return 1.0;
  const double dt = fabs(ti - tj);
  if (kind == SP_TEMPORAL_MATERN32) {
    const double x = 1.7320508075688772 * dt / tau;  // np.sqrt(3) * dt / tau
    return (1.0 + x) * exp(-x);
  }
  return exp(-(dt * dt) / (2.0 * tau));
}

// sums over the 16 lanes of a DPP row, N values at once, result in every lane: four exchange-and-add steps on the
// cross-lane data path (quad_perm xor 1, xor 2, row_half_mirror, row_mirror; two 32-bit moves per double).
// __shfl_xor(double, k, 16) compiles to ds_bpermute_b32 pairs -- a round trip through the LDS crossbar per step
// with a full lgkmcnt wait behind each: 16 dependent round trips per assembled tile (round 4).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void row16_sum(double (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] += dpp_move<0xB1>(x[k]);    // quad_perm [1, 0, 3, 2]
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] += dpp_move<0x4E>(x[k]);    // quad_perm [2, 3, 0, 1]
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] += dpp_move<0x141>(x[k]);   // row_half_mirror
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] += dpp_move<0x140>(x[k]);   // row_mirror
}

// spline lookup (flux.py:262-272)
//
// The segment index is integer work and must equal floor(fl(x / dx)) bit for
// bit.  An IEEE fp64 division costs ~12 dependent instructions, so the index is
// first taken from the product q = x * (1/dx) (within 2 ulp of the quotient)
// and the exact division is only evaluated when q lies within 1e-9 of an
// integer, the only case in which the two floors can differ.
typedef double dd2 __attribute__((ext_vector_type(2)));

struct SplineGen {
  const double *tab;   // LDS: {a0, a1} per segment (16 B each), then {a2, a3} per segment at
                       // tab + 2 np.  Two arrays of 16-byte entries rather than one of 32:
                       // neighbouring lanes look up neighbouring segments, and 16-byte entries put
                       // 16 consecutive segments on 16 different bank groups (32-byte ones: 8),
                       // which halves the bank conflicts of these gathers (they were 61 % of the
                       // LDS-busy cycles of the row sums, and the LDS was busy 65 % of the time)
  int np2;             // 2 np
  double dx, inv_dx;
  int covpts;
  __device__ __forceinline__ double operator()(double thi, double thj) const {
    // (no contraction whatever the translation unit's default: the index must be the reference's,
    //  and a tile formed at first touch must carry the bits the assembly would have written)
#pragma clang fp contract(off)
    const double x = fabs(thi - thj);
    const double q = x * inv_dx;
    // 0 <= x <= 2 pi, so the int64 index of the reference fits 32 bits (one v_cvt_i32_f64)
    int idx = (int)q;
    // x0 = (x - xp[idx + 1]) / dx with xp[k] = (k - 1) dx (flux.py:312-314): q - idx, equal to
    // 3e-14 absolute on [0, 1) and one LDS read shorter.  It also tells how close q is to an
    // integer: within 1e-9 of one the exact quotient decides the index.
    double x0 = q - (double)idx;
    if (fabs(x0 - 0.5) > 0.5 - 1.0e-9) {
      idx = (int)floor(x / dx);
      x0 = q - (double)idx;
    }
    idx = idx > covpts ? covpts : idx;      // (a conversion of |.| * inv_dx: never negative, 0 for a NaN)
    // two 16-byte LDS reads fetch the four coefficients of the segment (addresses by hand: see many())
    typedef const __attribute__((address_space(3))) dd2 *lds_dd2;
    const unsigned a1 = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)tab + ((unsigned)idx << 4);
    const dd2 c01 = *(lds_dd2)(size_t)a1;
    const dd2 c23 = *(lds_dd2)(size_t)(a1 + 8u * (unsigned)np2);
    // a0 + a1 x0 + a2 x0^2 + a3 x0^3 in Horner form (the value, unlike the index, only has
    // to agree to rounding: fused multiply-adds)
    return __builtin_fma(x0, __builtin_fma(x0, __builtin_fma(x0, c23.y, c23.x), c01.y), c01.x);
  }
  // N entries at once: entry e is operator()(thi[e], thj[e]), the same operations on the same operands
  // -- the same bits --, but the N index computations, the 2 N gathers and the N Horner chains are
  // straight-line code on either side of ONE rarely taken branch (some entry's quotient within 1e-9 of
  // an integer: the diagonal, x = 0, and little else), so that they overlap.  One entry at a time each
  // evaluation was a dependent chain of ~15 fp64 operations and two LDS round trips behind a branch of
  // its own, which nothing but other wavefronts could hide (round 4; the assembly ran at 0.65
  // evaluations per ns, a third of what its instruction count allows).
  template <int N>
  __device__ __forceinline__ void many(const double (&thi)[N], const double (&thj)[N], double (&out)[N]) const {
#pragma clang fp contract(off)
    double x0[N];
    int idx[N];
    bool odd = false;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const double q = fabs(thi[e] - thj[e]) * inv_dx;
      idx[e] = (int)q;
      x0[e] = q - (double)idx[e];
      odd |= fabs(x0[e] - 0.5) > 0.5 - 1.0e-9;
    }
    if (__builtin_expect(odd, 0)) {
#pragma unroll
      for (int e = 0; e < N; ++e)
        if (fabs(x0[e] - 0.5) > 0.5 - 1.0e-9) {
          const double x = fabs(thi[e] - thj[e]);
          idx[e] = (int)floor(x / dx);
          x0[e] = x * inv_dx - (double)idx[e];
        }
    }
    dd2 c01[N], c23[N];
    // 32-bit LDS addresses by hand: min, shift-add, add per entry.  (From `tab + 2 k` the compiler made eight
    // vector instructions and a two-cycle bubble per entry -- a signed clamp through a compare and a select, two
    // shifts, an add of the LDS base's relocation, an add of the second array's offset -- a third of everything the
    // assembly issued.  The index is a conversion of |.| * inv_dx: never negative, 0 for a NaN.)
    typedef const __attribute__((address_space(3))) dd2 *lds_dd2;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) double *)tab;
    const unsigned off2 = 8u * (unsigned)np2;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const int k = idx[e] > covpts ? covpts : idx[e];
      const unsigned a1 = base + ((unsigned)k << 4);
      c01[e] = *(lds_dd2)(size_t)a1;
      c23[e] = *(lds_dd2)(size_t)(a1 + off2);
    }
#pragma unroll
    for (int e = 0; e < N; ++e)
      out[e] = __builtin_fma(x0[e], __builtin_fma(x0[e], __builtin_fma(x0[e], c23[e].y, c23[e].x), c01[e].y), c01[e].x);
  }
};


// the star's packed table (theta_kernel: [np][2] = {a0, a1}, then [np][2] = {a2, a3}) into LDS, all 256 threads:
// every load is issued before the first store (a plain copy loop waits out one memory round trip per
// iteration -- the compiler keeps load, wait, store together)
__device__ __forceinline__ void spline_table_to_lds(const double *__restrict__ src, double *stage, int np, int tid) {
  const int n2 = 2 * np;                       // 16-byte entries
  dd2 v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = tid + 256 * c;
    v[c] = e < n2 ? *reinterpret_cast<const dd2 *>(src + 2 * e) : dd2{0.0, 0.0};
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = tid + 256 * c;
    if (e < n2) *reinterpret_cast<dd2 *>(stage + 2 * e) = v[c];
  }
  for (int e = tid + 1024; e < n2; e += 256)   // (covpts > 508: not a BASELINE configuration)
    *reinterpret_cast<dd2 *>(stage + 2 * e) = *reinterpret_cast<const dd2 *>(src + 2 * e);
}

// ---- tiles formed at first touch ---------------------------------------------------------------
// In the deferred-normalisation marginal path a tile of the system strictly below the diagonal whose
// rows are all covariance rows is a pure function of (phase_i, phase_j[, t_i, t_j]) until the
// factorisation first touches it.  With `LazyCov` the assembly does not write those tiles (it
// still takes their row / column sums) and the kernel that touches a tile first -- the panel
// kernel for the block columns of the first super-panel, the first trailing update for the rest --
// evaluates its entries instead of loading them: the same spline code, the coefficients gathered
// from a packed copy of the star's table in memory (L1 / L2), hence the same bits.  One write and
// one read of 3/4 of the matrix less per factorisation.

// the 16 entries of a lane of the 4 x (16 x 64) wavefront split: rows ri[r], columns cj[n]
// (absolute indices in the star's system), accumulator layout out[n][r]
// `stage`: LDS scratch of at least 4 (covpts + 4) doubles that the whole workgroup may use now: the
// star's packed table is copied there first (one memory round trip, in parallel with the phases';
// gathering the coefficients from memory instead made the entries wait for two round trips in a
// row: +3 us per tile).  All 256 threads call this together; two barriers inside.
#ifndef SP_LAZY_BATCH
#define SP_LAZY_BATCH 8
#endif
// MA row groups at once (the 128 x 64 tiles of large remainders: two; the 64 x 64 tiles: one): the table is copied and
// the column side is prepared once.
//
// The Matern-3/2 factor (1 + x) exp(-x), x = sqrt(3) |t_i - t_j| / tau (temporal.py:8-11), SEPARATES in a tile below
// the diagonal of a light curve whose cadences are in order (LazyCov.inorder, from the data plan):
//     exp(-c (t_i - t_j)) = exp(-c (t_i - b)) exp(c (t_j - b)),      c = sqrt(3) / tau,  b = the tile's first cadence
// -- 4 (MA + 1) exponentials per lane instead of 16 MA, and no division.  One exponential, one division and the
// polynomial per entry were three times the spline's own cost: the first trailing update of cfg5's shape ran at 0.70
// of the fp64 peak where the later ones, which load their tiles, reach 0.78.  A column whose factor would overflow
// (c (t_j - b) >= 600: a gap of years inside 64 cadences at a tau of hours), a star out of order, every other kernel:
// the entry-by-entry form.  The two forms agree to a few ulp (as in the assembly, sp_assemble.hip).
template <int MA, typename V4>
__device__ __forceinline__ void lazy_cov_tiles(const LazyCov &z, int star, const int (&ri)[MA][4],
                                               const int (&cj)[4], V4 (&out)[MA][4], double *stage) {
  const sp_star st = z.stars[star];
  const int nobs = star_nobs(st, z.K), np = z.covpts + 4;
  const double *th = z.theta + (size_t)star * z.K, *tt = z.t + (size_t)star * z.K;
  const bool tk = z.temporal != SP_TEMPORAL_NONE;
#ifdef SP_LAZY_NO_SEP
  const bool sep = false;      // (A/B: every entry its own exponential)
#else
  const bool sep = z.temporal == SP_TEMPORAL_MATERN32 && z.inorder && z.inorder[star] != 0.0;
#endif
  double thi[MA][4], ti[MA][4], thj[4], tj[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool oj = cj[k] < nobs;
    thj[k] = oj ? th[cj[k]] : 0.0;
    tj[k] = (oj && tk) ? tt[cj[k]] : 0.0;
#pragma unroll
    for (int m = 0; m < MA; ++m) {
      const bool oi = ri[m][k] < nobs;
      thi[m][k] = oi ? th[ri[m][k]] : 0.0;
      ti[m][k] = (oi && tk) ? tt[ri[m][k]] : 0.0;
    }
  }
  const int c0 = cj[0] & ~63;                                    // (tiles are aligned to 64 columns)
  const double tb = sep ? tt[c0 < nobs ? c0 : nobs - 1] : 0.0;
  spline_table_to_lds(z.ptab + (size_t)star * 4 * np, stage, np, threadIdx.x);
  __syncthreads();
  SplineGen g{stage, 2 * np, 6.283185307179586 / z.covpts,
              1.0 / (6.283185307179586 / z.covpts), z.covpts};
  const double cm = sep ? 1.7320508075688772 / st.tau : 0.0;
  double fc[4] = {1.0, 1.0, 1.0, 1.0};
  bool fok[4] = {false, false, false, false};
  if (sep) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const double a = cm * (tj[n] - tb);
      fok[n] = a < 600.0 && a >= 0.0;                             // (false for a NaN: tau = 0)
      fc[n] = exp(fok[n] ? a : 0.0);
    }
  }
#pragma unroll
  for (int m = 0; m < MA; ++m) {
    double er[4] = {1.0, 1.0, 1.0, 1.0};
    if (sep) {
#pragma unroll
      for (int r = 0; r < 4; ++r) er[r] = exp(-(cm * (ti[m][r] - tb)));
    }
    // (eight entries per batch: sixteen take the trailing update from 163 registers -- three workgroups per CU -- to 204)
#pragma unroll
    for (int n0 = 0; n0 < 4; n0 += SP_LAZY_BATCH / 4) {
      double a[SP_LAZY_BATCH], b[SP_LAZY_BATCH], v[SP_LAZY_BATCH];
#pragma unroll
      for (int e = 0; e < SP_LAZY_BATCH; ++e) {
        a[e] = thi[m][e & 3];
        b[e] = thj[n0 + (e >> 2)];
      }
      g.many<SP_LAZY_BATCH>(a, b, v);
#pragma unroll
      for (int e = 0; e < SP_LAZY_BATCH; ++e) {
        const int n = n0 + (e >> 2), r = e & 3;
        double w = 0.0;
        if (ri[m][r] < nobs && cj[n] < nobs) {
#pragma clang fp contract(off)
          if (sep && fok[n]) {
            const double x = cm * (ti[m][r] - tj[n]);             // (rows below columns, cadences in order: >= 0)
            w = v[e] * ((1.0 + x) * (er[r] * fc[n]));
          } else {
            w = v[e] * temporal_factor(z.temporal, ti[m][r], tj[n], st.tau);
          }
        }
        out[m][n][r] = w;
      }
    }
    if (z.rid) {
      // (a row tile that holds rows below the cadences: residuals, ones, variances -- the planned step; LazyCov.rid)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mr = ri[m][r] - z.K;
        if (mr < 0 || mr >= z.nrid) continue;
        const double *src = z.rid + ((size_t)star * (z.nrid + 1) + mr) * z.K;
        double x[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) x[n] = src[cj[n] < z.K ? cj[n] : z.K - 1];
#pragma unroll
        for (int n = 0; n < 4; ++n) out[m][n][r] = cj[n] < z.K ? x[n] : 0.0;
      }
    }
  }
  __syncthreads();   // the scratch goes back to its owner
}

// One 16 x 16 block of a DIAGONAL tile of the system in the accumulator layout (rows ri[r], column cj), formed at
// first touch (the planned step, LazyCov.dlazy): covariance entries + D / c1 on the diagonal (the row dd behind the
// riding rows of LazyCov.rid), the riding rows themselves where the tile holds them, identity on the padding.  The
// caller has the star's table in LDS (g) and every index < the padded size; loads are clamped to the light curve.
template <typename V4>
__device__ __forceinline__ void lazy_diag_block(const LazyCov &z, int star, const SplineGen &g, const sp_star &st,
                                                int nobs, const int (&ri)[4], int cj, V4 &out) {
  const double *th = z.theta + (size_t)star * z.K, *tt = z.t + (size_t)star * z.K;
  const double *rb = z.rid + (size_t)star * (z.nrid + 1) * z.K;
  const bool tk = z.temporal != SP_TEMPORAL_NONE;
  const int cc = cj < z.K ? cj : z.K - 1;
  const double thj = th[cc], tj = tk ? tt[cc] : 0.0;
  double thi[4], ti[4], dd[4], rv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rc = ri[r] < z.K ? ri[r] : z.K - 1;
    thi[r] = th[rc];
    ti[r] = tk ? tt[rc] : 0.0;
    dd[r] = rb[(size_t)z.nrid * z.K + rc];
    const int m = ri[r] - z.K;
    rv[r] = rb[(size_t)((m >= 0 && m < z.nrid) ? m : 0) * z.K + cc];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = ri[r] - z.K;
    double v;
    if (ri[r] < nobs && cj < nobs) {
#pragma clang fp contract(off)
      v = g(thi[r], thj) * temporal_factor(z.temporal, ti[r], tj, st.tau);
      if (ri[r] == cj) v += dd[r];
    } else if (m >= 0 && m < z.nrid && cj < z.K) {
      v = rv[r];
    } else {
      v = ri[r] == cj ? 1.0 : 0.0;
    }
    out[r] = v;
  }
}

// the same for a lane that holds ONE row and four groups of four consecutive columns (the panel
// kernel's transposed accumulator layout, sp_panel.hip): out[m][r] = entry (row ri, column
// c0 + 16 m + r).  No temporal kernel here: tiles are only left to their first touch without one
// (sp_lnlike_ensemble; the exp per entry, evaluated twice, costs more than the traffic saves), and
// sixteen inlined copies of exp() would be dead code in the panel kernel.
#ifndef SP_LAZY_ROW_BATCH
#define SP_LAZY_ROW_BATCH 0    // entries per SplineGen::many batch in the panel kernel's 64-row items (0: one at a time; the
                               // 128-row pair items always take one at a time).  Batches of four cost the lazy
                               // instantiations 26-31 spilled registers of their 168 whatever is done about it (the
                               // column phases through LDS, a scheduling barrier between batches): 107.9k against
                               // 110.6-112.8k in flight.  Pairs fit (154 registers) and measure the same as one at a
                               // time: 110.1-111.4k against 110.9-112.3k in flight, 0.816-0.821 against 0.821-0.828 ms
                               // alone.  (What the evaluation costs there: 0.052 ms of a 0.573 ms step with four in
                               // flight, 0.026 alone -- a build that fills these tiles with constants.)
#endif
template <int ROWB, typename V4>
__device__ __forceinline__ void lazy_cov_row(const LazyCov &z, int star, int ri, int c0, V4 (&out)[4],
                                             double *stage, int tid) {
  const sp_star st = z.stars[star];
  const int nobs = star_nobs(st, z.K), np = z.covpts + 4;
  const double *th = z.theta + (size_t)star * z.K;
  const bool oi = ri < nobs;
  const double thi = oi ? th[ri] : 0.0;
  double *s_thj = stage + 4 * np;      // the tile's 64 column phases, behind the table (zero beyond the cadences)
  {
    // the table in three 16-byte pieces per thread, all three loads (and the column phases') in flight before
    // the first store: a plain copy loop waits out one memory round trip per iteration -- 2.4 of them per item,
    // 5 us of a workgroup slot each time a tile is formed.  (Four pieces, spline_table_to_lds, cost this kernel
    // its register budget; tables beyond 764 segments take the loop for the rest.)
    const double *src = z.ptab + (size_t)star * 4 * np;
    const int n2 = 2 * np, cbase = c0 & ~63;
    dd2 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int e = tid + 256 * c;
      v[c] = e < n2 ? *reinterpret_cast<const dd2 *>(src + 2 * e) : dd2{0.0, 0.0};
    }
    const double tj = (tid < 64 && cbase + tid < nobs) ? th[cbase + tid] : 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int e = tid + 256 * c;
      if (e < n2) *reinterpret_cast<dd2 *>(stage + 2 * e) = v[c];
    }
    for (int e = tid + 768; e < n2; e += 256)
      *reinterpret_cast<dd2 *>(stage + 2 * e) = *reinterpret_cast<const dd2 *>(src + 2 * e);
    if (tid < 64) s_thj[tid] = tj;
  }
  __syncthreads();
  SplineGen g{stage, 2 * np, 6.283185307179586 / z.covpts,
              1.0 / (6.283185307179586 / z.covpts), z.covpts};
  const int cl = c0 & 63;
  // (unrolled -- a run-time index into the caller's accumulators would send them to scratch memory)
  // The column phases come from LDS, four at a time: fetched from memory all sixteen were in flight across the
  // whole evaluation (32 registers), which is what made batches spill in this kernel.
  if constexpr (ROWB == 2) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    V4 o;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const dd2 tt = *reinterpret_cast<const dd2 *>(s_thj + cl + 16 * m + 2 * half);
      const double a[2] = {thi, thi}, b[2] = {tt.x, tt.y};
      double v[2];
      g.many<2>(a, b, v);
      o[2 * half] = (oi && c0 + 16 * m + 2 * half < nobs) ? v[0] : 0.0;
      o[2 * half + 1] = (oi && c0 + 16 * m + 2 * half + 1 < nobs) ? v[1] : 0.0;
      __builtin_amdgcn_sched_barrier(0);
    }
    out[m] = o;
  }
  } else if constexpr (ROWB == 0) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const dd2 t01 = *reinterpret_cast<const dd2 *>(s_thj + cl + 16 * m), t23 = *reinterpret_cast<const dd2 *>(s_thj + cl + 16 * m + 2);
    const double thj[4] = {t01.x, t01.y, t23.x, t23.y};
    V4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (oi && c0 + 16 * m + r < nobs) ? g(thi, thj[r]) : 0.0;
    out[m] = o;
  }
  } else {
#pragma unroll
  for (int m0 = 0; m0 < 4; m0 += ROWB / 4) {
    double a[ROWB], b[ROWB], v[ROWB];
#pragma unroll
    for (int h = 0; h < ROWB / 4; ++h) {
      const dd2 t01 = *reinterpret_cast<const dd2 *>(s_thj + cl + 16 * (m0 + h));
      const dd2 t23 = *reinterpret_cast<const dd2 *>(s_thj + cl + 16 * (m0 + h) + 2);
      a[4 * h] = a[4 * h + 1] = a[4 * h + 2] = a[4 * h + 3] = thi;
      b[4 * h] = t01.x;
      b[4 * h + 1] = t01.y;
      b[4 * h + 2] = t23.x;
      b[4 * h + 3] = t23.y;
    }
    g.many<ROWB>(a, b, v);
#pragma unroll
    for (int h = 0; h < ROWB / 4; ++h) {
      V4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (oi && c0 + 16 * (m0 + h) + r < nobs) ? v[4 * h + r] : 0.0;
      out[m0 + h] = o;
    }
    // (one batch after the other: left to itself the scheduler starts all of them at once)
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  if (z.rid && !oi) {
    // (this lane's row lies below the cadences: residuals, ones, variances -- the planned step leaves the row tiles
    //  that hold them to their first touch too, LazyCov.rid; its sixteen columns' loads all issued before the first use)
    const int mr = ri - z.K;
    if (mr >= 0 && mr < z.nrid) {
      const double *src = z.rid + ((size_t)star * (z.nrid + 1) + mr) * z.K;
      double x[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = c0 + 16 * (e >> 2) + (e & 3);
        x[e] = src[c < z.K ? c : z.K - 1];        // (clamped: a row tile of riding rows only may reach columns beyond K)
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        V4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = c0 + 16 * m + r < z.K ? x[4 * m + r] : 0.0;
        out[m] = o;
      }
    }
  }
  __syncthreads();   // the scratch goes back to its owner
}

#endif
