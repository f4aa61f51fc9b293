// The planned step's assembly (round 5; include/starry_process_amd.h: sp_lnlike_ensemble_planned).  A translation unit of
// its own: the workgroup of tile (0, 0) goes on to FACTOR pivot block 0 (diag_block, sp_diag.h), whose code wants
// the default contraction and the MFMA accumulators in VGPRs like every other caller's -- sp_assemble.hip is
// compiled with -ffp-contract=off.  What must be the reference's bits here -- the segment index of an entry -- sits
// behind `#pragma clang fp contract(off)` inside SplineGen (sp_cov.h), whatever the unit's default.
#include <cstdlib>

#include "sp_internal.h"
#include "sp_cov.h"
#include "sp_asm.h"
#include "sp_tile.h"
#include "sp_paneldiag.h"

namespace {

typedef SpCoef Coef;

// ---- the planned step's assembly (round 5) -----------------------------------------------------------------
// With a data plan (sp_plan.hip) nothing of the normalisation needs the covariance's entries before the
// factorisation: m = yp . wbar / K^2, and the reduction takes q's Gram entries from rows that ride anyway
// (sp_reduce.h).  What is left of the assembly is to put in memory the tiles the factorisation wants THERE:
//   * with tiles formed at first touch (nfull > 0): the diagonal tiles (the eager updates read-modify-write them;
//     the data variance lands on them) and the row tiles from nfull on (residual rows, the rows 1 and d, identity
//     padding) -- 31 of cfg3's 136 lower tiles, each entry of the others evaluated ONCE, by the kernel that
//     touches it first (the first block column included: LazyCov.c0lazy; the kernel is bound by the bytes it
//     stores -- with the first block column 92 MB per 64-star step, 32 us);
//   * otherwise every lower tile, still without sums.
// Same long-lived workgroups as assemble_sums_kernel (table and the star's phases in LDS once, entries in batches
// of 16, no memory load in the tile loop); every workgroup derives m, z, alpha, beta, c1 from the table and wbar
// itself (304 multiply-adds; the same bits in every workgroup of a star: same code, same order) -- no launch of
// its own for five numbers -- and the star's first workgroup leaves the coefficients, the reduction's scalars,
// the packed table (for the kernels that form tiles) and the cleared flags in memory.
#ifndef SP_PLAN_OCC
#define SP_PLAN_OCC 2
#endif
template <int TK>
__global__ __launch_bounds__(256, SP_PLAN_OCC) void assemble_planned_kernel(
    int K, int M, int Kp, PlanDev plan, const double *__restrict__ t, const sp_star *__restrict__ stars, int covpts,
    const double *__restrict__ tab, const double *__restrict__ meanvar, const double *__restrict__ flux,
    const double *__restrict__ diag, double *__restrict__ out, long ldo, long strideo, int ntr, int nfull, int ncolw,
    int order,
    double zmax, Coef *__restrict__ coef, double *__restrict__ rscal, double *__restrict__ ptab,
    int32_t *__restrict__ info, uint32_t *__restrict__ status, int lds_phases, double *__restrict__ img, long lts,
    int fuse0, int S, int nchunk, double *__restrict__ rid, int dfrom, AsmChunks chunks) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // Workgroup -> (star, chunk).  The hardware deals consecutive workgroups to the 8 XCDs in turn and, inside an XCD,
  // to its 32 CUs in turn (sp_panel.hip: XCD-local indices k, k + 32, ... share a CU).  A batch that 8 divides gives
  // XCD x the stars x S/8 ... (the panel launches' own deal: a star's tiles are written into the L2 that reads
  // them), chunk 0 of its stars FIRST -- the workgroups that factor pivot block 0, the launch's longest, start at
  // once, one per CU (S/8 <= 32).  Chunk-major from one star's chunks to the next put every chunk-0 workgroup on
  // XCD 0, two per CU: 33 us for the launch instead of 24.
  int s, chunk;
  if ((S & 7) == 0) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3, spx = S >> 3;
    s = x * spx + k % spx;
    chunk = k / spx;
  } else {
    s = blockIdx.x / nchunk;
    chunk = blockIdx.x % nchunk;
  }
  const int np = covpts + 4, tid = threadIdx.x;
  const sp_star st = stars[s];
  const int t0 = chunks.start[chunk], t1 = chunks.start[chunk + 1];
  // (an empty chunk has nothing to do -- unless it is the star's LAST one, which writes the riding rows below:
  //  SP_PLAN_TILES=1 can leave that chunk without tiles)
  if (t0 >= t1 && chunk != 0 && !(rid && chunk == nchunk - 1)) return;
  double *s_tab = lds;                       // 4 np
  double *s_red = s_tab + 4 * np;            // 8
  double *s_th = s_red + 8;                  // [Kp] the star's phases (zero beyond K)            (lds_phases)
  double *s_tt = s_th + Kp;                  // [Kp] its times (temporal kernels only)
  const int nobs = star_nobs(st, K);
  const double *th = plan.theta + (size_t)s * K, *tt = t + (size_t)s * K;
  // Prologue: TWO memory round trips from a cold start, not one per dependent access (a kernel that lives for
  // microseconds pays 1-2 us for each).  First what depends on the star's index alone -- its parameters, the first
  // 1 024 phases (and times), the plan's weights --, then what depends on the star's table index -- the table, the
  // flux mean --, every load of a group in flight before the first use.
  const int cl = tid & 15, ri = tid >> 4;
  int unsorted = 0;
  double pa[4], pb[4], pbp[4], w[2];
  const bool want_phases = lds_phases || TK == SP_TEMPORAL_MATERN32;
  const double *wb = plan.wbar + (size_t)s * np;
  if (want_phases) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = tid + 256 * c;
      pa[c] = (lds_phases && i < K) ? th[i] : 0.0;
      pb[c] = (TK != SP_TEMPORAL_NONE && i < K) ? tt[i] : 0.0;
      pbp[c] = (TK == SP_TEMPORAL_MATERN32 && i > 0 && i < K) ? tt[i - 1] : -INFINITY;
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) w[c] = tid + 256 * c < np ? wb[tid + 256 * c] : 0.0;
  // the star's table: {a0, a1} pairs, then {a2, a3} pairs (SplineGen), and yp . wbar on the way
  double dot = 0.0, fmean, var1;
  {
    const double *src = tab + (size_t)st.table * 5 * np;
    double *pt = (chunk == 0 && ptab) ? ptab + (size_t)s * 4 * np : nullptr;
    double y[2], a0[2], a1[2], a2[2], a3[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int e = tid + 256 * c;
      const bool ok = e < np;
      y[c] = ok ? src[e] : 0.0;
      a0[c] = ok ? src[np + e] : 0.0;
      a1[c] = ok ? src[2 * np + e] : 0.0;
      a2[c] = ok ? src[3 * np + e] : 0.0;
      a3[c] = ok ? src[4 * np + e] : 0.0;
    }
    fmean = meanvar[2 * st.table];
    var1 = meanvar[2 * st.table + 1];
    if (want_phases) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = tid + 256 * c;
        if (lds_phases && i < Kp) {
          s_th[i] = pa[c];
          if (TK != SP_TEMPORAL_NONE) s_tt[i] = pb[c];
        }
        if (TK == SP_TEMPORAL_MATERN32 && i > 0 && i < nobs && !(pb[c] >= pbp[c])) unsorted = 1;
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int e = tid + 256 * c;
      if (e < np) {
        *reinterpret_cast<dd2 *>(s_tab + 2 * e) = dd2{a0[c], a1[c]};
        *reinterpret_cast<dd2 *>(s_tab + 2 * np + 2 * e) = dd2{a2[c], a3[c]};
        if (pt) {
          *reinterpret_cast<dd2 *>(pt + 2 * e) = dd2{a0[c], a1[c]};
          *reinterpret_cast<dd2 *>(pt + 2 * np + 2 * e) = dd2{a2[c], a3[c]};
        }
        dot += y[c] * w[c];
      }
    }
    for (int e = tid + 512; e < np; e += 256) {    // (covpts > 508: calibrate's covpts = K - 1)
      const dd2 c01 = dd2{src[np + e], src[2 * np + e]}, c23 = dd2{src[3 * np + e], src[4 * np + e]};
      *reinterpret_cast<dd2 *>(s_tab + 2 * e) = c01;
      *reinterpret_cast<dd2 *>(s_tab + 2 * np + 2 * e) = c23;
      if (pt) {
        *reinterpret_cast<dd2 *>(pt + 2 * e) = c01;
        *reinterpret_cast<dd2 *>(pt + 2 * np + 2 * e) = c23;
      }
      dot += src[e] * wb[e];
    }
  }
  // (light curves beyond 1 024 cadences: the rest of the phases, four loads per thread in flight at a time)
  if (want_phases) {
    double a[4], b[4], bp[4];
    for (int base = 1024; base < Kp; base += 1024) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = base + tid + 256 * c;
        a[c] = (lds_phases && i < K) ? th[i] : 0.0;
        b[c] = (TK != SP_TEMPORAL_NONE && i < K) ? tt[i] : 0.0;
        bp[c] = (TK == SP_TEMPORAL_MATERN32 && i < nobs) ? tt[i - 1] : -INFINITY;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = base + tid + 256 * c;
        if (lds_phases && i < Kp) {
          s_th[i] = a[c];
          if (TK != SP_TEMPORAL_NONE) s_tt[i] = b[c];
        }
        if (TK == SP_TEMPORAL_MATERN32 && i < nobs && !(b[c] >= bp[c])) unsorted = 1;
      }
    }
  }
  // m = yp . wbar / K^2 over the workgroup: wavefront sums, then the four of them in order
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = dot;
  const bool in_order = TK == SP_TEMPORAL_MATERN32 ? !__syncthreads_or(unsorted) : (__syncthreads(), false);
  const double total = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  // (a single cadence: the covariance is the variance, flux.py:274-275)
  const double m = nobs == 1 ? var1 : total / ((double)nobs * (double)nobs);
  const Coef c = defer_coef(m, fmean, order, st.baseline_var);
  const double inv_c1 = 1.0 / c.c1;
  if (chunk == 0) {
    double *rs = rscal + (size_t)s * (SP_RSCAL_HEAD + M);
    for (int mm = tid; mm < M; mm += 256)
      rs[SP_RSCAL_HEAD + mm] = plan.sflux[(size_t)s * M + mm] - (double)nobs * st.baseline_mean;
    if (tid == 0) {
      const double delta = st.data_var / c.c1;
      rs[0] = (double)nobs * m;
      rs[1] = diag ? plan.sdv[s] / c.c1 : (double)nobs * delta;
      rs[2] = delta;
      coef[s] = c;
      const double *key = plan.key + 3 * (size_t)s;
      const bool stale = !(key[0] == st.period) || (TK != SP_TEMPORAL_NONE && !(key[1] == st.tau)) || key[2] != (double)nobs;
      if (info) info[s] = 0;
      if (status) status[s] = (c.z > zmax ? SP_STAR_ZMAX : 0u) | (stale ? SP_STAR_STALE_PLAN : 0u);
    }
  }
  if (rid && chunk == nchunk - 1) {
    // The rows below the cadences as the row tiles that hold them would carry them LEFT of the diagonal --
    // rid[s][m][col]: the residuals, the row of ones, the variances / c1 -- for the kernels that form those tiles at
    // their first touch (LazyCov.rid).  A few K numbers per star, by the star's last workgroup (not the one that
    // factors pivot block 0).
    // One more row behind them: dd[col] = D_col / c1, what the diagonal carries on top of the covariance -- for the
    // kernels that form DIAGONAL tiles (LazyCov.dlazy).
    const int nrid = M + (diag ? 2 : 1);
    double *dst = rid + (size_t)s * (nrid + 1) * K;
    for (int e = tid; e < (nrid + 1) * K; e += 256) {
      const int m = e / K, col = e - m * K;
      double val = 0.0;
      if (col < nobs) {
        if (m < M) val = flux[((size_t)s * M + m) * K + col] - st.baseline_mean;
        else if (m == M) val = 1.0;
        else {
#pragma clang fp contract(off)
          val = (diag ? diag[(size_t)s * K + col] : st.data_var) * inv_c1;     // (rows M + 1 with variances, and dd)
        }
      }
      dst[e] = val;
    }
  }
  if (t0 >= t1) return;
#ifdef P_PLAN_NOTILES
  if (chunk != 0) return;                                      // (timing probe: only pivot block 0's workgroups work)
#endif
  // strip-major tile order: strip tj holds the tiles ti = tj .. ntr - 1; of those only the WRITTEN ones are visited
  int tj = 0, ti;
  {
    int rem = t0;
    while (rem >= ntr - tj) {
      rem -= ntr - tj;
      ++tj;
    }
    ti = tj + rem;
  }
  auto phase_of = [&](int i) { return lds_phases ? s_th[i] : (i < K ? th[i] : 0.0); };
  auto time_of = [&](int i) { return TK == SP_TEMPORAL_NONE ? 0.0 : (lds_phases ? s_tt[i] : (i < K ? tt[i] : 0.0)); };
  double thj[4], tmj[4], thi[4], tmi[4];
  const double cm = TK == SP_TEMPORAL_MATERN32 ? 1.7320508075688772 / st.tau : 0.0;
  double bref = 0.0, fc[4] = {1.0, 1.0, 1.0, 1.0};
  bool sep_strip = false;
  auto load_cols = [&](int tjj) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 64 * tjj + cl + 16 * e;
      thj[e] = phase_of(j);
      tmj[e] = time_of(j);
    }
    // (the separated Matern-3/2 factor of a strip: see assemble_sums_kernel)
    if (TK == SP_TEMPORAL_MATERN32) {
      const int j0s = 64 * tjj, j1s = (j0s + 63 < nobs ? j0s + 63 : nobs - 1);
      bref = time_of(j0s);
      sep_strip = in_order && j1s >= j0s && cm * (time_of(j1s) - bref) < 600.0 && cm > 0.0;
      if (sep_strip) {
#pragma unroll
        for (int e = 0; e < 4; ++e) fc[e] = exp(cm * (tmj[e] - bref));
      }
    }
  };
  load_cols(tj);
  SplineGen g{s_tab, 2 * np, 6.283185307179586 / covpts, 1.0 / (6.283185307179586 / covpts), covpts};
  double *ob = out + (size_t)s * strideo;
  int tile = t0;
  while (tile < t1) {
    const int i0 = 64 * ti, j0 = 64 * tj;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int i = i0 + ri + 16 * pass;
      thi[pass] = phase_of(i);
      tmi[pass] = time_of(i);
    }
    // What the tile needs from memory is requested BEFORE the evaluation, unconditionally, from addresses that are
    // always valid: a load inside a branch is waited for where the branch ends (one round trip per residual entry:
    // a last-row tile took 7.6 us, 3.8 times a plain one).  below: the tile holds rows beyond the cadences --
    // residuals, the row of ones, the variances' row, identity padding.
    const bool below = i0 + 64 > K;                  // (uniform)
    double dvp[4] = {st.data_var, st.data_var, st.data_var, st.data_var}, fl[16];
    if (diag && ti == tj) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int i = i0 + ri + 16 * pass;
        dvp[pass] = diag[(size_t)s * K + (i < K ? i : K - 1)];
      }
    }
    if (below) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int mrow = i0 + ri + 16 * pass - K;     // 0 .. M - 1: residuals; M: ones; M + 1: variances
        const double *src = (mrow >= 0 && mrow < M) ? flux + ((size_t)s * M + mrow) * K
                                                    : ((diag && mrow == M + 1) ? diag + (size_t)s * K : th);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = j0 + cl + 16 * e;
          fl[4 * pass + e] = src[j < K ? j : K - 1];
        }
      }
    }
    double v[16];
#ifndef SP_PLAN_BATCH
#define SP_PLAN_BATCH 16     // entries per SplineGen::many batch (4, 8 or 16)
#endif
#pragma unroll
    for (int p0 = 0; p0 < 4; p0 += SP_PLAN_BATCH / 4) {
      double a[SP_PLAN_BATCH], b[SP_PLAN_BATCH], o[SP_PLAN_BATCH];
#pragma unroll
      for (int k = 0; k < SP_PLAN_BATCH; ++k) {
        a[k] = thi[p0 + (k >> 2)];
        b[k] = thj[k & 3];
      }
#ifdef P_PLAN_NOEVAL
#pragma unroll
      for (int k = 0; k < SP_PLAN_BATCH; ++k) o[k] = a[k] - b[k];        // (timing probe: results are garbage)
#else
      g.many<SP_PLAN_BATCH>(a, b, o);
#endif
#pragma unroll
      for (int k = 0; k < SP_PLAN_BATCH; ++k) v[4 * p0 + k] = o[k];
    }
    if (nobs == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = var1;
    }
    if (TK == SP_TEMPORAL_MATERN32 && sep_strip && ti > tj && ti < ntr - 1) {
      double er[4];
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) er[pass] = exp(-(cm * (tmi[pass] - bref)));
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double x = cm * (tmi[pass] - tmj[e]);
          v[4 * pass + e] *= (1.0 + x) * (er[pass] * fc[e]);
        }
    } else if (TK != SP_TEMPORAL_NONE) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * pass + e] *= temporal_factor(TK, tmi[pass], tmj[e], st.tau);
    }
    double w[16];
    if (i0 + 64 <= nobs && j0 + 64 <= nobs) {
      // a tile of valid cadences only: no masks
#pragma unroll
      for (int k = 0; k < 16; ++k) w[k] = v[k];
      if (ti == tj && ri == cl) {
        // the diagonal entries of the tile are this thread's (pass, pass): B = Sigma + D / c1
        // (product rounded, then added -- no contraction: the kernels that form a diagonal tile at its first touch add
        //  the same rounded D / c1, LazyCov.rid's last row, and a tile must carry the same bits either way)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
#pragma clang fp contract(off)
          const double dc = dvp[pass] * inv_c1;
          w[5 * pass] = w[5 * pass] + dc;
        }
      }
    } else {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int i = i0 + ri + 16 * pass;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = j0 + cl + 16 * e;
          double val = 0.0;
          if (i < nobs && j < nobs) {
#pragma clang fp contract(off)
            const double dc = dvp[pass] * inv_c1;
            val = v[4 * pass + e];
            if (i == j) val = val + dc;
          } else if (below && i >= K && i < K + M && j < nobs) {
            val = fl[4 * pass + e] - st.baseline_mean;      // (the GP mean of the normalised process is 0)
          } else if (i == K + M && j < nobs) {
            val = 1.0;                                      // L^-1 1 rides here
          } else if (below && diag && i == K + M + 1 && j < nobs) {
            val = fl[4 * pass + e] * inv_c1;                // L^-1 d
          } else if (i == j) {
            val = 1.0;
          }
          w[4 * pass + e] = val;
        }
      }
    }
    if (fuse0 && tile == 0) {
      // Pivot block 0, complete and in registers: factored HERE (the star's first workgroup holds nothing but this
      // tile: plan_chunks), beside the launch's other tiles -- rounds 1-4 spent a launch of 64 workgroups on it
      // behind the assembly (18 us one step at a time).  L_d to the system, L_d^-1 to the star's image slot 0.
      __syncthreads();                       // (the table and the phases in LDS have been read by every wavefront)
      double *sD = lds;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int li = ri + 16 * pass, lj = cl + 16 * e;
          sD[li * BLD + lj] = lj > li ? 0.0 : w[4 * pass + e];
        }
      __builtin_amdgcn_s_setprio(3);
      panel_diag_core(ob, ldo, 64, img + (size_t)s * lts + sp_img_off(0), info ? info + s : nullptr, lds, tid);
      return;
    }
#ifdef P_PLAN_NOSTORE
    if (w[3] == 1.2345e300)                                    // (timing probe: nothing is stored)
#endif
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      double *dst = ob + (size_t)(i0 + ri + 16 * pass) * ldo + j0 + cl;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[16 * e] = w[4 * pass + e];
    }
    // the next written tile of the strip, or the next strip's diagonal tile
    int nti = ti + 1;
    if (nti < nfull && tj >= ncolw) nti = nfull;
    if (nti >= ntr) {
      tile += ntr - ti;
      ++tj;
      ti = tj;
      // (diagonal tiles from dfrom on are formed by the first trailing update: strips that hold nothing else are skipped)
      while (tj >= dfrom && tj >= ncolw && nfull >= ntr && tj < ntr) {
        tile += ntr - tj;
        ++tj;
        ti = tj;
      }
      if (tile < t1) load_cols(tj);
    } else {
      tile += nti - ti;
      ti = nti;
    }
  }
}

}  // namespace

static size_t attr_lds_limit = 150 * 1024;

template <typename F>
static void allow_big_lds(F f) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(f),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)attr_lds_limit);
}

// Chunks of the planned assembly: the WRITTEN tiles of a star (strip-major; written(a, b): a == b,
// a >= nfull or b < ncolw) cut into nchunk runs of equal cost.  start[c] = index of chunk c's first tile in the numbering of ALL
// lower tiles.  No sums are taken in this kernel: the cut has no influence on any bit of the result.
// fuse0: chunk 0 is tile (0, 0) alone (its workgroup goes on to factor pivot block 0).
static AsmChunks plan_chunks(int ntr, int nfull, int ncolw, int nchunk, int fuse0, int dfrom) {
  static thread_local int have_ntr = -1, have_nfull = -1, have_ncolw = -1, have_nchunk = -1, have_fuse0 = -1, have_dfrom = -1;
  static thread_local AsmChunks have;
  if (ntr == have_ntr && nfull == have_nfull && ncolw == have_ncolw && nchunk == have_nchunk && fuse0 == have_fuse0 &&
      dfrom == have_dfrom)
    return have;
  const auto written = [&](int a, int b) { return (a == b && a < dfrom) || a >= nfull || b < ncolw; };
  const auto weight = [&](int a, int b) { return a == ntr - 1 ? 20 : (a == b ? 11 : 10); };
  const bool own0 = fuse0 && nchunk >= 2;       // tile (0, 0) in a chunk of its own, the others over nchunk - 1
  const int nc = own0 ? nchunk - 1 : nchunk;
  long total = 0;
  for (int b = 0; b < ntr; ++b)
    for (int a = b; a < ntr; ++a)
      if (written(a, b) && !(own0 && a == 0)) total += weight(a, b);
  AsmChunks c;
  int chunk = 0, tile = 0;
  long cum = 0;
  if (own0) c.start[0] = 0;
  for (int b = 0; b < ntr; ++b)
    for (int a = b; a < ntr; ++a, ++tile) {
      if (!written(a, b) || (own0 && a == 0)) continue;
      while (chunk <= nc && cum >= (long)chunk * total / nc) c.start[(own0 ? 1 : 0) + chunk++] = (unsigned short)tile;
      cum += weight(a, b);
    }
  while (chunk <= nc) c.start[(own0 ? 1 : 0) + chunk++] = (unsigned short)tile;
  have = c;
  have_ntr = ntr;
  have_nfull = nfull;
  have_ncolw = ncolw;
  have_nchunk = nchunk;
  have_fuse0 = fuse0;
  have_dfrom = dfrom;
  return c;
}

// LDS of the planned assembly: the star's table, a reduction's scratch, its phases (and times)
static size_t assemble_planned_lds(int Kp, int covpts, int temporal, int lds_phases, int fuse0) {
  const size_t own = 4 * (size_t)(covpts + 4) + 8 + (lds_phases ? (size_t)Kp * (temporal == SP_TEMPORAL_NONE ? 1 : 2) : 0);
  // (the workgroup that factors pivot block 0 does it in this LDS)
  return sizeof(double) * ((fuse0 && own < SP_DIAG_LDS_DOUBLES) ? (size_t)SP_DIAG_LDS_DOUBLES : own);
}

int sp_launch_assemble_planned(int S, int K, int M, int Kp, const PlanDev &plan, const double *t,
                               const sp_star *stars, int covpts, const double *tab, const double *meanvar,
                               int temporal, const double *flux, const double *diag, double *sys, int nfull,
                               int ncolw, int order, double zmax, void *coef, double *rscal, double *ptab, int32_t *info,
                               uint32_t *status, hipStream_t st, double *img, long lts, int fuse0, double *rid, int dfrom) {
  const int ntr = Kp / 64, ntiles = ntr * (ntr + 1) / 2;
  if (ntiles > 65535 || !coef || !rscal || (fuse0 && (!img || K < 64))) return SP_ERR_INVALID;
  int lds_phases = 1;
  size_t lds = assemble_planned_lds(Kp, covpts, temporal, 1, fuse0);
  if (lds > SP_ASM_LDS_MAX) {
    lds_phases = 0;
    lds = assemble_planned_lds(Kp, covpts, temporal, 0, fuse0);
    if (lds > attr_lds_limit) return SP_ERR_INVALID;
  }
  int nwritten = 0;
  for (int b = 0; b < ntr; ++b)
    for (int a = b; a < ntr; ++a) nwritten += ((a == b && a < dfrom) || a >= nfull || b < ncolw) ? 1 : 0;
  // Written tiles per workgroup: one round of two workgroups per CU where that leaves a workgroup at least three
  // tiles (cfg3: 16 diagonal tiles x 64 stars: 3 per workgroup, 0.755 ms per step one at a time against 0.762 with
  // 2 or 4), never more than 24 (cfg5's shape: 1 128 tiles per star, three
  // rounds -- a workgroup's prologue copies the star's phases and times, 48 KB there).  No sums are taken here: the
  // cut changes no bit of the result.  SP_PLAN_TILES overrides.
  static const int per_env = [] {
    const char *e = getenv("SP_PLAN_TILES");
    return e ? atoi(e) : 0;
  }();
  int per = per_env > 0 ? per_env : (int)(((long)nwritten * S + 511) / 512);
  if (per_env <= 0) per = per < 3 ? 3 : (per > 24 ? 24 : per);
  int nchunk = (nwritten + per - 1) / per;
  if (nchunk > SP_ASM_MAX_CHUNKS) nchunk = SP_ASM_MAX_CHUNKS;
  if (nchunk < 1) nchunk = 1;
  if (fuse0 && nchunk < 2) nchunk = 2;
  // (dfrom < ntr: the diagonal tiles from dfrom on are left to the first trailing update -- only with every other
  //  tile of their strips left to its first touch too)
  if (dfrom < ntr && (nfull < ntr || !rid)) return SP_ERR_INVALID;
  const AsmChunks chunks = plan_chunks(ntr, nfull, ncolw, nchunk, fuse0, dfrom);
  dim3 grid((unsigned)(nchunk * S));
#define SP_ASMP(TK)                                                                                       \
  do {                                                                                                    \
    allow_big_lds(assemble_planned_kernel<TK>);                                                           \
    hipLaunchKernelGGL((assemble_planned_kernel<TK>), grid, dim3(256), lds, st, K, M, Kp, plan, t, stars, \
                       covpts, tab, meanvar, flux, diag, sys, (long)Kp, (long)Kp * Kp, ntr, nfull, ncolw, order, \
                       zmax, (Coef *)coef, rscal, ptab, info, status, lds_phases, img, lts, fuse0, S, nchunk, rid, dfrom, chunks); \
  } while (0)
  if (temporal == SP_TEMPORAL_NONE) SP_ASMP(SP_TEMPORAL_NONE);
  else if (temporal == SP_TEMPORAL_MATERN32) SP_ASMP(SP_TEMPORAL_MATERN32);
  else if (temporal == SP_TEMPORAL_EXPSQUARED) SP_ASMP(SP_TEMPORAL_EXPSQUARED);
  else return SP_ERR_INVALID;
#undef SP_ASMP
  SP_LAUNCH_CHECK();
  return SP_OK;
}

