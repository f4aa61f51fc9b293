// Short light curves (K <= 128 cadences) in ONE kernel, one workgroup per star (round 6).
//
// The blocked path spends a launch per 64-column panel and a workgroup per (star, row tile) and launch: at K = 128 a
// step of 1 688 stars is ~10 000 workgroups, each at least one 12 us diagonal-block chain long, over four launches,
// with the 192-row padded system (2.25 x the entries) written to and read from HBM in between -- 0.31 ms per step,
// 0.07 of the fp64 peak (bench.py's K sweep; the reference's own benchmark sweeps K, joss/figures/speed.py:22-37).
// Here a star's whole evaluation -- the planned step's assembly (sp_planasm.hip: phases and weights from the data plan,
// the normalisation's coefficients from yp . wbar), the factorisation, the riding rows' solves, the reduction of
// sp_reduce.h -- happens in the LDS and registers of one workgroup; nothing of the system ever reaches memory:
//
//   tile (0, 0) -> LDS, factored in place by diag_block (sp_diag.h) -- WITHOUT its inverse for K <= 64;
//   K > 64: tile (1, 0) is assembled after pivot block 0, straight into the MFMA A fragments of the wavefront's own
//           sixteen rows; X = T10 L00^-T on the matrix cores (the one block whose inverse is used) takes the pivot
//           block's place in the LDS once L00 is dead; tile (1, 1) is assembled into accumulators, -= X X^T there, stored
//           over X and factored without an inverse;
//   the riding rows [r_0 .. r_{M-1}, 1, (d)] in LDS: against a block with an inverse a product
//           with L^-T, against one without a substitution (ride_subst: one wavefront per row, lane = cadence);
//   lnlike_reduce_src on the LDS copies, in the first wavefront alone.
//
// ONE 64 x 64 tile of LDS per workgroup: 40 KB for K <= 64 (the spline table lies in the tile's place until the
// assembly is through: four workgroups a CU), 52 KB for K > 64 (a table region of its own, since two tiles are assembled
// after the first factorisation: three).  What decided the form was an instruction budget by phase (tools/small_k_pmc.sh:
// builds that return at a phase boundary, -DSMK_STOP=k, and SQ_INSTS_VALU differences): half of a K = 64 star's vector
// instructions formed an inverse nobody multiplied with, a quarter were the reduction run by four wavefronts.  The same
// values as the blocked planned step to rounding (tests/test_gpu_small.py: 1e-10 against it, 1e-8 against the oracle).
#include "sp_internal.h"
#include "sp_cov.h"
#include "sp_asm.h"
#include "sp_tile.h"
#include "sp_reduce.h"

#ifdef SP_SMALL_TRACE
// (debug builds, tools/ab_build.sh trace -DSP_SMALL_TRACE: wall-clock stamps of the first 64 workgroups' phases)
__device__ long long g_small_trace[64 * 16];
__device__ long long g_small_diag[64];      // diag_block's own stamps (shader clock) of workgroup 0's first pivot block
#define SMK_STAMP(k)                                                          \
  do {                                                                        \
    if (blockIdx.x < 64 && threadIdx.x == 0) g_small_trace[blockIdx.x * 16 + (k)] = wall_clock64(); \
  } while (0)
extern "C" int sp_debug_small_trace(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_trace), sizeof(long long) * 64 * 16) == hipSuccess ? SP_OK : SP_ERR_HIP;
}
extern "C" int sp_debug_small_diag(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_diag), sizeof(long long) * 64) == hipSuccess ? SP_OK : SP_ERR_HIP;
}
#define SMK_DIAGDBG (blockIdx.x == 0 ? g_small_diag : nullptr)
#else
#define SMK_DIAGDBG nullptr
#endif
#if defined(SMK_STOP)
// (debug builds, -DSMK_STOP=k: the workgroup returns at phase boundary k -- instruction counts of the phases from PMC
//  differences, tools/small_k_pmc.sh; results are garbage)
#define SMK_STAMP(k)        \
  do {                      \
    if ((k) == SMK_STOP) return; \
  } while (0)
#elif !defined(SP_SMALL_TRACE)
#define SMK_STAMP(k)
#endif

namespace {

typedef SpCoef Coef;
constexpr int SMK_MAXR = 4;          // riding rows: M + 1 (scalar variance) or M + 2 (per-cadence variances)

struct SmallSrc {                    // lnlike_reduce_src's view of the factored star
  const double *dg;                  // [64 NB] L_ii
  const double *rows;                // [nr][64 NB] riding rows
  int ldr;
  __device__ __forceinline__ double diag(int i) const { return dg[i]; }
  __device__ __forceinline__ double row(int m, int k) const { return rows[m * ldr + k]; }
};

// (wavefronts per SIMD the register budget is held to: three workgroups a CU for K > 64, four for K <= 64 -- what their LDS allows)
template <int NB, int TK>
__global__ __launch_bounds__(256, NB == 2 ? 3 : 4) void small_lnlike_kernel(
    int K, int M, PlanDev plan, const double *__restrict__ t, const sp_star *__restrict__ stars, int covpts,
    const double *__restrict__ tab, const double *__restrict__ meanvar, const double *__restrict__ flux,
    const double *__restrict__ diag, int order, double zmax, double *__restrict__ lnlike,
    uint32_t *__restrict__ status_out, int regionB) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int KC = 64 * NB;
  const int s = blockIdx.x, tid = threadIdx.x, np = covpts + 4;
  const int nr = M + (diag ? 2 : 1);
  double *sD = lds;                              // 64 BLD  | sRd 64 | leaf 256           (SP_DIAG_LDS_DOUBLES)
  double *sRd = sD + 64 * BLD;
  // region B: the spline table (NB == 2: tiles (1, 0) and (1, 1) are assembled after pivot block 0).  NB == 1: no region
  // of its own -- the table lies in the pivot block's place until the assembly is through (the tile is assembled in
  // registers): 40 KB, four workgroups a CU
  double *sB = NB == 2 ? lds + SP_DIAG_LDS_DOUBLES : lds;
  double *sTh = lds + SP_DIAG_LDS_DOUBLES + regionB;   // [KC] phases
  double *sTt = sTh + KC;                        // [KC] times
  double *sR = sTt + KC;                         // [nr][KC] riding rows
  double *sDg = sR + nr * KC;                    // [KC] L_ii
  double *sRed = sDg + KC;                       // 48
  double *sRs = sRed + 48;                       // [SP_RSCAL_HEAD + M] the reduction's scalars
  Coef *sCoef = reinterpret_cast<Coef *>(sRs + SP_RSCAL_HEAD + M + ((SP_RSCAL_HEAD + M) & 1));   // 8 doubles
  uint32_t *sStat = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(sCoef) + 8);
  SMK_STAMP(0);
  const sp_star st = stars[s];
  const int nobs = star_nobs(st, K);
  const double *th = plan.theta + (size_t)s * K, *tt = t + (size_t)s * K;
  // ---- prologue: phases, table, m = yp . wbar / nobs^2 (the planned assembly's, sp_planasm.hip) --------------------
  if (tid < KC) {
    sTh[tid] = tid < K ? th[tid] : 0.0;
    sTt[tid] = (TK != SP_TEMPORAL_NONE && tid < K) ? tt[tid] : 0.0;
  }
  double dot = 0.0;
  {
    const double *src = tab + (size_t)st.table * 5 * np, *wb = plan.wbar + (size_t)s * np;
    for (int e = tid; e < np; e += 256) {
      *reinterpret_cast<dd2 *>(sB + 2 * e) = dd2{src[np + e], src[2 * np + e]};
      *reinterpret_cast<dd2 *>(sB + 2 * np + 2 * e) = dd2{src[3 * np + e], src[4 * np + e]};
      dot += src[e] * wb[e];
    }
  }
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
  if ((tid & 63) == 0) sRed[tid >> 6] = dot;
  __syncthreads();
  const double total = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
  const double fmean = meanvar[2 * st.table], var1 = meanvar[2 * st.table + 1];
  const double m = nobs == 1 ? var1 : total / ((double)nobs * (double)nobs);
  const Coef c = defer_coef(m, fmean, order, st.baseline_var);
  const double inv_c1 = 1.0 / c.c1;
  if (tid == 0) {
    const double delta = st.data_var / c.c1;
    sRs[0] = (double)nobs * m;
    sRs[1] = diag ? plan.sdv[s] / c.c1 : (double)nobs * delta;
    sRs[2] = delta;
    *sCoef = c;
    const double *key = plan.key + 3 * (size_t)s;
    const bool stale = !(key[0] == st.period) || (TK != SP_TEMPORAL_NONE && !(key[1] == st.tau)) || key[2] != (double)nobs;
    *sStat = (c.z > zmax ? SP_STAR_ZMAX : 0u) | (stale ? SP_STAR_STALE_PLAN : 0u);
  }
  for (int mm = tid; mm < M; mm += 256) sRs[SP_RSCAL_HEAD + mm] = plan.sflux[(size_t)s * M + mm] - (double)nobs * st.baseline_mean;
  // riding rows: residuals, ones, variances / c1 (zero beyond the valid cadences)
  for (int e = tid; e < nr * KC; e += 256) {
    const int mr = e / KC, col = e - mr * KC;
    double v = 0.0;
    if (col < nobs) {
      if (mr < M) v = flux[((size_t)s * M + mr) * K + col] - st.baseline_mean;
      else if (mr == M) v = 1.0;
      else v = diag[(size_t)s * K + col] * inv_c1;
    }
    sR[e] = v;
  }
  SMK_STAMP(1);
  // ---- assembly: sixteen entries of a thread, where a layout wants them ------------------------------------------------
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  SplineGen g{sB, 2 * np, 6.283185307179586 / covpts, 1.0 / (6.283185307179586 / covpts), covpts};
  // (entries in batches of BT through SplineGen::many, one batch after the other: sixteen at once keep ~90 doubles of
  //  gather temporaries alive)
  constexpr int BT = NB == 2 ? 4 : 8;
  // entry e of the thread is (rowf(e), colf(e)) of the star's matrix; skipf(e0): the batch from e0 is not wanted (zeros)
  auto entries = [&](auto rowf, auto colf, auto skipf, double (&w)[16]) {
#pragma unroll
    for (int e0 = 0; e0 < 16; e0 += BT) {
      if (skipf(e0)) {                    // (wavefront-uniform)
#pragma unroll
        for (int q = 0; q < BT; ++q) w[e0 + q] = 0.0;
        continue;
      }
      double a[BT], b[BT], o[BT];
#pragma unroll
      for (int q = 0; q < BT; ++q) {
        a[q] = sTh[rowf(e0 + q)];
        b[q] = sTh[colf(e0 + q)];
      }
      g.many<BT>(a, b, o);
#pragma unroll
      for (int q = 0; q < BT; ++q) {
        const int i = rowf(e0 + q), j = colf(e0 + q);
        double v = nobs == 1 ? var1 : o[q];
        if (TK != SP_TEMPORAL_NONE) v *= temporal_factor(TK, sTt[i], sTt[j], st.tau);
        if (i < nobs && j < nobs) {
          if (i == j) v += (diag ? diag[(size_t)s * K + i] : st.data_var) * inv_c1;
        } else {
          v = i == j ? 1.0 : 0.0;
        }
        w[e0 + q] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // a diagonal tile in the MFMA accumulator layout (e = 4 nb + r: row 16 wave + fg + 4 r, column 16 nb + fr): the
  // wavefront's row strip is block row `wave`, the 16 x 16 blocks right of the diagonal are never read
  auto diag_tile = [&](int ti, double (&w)[16]) {
    entries([&](int e) { return 64 * ti + 16 * wave + fg + 4 * (e & 3); }, [&](int e) { return 64 * ti + 16 * (e >> 2) + fr; },
            [&](int e0) { return (e0 >> 2) > wave; }, w);
  };
  auto store_lower = [&](const double (&w)[16]) {     // ... to the pivot block's place, zeros above the diagonal
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = 16 * wave + fg + 4 * (e & 3), col = 16 * (e >> 2) + fr;
      sD[row * BLD + col] = col > row ? 0.0 : w[e];
    }
  };
  {
    double w00[16];
    diag_tile(0, w00);
    if (NB == 1) __syncthreads();        // (the table, in the block's place, has been read by everybody)
    store_lower(w00);
  }
  __syncthreads();
  SMK_STAMP(2);
  // ---- pivot block 0 ------------------------------------------------------------------------------------------------
  // (wavefront w of diag_block owns block column w: workgroups that share a CU -- 256 apart in the grid, the hardware
  //  deals consecutive ones to the XCDs and their CUs in turn -- start in step, and with the same roles their leaf
  //  chains would queue on ONE SIMD while three idle: the roles are rotated from workgroup to workgroup)
  const int tid_rot = (tid + 64 * ((blockIdx.x >> 8) & 3)) & 255;
  // (K <= 64: nobody multiplies with the block's inverse -- the riding rows are substituted, ride_subst)
  int notpd = diag_block<NB == 2>(sD, sRd, tid_rot, SMK_DIAGDBG);  // (ends behind a barrier)
  SMK_STAMP(3);
  if (tid < 64) sDg[tid] = sD[tid * BLD + tid];
  // The riding rows of a block that has no inverse, y = L^-1 r by substitution: wavefront mr takes row mr, lane i holds
  // s_i = r_i / L_ii; step c hands y_c = s_c to every lane (v_readlane) and s_i -= (L_ic / L_ii) y_c.  The block's
  // diagonal is ZEROED first (it is in sDg) and its upper triangle is zero (diag_block<false>), so lanes i <= c are
  // left alone and lane i ends with y_i: four vector instructions per step and row, no select.
  auto ride_subst = [&](int c0) {
    if (tid < 64) sD[tid * BLD + tid] = 0.0;
    __syncthreads();
    if (wave < nr) {
      const double rdi = sRd[lane];
      const double *Lrow = sD + lane * BLD;
      double sv = sR[wave * KC + c0 + lane] * rdi;
#pragma unroll
      for (int c = 0; c < 64; c += 2) {
        const d2v l = *reinterpret_cast<const d2v *>(Lrow + c);
        sv = fma(-(l.x * rdi), read_lane(sv, c), sv);
        sv = fma(-(l.y * rdi), read_lane(sv, c + 1), sv);
      }
      sR[wave * KC + c0 + lane] = sv;
    }
  };
  if (NB == 1) {
    ride_subst(0);
    SMK_STAMP(8);
  } else {
    // Linv[n][k] of the block in sD / sRd (sp_diag.h: L^-T above the diagonal, the reciprocal diagonal apart)
    auto linv = [&](int n, int k) { return k < n ? sD[k * BLD + n] : (k == n ? sRd[k] : 0.0); };
    // R[:, .. 63] <- R[:, .. 63] L00^-T (value kept, stored behind the barrier)
    double rnew = 0.0;
    if (tid < nr * 64) {
      const int mr = tid >> 6, n = tid & 63;
      const double *row = sR + mr * KC;
      rnew = row[n] * sRd[n];
#pragma unroll 4
      for (int k = 0; k < n; ++k) rnew = fma(row[k], sD[k * BLD + n], rnew);
    }
    SMK_STAMP(8);
    // X = T10 L00^-T: tile (1, 0) is assembled NOW and straight into the A fragments of the wavefront's own sixteen
    // rows (entry ks: row 64 + 16 wave + fr, column 4 ks + fg) -- it never lies in the LDS, and nothing of it is held
    // across the pivot block: one tile of LDS per workgroup instead of two, three workgroups a CU.  Every k step's
    // fragment feeds the column blocks it reaches (L00^-T is triangular), four steps per fence so that the B
    // fragments' reads are not all gathered up front.
    d4 x[4];
    {
      double af[16];
      entries([&](int e) { return 64 + 16 * wave + fr; }, [&](int e) { return 4 * e + fg; }, [](int) { return false; }, af);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) x[nb] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int nb = ks >> 2; nb < 4; ++nb)
          x[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], linv(16 * nb + fr, 4 * ks + fg), x[nb], 0, 0, 0);
        if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    SMK_STAMP(9);
    __syncthreads();                     // (every reader of R's first half and of L00 / L00^-T is through)
    if (tid < nr * 64) sR[(tid >> 6) * KC + (tid & 63)] = rnew;
    // X takes the pivot block's place
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) sD[(16 * wave + fg + 4 * r) * BLD + 16 * nb + fr] = x[nb][r];
    __syncthreads();                     // X and the solved first half of R are in place
    SMK_STAMP(10);
    // R[:, 64 ..] -= R[:, .. 63] X^T
    if (tid < nr * 64) {
      const int mr = tid >> 6, cc = tid & 63;
      const double *r0 = sR + mr * KC, *xr = sD + cc * BLD;
      double acc = r0[64 + cc];
#pragma unroll 4
      for (int n = 0; n < 64; ++n) acc = fma(-r0[n], xr[n], acc);
      sR[mr * KC + 64 + cc] = acc;
    }
    SMK_STAMP(11);
    // T11 (assembled now, in accumulators) -= X X^T: the blocks on or below the diagonal
    double w11[16];
    diag_tile(1, w11);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      if (nb > wave) continue;
      d4 acc = d4{w11[4 * nb], w11[4 * nb + 1], w11[4 * nb + 2], w11[4 * nb + 3]};
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const int k = 4 * ks + fg;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sD[(16 * wave + fr) * BLD + k], sD[(16 * nb + fr) * BLD + k], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) w11[4 * nb + r] = acc[r];
    }
    SMK_STAMP(12);
    __syncthreads();                     // (every reader of X is through)
    store_lower(w11);
    __syncthreads();                     // (diag_block wants a barrier behind the block's stores)
    SMK_STAMP(4);
    notpd |= diag_block<false>(sD, sRd, tid_rot);
    SMK_STAMP(5);
    if (tid < 64) sDg[64 + tid] = sD[tid * BLD + tid];
    ride_subst(64);
  }
  notpd = __syncthreads_or(notpd);
  SMK_STAMP(6);
  // ---- reduction ------------------------------------------------------------------------------------------------------
  // (the first wavefront alone: at most two entries a lane)
  if (tid >= 64) return;
  lnlike_reduce_src<false, SmallSrc, 64, NB>(SmallSrc{sDg, sR, KC}, K, M, nullptr, lnlike + s, sStat,
                                             status_out ? status_out + s : nullptr, stars + s, sCoef, sRs, diag ? 1 : 0,
                                             sRed, tid, notpd);
  SMK_STAMP(7);
}

}  // namespace

// can the planned step of this shape run in the small-K kernel?  (sp_lnlike_ensemble_planned asks)
bool sp_small_k_serves(int K, int M, int covpts, bool has_diag) {
  const int nr = M + (has_diag ? 2 : 1);
  if (K < 2 || K > 128 || nr > SMK_MAXR) return false;
  const int np = covpts + 4;
  // the table: in the pivot block's place (K <= 64) or in a region of its own, no larger (K > 64: 54 KB, three
  // workgroups a CU, up to covpts = 390 with two riding rows)
  return 4 * np <= 64 * BLD;
}

int sp_launch_small_lnlike(int S, int K, int M, const PlanDev &plan, const double *t, const sp_star *stars, int covpts,
                           const double *tab, const double *meanvar, int temporal, const double *flux, const double *diag,
                           int order, double zmax, double *lnlike, uint32_t *status_out, hipStream_t st) {
  if (!sp_small_k_serves(K, M, covpts, diag != nullptr)) return SP_ERR_INVALID;
  const int NB = K > 64 ? 2 : 1, KC = 64 * NB, np = covpts + 4, nr = M + (diag ? 2 : 1);
  // (K <= 64: the table lies in the pivot block's place -- no region of its own)
  const int regionB = NB == 2 ? 4 * np : 0;
  const size_t doubles = (size_t)SP_DIAG_LDS_DOUBLES + regionB + 2 * KC + (size_t)nr * KC + KC + 48 + SP_RSCAL_HEAD + M + 1 + 8 + 2;
  const size_t lds = sizeof(double) * doubles;
#define SP_SMALL(NBV, TKV)                                                                                        \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(small_lnlike_kernel<NBV, TKV>),                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                            \
    hipLaunchKernelGGL((small_lnlike_kernel<NBV, TKV>), dim3(S), dim3(256), lds, st, K, M, plan, t, stars, covpts, tab, \
                       meanvar, flux, diag, order, zmax, lnlike, status_out, regionB);                            \
  } while (0)
#define SP_SMALL_TK(NBV)                                                                \
  do {                                                                                  \
    if (temporal == SP_TEMPORAL_NONE) SP_SMALL(NBV, SP_TEMPORAL_NONE);                  \
    else if (temporal == SP_TEMPORAL_MATERN32) SP_SMALL(NBV, SP_TEMPORAL_MATERN32);     \
    else if (temporal == SP_TEMPORAL_EXPSQUARED) SP_SMALL(NBV, SP_TEMPORAL_EXPSQUARED); \
    else return SP_ERR_INVALID;                                                         \
  } while (0)
  if (NB == 2) SP_SMALL_TK(2);
  else SP_SMALL_TK(1);
#undef SP_SMALL_TK
#undef SP_SMALL
  SP_LAUNCH_CHECK();
  return SP_OK;
}
