// Covariance assembly: HBM-bound kernels.
//
//   theta_kernel      theta = 2 pi mod(t / p, 1)                  flux.py:262
//   rowsum_kernel     row sums of the raw covariance (for _normalize)
//   norm_coef_kernel  m, q, z, alpha(z), beta(z)                  sp.py:705-727
//   assemble_kernel   final matrix: normalisation fix-up + data variance +
//                     baseline variance (sp.py:1135-1151), written once.
//
// The marginal-path covariance is a function of the phase lag only,
// cov_ij = spline(|theta_i - theta_j|) [* temporal(|t_i - t_j|)], so it is
// never stored in raw form: the row sums re-evaluate the spline (pure compute,
// no HBM traffic) and the assembly writes the normalised matrix exactly once
// (lower triangle only when it feeds the Cholesky).  The conditional path reads
// its raw covariance from the GEMM output instead (template parameter).
//
// Compiled with -ffp-contract=off: the int64 interpolation index
// floor(x / dx) must agree bit for bit with the reference (flux.py:262-265).
#include <cstdlib>

#include "sp_internal.h"
#include "sp_cov.h"
#include "sp_asm.h"

namespace {

typedef SpCoef Coef;   // per-star normalisation coefficients (sp_internal.h)

__global__ __launch_bounds__(256) void theta_kernel(
    int K, const double *__restrict__ t, const sp_star *__restrict__ stars,
    double *__restrict__ theta, int32_t *__restrict__ info, uint32_t *__restrict__ status,
    const double *__restrict__ tab, int np, double *__restrict__ ptab) {
  const int s = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (ptab && blockIdx.x == 0) {
    // the star's table in the layout SplineGen reads (load_tables below), for the kernels that
    // form covariance tiles at first touch (LazyCov, sp_cov.h)
    const double *src = tab + (size_t)stars[s].table * 5 * np;
    for (int e = threadIdx.x; e < 4 * np; e += 256) {
      const int half = e >= 2 * np, f = half ? e - 2 * np : e;
      ptab[(size_t)s * 4 * np + e] = src[(1 + 2 * half + (f & 1)) * np + (f >> 1)];
    }
  }
  if (i == 0) {  // first kernel of a likelihood call: also clears the per-star flags
    if (info) info[s] = 0;
    if (status) status[s] = 0u;
  }
  if (i >= K) return;
  const double a = t[(size_t)s * K + i] / stars[s].period;
  double m = fmod(a, 1.0);
  if (m != 0.0 && m < 0.0) m += 1.0;
  theta[(size_t)s * K + i] = 6.283185307179586 * m;
}

// int64 interpolation indices of every (i, j) pair, exposed for parity tests
__global__ __launch_bounds__(256) void spline_index_kernel(
    int K, const double *__restrict__ theta, double dx, long long *__restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)K * K) return;
  const int i = e / K, j = e % K;
  const double x = fabs(theta[i] - theta[j]);
  out[e] = (long long)floor(x / dx);
}

__device__ __forceinline__ void load_tables(const double *__restrict__ tab, int np,
                                            const double *__restrict__ xp,
                                            double *s_tab) {
  // s_tab: [np][2] = {a0, a1} per segment, then [np][2] = {a2, a3} (the lag grid itself is not
  // needed: xp[k] = (k - 1) dx, SplineGen)
  for (int i = threadIdx.x; i < 4 * np; i += blockDim.x) {
    const int half = i >= 2 * np, e = half ? i - 2 * np : i;
    const int seg = e >> 1, k = 2 * half + (e & 1);
    s_tab[i] = tab[(1 + k) * np + seg];
  }
}

// Row sums of the raw covariance.  grid (ceil(K/64), S), 256 threads: thread
// (r = tid & 63, q = tid >> 6) sums columns j = q, q+4, ... of row r.  The
// columns' phases (and times) pass through LDS in chunks of `chunk` (= K when
// one star's fit, which is every configuration of BASELINE.json).
template <bool FROM_MATRIX>
__global__ __launch_bounds__(256) void rowsum_kernel(
    int K, int chunk, const double *__restrict__ theta, const double *__restrict__ t,
    const sp_star *__restrict__ stars, int covpts, const double *__restrict__ tab,
    const double *__restrict__ meanvar, const double *__restrict__ xp,
    int temporal, const double *raw, double *__restrict__ rowsum) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int s = blockIdx.y, np = covpts + 4;
  const sp_star st = stars[s];
  double *s_tab = lds;            // 4 * np (unused when FROM_MATRIX)
  double *s_th = lds + (FROM_MATRIX ? 0 : 4 * np);  // chunk
  double *s_t = s_th + chunk;     // chunk
  double *s_red = s_t + chunk;    // 256
  if (!FROM_MATRIX) load_tables(tab + (size_t)st.table * 5 * np, np, xp, s_tab);
  const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;
  const int nobs = star_nobs(st, K);
  const bool tk = temporal != SP_TEMPORAL_NONE;
  const bool live = i < nobs;
  const double thi = (live && !FROM_MATRIX) ? theta[(size_t)s * K + i] : 0.0;
  const double ti = (live && tk) ? t[(size_t)s * K + i] : 0.0;
  SplineGen g{s_tab, 2 * np, 6.283185307179586 / covpts,
              1.0 / (6.283185307179586 / covpts), covpts};
  double acc = 0.0;
  for (int c0 = 0; c0 < nobs; c0 += chunk) {
    const int n = nobs - c0 < chunk ? nobs - c0 : chunk;
    if (c0) __syncthreads();
    if (!FROM_MATRIX)
      for (int j = threadIdx.x; j < n; j += 256) s_th[j] = theta[(size_t)s * K + c0 + j];
    if (tk)
      for (int j = threadIdx.x; j < n; j += 256) s_t[j] = t[(size_t)s * K + c0 + j];
    __syncthreads();
    if (!live) continue;
    if (FROM_MATRIX) {
      const double *row = raw + ((size_t)s * K + i) * K + c0;
      for (int j = q; j < n; j += 4)
        acc += row[j] * temporal_factor(temporal, ti, tk ? s_t[j] : 0.0, st.tau);
    } else if (nobs == 1) {
      acc = q == 0 ? meanvar[2 * st.table + 1] : 0.0;
    } else if (!tk) {
      for (int j = q; j < n; j += 4) acc += g(thi, s_th[j]);
    } else {
      for (int j = q; j < n; j += 4)
        acc += g(thi, s_th[j]) * temporal_factor(temporal, ti, s_t[j], st.tau);
    }
  }
  s_red[threadIdx.x] = acc;
  __syncthreads();
  if (q == 0 && i < K)
    rowsum[(size_t)s * K + i] = (s_red[r] + s_red[64 + r]) + (s_red[128 + r] + s_red[192 + r]);
}

// one workgroup per star (sp.py:705-727, ops/norm/norm.py:26-44)
__global__ __launch_bounds__(256) void norm_coef_kernel(
    int K, const sp_star *__restrict__ stars, const double *__restrict__ meanvar,
    const double *__restrict__ condmean, int normalized, int order, double zmax,
    const double *__restrict__ rowsum, double *__restrict__ qv, Coef *__restrict__ coef,
    uint32_t *__restrict__ status) {
  __shared__ double red[4];
  const int s = blockIdx.x;
  const double fmean = condmean ? condmean[s] : meanvar[2 * stars[s].table];
  Coef c;
  c.c1 = 1.0;
  c.zab = 0.0;
  c.za = 0.0;
  c.z = 0.0;
  c.gpmean = normalized ? 0.0 : fmean;
  c.m = 0.0;
  c.mu = 1.0 + fmean;
  c.d1 = 0.0;
  const int nobs = star_nobs(stars[s], K);
  if (normalized) {
    double part = 0.0;
    for (int i = threadIdx.x; i < nobs; i += 256) part += rowsum[(size_t)s * K + i];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const double total = (red[0] + red[1]) + (red[2] + red[3]);
    const double m = total / ((double)nobs * (double)nobs);
    const double mu = c.mu;
    const double z = m / (mu * mu);
    double fac = 1.0, alpha = 0.0, beta = 0.0;
    for (int n = 0; n <= order; ++n) {
      alpha += fac;
      beta += 2 * n * fac;
      fac *= z * (2 * n + 3);
    }
    c.c1 = alpha / (mu * mu);
    c.zab = alpha + beta;
    c.za = alpha;
    c.z = z;
    c.m = m;
    const double km = (double)nobs * m;
    for (int i = threadIdx.x; i < K; i += 256)
      qv[(size_t)s * K + i] = i < nobs ? rowsum[(size_t)s * K + i] / km : 0.0;
    if (threadIdx.x == 0 && status && z > zmax) atomicOr(&status[s], SP_STAR_ZMAX);
  }
  if (threadIdx.x == 0) coef[s] = c;
}

// Writes 64 x 64 tiles of the final matrix.
//   SYSTEM = false: plain K x K covariance, every tile (sp.cov()).
//   SYSTEM = true : the padded (Kp x Kp) Cholesky system: lower-triangle tiles
//                   only, residual rows K..K+M-1, unit diagonal on the padding.
//   DEFER  = true : (SYSTEM only) deferred normalisation.  The tile is written RAW (spline x temporal
//                   factor, no normalisation, no noise) and its row sums -- and, for tiles below
//                   the diagonal, its column sums, which are the row sums of the mirror tile --
//                   go to part[s][column tile][row]: the separate row-sum pass over all K^2
//                   entries disappears.  defer_finish_kernel turns the sums into the
//                   normalisation's vectors, which then ride through the factorisation as three
//                   more rows (sp.py:705-727 is  c1 Sigma + rank 2;  lnlike_reduce_kernel applies
//                   the rank-2 (+ baseline) part by the matrix determinant / Woodbury identities).
template <bool FROM_MATRIX, bool SYSTEM, bool DEFER = false>
__global__ __launch_bounds__(256) void assemble_kernel(
    int K, int M, int Kp, const double *__restrict__ theta,
    const double *__restrict__ t, const sp_star *__restrict__ stars, int covpts,
    const double *__restrict__ tab, const double *__restrict__ meanvar,
    const double *__restrict__ xp, int temporal, const double *__restrict__ raw,
    int normalized, const double *__restrict__ qv, const Coef *__restrict__ coef,
    const double *__restrict__ diag, int add_noise, const double *__restrict__ flux,
    double *__restrict__ out, long ldo, long strideo, int ntr, double *__restrict__ part,
    int lazy_nfull) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int s = blockIdx.y, np = covpts + 4;
  const sp_star st = stars[s];
  Coef c;
  if (DEFER) {
    c.c1 = 1.0; c.zab = 0.0; c.za = 0.0; c.z = 0.0; c.gpmean = 0.0; c.m = 0.0; c.mu = 1.0; c.d1 = 0.0;
  } else {
    c = coef[s];
  }
  int ti, tj;
  if (SYSTEM) {
    const int tile = blockIdx.x;
    ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    tj = tile - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x / ntr;
    tj = blockIdx.x % ntr;
  }
  const int i0 = ti * 64, j0 = tj * 64;
  double *s_tab = lds;                                  // 4 np
  double *s_thi = lds + (FROM_MATRIX ? 0 : 4 * np);     // 64 each below
  double *s_thj = s_thi + 64, *s_ti = s_thj + 64, *s_tj = s_ti + 64;
  double *s_qi = s_tj + 64, *s_qj = s_qi + 64;
  double *s_col = s_qj + 64;   // DEFER: [16][64] column-sum partials
  if (!FROM_MATRIX) load_tables(tab + (size_t)st.table * 5 * np, np, xp, s_tab);
  if (threadIdx.x < 64) {
    const int i = i0 + threadIdx.x;
    const bool ok = i < K;
    s_thi[threadIdx.x] = (ok && !FROM_MATRIX) ? theta[(size_t)s * K + i] : 0.0;
    s_ti[threadIdx.x] = (ok && temporal != SP_TEMPORAL_NONE) ? t[(size_t)s * K + i] : 0.0;
    s_qi[threadIdx.x] = (ok && normalized && !DEFER) ? qv[(size_t)s * K + i] : 0.0;
  } else if (threadIdx.x < 128) {
    const int l = threadIdx.x - 64, j = j0 + l;
    const bool ok = j < K;
    s_thj[l] = (ok && !FROM_MATRIX) ? theta[(size_t)s * K + j] : 0.0;
    s_tj[l] = (ok && temporal != SP_TEMPORAL_NONE) ? t[(size_t)s * K + j] : 0.0;
    s_qj[l] = (ok && normalized && !DEFER) ? qv[(size_t)s * K + j] : 0.0;
  }
  __syncthreads();
  SplineGen g{s_tab, 2 * np, 6.283185307179586 / covpts,
              1.0 / (6.283185307179586 / covpts), covpts};
  const int nobs = star_nobs(st, K);
  const double var1 = (!FROM_MATRIX && nobs == 1) ? meanvar[2 * st.table + 1] : 0.0;
  double *ob = out + (size_t)s * strideo;
  // thread -> columns cl, cl + 16, cl + 32, cl + 48 of a row, 16 rows per pass: the 16 lanes of a row look up
  // ADJACENT columns -- neighbouring phases, neighbouring table segments, which the two arrays of 16-byte entries
  // put on distinct bank groups (four consecutive columns per thread spread a row's lanes over ~80 segments: 37 %
  // of the LDS cycles of this kernel were bank conflicts)
  const int cl = threadIdx.x & 15, ri = threadIdx.x >> 4;
  const int lim = SYSTEM ? Kp : K;
  double csum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int li = ri + 16 * pass, i = i0 + li;
    if (!DEFER && i >= lim) continue;
    double v[4];
    double rsum = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int lj = cl + 16 * e, j = j0 + lj;
      double val = 0.0;
      if (i < nobs && j < nobs) {
        double rawv;
        if (FROM_MATRIX)
          rawv = raw[((size_t)s * K + i) * K + j];
        else if (nobs == 1)
          rawv = var1;
        else
          rawv = g(s_thi[li], s_thj[lj]);
        rawv *= temporal_factor(temporal, s_ti[li], s_tj[lj], st.tau);
        if (DEFER) {
          val = rawv;
          rsum += rawv;
          csum[e] += rawv;
        } else if (normalized) {
          const double qi = s_qi[li], qj = s_qj[lj];
          const double pp = (1.0 - qi) * (1.0 - qj), qq = qi * qj;
          val = c.c1 * rawv + c.z * (c.zab * pp - c.za * qq);
        } else {
          val = rawv;
        }
        if (add_noise && !DEFER) {
          if (i == j) val += diag ? diag[(size_t)s * K + i] : st.data_var;
          val += st.baseline_var;
        }
      } else if (SYSTEM) {
        if (i >= K && i < K + M && j < nobs)
          val = flux[((size_t)s * M + (i - K)) * K + j] - (c.gpmean + st.baseline_mean);
        else if (i == j)
          val = 1.0;
      }
      v[e] = val;
    }
    if (DEFER) {
      // row sum of this tile's 64 columns: the 16 lanes that share the row
      {
        double one[1] = {rsum};
        row16_sum(one);
        rsum = one[0];
      }
      if ((threadIdx.x & 15) == 0 && i < K) part[((size_t)s * ntr + tj) * K + i] = rsum;
      if (i >= lim) continue;
    }
    // (tiles the factorisation forms itself at first touch: sums taken above, nothing written)
    // (not the first block column: its panel launch has no product to form the tile behind)
    if (DEFER && ti > tj && tj > 0 && ti < lazy_nfull) continue;
    double *dst = ob + (size_t)i * ldo + j0 + cl;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (j0 + cl + 16 * e < lim) dst[16 * e] = v[e];
  }
  if (DEFER && ti > tj) {
    // column sums = row sums of the mirror tile (tj, ti), which is never formed
#pragma unroll
    for (int e = 0; e < 4; ++e) s_col[ri * 64 + cl + 16 * e] = csum[e];
    __syncthreads();
    if (threadIdx.x < 64) {
      double a = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += s_col[r * 64 + threadIdx.x];
      const int j = j0 + threadIdx.x;
      if (j < K) part[((size_t)s * ntr + ti) * K + j] = a;
    }
  }
}

// The hot form of the assembly -- marginal branch, deferred normalisation (what assemble_kernel<false, true,
// true> did until round 4): the RAW lower tiles of the padded system and their row / column sums.
//
// Round 3's kernel spent one workgroup per 64 x 64 tile (8 704 per 64-star step): timed with the spline
// evaluation compiled out it still took 33 of its 55 us -- tile decode, a transposing copy of the table that
// waits out five memory round trips in a row, two barriers, a column reduction through LDS per tile.  Here a
// workgroup is long-lived: it takes a CHUNK of its star's lower tiles in column-strip order (tiles (tj, tj),
// (tj + 1, tj), ... (ntr - 1, tj), then the next strip), so that
//   * the star's table (packed by theta_kernel) is copied to LDS once per workgroup, every load of the
//     prologue in flight before the first wait;
//   * the phases of a strip's 64 columns stay in four registers per thread, the column sums accumulate in
//     four more over the strip and go through LDS once per strip, not once per tile;
//   * the phases of a tile's rows are fetched one tile ahead; the tile loop has no barrier at all;
//   * the 16 entries of a thread are evaluated as ONE batch (SplineGen::many) and the temporal kernel is a
//     template parameter: no run-time switch (and no inlined exp) between entries.
// Same operations per entry as the general kernel and as the tiles formed at first touch: same bits.
// Round 5: the reduction needs nothing of the covariance's row sums any more (sp_reduce.h) -- only the star's TOTAL,
// for m = mean(Sigma).  A thread adds up its entries (a tile below the diagonal counts twice: its mirror is never
// formed), a workgroup leaves ONE number, part[s][chunk]; rounds 2-4 took row sums by DPP and column sums through LDS
// per strip (a third of the kernel's cycles) and stored K numbers per tile row.
#ifdef SP_ASM_STAMPS
__device__ long long sp_asm_dbg[8 * 8192];
#endif
#ifndef SP_ASM_BATCH
#define SP_ASM_BATCH 16      // entries per SplineGen::many batch (4, 8 or 16)
#endif
#ifndef SP_ASM_OCC
#define SP_ASM_OCC 2         // workgroups per CU the register budget is cut for
#endif
template <int TK>
__global__ __launch_bounds__(256, SP_ASM_OCC) void assemble_sums_kernel(
    int K, int M, int Kp, const double *__restrict__ theta, const double *__restrict__ t,
    const sp_star *__restrict__ stars, int covpts, const double *__restrict__ ptab,
    const double *__restrict__ meanvar, const double *__restrict__ flux, double *__restrict__ out, long ldo,
    long strideo, int ntr, double *__restrict__ part, int lazy_nfull, AsmChunks chunks) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int s = blockIdx.y, np = covpts + 4, tid = threadIdx.x;
#ifdef SP_ASM_STAMPS
  const long long stamp_begin = __builtin_readcyclecounter();
  const long long wall_begin = wall_clock64();
#endif
  const sp_star st = stars[s];
  const int t0 = chunks.start[blockIdx.x], t1 = chunks.start[blockIdx.x + 1];
  if (t0 >= t1) {
    if (threadIdx.x == 0) part[(size_t)s * gridDim.x + blockIdx.x] = 0.0;
    return;
  }
  // strip-major tile order: strip tj holds the tiles ti = tj .. ntr - 1
  int tj = 0, ti;
  {
    int rem = t0;
    while (rem >= ntr - tj) {
      rem -= ntr - tj;
      ++tj;
    }
    ti = tj + rem;
  }
  double *s_tab = lds;                       // 4 np
  double *s_red = s_tab + 4 * np;            // 8: the workgroup's sum
  double *s_th = s_red + 8;                  // [Kp] the star's phases (zero beyond K)
  double *s_tt = s_th + Kp;                  // [Kp] its times (temporal kernels only)
  const int nobs = star_nobs(st, K);
  const double *th = theta + (size_t)s * K, *tt = t + (size_t)s * K;
  // thread -> columns cl, cl + 16, cl + 32, cl + 48 of rows ri, ri + 16, ri + 32, ri + 48 of a tile (the 16 lanes of
  // a row look up ADJACENT columns: neighbouring table segments on distinct bank groups)
  const int cl = tid & 15, ri = tid >> 4;
  // The phases (and times) of the whole star go to LDS once: the tile loop then issues NO vector-memory load.
  // Fetched from memory one tile ahead -- the first long-lived form -- every tile waited for its predecessor's
  // STORES: loads and stores share one in-order counter, so the wait for the prefetched rows (the youngest
  // operations) was a wait for the row-sum and tile stores issued before them: 2.75 us per tile for 0.6 us of
  // arithmetic, whatever the arithmetic was (a gather with five instructions less per entry changed nothing).
  int unsorted = 0;       // (Matern-3/2 only) some cadence of this star earlier than its predecessor
  {
    double a[4], b[4], bp[4];
    for (int base = 0; base < Kp; base += 1024) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = base + tid + 256 * c;
        a[c] = i < K ? th[i] : 0.0;
        b[c] = (TK != SP_TEMPORAL_NONE && i < K) ? tt[i] : 0.0;
        bp[c] = (TK == SP_TEMPORAL_MATERN32 && i > 0 && i < nobs) ? tt[i - 1] : -INFINITY;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = base + tid + 256 * c;
        if (i < Kp) {
          s_th[i] = a[c];
          if (TK != SP_TEMPORAL_NONE) s_tt[i] = b[c];
        }
        if (TK == SP_TEMPORAL_MATERN32 && i < nobs && !(b[c] >= bp[c])) unsorted = 1;
      }
    }
  }
  double thj[4], tmj[4], thi[4], tmi[4];
  auto load_cols = [&](int tjj) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 64 * tjj + cl + 16 * e;
      thj[e] = s_th[j];
      tmj[e] = TK != SP_TEMPORAL_NONE ? s_tt[j] : 0.0;
    }
  };
  auto load_rows = [&](int tii, double (&a)[4], double (&b)[4]) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int i = 64 * tii + ri + 16 * pass;
      a[pass] = s_th[i];
      b[pass] = TK != SP_TEMPORAL_NONE ? s_tt[i] : 0.0;
    }
  };
  spline_table_to_lds(ptab + (size_t)s * 4 * np, s_tab, np, tid);
  // The Matern-3/2 factor (1 + x) exp(-x), x = sqrt(3) |t_i - t_j| / tau (temporal.py:8-11), SEPARATES below the
  // diagonal of a light curve whose cadences are in order: exp(-c (t_i - t_j)) = exp(-c (t_i - b)) exp(c (t_j - b))
  // for any b -- four exponentials per thread and tile (its rows; the columns' are a strip's) instead of sixteen,
  // and no division (c = sqrt(3) / tau once).  A division, an exponential and the polynomial per entry were three
  // times the spline's own cost: cfg5's assembly 351 us.  b = the strip's first cadence, so that the column
  // factors are exp(c (span of 64 cadences)): a strip whose span would overflow that (c span > 600: a gap of
  // years at a tau of hours), a light curve out of order, the diagonal tile (|t_i - t_j| both ways) and the last
  // row tile (padding) take the entry-by-entry form.  The two forms agree to a few ulp, not bit for bit.
  const bool in_order = TK == SP_TEMPORAL_MATERN32 ? !__syncthreads_or(unsorted) : (__syncthreads(), false);
  const double cm = TK == SP_TEMPORAL_MATERN32 ? 1.7320508075688772 / st.tau : 0.0;
  double bref = 0.0, fc[4] = {1.0, 1.0, 1.0, 1.0};
  bool sep_strip = false;
  auto strip_factors = [&](int tjj) {
    if (TK != SP_TEMPORAL_MATERN32) return;
    const int j0s = 64 * tjj, j1s = (j0s + 63 < nobs ? j0s + 63 : nobs - 1);
    bref = s_tt[j0s];
#ifdef SP_ASM_NO_SEP_MATERN
    sep_strip = false;      // (A/B: every entry its own exponential)
#else
    sep_strip = in_order && j1s >= j0s && cm * (s_tt[j1s] - bref) < 600.0 && cm > 0.0;
#endif
    if (sep_strip) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fc[e] = exp(cm * (tmj[e] - bref));
    }
  };
  load_cols(tj);
  strip_factors(tj);
  load_rows(ti, thi, tmi);
  SplineGen g{s_tab, 2 * np, 6.283185307179586 / covpts, 1.0 / (6.283185307179586 / covpts), covpts};
  const double var1 = nobs == 1 ? meanvar[2 * st.table + 1] : 0.0;
  double *ob = out + (size_t)s * strideo;
  double tsum = 0.0;                         // this thread's share of the star's total (off-diagonal tiles twice)
#ifdef SP_ASM_STAMPS
  long long stamp_eval = 0, stamp_sums = 0, stamp_rest = 0, stamp_loop0 = __builtin_readcyclecounter();
#define SP_STAMP(acc, since) do { const long long now_ = __builtin_readcyclecounter(); acc += now_ - since; since = now_; } while (0)
  long long stamp_t = stamp_loop0;
#else
#define SP_STAMP(acc, since)
#endif
  for (int tile = t0; tile < t1; ++tile) {
    const int i0 = 64 * ti, j0 = 64 * tj;
    const bool strip_ends = ti == ntr - 1 || tile == t1 - 1;
    const int nti = ti == ntr - 1 ? tj + 1 : ti + 1;
    double v[16];
#pragma unroll
    for (int p0 = 0; p0 < 4; p0 += SP_ASM_BATCH / 4) {
      double a[SP_ASM_BATCH], b[SP_ASM_BATCH], o[SP_ASM_BATCH];
#pragma unroll
      for (int k = 0; k < SP_ASM_BATCH; ++k) {
        a[k] = thi[p0 + (k >> 2)];
        b[k] = thj[k & 3];
      }
      g.many<SP_ASM_BATCH>(a, b, o);
#pragma unroll
      for (int k = 0; k < SP_ASM_BATCH; ++k) v[4 * p0 + k] = o[k];
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
        for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
          v[4 * pass + e] *= temporal_factor(TK, tmi[pass], tmj[e], st.tau);
        }
    }
    asm volatile("" : "+v"(v[0]), "+v"(v[5]), "+v"(v[10]), "+v"(v[15]));
    SP_STAMP(stamp_eval, stamp_t);
    // the next tile's rows (LDS)
    if (tile + 1 < t1) load_rows(nti, thi, tmi);
    // (tiles the factorisation forms itself at first touch: sums taken, nothing written -- not the first block
    //  column: its panel launch has no product to form the tile behind)
    const bool skip_write = ti > tj && tj > 0 && ti < lazy_nfull;
    const double twice = ti > tj ? 2.0 : 1.0;
    if (i0 + 64 <= nobs && j0 + 64 <= nobs) {
      // a tile of valid cadences only (all but the last row / column tile of a system): no masks -- the general
      // form below spends five instructions on predicates and selects for every one of the evaluation
      double ts = 0.0;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass)
        ts += (v[4 * pass] + v[4 * pass + 1]) + (v[4 * pass + 2] + v[4 * pass + 3]);
      tsum += twice * ts;
      if (!skip_write) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          double *dst = ob + (size_t)(i0 + ri + 16 * pass) * ldo + j0 + cl;
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[16 * e] = v[4 * pass + e];
        }
      }
    } else {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int i = i0 + ri + 16 * pass;
        double w[4];
        double rsum = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = j0 + cl + 16 * e;
          double val = 0.0;
          if (i < nobs && j < nobs) {
            val = v[4 * pass + e];
            rsum += val;
          } else if (i >= K && i < K + M && j < nobs) {
            val = flux[((size_t)s * M + (i - K)) * K + j] - st.baseline_mean;   // (the GP mean of the normalised process is 0)
          } else if (i == j) {
            val = 1.0;
          }
          w[e] = val;
        }
        tsum += twice * rsum;
        if (i >= Kp || skip_write) continue;
        double *dst = ob + (size_t)i * ldo + j0 + cl;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (j0 + cl + 16 * e < Kp) dst[16 * e] = w[e];
      }
    }
    SP_STAMP(stamp_sums, stamp_t);
    if (strip_ends && tile + 1 < t1) {
      load_cols(tj + 1);
      strip_factors(tj + 1);
    }
    if (ti == ntr - 1) {
      ++tj;
      ti = tj;
    } else {
      ++ti;
    }
    SP_STAMP(stamp_rest, stamp_t);
  }
  // the workgroup's sum: wavefront sums, then the four of them in a fixed order
  for (int off = 32; off > 0; off >>= 1) tsum += __shfl_down(tsum, off, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = tsum;
  __syncthreads();
  if (tid == 0) part[(size_t)s * gridDim.x + blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
#ifdef SP_ASM_STAMPS
  if (tid == 0) {
    long long *o = sp_asm_dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = blockIdx.x; o[1] = blockIdx.y; o[2] = wall_begin; o[3] = wall_clock64();
    o[4] = stamp_loop0 - stamp_begin; o[5] = stamp_eval; o[6] = stamp_sums; o[7] = stamp_rest;
  }
  if (false && (tid & 63) == 0 && blockIdx.y == 5 && (blockIdx.x == 3 || blockIdx.x == 9))
    printf("asm stamps wg %d wave %d: tiles %d  prologue %lld  eval %lld  sums+stores %lld  rest %lld  (ticks, per tile: %lld %lld %lld)\n",
           (int)blockIdx.x, tid >> 6, t1 - t0, stamp_loop0 - stamp_begin, stamp_eval, stamp_sums, stamp_rest,
           stamp_eval / (t1 - t0), stamp_sums / (t1 - t0), stamp_rest / (t1 - t0));
#endif
}

// Deferred normalisation, second half: one workgroup per star.
//   total = sum of the tiles' row / column partial sums of part[s][c][r]   (fixed order: deterministic)
//   m, z, alpha(z), beta(z), c1 = alpha / mu^2                       (sp.py:705-727, ops/norm/norm.py:26-44)
//   the system holds  B = Sigma + D / c1  (D: data variance): the diagonal gets D / c1 here
//   one more row below the residuals: 1 (valid cadences); with per-cadence variances a second one: d = diag(D) / c1
//   rscal[s] = {K m, sum(d), delta, sum(r_0), ...}: what the reduction needs of q = Sigma 1 / (K m) without
//   ever forming it (sp_reduce.h; rounds 2-4 wrote p = 1 - q and q as rows here)
// C = c1 (B + d_p p p^T + d_q q q^T + d_1 1 1^T) with d_p = z (alpha + beta) / c1,
// d_q = -z alpha / c1, d_1 = baseline_var / c1: the reduction finishes the job.
__device__ __forceinline__ double block_sum_1024(double v, double *red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();                       // (red may still be read from the previous sum)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double total = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) total += red[w];
  return total;
}

// nflat > 0: part[s][0 .. nflat) holds partial TOTALS (assemble_sums_kernel); else part[s][column tile][row] holds
// the tiles' row / column sums (assemble_kernel<.., DEFER>, cond_system_kernel)
__global__ __launch_bounds__(1024) void defer_finish_kernel(
    int K, int M, int Kp, int ntr, int nflat, const sp_star *__restrict__ stars,
    const double *__restrict__ meanvar, const double *__restrict__ condmean, int order, double zmax,
    const double *__restrict__ part, const double *__restrict__ diag, const double *__restrict__ flux,
    double *__restrict__ sys, Coef *__restrict__ coef, double *__restrict__ rscal,
    uint32_t *__restrict__ status) {
  __shared__ double red[16];
  const int s = blockIdx.x;
  const sp_star st = stars[s];
  const int nobs = star_nobs(st, K);
  double *Mx = sys + (size_t)s * Kp * Kp;
  const double *P = part + (size_t)s * (nflat > 0 ? nflat : ntr * K);
  // (ragged stars: tiles beyond nobs contribute zeros)
  double mine = 0.0;
  if (nflat > 0) {
    // (one thread, fixed order: a few dozen numbers)
    if (threadIdx.x == 0)
      for (int c = 0; c < nflat; ++c) mine += P[c];
  } else
  for (int r = threadIdx.x; r < K; r += 1024) {
    // the column tiles eight at a time: the loads of a group are independent, the sum keeps its order
    double a = 0.0;
    for (int c0 = 0; c0 < ntr; c0 += 8) {
      double v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = c0 + c < ntr ? P[(size_t)(c0 + c) * K + r] : 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) a += v[c];
    }
    if (r >= nobs) a = 0.0;
    mine += a;
  }
  const double total = block_sum_1024(mine, red);
  const double fmean = condmean ? condmean[s] : meanvar[2 * st.table];
  const double m = total / ((double)nobs * (double)nobs);
  const Coef c = defer_coef(m, fmean, order, st.baseline_var);
  const double c1 = c.c1;
  double *row1 = Mx + (size_t)(K + M) * Kp, *rowd = row1 + Kp;
  double sdm = 0.0;
  for (int r = threadIdx.x; r < K; r += 1024) {
    const bool ok = r < nobs;
    const double d = (diag ? diag[(size_t)s * K + r] : st.data_var) / c1;
    row1[r] = ok ? 1.0 : 0.0;
    if (diag) rowd[r] = ok ? d : 0.0;
    if (ok) {
      Mx[(size_t)r * Kp + r] += d;
      sdm += d;
    }
  }
  double *rs = rscal + (size_t)s * (SP_RSCAL_HEAD + M);
  const double delta = st.data_var / c1;
  const double sd = diag ? block_sum_1024(sdm, red) : (double)nobs * delta;
  for (int mm = 0; mm < M; ++mm) {
    const double *f = flux + ((size_t)s * M + mm) * K;
    double a = 0.0;
    for (int r = threadIdx.x; r < nobs; r += 1024) a += f[r];
    const double sf = block_sum_1024(a, red);
    if (threadIdx.x == 0) rs[SP_RSCAL_HEAD + mm] = sf - (double)nobs * st.baseline_mean;
  }
  if (threadIdx.x == 0) {
    rs[0] = (double)nobs * m;
    rs[1] = sd;
    rs[2] = delta;
    coef[s] = c;
    if (status && c.z > zmax) atomicOr(&status[s], SP_STAR_ZMAX);
  }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------

static size_t attr_lds_limit = 150 * 1024;

template <typename F>
static void allow_big_lds(F f) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(f),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)attr_lds_limit);
}

int sp_launch_theta(int S, int K, const double *t, const sp_star *stars,
                    double *theta, hipStream_t st, int32_t *info, uint32_t *status,
                    const double *tab, int covpts, double *ptab) {
  hipLaunchKernelGGL(theta_kernel, dim3((K + 255) / 256, S), dim3(256), 0, st, K,
                     t, stars, theta, info, status, tab, covpts + 4, ptab);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_spline_index(int K, const double *theta, double dx, long long *out,
                           hipStream_t st) {
  const long n = (long)K * K;
  hipLaunchKernelGGL(spline_index_kernel, dim3((unsigned)((n + 255) / 256)),
                     dim3(256), 0, st, K, theta, dx, out);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_rowsum(int S, int K, const double *theta, const double *t,
                     const sp_star *stars, int covpts, const double *tab,
                     const double *meanvar, const double *xp, int temporal,
                     const double *raw, double *rowsum, hipStream_t st) {
  const int np = covpts + 4;
  const size_t fixed = sizeof(double) * ((raw ? 0 : 4 * (size_t)np) + 256);
  if (fixed + sizeof(double) * 2 * 64 > attr_lds_limit) return SP_ERR_INVALID;
  // columns per LDS pass: all of them when they fit (K <= 4096 keeps every BASELINE
  // configuration at one pass and at the occupancy it was measured with)
  const int chunk = K < 4096 ? K : 4096;
  const size_t lds = fixed + sizeof(double) * 2 * (size_t)chunk;
  if (lds > attr_lds_limit) return SP_ERR_INVALID;
  dim3 grid((K + 63) / 64, S);
  if (raw) {
    allow_big_lds(rowsum_kernel<true>);
    hipLaunchKernelGGL(rowsum_kernel<true>, grid, dim3(256), lds, st, K, chunk, theta, t,
                       stars, covpts, tab, meanvar, xp, temporal, raw, rowsum);
  } else {
    allow_big_lds(rowsum_kernel<false>);
    hipLaunchKernelGGL(rowsum_kernel<false>, grid, dim3(256), lds, st, K, chunk, theta,
                       t, stars, covpts, tab, meanvar, xp, temporal, raw, rowsum);
  }
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_norm_coef(int S, int K, const sp_star *stars, const double *meanvar,
                        const double *condmean, int normalized, int order,
                        double zmax, const double *rowsum, double *qv, void *coef,
                        uint32_t *status, hipStream_t st) {
  hipLaunchKernelGGL(norm_coef_kernel, dim3(S), dim3(256), 0, st, K, stars,
                     meanvar, condmean, normalized, order, zmax, rowsum, qv,
                     (Coef *)coef, status);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_assemble(int S, int K, int M, int Kp, int system,
                       const double *theta, const double *t, const sp_star *stars,
                       int covpts, const double *tab, const double *meanvar,
                       const double *xp, int temporal, const double *raw,
                       int normalized, const double *qv, const void *coef,
                       const double *diag, int add_noise, const double *flux,
                       double *out, long ldo, long strideo, hipStream_t st, double *part,
                       int lazy_nfull) {
  const int np = covpts + 4;
  const size_t lds = sizeof(double) * ((raw ? 0 : 4 * (size_t)np) + 6 * 64 + (part ? 16 * 64 : 0));
  if (lds > attr_lds_limit) return SP_ERR_INVALID;
  const int ntr = ((system ? Kp : K) + 63) / 64;
  const int ntiles = system ? ntr * (ntr + 1) / 2 : ntr * ntr;
  dim3 grid(ntiles, S);
#define SP_ASM(FM, SY, ...)                                                                \
  do {                                                                                     \
    allow_big_lds(assemble_kernel<FM, SY, ##__VA_ARGS__>);                                 \
    hipLaunchKernelGGL((assemble_kernel<FM, SY, ##__VA_ARGS__>), grid, dim3(256), lds, st, K, \
                       M, Kp, theta, t, stars, covpts, tab, meanvar, xp,                   \
                       temporal, raw, normalized, qv, (const Coef *)coef, diag,            \
                       add_noise, flux, out, ldo, strideo, ntr, part, lazy_nfull);         \
  } while (0)
  if (part && !system) return SP_ERR_INVALID;
  if (part && raw)
    SP_ASM(true, true, true);
  else if (part)
    SP_ASM(false, true, true);
  else if (raw && system)
    SP_ASM(true, true);
  else if (raw)
    SP_ASM(true, false);
  else if (system)
    SP_ASM(false, true);
  else
    SP_ASM(false, false);
#undef SP_ASM
  SP_LAUNCH_CHECK();
  return SP_OK;
}

#ifdef SP_ASM_STAMPS
extern "C" int sp_debug_asm_stamps(long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_asm_dbg), sizeof(long long) * n);
}
#endif

// Chunks of equal COST, not of equal length: a tile of the last row tile (masks, residual rows, identity padding; it
// also ends its strip: the column sums' trip through LDS) costs 3.8 tiles, a tile that is written (the diagonal one,
// any tile of strip 0) 1.2 -- per-workgroup stamps, tools/attic/asm_wall.py: with 17 tiles each the workgroup of the last
// strips (five last-row tiles) ran for 50-58 us, the others for 33-37, and the launch lasted as long as it did.  The
// weights depend on the shape alone -- not on the batch, not on which tiles are left to their first touch: a star's
// sums are the same bits whoever shares its launch and whichever way its tiles are formed.
// Chunk c = the tiles whose cumulative weight BEFORE them lies in [c total / nchunk, (c + 1) total / nchunk).
static AsmChunks asm_chunks(int ntr, int nchunk) {
  static thread_local int have_ntr = -1, have_nchunk = -1;
  static thread_local AsmChunks have;
  if (ntr == have_ntr && nchunk == have_nchunk) return have;
  const auto weight = [&](int a, int b) { return a == ntr - 1 ? 38 : ((b == 0 || a == b) ? 12 : 10); };
  long total = 0;
  for (int b = 0; b < ntr; ++b)
    for (int a = b; a < ntr; ++a) total += weight(a, b);
  AsmChunks c;
  int chunk = 0, tile = 0;
  long cum = 0;
  for (int b = 0; b < ntr; ++b)
    for (int a = b; a < ntr; ++a, ++tile) {
      while (chunk <= nchunk && cum >= (long)chunk * total / nchunk) c.start[chunk++] = (unsigned short)tile;
      cum += weight(a, b);
    }
  while (chunk <= nchunk) c.start[chunk++] = (unsigned short)tile;
  have = c;
  have_ntr = ntr;
  have_nchunk = nchunk;
  return c;
}

extern "C" int sp_debug_asm_chunks(int ntr, int nchunk, int *start_host) {
  if (ntr < 1 || nchunk < 1 || nchunk > SP_ASM_MAX_CHUNKS || ntr * (ntr + 1) / 2 > 65535 || !start_host) return SP_ERR_INVALID;
  const AsmChunks c = asm_chunks(ntr, nchunk);
  for (int k = 0; k <= nchunk; ++k) start_host[k] = c.start[k];
  return SP_OK;
}

// LDS of the hot form: the star's table, the column-sum partials, its phases (and times); two workgroups per CU
size_t sp_assemble_sums_lds(int Kp, int covpts, int temporal) {
  return sizeof(double) * (4 * (size_t)(covpts + 4) + 8 + (size_t)Kp * (temporal == SP_TEMPORAL_NONE ? 1 : 2));
}

int sp_launch_assemble_sums(int S, int K, int M, int Kp, const double *theta, const double *t,
                            const sp_star *stars, int covpts, const double *ptab, const double *meanvar,
                            int temporal, const double *flux, double *sys, hipStream_t st, double *part,
                            int lazy_nfull, int *nflat) {
  const size_t lds = sp_assemble_sums_lds(Kp, covpts, temporal);
  if (lds > SP_ASM_LDS_MAX || !ptab || !part) return SP_ERR_INVALID;
  const int ntr = Kp / 64, ntiles = ntr * (ntr + 1) / 2;
  // tiles per workgroup: a function of nothing but the build -- the column sums of a strip segment are added
  // in the segment's order, and a star's value must not depend on how many stars share its launch
  static const int per = [] {
    const char *e = getenv("SP_ASM_TILES");
    const int v = e ? atoi(e) : 17;
    return v < 1 ? 1 : v;
  }();
  int nchunk = (ntiles + per - 1) / per;
  if (nchunk > SP_ASM_MAX_CHUNKS) nchunk = SP_ASM_MAX_CHUNKS;
  if (ntiles > 65535) return SP_ERR_INVALID;
  const AsmChunks chunks = asm_chunks(ntr, nchunk);
  dim3 grid(nchunk, S);
  if ((size_t)nchunk > (size_t)ntr * K) return SP_ERR_INVALID;   // (part holds ntr K doubles per star)
  if (nflat) *nflat = nchunk;       // part[s][0 .. nchunk): one partial total per workgroup
#ifdef SP_PROBE
  // (what would a free assembly be worth?  results are garbage; timing probe only)
  static const bool skip = getenv("SP_PROBE_SKIP_ASM") != nullptr;
  if (skip) return SP_OK;
#endif
#define SP_ASMS(TK)                                                                                     \
  do {                                                                                                  \
    allow_big_lds(assemble_sums_kernel<TK>);                                                            \
    hipLaunchKernelGGL((assemble_sums_kernel<TK>), grid, dim3(256), lds, st, K, M, Kp, theta, t, stars, \
                       covpts, ptab, meanvar, flux, sys, (long)Kp, (long)Kp * Kp, ntr, part, lazy_nfull, chunks); \
  } while (0)
  if (temporal == SP_TEMPORAL_NONE) SP_ASMS(SP_TEMPORAL_NONE);
  else if (temporal == SP_TEMPORAL_MATERN32) SP_ASMS(SP_TEMPORAL_MATERN32);
  else if (temporal == SP_TEMPORAL_EXPSQUARED) SP_ASMS(SP_TEMPORAL_EXPSQUARED);
  else return SP_ERR_INVALID;
#undef SP_ASMS
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_defer_finish(int S, int K, int M, int Kp, const sp_star *stars, const double *meanvar,
                           const double *condmean, int order, double zmax, const double *part,
                           const double *diag, const double *flux, double *sys, void *coef, double *rscal,
                           uint32_t *status, hipStream_t st, int nflat) {
  hipLaunchKernelGGL(defer_finish_kernel, dim3(S), dim3(1024), 0, st, K, M, Kp, Kp / 64, nflat, stars, meanvar,
                     condmean, order, zmax, part, diag, flux, sys, (Coef *)coef, rscal, status);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
