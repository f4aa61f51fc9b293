// Batched fp64 Cholesky, triangular solves and the likelihood reduction
//  for gfx950.
//
// Blocked factorisation, panels of SP_NB = 64 columns, super-panels of 8 panels (driver:
// sp_launch_cholesky_groups below):
//   panel_kernel (sp_panel.hip)   one launch per panel: left-looking product, triangular solve on the
//                  matrix cores, eager update of the coming diagonal tiles, the next pivot block
//                  factored in the tail of the first item (diag_block, sp_diag.h / sp_paneldiag.h);
//   sp_launch_syrk_diag (sp_gemm.hip)  the rank-512 trailing update between two super-panels, whose
//                  tile-(0, 0) workgroup factors the next super-panel's first pivot block.
// The systems are padded to a multiple of 64 rows and carry the residual
// vectors as EXTRA ROWS below the matrix: factoring
//     [ C   . ]          gives          [ L   . ]
//     [ r^T . ]                         [ y^T . ] ,   y = L^-1 r,
// so r^T C^-1 r = |y|^2 falls out of the factorisation and no separate
// triangular solve is needed for the likelihood.
#include <cstdio>
#include <vector>

#include "sp_internal.h"

#include "sp_tile.h"
#include "sp_paneldiag.h"
#include "sp_reduce.h"

#define DLD 65  // padded row length of a diagonal block in cho_solve_kernel

namespace {

// one workgroup per star (sp_reduce.h)
__global__ __launch_bounds__(256) void lnlike_reduce_kernel(
    const double *__restrict__ sys, long ld, long stride, int K, int M,
    const int32_t *__restrict__ info, double *__restrict__ lnlike,
    uint32_t *__restrict__ status, uint32_t *__restrict__ status_out,
    const sp_star *__restrict__ stars, const RedCoef *__restrict__ coef, const double *__restrict__ rscal,
    int dvec) {
  __shared__ double red[48];
  const int s = blockIdx.x;
  lnlike_reduce_body<false>(sys + (size_t)s * stride, ld, K, M, info ? info + s : nullptr, lnlike + s,
                            status ? status + s : nullptr, status_out ? status_out + s : nullptr,
                            stars ? stars + s : nullptr, coef ? coef + s : nullptr,
                            rscal ? rscal + (size_t)s * (SP_RSCAL_HEAD + M) : nullptr, dvec, red, threadIdx.x);
}

// copy a batch of K x K matrices into zero/identity padded Kp x Kp systems
__global__ __launch_bounds__(256) void pad_in_kernel(const double *__restrict__ A,
                                                     int K, long lda, long strideA,
                                                     double *__restrict__ sys, int Kp,
                                                     long strideS, int M,
                                                     const double *__restrict__ resid, int ident) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Kp) return;
  double v = 0.0;
  if (i < K && j < K)
    v = A[(size_t)s * strideA + (size_t)i * lda + j];
  else if (i >= K && i < K + M && j < K && resid)
    v = resid[((size_t)s * M + (i - K)) * K + j];
  else if (ident && i >= K + M)
    v = (j == i - (K + M) && j < K) ? 1.0 : 0.0;   // the identity riding along below the M residual rows (no unit diagonal
                                             // of their own: those rows are never pivots, and their columns beyond K
                                             // must stay zero for the product that forms the inverse)
  else if (i == j)
    v = 1.0;
  sys[(size_t)s * strideS + (size_t)i * Kp + j] = v;
}

// copy the lower factor back, zero the strict upper triangle, NaN on failure
__global__ __launch_bounds__(256) void pad_out_kernel(const double *__restrict__ sys,
                                                      int Kp, long strideS,
                                                      double *__restrict__ A, int K,
                                                      long lda, long strideA,
                                                      const int32_t *__restrict__ info) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  double v = j <= i ? sys[(size_t)s * strideS + (size_t)i * Kp + j] : 0.0;
  if (info && info[s]) v = __builtin_nan("");
  A[(size_t)s * strideA + (size_t)i * lda + j] = v;
}

// x = (L L^T)^-1 b (math.py:97-100).  grid (nrhs, batch), one right-hand side
// per workgroup; b is [K, nrhs] row-major per matrix.  Blocked substitution:
// the 64x64 diagonal blocks are solved by one wavefront with lane shuffles, the
// off-diagonal updates are one row (forward) / one column (backward) per thread.
//
// mode: 0 both sweeps, 1 forward only (L y = b), 2 backward only (L^T x = b) -- the two
// Solve ops of math.py:97-100 taken one at a time, which their reverse mode needs
// (math.py:40-72).  Element (i, rhs) of a right-hand side lives at B[i * rs + rhs * cs].
__global__ __launch_bounds__(256) void cho_solve_kernel(
    const double *__restrict__ Lall, int K, long ldl, long strideL,
    double *__restrict__ Ball, long strideB, long rs, long cs, int mode) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int Kr = ((K + 63) / 64) * 64;
  double *x = lds;        // Kr
  double *sL = lds + Kr;  // 64 * DLD
  const double *L = Lall + (size_t)blockIdx.y * strideL;
  double *B = Ball + (size_t)blockIdx.y * strideB + (size_t)blockIdx.x * cs;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < Kr; i += 256) x[i] = i < K ? B[(size_t)i * rs] : 0.0;
  const int nb = Kr / 64;
  __syncthreads();
  // forward: L y = b
  for (int blk = 0; blk < nb && mode != 2; ++blk) {
    const int c0 = blk * 64, n = K - c0 < 64 ? K - c0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      sL[r * DLD + c] = (c0 + r < K && c <= r) ? L[(size_t)(c0 + r) * ldl + c0 + c]
                                               : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = x[c0 + lane];
      for (int c = 0; c < n; ++c) {
        const double xc = __shfl(v, c, 64) / sL[c * DLD + c];
        if (lane == c)
          v = xc;
        else if (lane > c)
          v -= sL[lane * DLD + c] * xc;
      }
      x[c0 + lane] = v;
    }
    __syncthreads();
    for (int j = c0 + 64 + tid; j < K; j += 256) {
      const double *row = L + (size_t)j * ldl + c0;
      double acc = x[j];
      for (int k = 0; k < n; ++k) acc -= row[k] * x[c0 + k];
      x[j] = acc;
    }
    __syncthreads();
  }
  // backward: L^T x = y
  for (int blk = nb - 1; blk >= 0 && mode != 1; --blk) {
    const int c0 = blk * 64, n = K - c0 < 64 ? K - c0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      sL[r * DLD + c] = (c0 + r < K && c <= r) ? L[(size_t)(c0 + r) * ldl + c0 + c]
                                               : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = x[c0 + lane];
      for (int c = n - 1; c >= 0; --c) {
        const double xc = __shfl(v, c, 64) / sL[c * DLD + c];
        if (lane == c)
          v = xc;
        else if (lane < c)
          v -= sL[c * DLD + lane] * xc;
      }
      x[c0 + lane] = v;
    }
    __syncthreads();
    for (int j = tid; j < c0; j += 256) {
      double acc = x[j];
      for (int k = 0; k < n; ++k) acc -= L[(size_t)(c0 + k) * ldl + j] * x[c0 + k];
      x[j] = acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < K; i += 256) B[(size_t)i * rs] = x[i];
}

// out = in^T per matrix (K x K, row-major; out has leading dimension K), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ in, long ldi,
                                                        long stridei, double *__restrict__ out,
                                                        int K) {
  __shared__ double tile[64][65];
  const double *A = in + (size_t)blockIdx.z * stridei;
  double *O = out + (size_t)blockIdx.z * K * K;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  for (int r = r4; r < 64; r += 4)
    tile[r][c] = (i0 + r < K && j0 + c < K) ? A[(size_t)(i0 + r) * ldi + j0 + c] : 0.0;
  __syncthreads();
  for (int r = r4; r < 64; r += 4)
    if (j0 + r < K && i0 + c < K) O[(size_t)(j0 + r) * K + i0 + c] = tile[c][r];
}

// keep one triangle of every K x K matrix, scale its diagonal:  A <- tri(A), diag *= dscale
// (tril / triu of Solve.L_op, math.py:66-69; tril_and_halve_diagonal of Cholesky.L_op)
__global__ __launch_bounds__(256) void tri_mask_kernel(double *__restrict__ A, int K, int upper,
                                                       double dscale) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)K * K) return;
  const int i = e / K, j = e % K;
  double *p = A + (size_t)blockIdx.y * K * K + e;
  if (i == j)
    *p *= dscale;
  else if ((j > i) != (upper != 0))
    *p = 0.0;
}

// C_bar = tril(S + S^T) - diag(S); all NaN when the factor itself is NaN (on_error = "nan")
__global__ __launch_bounds__(256) void chol_rev_finish_kernel(const double *__restrict__ S,
                                                              const double *__restrict__ L, long ldl,
                                                              long strideL, double *__restrict__ out,
                                                              int K) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)K * K) return;
  const int i = e / K, j = e % K;
  const double *Sb = S + (size_t)blockIdx.y * K * K;
  double v = 0.0;
  if (i > j)
    v = Sb[(size_t)i * K + j] + Sb[(size_t)j * K + i];
  else if (i == j)
    v = Sb[e];
  const double l00 = L[(size_t)blockIdx.y * strideL];
  if (l00 != l00) v = __builtin_nan("");
  out[(size_t)blockIdx.y * K * K + e] = v;
}

}  // namespace

// ---- drivers and launchers -----------------------------------------------------

// ---- the driver: ONE launch per panel (sp_panel.hip) -------------------------------------------------
// Super-panels of w pivot blocks as above.  Launch j = T items (left-looking product over the q
// panels of the super-panel before it, solve on the matrix cores, eager update of the coming
// diagonal tiles).  Who factors pivot block j (a 12-17 us latency chain):
//   j = 0                      a launch of its own (64 workgroups);
//   q = 0, later super-panels  the tile-(0, 0) workgroup of the trailing update (sp_launch_syrk_diag);
//   q > 0                      the workgroup that solved row tile j in launch j - 1, at its end (SP_PANEL_TAILD).
// Look-ahead (SP_PANEL_LA, default on): the first tile of launch j + 1 -- tile (j + 2, j + 1), whose solve
// and eager update stand between this launch and pivot block j + 2 -- is brought up to date with the
// column blocks s0 .. j - 1 by an item of THIS launch; launch j + 1 then only adds the rank-64 update
// with column block j: the first item's product leaves the critical path, and the diagonal block
// behind it runs under the products of the launch's other items.
static int cholesky_panel2(sp_handle *h, int ngroups, const sp_chol_group *grp, int K, int Kp, int w) {
  const long ld = Kp, stride = (long)Kp * Kp, lts = sp_lt_stride(Kp);
  const int nsteps = (K + SP_NB - 1) / SP_NB, ntile_all = Kp / SP_NB;
  const bool la_on = h->look_ahead != 0;
  // row tiles a launch of pivot block j has to take (all of them, unless an identity rides along: tri0)
  const int tri0 = grp[0].tri0;
  auto ntile_of = [&](int j) {
    if (tri0 < 0) return ntile_all;
    const int need = (tri0 + SP_NB * (j + 1) + SP_NB - 1) / SP_NB;
    return need < ntile_all ? need : ntile_all;
  };
  const bool fuse_reduce = sp_panel_fuses_reduce(h, K, Kp);
  auto nact_of = [&](int j) { return K - j * SP_NB < SP_NB ? K - j * SP_NB : SP_NB; };
  for (int s0 = 0; s0 < nsteps; s0 += w) {
    {
      SpProfScope sp_scope(ngroups == 1 ? h : nullptr, grp[0].st, SP_PROF_PANELS, 0.0, 0);
      for (int q = 0; q < w && s0 + q < nsteps; ++q) {
        const int j = s0 + q;
        const int ntile = ntile_of(j);
        int last = s0 + w;
        if (last > nsteps - 1) last = nsteps - 1;
        const int neager = last > j ? last - j : 0;
        const double rows = (double)(ntile - j - 1) * SP_NB;
        for (int g = 0; g < ngroups; ++g) {
          const sp_chol_group &G = grp[g];
          const LazyCov *lzp = (G.lazy.theta && s0 == 0 && !G.lazy.no_panels) ? &G.lazy : nullptr;
          // Work of the launch: left-looking product, triangular solve, eager rank-64 updates, the diagonal block.
          // flp: on the PADDED system, every block 64 wide, all (ntile - j - 1) 64 rows below it -- what the launch
          // executes (rounds 1-5 reported this as the algorithmic count: 7 % high at K = 1000, Kp = 1024).
          // fl: the ALGORITHMIC count -- the K cadences' rows and the M riding residual rows (the forward solve of
          // math.py:98) below a block of nact = min(64, K - 64 j) columns; padding rows and the normalisation's
          // riding rows are the implementation's.
          const double flp = (double)G.S * (2.0 * rows * 64 * (q * 64.0) + rows * 64 * 64 +
                                            neager * 64.0 * 64 * 64 + 64.0 * 64 * 64 / 3);
          const double na = nact_of(j), rk0 = (double)(K - (j + 1) * SP_NB);
          const double rowsK = (rk0 > 0 ? rk0 : 0.0) + (rows > 0 ? (double)G.red.M : 0.0);
          double eag = 0.0;      // eager updates: the diagonal blocks j + 1 .. last take a rank-nact update each
          for (int i2 = j + 1; i2 <= last && i2 < nsteps; ++i2) eag += (double)nact_of(i2) * nact_of(i2) * na;
          const double fl = (double)G.S * (2.0 * rowsK * na * (q * 64.0) + rowsK * na * na + eag + na * na * na / 3);
          const bool d_alone = j == 0;                         // block 0: nobody before it
          // block j + 1 in the tail of this launch (same super-panel: the next one's first block
          // belongs to the trailing update)
          const bool tail = rows > 0 && j + 1 < nsteps && q + 1 < w;
          const bool la = la_on && q >= 1 && j + 2 < ntile && j + 1 < nsteps && q + 1 < w && rows > 0;
          const bool first_la = la_on && q >= 2 && rows > 0;   // (launch j - 1 qualified: q - 1 >= 1, j + 1 < ntile)
          const int nl = (d_alone && !G.block0_done && rows > 0) ? 2 : 1;
          if (!d_alone && rows <= 0) {   // the last pivot block with no row tile below it: factored in launch j - 1's tail
            sp_scope.add(fl, 0, flp);
            continue;
          }
          SpProfScope prof(h, G.st, SP_PROF_CHAIN, fl, nl, flp);
          SpProfScope prof1(h, G.st, SP_PROF_PANEL_LAUNCH, fl, nl, flp);
          sp_scope.add(fl, nl, flp);
          int rc = SP_OK;
          if (d_alone && !G.block0_done)
            rc = sp_launch_panel2(h->panel_layout | (tri0 >= 0 ? (2 | (tri0 << 8)) : 0), nullptr, G.sys, ld, stride, G.S, ntile, j, s0, nact_of(j), 0, last, SP_PANEL_D,
                                  h->ncu, G.invL, lts, G.info, G.st, nullptr);
          const int what = rows > 0 ? (SP_PANEL_T | (tail ? SP_PANEL_TAILD : 0) | (la ? SP_PANEL_LA : 0) |
                                       (first_la ? SP_PANEL_FIRSTLA : 0))
                                    : 0;
          // (the launch whose tail factors the LAST pivot block carries the reduction, if there is one)
          const SpReduceArgs *red = (tail && j + 1 == nsteps - 1 && fuse_reduce) ? &G.red : nullptr;
#ifdef SP_PROBE
          // (what is each kind of launch worth with several steps in flight?  results are garbage; timing probe only)
          static const int probe_skip_p = getenv("SP_PROBE_SKIP_PANELS") ? atoi(getenv("SP_PROBE_SKIP_PANELS")) : 0;
          if (probe_skip_p == 3 || (probe_skip_p == 1 && s0 == 0) || (probe_skip_p == 2 && s0 > 0)) continue;
#endif
          if (rc == SP_OK && what)
            rc = sp_launch_panel2(h->panel_layout | (tri0 >= 0 ? (2 | (tri0 << 8)) : 0), red, G.sys, ld, stride, G.S, ntile, j, s0, nact_of(j), tail ? nact_of(j + 1) : 0,
                                  last, what, h->ncu, G.invL, lts, G.info, G.st, lzp);
          if (rc != SP_OK) return rc;
        }
      }
    }   // (sp_scope ends here: the trailing update has its own pair)
    const int jE = s0 + w, cE = jE * SP_NB;
    if (cE < K) {
      // (an identity riding along: rows up to what column block jE - 1 has reached, pivot columns only)
      const int n = ntile_of(jE - 1) * SP_NB - cE, kd = w * SP_NB, cS = s0 * SP_NB;
      const int tj_limit = tri0 >= 0 ? nsteps - jE : 0;
      for (int g = 0; g < ngroups; ++g) {
        const sp_chol_group &G = grp[g];
        LazyCov lzv = G.lazy;
        lzv.tr0 = lzv.tc0 = jE;
        DiagFuse df{G.sys, ld, stride, jE, nact_of(jE), G.invL, lts, G.info, tri0, s0};
        // (algorithmic: the rows of the K cadences and the M residual rows; executed: every row of the padded system)
        const double nK = (double)(K + G.red.M - cE) > 0 ? (double)(K + G.red.M - cE) : 0.0;
        SpProfScope prof(h, G.st, SP_PROF_SYRK, (double)G.S * nK * (nK + 1) * kd, 1, (double)G.S * (double)n * (n + 1) * kd);
#ifdef SP_PROBE
        static const bool probe_skip_mm = getenv("SP_PROBE_SKIP_MM2") != nullptr;
        if (probe_skip_mm) continue;
#endif
        int rc = sp_launch_syrk_diag(G.sys + (size_t)cE * ld + cS, ld, stride, G.sys + (size_t)cE * ld + cE,
                                     n, kd, G.S, G.st, (G.lazy.theta && s0 == 0) ? &lzv : nullptr, &df, tj_limit);
        if (rc != SP_OK) return rc;
      }
    }
  }
  return SP_OK;
}

// In-place factorisation of S padded systems (Kp x Kp, ld = Kp): the leading
// K x K part is factored, rows K..Kp-1 only receive the triangular solve.
// Two-level blocking: panels (64 columns) are grouped in super-panels of w panels.  Inside a
// super-panel a block column is brought up to date left-looking (ONE narrow product over the q
// previous panels of the group, k = 64 q) by the launch that solves it; the big trailing matrix is
// touched once per super-panel with a rank-64w update instead of w rank-64 updates (at k = 64 that
// update is HBM-bound: 8 flop per byte of C traffic; k = 64 w divides the traffic by w).
static int superpanel_of(const sp_handle *h, int K) {
  const int nsteps = (K + SP_NB - 1) / SP_NB;
  return h->superpanel > 0 ? h->superpanel : (nsteps >= 16 ? 8 : 4);
}

// The last pivot block is partial and its row tile holds the rows below the matrix (nsteps == ntile), and it is
// factored in the tail of launch nsteps - 2 (it is not the first block of a super-panel).
int sp_superpanel_width(const sp_handle *h, int K) { return superpanel_of(h, K); }

bool sp_panel_fuses_reduce(const sp_handle *h, int K, int Kp) {
  const int nsteps = (K + SP_NB - 1) / SP_NB, ntile = Kp / SP_NB;
  return h->fuse_reduce && nsteps >= 2 && nsteps == ntile && (nsteps - 1) % superpanel_of(h, K) != 0;
}

int sp_launch_cholesky_groups(sp_handle *h, int ngroups, const sp_chol_group *grp, int K,
                              int Kp) {
  if (!h) return SP_ERR_INVALID;
  const int nsteps = (K + SP_NB - 1) / SP_NB;
  // panels per super-panel: wider super-panels raise the arithmetic intensity of the trailing update
  // at the price of more left-looking work per block column; measured (round 1):
  // K = 1000 (16 panels) w = 2 / 4 / 6 / 8 / 12 / 16 -> 1.17 / 1.10 / 1.085 / 1.08 / 1.12 / 1.14 ms per
  // step; K = 3000 (47 panels): 8 best as well
  (void)nsteps;
  return cholesky_panel2(h, ngroups, grp, K, Kp, superpanel_of(h, K));
}

int sp_launch_cholesky_systems(sp_handle *h, double *sys, int S, int K, int Kp,
                               int32_t *info, double *invL, hipStream_t st) {
  sp_chol_group g{sys, info, invL, S, st, LazyCov{}, SpReduceArgs{}};
  return sp_launch_cholesky_groups(h, 1, &g, K, Kp);
}

int sp_launch_lnlike_reduce(const double *sys, int S, int K, int M, int Kp,
                            const int32_t *info, double *lnlike, uint32_t *status,
                            hipStream_t st, uint32_t *status_out, const sp_star *stars,
                            const void *defer_coef, const double *rscal, int dvec) {
  if (defer_coef && !rscal) return SP_ERR_INVALID;
  hipLaunchKernelGGL(lnlike_reduce_kernel, dim3(S), dim3(256), 0, st, sys,
                     (long)Kp, (long)Kp * Kp, K, M, info, lnlike, status, status_out, stars,
                     (const RedCoef *)defer_coef, rscal, dvec);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_pad_in(const double *A, int K, long lda, long strideA, double *sys,
                     int Kp, int M, const double *resid, int S, hipStream_t st, int ident) {
  hipLaunchKernelGGL(pad_in_kernel, dim3((Kp + 255) / 256, Kp, S), dim3(256), 0,
                     st, A, K, lda, strideA, sys, Kp, (long)Kp * Kp, M, resid, ident);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_pad_out(const double *sys, int Kp, double *A, int K, long lda,
                      long strideA, const int32_t *info, int S, hipStream_t st) {
  hipLaunchKernelGGL(pad_out_kernel, dim3((K + 255) / 256, K, S), dim3(256), 0, st,
                     sys, Kp, (long)Kp * Kp, A, K, lda, strideA, info);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_tri_solve(const double *L, int K, long ldl, long strideL, double *B,
                        long strideB, long rs, long cs, int nrhs, int batch, int mode,
                        hipStream_t st) {
  const int Kr = ((K + 63) / 64) * 64;
  const size_t lds = sizeof(double) * ((size_t)Kr + 64 * DLD);
  if (lds > 150 * 1024) return SP_ERR_INVALID;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cho_solve_kernel),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(cho_solve_kernel, dim3(nrhs, batch), dim3(256), lds, st, L, K,
                     ldl, strideL, B, strideB, rs, cs, mode);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_cho_solve(const double *L, int K, long ldl, long strideL, double *B,
                        int nrhs, int batch, hipStream_t st) {
  return sp_launch_tri_solve(L, K, ldl, strideL, B, (long)K * nrhs, nrhs, 1, nrhs, batch, 0, st);
}

int sp_launch_transpose(const double *in, long ldi, long stridei, double *out, int K, int batch,
                        hipStream_t st) {
  const int nt = (K + 63) / 64;
  hipLaunchKernelGGL(transpose_kernel, dim3(nt, nt, batch), dim3(256), 0, st, in, ldi, stridei, out,
                     K);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_tri_mask(double *A, int K, int batch, int upper, double dscale, hipStream_t st) {
  const long n = (long)K * K;
  hipLaunchKernelGGL(tri_mask_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, st, A,
                     K, upper, dscale);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_chol_rev_finish(const double *S, const double *L, long ldl, long strideL, double *out,
                              int K, int batch, hipStream_t st) {
  const long n = (long)K * K;
  hipLaunchKernelGGL(chol_rev_finish_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0,
                     st, S, L, ldl, strideL, out, K);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
