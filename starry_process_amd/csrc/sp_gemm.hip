// Batched fp64 "NT" matrix product on the CDNA4 matrix cores:
//
//     C[b] = beta * C[b] + alpha * A[b] . B[b]^T          (beta in {0, 1})
//
// This one kernel carries the two GEMM-shaped steps of the hot path:
//   - the trailing update of the blocked Cholesky, C -= X X^T   
//   - the conditional covariance A Sigma_y A^T as two products  
//
// Tiling: one 256-thread workgroup (4 wavefronts, one per SIMD)
// owns a 64 x 64 tile of C; wavefront w owns rows 16w..16w+15 as four
// 16 x 16 accumulators of v_mfma_f64_16x16x4_f64.  A and B^T row panels are
// staged through LDS in 32-deep slices, rows padded to 34 doubles (68 dwords =
// 4 mod 64 banks) so that the per-lane fragment reads (16 rows x 2 k per
// 32-lane half) are bank-conflict free with ds_read_b64.
//
// Workgroup -> tile mapping is XCD-aware: blockIdx.x % 8 selects the XCD
// (round-robin dispatch), and each XCD is given a contiguous run of (matrix, tile)
// items -- whole matrices when the batch is a multiple of 8 -- so that the row
// panels neighbouring tiles re-read stay in that XCD's 4 MiB L2 (sp_xcd_decode).
#include <cstdlib>

#include "sp_internal.h"
#include "sp_tile.h"
#include "sp_cov.h"
#include "sp_mm.h"
#include "sp_wt.h"
#include "sp_stage.h"
#include "sp_paneldiag.h"

typedef double d2 __attribute__((ext_vector_type(2)));

#define GT SP_GT   // tile edge
// one-launch-per-panel kernel: workgroups per CU asked of the compiler (0: whatever it needs), and
// whether the image of the solve is fetched ahead of the product (32 more registers)
#ifndef SP_PANEL_WGS
#define SP_PANEL_WGS 0
#endif
#ifndef SP_PANEL_PREFETCH
#define SP_PANEL_PREFETCH 0
#endif

#ifdef SP_PANEL_TRACE
// (variant build only, tools/ab_build.sh -DSP_PANEL_TRACE: wall-clock stamps of the workgroups of
//  star 0 in the one-launch-per-panel kernel; rows = launches in order, read by sp_debug_panel_trace)
__device__ long long g_panel_trace[64 * 4 * 16];
__device__ int g_panel_trace_n;
#define PT_STAMP(k)                                                                              \
  do {                                                                                           \
    if (FUSE == 2 && mtx == 0 && (ti == 0 || ti == 3) && threadIdx.x == 0 && pt_row < 64)         \
      g_panel_trace[(pt_row * 4 + (ti == 0 ? 0 : 1)) * 16 + (k)] = wall_clock64();                \
  } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif

namespace {

// (A and C may alias: the triangular solve X = P L^-T runs in place, each
//  workgroup reads its whole A row-tile before it stores the same C tile.)
//
// Template parameters: BK = depth of one LDS stage (32 or 64); DEFER_C = issue
// the C-tile loads before staging but consume them only after the MFMA loop, so
// their HBM latency hides behind the panel staging and the matrix work.
//
// FUSE == 1: tile (0, 0) of the launch is the diagonal block of the next panel.
// The workgroup that owns it does not stop after its tile: it keeps the updated
// block in LDS and factors it (diag_block, sp_diag.h), writing L_d and L_d^T.
// The latency-bound factorisation then runs concurrently with the other
// tiles of the same launch instead of as a kernel of its own between launches.
// ABL (debug only, tools/microbench.py): ablations that locate the bound of the
// trailing update -- 1: operand slices fetched from global memory once, 2: also no
// LDS staging stores / barriers in the loop, 3: also no LDS fragment reads (MFMA
// issue only), 4: everything but the C tile load / store.  Results are garbage.
template <int BK, bool DEFER_C, int FUSE, int ABL = 0, bool FAST = false>
__global__ __launch_bounds__(256, (FUSE == 2 && SP_PANEL_WGS > 0) ? SP_PANEL_WGS : 1) void gemm_nt_kernel(
    const double *A, long lda, long strideA,
    const double *__restrict__ B, long ldb, long strideB, double *C,
    long ldc, long strideC, int Mrows, int Nrows, int Kd, double alpha,
    int beta, int lower_only, int batch, int ntm, int ntn, int ntiles, int nact,
    double *invL_all, int32_t *info, int skip00, const double *lt_in, long lts, LazyCov lz) {
  // Padded LDS row of BK + 1 doubles.  hipcc fuses the per-k-step fragment reads
  // into ds_read2_b64, which is banked mod 32 dwords in 16-lane groups: an ODD
  // row length puts the 16 rows of a group on 16 distinct bank pairs.  (An even
  // row length of BK + 2 is conflict-free only for plain ds_read_b64 and cost
  // 42% extra LDS cycles here: SQ_LDS_BANK_CONFLICT, profiles/r01_v3_pmc.txt.)
  constexpr int LDW = BK + 1;
  constexpr int NLDS = FUSE ? (2 * GT * LDW > SP_TILE_LDS_DOUBLES ? 2 * GT * LDW
                                                                  : SP_TILE_LDS_DOUBLES)
                            : 2 * GT * LDW;
  __shared__ __attribute__((aligned(16))) double smem[NLDS];
  double *sA = smem, *sB = smem + GT * LDW;

  // XCD-aware decode: blocks b and b+8 share an XCD (sp_tile.h)
  int mtx, tile;
  if (!sp_xcd_decode(blockIdx.x, batch, ntiles, mtx, tile)) return;
  int ti, tj;
  if (lower_only) {
    ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    tj = tile - ti * (ti + 1) / 2;
  } else {
    ti = tile / ntn;
    tj = tile % ntn;
  }
  const int row0 = ti * GT, col0 = tj * GT;
#ifdef SP_PANEL_TRACE
  const int pt_row = (FUSE == 2) ? __hip_atomic_load(&g_panel_trace_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 64;
#endif
  PT_STAMP(0);
  // (one-launch-per-panel mode: the first diagonal block of the next super-panel has
  //  been updated eagerly AND factored by the last panel launch -- leave it alone)
  if (FUSE == 0 && skip00 && lower_only && ti == 0 && tj == 0) return;
  // the workgroup that will factor the diagonal block is the critical path of the
  // launch: let its wavefronts win the issue arbitration on their SIMDs
  if ((FUSE == 1 && ti == 0 && tj == 0) || (FUSE == 2 && ti == 0)) __builtin_amdgcn_s_setprio(3);
  const double *Ab = A + (size_t)mtx * strideA;
  const double *Bb = B + (size_t)mtx * strideB;
  double *Cb = C + (size_t)mtx * strideC;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;

  d4 acc[4], cin[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    acc[n] = d4{0.0, 0.0, 0.0, 0.0};
    cin[n] = d4{0.0, 0.0, 0.0, 0.0};
  }
  const bool full = FAST || (row0 + GT <= Mrows && col0 + GT <= Nrows);
  // (one launch per panel, first super-panel: a tile below the diagonal made of covariance rows
  //  has not been written by the assembly -- its entries are evaluated below, sp_cov.h)
  const bool lazy = FUSE == 2 && beta && lz.theta && lz.tr0 + ti > lz.tc0 + tj && lz.tc0 + tj > 0 &&
                    lz.tr0 + ti < lz.nfull;
  if (beta && ABL != 4 && !lazy) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = row0 + 16 * wave + fk + 4 * r, gj = col0 + 16 * n + fr;
        if (full || (gi < Mrows && gj < Nrows)) cin[n][r] = Cb[(size_t)gi * ldc + gj];
      }
    if (!DEFER_C) {
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = cin[n];
    }
  }

  // (one launch per panel) the image of the solve is in memory since the previous launch: its
  // fetch rides behind the product instead of standing between the product and the solve
  LtRegs lt_pre;
  if (FUSE == 2 && FAST && SP_PANEL_PREFETCH) lt_load(lt_pre, lt_in + (size_t)mtx * lts);

  const bool vecA = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(Ab) & 15) == 0);
  const bool vecB = ((ldb & 1) == 0) && ((reinterpret_cast<uintptr_t>(Bb) & 15) == 0);

  // skip00: the diagonal block of the next panel already carries every update (the panel
  // solves applied them eagerly, trsm_quad_kernel): its workgroup goes straight to the
  // factorisation
  const int Kloop = (FUSE == 1 && skip00 && ti == 0 && tj == 0) ? 0 : Kd;
  PanelRegs<BK> ra, rb;
  if (Kloop > 0) {
    if (FAST) {
      stage_load_fast<BK>(Ab, lda, row0, 0, ra);
      stage_load_fast<BK>(Bb, ldb, col0, 0, rb);
    } else {
      stage_load<BK>(Ab, lda, row0, Mrows, 0, Kd, vecA, ra);
      stage_load<BK>(Bb, ldb, col0, Nrows, 0, Kd, vecB, rb);
    }
  } else {
#pragma unroll
    for (int i = 0; i < (int)(sizeof(ra.v) / sizeof(ra.v[0])); ++i) ra.v[i] = rb.v[i] = d2{0.0, 0.0};
  }
  if (lazy) {
    // the first operand slices are on their way to registers; the LDS is free until they are
    // stored: the star's table passes through it and the tile is evaluated meanwhile
    int ri[4], cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ri[k] = GT * (lz.tr0 + ti) + 16 * wave + fk + 4 * k;
      cj[k] = GT * (lz.tc0 + tj) + 16 * k + fr;
    }
    PT_STAMP(8);
    lazy_cov_tile(lz, mtx, ri, cj, cin, smem);
    PT_STAMP(9);
    if (!DEFER_C) {
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = cin[n];
    }
  }
  if (ABL == 2 || ABL == 3) {
    stage_store<BK>(ra, alpha, sA);
    stage_store<BK>(rb, 1.0, sB);
    __syncthreads();
  }
  double fa = ra.v[0].x, fb0 = rb.v[0].x, fb1 = rb.v[0].y, fb2 = rb.v[1].x, fb3 = rb.v[1].y;
  for (int k0 = 0; k0 < Kloop; k0 += BK) {
    if (ABL < 2 || ABL == 4) {
      stage_store<BK>(ra, alpha, sA);
      stage_store<BK>(rb, 1.0, sB);
      __syncthreads();
    }
    if (k0 + BK < Kd && (ABL == 0 || ABL == 4)) {  // next slice: loads fly while this slice is multiplied
      if (FAST) {
        stage_load_fast<BK>(Ab, lda, row0, k0 + BK, ra);
        stage_load_fast<BK>(Bb, ldb, col0, k0 + BK, rb);
      } else {
        stage_load<BK>(Ab, lda, row0, Mrows, k0 + BK, Kd, vecA, ra);
        stage_load<BK>(Bb, ldb, col0, Nrows, k0 + BK, Kd, vecB, rb);
      }
    }
    const double *pa = sA + (16 * wave + fr) * LDW + fk;
    const double *pb = sB + fr * LDW + fk;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      const double a = ABL == 3 ? fa : pa[kk];
      const double b0 = ABL == 3 ? fb0 : pb[kk];
      const double b1 = ABL == 3 ? fb1 : pb[16 * LDW + kk];
      const double b2 = ABL == 3 ? fb2 : pb[32 * LDW + kk];
      const double b3 = ABL == 3 ? fb3 : pb[48 * LDW + kk];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[3], 0, 0, 0);
    }
    if (ABL < 2 || ABL == 4) __syncthreads();
  }
  if (DEFER_C && beta) {
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] += cin[n];
  }
  PT_STAMP(1);

  if (FUSE == 2) {
    // One launch per panel: the updated 64 x 64 tile never goes back to memory unsolved.
    //   accumulators -> LDS -> four lanes per row -> X = P L_d^-T against the image the
    //   previous launch left (lt_in) -> solved rows stored; the leading row tiles then apply
    //   their eager update to their own diagonal tile, and the first of them, whose diagonal
    //   tile is the next pivot block and is now complete, factors it (image -> invL_all, the
    //   other parity, while the rest of this launch still reads lt_in).
    constexpr int XW = 65;
    double *sT = smem;                         // 64 x 65 doubles <= SP_TILE_LDS_DOUBLES
#if !SP_PANEL_MFMA_SOLVE
    const int tid = threadIdx.x, q = tid & 3, lrow = tid >> 2;
#endif
#if SP_PANEL_MFMA_SOLVE == 2
    // X = T L_d^-T as a block substitution on the matrix cores (diag_solve_operand, sp_tile.h):
    //   X_c = (T_c - sum_{k<c} X_k L_ck^T) M_c^T,   c = 0..3 (16 columns each),
    // with the 16 x 16 blocks L_ck and the leaf inverses M_c read as MFMA operands straight from
    // memory (L2) into registers.  A wavefront works on its own 16 rows throughout: T_c is its
    // accumulator n = c, X_k goes through ITS rows of the LDS tile to become an A operand -- no
    // workgroup barrier, no image staged in LDS, 40 MFMAs instead of 64 dependent vector steps.
    const double *Wop = lt_in + (size_t)mtx * lts + SP_LT_IMG + (size_t)fr * 64 + fk;
    double bq[40];
    {
      int e = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k <= c; ++k)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) bq[e++] = Wop[(16 * c) * 64 + 16 * k + 4 * s4];
    }
    {
      double *mine = sT + (16 * wave) * XW;         // this wavefront's 16 rows of the tile
      int e = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d4 u = acc[c];
#pragma unroll
        for (int k = 0; k < c; ++k)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            u = __builtin_amdgcn_mfma_f64_16x16x4f64(-mine[fr * XW + 16 * k + 4 * s4 + fk], bq[e++], u, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(fk + 4 * r) * XW + 16 * c + fr] = u[r];
        d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          x = __builtin_amdgcn_mfma_f64_16x16x4f64(mine[fr * XW + 16 * c + 4 * s4 + fk], bq[e++], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          mine[(fk + 4 * r) * XW + 16 * c + fr] = x[r];
          const int gi = row0 + 16 * wave + fk + 4 * r;
          if (FAST || gi < Mrows) Cb[(size_t)gi * ldc + col0 + 16 * c + fr] = x[r];
        }
      }
    }
    const int neager = skip00;
    if (ti >= neager) return;
    __syncthreads();                           // the solved tile, row layout, complete in LDS
#elif SP_PANEL_MFMA_SOLVE
    // X = T L_d^-T as a product: L_d^-T (row k, column n; zero for n < k) was left behind the
    // image by the workgroup that factored the block (diag_inverse, sp_tile.h).  Its fragments
    // come straight from memory (L2) into registers while T goes through LDS to become the A
    // operand; the 16 x 16 blocks below the diagonal are skipped: 40 MFMAs per wavefront instead
    // of 64 dependent vector steps that kept the LDS pipe busy for every resident workgroup.
    const double *iv = lt_in + (size_t)mtx * lts + SP_LT_IMG + fk * 64 + fr;
    double bq[40];
    {
      int e = 0;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            if (n >= kb) bq[e++] = iv[(16 * kb + 4 * s4) * 64 + 16 * n];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) sT[(16 * wave + fk + 4 * r) * XW + 16 * n + fr] = acc[n][r];
    __syncthreads();
    d4 xo[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) xo[n] = d4{0.0, 0.0, 0.0, 0.0};
    {
      const double *pa = sT + (16 * wave + fr) * XW + fk;
      int e = 0;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const double a = pa[16 * kb + 4 * s4];
#pragma unroll
          for (int n = 0; n < 4; ++n)
            if (n >= kb) xo[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq[e++], xo[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = row0 + 16 * wave + fk + 4 * r;
        if (FAST || gi < Mrows) Cb[(size_t)gi * ldc + col0 + 16 * n + fr] = xo[n][r];
      }
    const int neager = skip00;
    if (ti >= neager) return;
    __syncthreads();                           // T has been read by every wavefront
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) sT[(16 * wave + fk + 4 * r) * XW + 16 * n + fr] = xo[n][r];
    __syncthreads();
#else
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) sT[(16 * wave + fk + 4 * r) * XW + 16 * n + fr] = acc[n][r];
    if (!(FAST && SP_PANEL_PREFETCH)) lt_load(lt_pre, lt_in + (size_t)mtx * lts);
    const LtRegs &lt = lt_pre;
    __syncthreads();
    double x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[2 * i] = sT[lrow * XW + 8 * i + 2 * q];
      x[2 * i + 1] = sT[lrow * XW + 8 * i + 2 * q + 1];
    }
    __syncthreads();
    lt_store(lt, sT, sT + 4096);
    __syncthreads();
    PT_STAMP(2);
    const bool valid = row0 + lrow < Mrows;
    double *prow = Cb + (size_t)(row0 + (valid ? lrow : 0)) * ldc + col0 + 2 * q;
    quad_solve_store(x, sT, sT + 4096, prow, valid);
    PT_STAMP(3);
    const int neager = skip00;
    if (ti >= neager) return;
    __syncthreads();
    {
      double *row = sT + lrow * XW + 2 * q;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        row[8 * i] = x[2 * i];
        row[8 * i + 1] = x[2 * i + 1];
      }
    }
    __syncthreads();
#endif
    PT_STAMP(3);
    // my own diagonal tile: rows / columns (GT + row0 ..) relative to the block column
    double *D = Cb + (size_t)row0 * ldc + GT + row0;
    d4 dac[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dac[n][r] = D[(size_t)(16 * wave + fk + 4 * r) * ldc + 16 * n + fr];
    {
      const double *pa = sT + (16 * wave + fr) * XW + fk;
      const double *pb = sT + fr * XW + fk;
#pragma unroll
      for (int kk = 0; kk < 64; kk += 4) {
        const double a = -pa[kk];
        dac[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[kk], dac[0], 0, 0, 0);
        dac[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[16 * XW + kk], dac[1], 0, 0, 0);
        dac[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[32 * XW + kk], dac[2], 0, 0, 0);
        dac[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[48 * XW + kk], dac[3], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        D[(size_t)(16 * wave + fk + 4 * r) * ldc + 16 * n + fr] = dac[n][r];
    PT_STAMP(4);
    if (ti > 0 || nact <= 0) return;
    // the next pivot block: complete now -- factor it (nact = its active columns; rows and
    // columns beyond them keep the updated values just stored)
    __syncthreads();
    double *sD = smem, *sRd = smem + 64 * BLD;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int li = 16 * wave + fk + 4 * r, lj = 16 * n + fr;
        double v = (li < nact && lj < nact) ? dac[n][r] : (li == lj ? 1.0 : 0.0);
        if (lj > li) v = 0.0;
        sD[li * BLD + lj] = v;
      }
    __syncthreads();
    PT_STAMP(5);
    const int notpd = diag_block(sD, sRd, invL_all + (size_t)mtx * lts, nullptr, threadIdx.x,
                                 SP_PANEL_MFMA_SOLVE == 1 ? invL_all + (size_t)mtx * lts + SP_LT_IMG : nullptr);
    PT_STAMP(6);
    if (notpd && info) info[mtx] = 1;
    {
      const int cj = (threadIdx.x & 15) * 4, ri = threadIdx.x >> 4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = ri + 16 * pass;
        double *dst = D + (size_t)r * ldc + cj;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r < nact && cj + e <= r) dst[e] = sD[r * BLD + cj + e];
      }
    }
#if SP_PANEL_MFMA_SOLVE == 2
    // ... and the operand of the next launch's solves (L's blocks, the leaves inverted), behind the image
    diag_solve_operand(sD, sRd, invL_all + (size_t)mtx * lts + SP_LT_IMG);
#elif SP_PANEL_MFMA_SOLVE
    // (its L_d^-T for the next launch's solves was formed inside diag_block)
#else
    if (nact < GT) {
      // partial last block: the rows of this tile below the active ones (residual rows,
      // padding) carry every update already (the eager updates cover the whole tile) and are
      // solved against the block just factored, here, instead of by a launch of their own
      // (identity padding: their columns >= nact stay as they are)
      const int tid2 = threadIdx.x, q2 = tid2 & 3, lrow2 = tid2 >> 2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile and diag_block's image are in memory
      __syncthreads();                                   // ... and L has been copied out of sD
      LtRegs lt2;
      lt_load(lt2, invL_all + (size_t)mtx * lts);
      double x2[16];
      {
        const double *prow = D + (size_t)lrow2 * ldc + 2 * q2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const d2v v = *reinterpret_cast<const d2v *>(prow + 8 * i);
          x2[2 * i] = v.x;
          x2[2 * i + 1] = v.y;
        }
      }
      lt_store(lt2, sT, sT + 4096);
      __syncthreads();
      quad_solve_store(x2, sT, sT + 4096, D + (size_t)lrow2 * ldc + 2 * q2, lrow2 >= nact);
    }
#endif
    PT_STAMP(7);
#ifdef SP_PANEL_TRACE
    if (mtx == 0 && threadIdx.x == 0) atomicAdd(&g_panel_trace_n, 1);
#endif
    return;
  }

#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = row0 + 16 * wave + fk + 4 * r, gj = col0 + 16 * n + fr;
      if ((ABL != 4 || acc[n][r] == 123.456) && (full || (gi < Mrows && gj < Nrows)))
        Cb[(size_t)gi * ldc + gj] = acc[n][r];
    }

  if (FUSE == 1 && ti == 0 && tj == 0) {
    // the updated tile is the next diagonal block: factor it right here
    double *sD = smem, *sRd = smem + 64 * BLD;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int li = 16 * wave + fk + 4 * r, lj = 16 * n + fr;
        double v = (li < nact && lj < nact) ? acc[n][r] : (li == lj ? 1.0 : 0.0);
        if (lj > li) v = 0.0;
        sD[li * BLD + lj] = v;
      }
    __syncthreads();
    const int notpd = diag_block(sD, sRd, invL_all + (size_t)mtx * lts);
    if (notpd && info) info[mtx] = 1;
    const int cj = (threadIdx.x & 15) * 4, ri = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      double *dst = Cb + (size_t)r * ldc + cj;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (cj + e <= r && r < nact) dst[e] = sD[r * BLD + cj + e];
    }
  }
}

}  // namespace


// ---- pipelined kernel (sp_mm.h): full tiles, Kd a multiple of BK, aligned operands ------------
// C[b] = beta C[b] + alpha A[b] B[b]^T on TM x TN tiles; the C tile is fetched up front and
// added after the loop (its latency hides behind the product).
namespace {
// SGN: alpha is +1 or -1 and the accumulators start from +-C (no second register tile, exact);
// otherwise the C tile waits in registers and is combined after the loop.
template <class Core, bool SGN>
__global__ __launch_bounds__(256) void mm_nt_kernel(
    const double *__restrict__ A, long lda, long strideA, const double *__restrict__ B, long ldb,
    long strideB, double *__restrict__ C, long ldc, long strideC, int Kd, double alpha, int beta,
    int lower_only, int batch, int ntn, int ntiles, int skip00, LazyCov lz, DiagFuse df) {
  constexpr int TM = Core::TM_, TN = Core::TN_;
  static_assert(Core::LDS_DOUBLES >= SP_DIAG_LDS_DOUBLES, "the tile-(0,0) workgroup factors a pivot block in this LDS");
  __shared__ __attribute__((aligned(16))) double lds[Core::LDS_DOUBLES];
  int mtx, tile;
  if (!sp_xcd_decode(blockIdx.x, batch, ntiles, mtx, tile)) return;
  int ti, tj;
  if (lower_only) {
    ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    tj = tile - ti * (ti + 1) / 2;
  } else {
    ti = tile / ntn;
    tj = tile % ntn;
  }
  if (skip00 && lower_only && ti == 0 && tj == 0) {
    // tile (0, 0) -- the next pivot block -- carries every update already (the panel kernels keep it
    // up to date); its workgroup factors it beside the products of this launch (sp_paneldiag.h)
    if (df.sys) {
      __builtin_amdgcn_s_setprio(3);
      panel_diag_item(df.sys + (size_t)mtx * df.stride, df.ld, df.j, df.nact,
                      df.img + (size_t)mtx * df.lts + (size_t)(df.j & 1) * 2 * SP_LT_IMG,
                      df.info ? df.info + mtx : nullptr, lds, threadIdx.x);
    }
    return;
  }
  const double *Ab = A + (size_t)mtx * strideA + (size_t)ti * TM * lda;
  const double *Bb = B + (size_t)mtx * strideB + (size_t)tj * TN * ldb;
  double *Cb = C + (size_t)mtx * strideC + (size_t)ti * TM * ldc + (size_t)tj * TN;
  Core mm;
  mm.init(Ab, lda, Bb, ldb);
  mm_d4 acc[Core::MA][Core::NA], cin[SGN ? 1 : Core::MA][SGN ? 1 : Core::NA];
  // (first trailing update of a factorisation whose assembly left the tiles below the diagonal
  //  to their first touch: a tile of covariance rows is evaluated, not loaded -- sp_cov.h; the
  //  star's table passes through the LDS stages before the product claims them)
  constexpr bool CAN_LAZY = Core::MA == 1 && Core::NA == 4 && TM == 64 && TN == 64;
  const bool lazy = CAN_LAZY && beta && lz.theta && lz.tr0 + ti > lz.tc0 + tj && lz.tc0 + tj > 0 &&
                    lz.tr0 + ti < lz.nfull;
  mm_d4 cz[4];
  if (!lazy) mm.prologue(lds, 0, Kd);   // the first slices are on their way while the C tile is fetched
  if (lazy) {
    int ri[4], cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ri[k] = 64 * (lz.tr0 + ti) + mm.acc_row(0, k);
      cj[k] = 64 * (lz.tc0 + tj) + mm.acc_col(k < Core::NA ? k : 0);
    }
    lazy_cov_tile(lz, mtx, ri, cj, cz, lds);
    mm.prologue(lds, 0, Kd);
  }
#pragma unroll
  for (int m = 0; m < Core::MA; ++m)
#pragma unroll
    for (int n = 0; n < Core::NA; ++n) {
      mm_d4 c = mm_d4{0.0, 0.0, 0.0, 0.0};
      if (lazy) {
        c = cz[n & 3];
      } else if (beta) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = Cb[(size_t)mm.acc_row(m, r) * ldc + mm.acc_col(n)];
      }
      if (SGN) {
        acc[m][n] = alpha < 0.0 ? -c : c;
      } else {
        acc[m][n] = mm_d4{0.0, 0.0, 0.0, 0.0};
        cin[m][n] = c;
      }
    }
  mm.loop(lds, 0, Kd, acc);
#pragma unroll
  for (int m = 0; m < Core::MA; ++m)
#pragma unroll
    for (int n = 0; n < Core::NA; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v;
        if (SGN)
          v = alpha < 0.0 ? -acc[m][n][r] : acc[m][n][r];
        else
          v = fma(alpha, acc[m][n][r], cin[m][n][r]);
        Cb[(size_t)mm.acc_row(m, r) * ldc + mm.acc_col(n)] = v;
      }
}


// ---- wave-tile kernel (sp_wt.h): each wavefront a 64 x 64 tile fed straight from L2 -----------
// A workgroup is a 2 x 2 arrangement of wave tiles (a 128 x 128 super-tile: the two wavefronts of
// a tile row read the same A panel, those of a tile column the same B panel, within one CU's L1).
// lower_only: super-tiles on or below the diagonal; inside a diagonal super-tile the wavefront
// above the diagonal has nothing to do, and a diagonal wave tile of a symmetric update (same_ab:
// A and B are the same rows) loads its panel once and forms only the blocks on or below its own
// diagonal.  Tiles beyond an odd tile count are skipped by their wavefront.
template <bool SGN>
__global__ __launch_bounds__(256, 2) void wt_nt_kernel(
    const double *__restrict__ A, long lda, long strideA, const double *__restrict__ B, long ldb,
    long strideB, double *__restrict__ C, long ldc, long strideC, int ntm, int ntn, int Kd,
    double alpha, int beta, int lower_only, int same_ab, int batch, int nsn, int nsuper,
    int skip00) {
  int mtx, st;
  if (!sp_xcd_decode(blockIdx.x, batch, nsuper, mtx, st)) return;
  int si, sj;
  if (lower_only) {
    si = (int)((sqrt(8.0 * st + 1.0) - 1.0) * 0.5);
    while (si * (si + 1) / 2 > st) --si;
    while ((si + 1) * (si + 2) / 2 <= st) ++si;
    sj = st - si * (si + 1) / 2;
  } else {
    si = st / nsn;
    sj = st % nsn;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ti = 2 * si + (wave >> 1), tj = 2 * sj + (wave & 1);
  if (ti >= ntm || tj >= ntn) return;
  if (lower_only && tj > ti) return;
  if (skip00 && lower_only && ti == 0 && tj == 0) return;
  const double *Ab = A + (size_t)mtx * strideA + (size_t)ti * 64 * lda;
  const double *Bb = B + (size_t)mtx * strideB + (size_t)tj * 64 * ldb;
  double *Cb = C + (size_t)mtx * strideC + (size_t)ti * 64 * ldc + (size_t)tj * 64;
  wt_d4 acc[4][4];
  (void)same_ab;
  if (SGN && beta) {   // (one uniform branch around ALL the loads: a select per element serialises them)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = Cb[(size_t)wt_row(m, r) * ldc + wt_col(n)];
    if (alpha < 0.0) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = -acc[m][n];
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = wt_d4{0.0, 0.0, 0.0, 0.0};
  }
  WT<false> w;
  w.init(Ab, lda, Bb, ldb);
  w.run(Kd, acc);
  if (SGN) {
    if (alpha < 0.0) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = -acc[m][n];
    }
  } else if (beta) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[m][n][r] = fma(alpha, acc[m][n][r], Cb[(size_t)wt_row(m, r) * ldc + wt_col(n)]);
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = alpha * acc[m][n];
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cb[(size_t)wt_row(m, r) * ldc + wt_col(n)] = acc[m][n][r];
}

static int wt_launch(const double *A, long lda, long strideA, const double *B, long ldb, long strideB,
                     double *C, long ldc, long strideC, int Mrows, int Nrows, int Kd, double alpha,
                     int beta, int lower_only, int batch, hipStream_t st, int skip00) {
  const int ntm = Mrows / 64, ntn = Nrows / 64;
  const int nsm = (ntm + 1) / 2, nsn = (ntn + 1) / 2;
  const int nsuper = lower_only ? nsm * (nsm + 1) / 2 : nsm * nsn;
  const long nblk = sp_xcd_grid(batch, nsuper);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  const int same_ab = (A == B && lda == ldb && strideA == strideB) ? 1 : 0;
  if (alpha != 1.0 && alpha != -1.0) return SP_ERR_INVALID;   // (general alpha: mm_nt_kernel)
  hipLaunchKernelGGL((wt_nt_kernel<true>), dim3((unsigned)nblk), dim3(256), 0, st, A, lda, strideA, B,
                     ldb, strideB, C, ldc, strideC, ntm, ntn, Kd, alpha, beta, lower_only, same_ab,
                     batch, nsn, nsuper, skip00);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

template <class Core>
int mm_launch(const double *A, long lda, long strideA, const double *B, long ldb, long strideB,
              double *C, long ldc, long strideC, int Mrows, int Nrows, int Kd, double alpha, int beta,
              int lower_only, int batch, hipStream_t st, int skip00, const LazyCov *lazy = nullptr,
              const DiagFuse *dfp = nullptr) {
  const LazyCov lz = lazy ? *lazy : LazyCov{};
  const DiagFuse df = dfp ? *dfp : DiagFuse{nullptr, 0, 0, 0, 0, nullptr, 0, nullptr};
  const int ntm = Mrows / Core::TM_, ntn = Nrows / Core::TN_;
  const int ntiles = lower_only ? ntm * (ntm + 1) / 2 : ntm * ntn;
  const long nblk = sp_xcd_grid(batch, ntiles);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  if (alpha == 1.0 || alpha == -1.0)
    hipLaunchKernelGGL((mm_nt_kernel<Core, true>), dim3((unsigned)nblk), dim3(256), 0, st,
                       A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Kd, alpha, beta, lower_only,
                       batch, ntn, ntiles, skip00, lz, df);
  else
    hipLaunchKernelGGL((mm_nt_kernel<Core, false>), dim3((unsigned)nblk), dim3(256), 0, st,
                       A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Kd, alpha, beta, lower_only,
                       batch, ntn, ntiles, skip00, lz, df);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
}  // namespace

// tile shape of the pipelined kernel (SP_MM, or sp_debug_set_mm_variant for the microbenchmarks);
// 0 = the register-staged gemm_nt_kernel
static int g_mm_variant = -1;
void sp_set_mm_variant(int v) { g_mm_variant = v; }

static int launch_gemm(const double *A, long lda, long strideA, const double *B, long ldb,
                       long strideB, double *C, long ldc, long strideC, int Mrows, int Nrows,
                       int Kd, double alpha, int beta, int lower_only, int batch, int fuse,
                       int nact, double *invL, long lts, int32_t *info, hipStream_t st, int skip00 = 0,
                       const LazyCov *lazy = nullptr) {
  if (Mrows <= 0 || Nrows <= 0 || batch <= 0) return SP_OK;
  if (Kd < 0) return SP_ERR_INVALID;
  const int ntm = (Mrows + GT - 1) / GT, ntn = (Nrows + GT - 1) / GT;
  if (lower_only && ntm != ntn) return SP_ERR_INVALID;
  const int ntiles = lower_only ? ntm * (ntm + 1) / 2 : ntm * ntn;
  const long nblk = sp_xcd_grid(batch, ntiles);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  static int variant = -1;
  if (variant < 0) {
    const char *e = getenv("SP_GEMM_VARIANT");
    variant = e ? atoi(e) : 0;
  }
  const bool fast = (Mrows % GT) == 0 && (Nrows % GT) == 0 && (Kd % 32) == 0 && Kd > 0 &&
                    ((lda | ldb) & 1) == 0 && (strideA & 1) == 0 && (strideB & 1) == 0 &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
#define SP_GO_FAST(FD)                                                                       \
  hipLaunchKernelGGL((gemm_nt_kernel<32, false, FD, 0, true>), dim3((unsigned)nblk), dim3(256), 0, \
                     st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd,  \
                     alpha, beta, lower_only, batch, ntm, ntn, ntiles, nact, invL, info, skip00, nullptr, lts, LazyCov{})
#define SP_GO(BK, DC, FD)                                                              \
  hipLaunchKernelGGL((gemm_nt_kernel<BK, DC, FD>), dim3((unsigned)nblk), dim3(256), 0, st, \
                     A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd, \
                     alpha, beta, lower_only, batch, ntm, ntn, ntiles, nact, invL, info, skip00, nullptr, lts, LazyCov{})
  static int abl = -1;
  if (abl < 0) {
    const char *e = getenv("SP_GEMM_ABL");
    abl = e ? atoi(e) : 0;
  }
#define SP_GO_ABL(N)                                                                          \
  hipLaunchKernelGGL((gemm_nt_kernel<32, false, 0, N>), dim3((unsigned)nblk), dim3(256), 0, st, \
                     A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd,       \
                     alpha, beta, lower_only, batch, ntm, ntn, ntiles, nact, invL, info, skip00, nullptr, lts, LazyCov{})
  if (g_mm_variant < 0) {
    const char *e = getenv("SP_MM");
    g_mm_variant = e ? atoi(e) : 11;
  }
  const int mmv = g_mm_variant;
  if (lazy && lazy->theta) {
    // tiles formed at first touch: only the 64 x 64 pipelined kernel knows how
    if (!(fast && !fuse && (Kd % 16) == 0)) return SP_ERR_INVALID;
    return mm_launch<MM2<64, 64, 8, 6, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows,
                                           Kd, alpha, beta, lower_only, batch, st, skip00, lazy);
  }
  if (fast && !fuse && abl == 0 && mmv > 0) {
    // pipelined kernels (sp_mm.h); tile shape by SP_MM (tools/microbench.py compares them)
#define SP_MM_GO(TM, TN, BK, NS, WR)                                                              \
  return mm_launch<MM<TM, TN, BK, NS, WR>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, \
                                       Nrows, Kd, alpha, beta, lower_only, batch, st, skip00)
    if (mmv == 9 && (Kd % 16) == 0 && (alpha == 1.0 || alpha == -1.0))
      return wt_launch(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd, alpha, beta,
                       lower_only, batch, st, skip00);
#define SP_MM2_GO(TM, TN, BK, NS, WR)                                                              \
  return mm_launch<MM2<TM, TN, BK, NS, WR>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, \
                                            Nrows, Kd, alpha, beta, lower_only, batch, st, skip00)
    if (mmv == 10) SP_MM2_GO(64, 64, 16, 4, 4);
    // default (11): 128 x 128 tiles for full products of that granularity (0.80 of peak against
    // 0.73), 64 x 64 tiles otherwise (lower-triangular updates: no wasted half tiles)
    if ((mmv == 12 || (mmv == 11 && !lower_only)) && (Mrows % 128) == 0 && (Nrows % 128) == 0)
      SP_MM2_GO(128, 128, 8, 4, 2);
    if (mmv == 11) SP_MM2_GO(64, 64, 8, 6, 4);
#undef SP_MM2_GO
    const bool big = (Mrows % 128) == 0 && (Nrows % 128) == 0;
    if (mmv == 3 && big) SP_MM_GO(128, 128, 8, 4, 2);
    if (mmv == 8 && big) SP_MM_GO(128, 128, 8, 3, 2);
    if (mmv == 5 && big) SP_MM_GO(128, 128, 16, 3, 2);
    if (mmv == 4 && (Mrows % 128) == 0 && !lower_only) SP_MM_GO(128, 64, 16, 3, 4);
    if (mmv == 2) SP_MM_GO(64, 64, 32, 3, 4);
    if (mmv == 7) SP_MM_GO(64, 64, 8, 6, 4);
    if (mmv == 1) SP_MM_GO(64, 64, 16, 4, 4);
    SP_MM_GO(64, 64, 16, 3, 4);
#undef SP_MM_GO
  }
  if (abl > 0 && !fuse) {
    switch (abl) {
      case 1: SP_GO_ABL(1); break;
      case 2: SP_GO_ABL(2); break;
      case 3: SP_GO_ABL(3); break;
      default: SP_GO_ABL(4); break;
    }
  } else if (fuse && fast) {
    SP_GO_FAST(1);
  } else if (fuse) {
    SP_GO(32, false, 1);
  } else if (fast && variant == 0) {
    SP_GO_FAST(0);
  } else if (fast && variant == 2 && (Kd % 64) == 0) {
    hipLaunchKernelGGL((gemm_nt_kernel<64, false, 0, 0, true>), dim3((unsigned)nblk), dim3(256), 0,
                       st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd,
                       alpha, beta, lower_only, batch, ntm, ntn, ntiles, nact, invL, info, skip00, nullptr, lts, LazyCov{});
  } else {
    switch (variant) {
      case 1: SP_GO(32, true, 0); break;
      case 2: SP_GO(64, false, 0); break;
      case 3: SP_GO(64, true, 0); break;
      default: SP_GO(32, false, 0); break;
    }
  }
#undef SP_GO
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// One launch per panel (FUSE = 2): rows r1.. of block column c0 are updated with the Kd
// columns of the panels before it in the super-panel (A: those rows, B: the 64 rows of the
// pivot block), solved against lt_in, eagerly applied to the leading `neager` diagonal
// tiles, and the first of those is factored (next_nact > 0) into lt_out.
// (debug) stamps of the one-launch-per-panel kernel, variant builds with -DSP_PANEL_TRACE only:
// reset (out == null) or copy out 64 x 4 x 16 int64
int sp_debug_panel_trace(long long *out) {
#ifdef SP_PANEL_TRACE
  if (!out) {
    static long long zeros[64 * 4 * 16];
    int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_panel_trace), zeros, sizeof(zeros)) != hipSuccess) return SP_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_panel_trace_n), &z, sizeof(z)) != hipSuccess) return SP_ERR_HIP;
    return SP_OK;
  }
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_trace), sizeof(long long) * 64 * 4 * 16) != hipSuccess)
    return SP_ERR_HIP;
  return SP_OK;
#else
  (void)out;
  return SP_ERR_INVALID;
#endif
}

int sp_launch_panel(const double *A, long lda, const double *B, long ldb, double *C, long ldc,
                    long stride, int Mrows, int Kd, int batch, const double *lt_in, double *lt_out,
                    long lts, int neager, int next_nact, int32_t *info, hipStream_t st,
                    const LazyCov *lazy) {
  if (Mrows <= 0 || batch <= 0) return SP_OK;
  const LazyCov lz = lazy ? *lazy : LazyCov{};
  const int ntm = (Mrows + GT - 1) / GT, ntn = 1, ntiles = ntm;
  const long nblk = sp_xcd_grid(batch, ntiles);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  const bool fast = (Mrows % GT) == 0 && (Kd % 32) == 0 && Kd > 0 && ((lda | ldb) & 1) == 0 &&
                    (stride & 1) == 0 &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (fast)
    hipLaunchKernelGGL((gemm_nt_kernel<32, false, 2, 0, true>), dim3((unsigned)nblk), dim3(256), 0,
                       st, A, lda, stride, B, ldb, stride, C, ldc, stride, Mrows, GT, Kd, -1.0, 1, 0,
                       batch, ntm, ntn, ntiles, next_nact, lt_out, info, neager, lt_in, lts, lz);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<32, false, 2, 0, false>), dim3((unsigned)nblk), dim3(256), 0,
                       st, A, lda, stride, B, ldb, stride, C, ldc, stride, Mrows, GT, Kd, -1.0, 1, 0,
                       batch, ntm, ntn, ntiles, next_nact, lt_out, info, neager, lt_in, lts, lz);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// C -= X X^T on the lower 64 x 64 tiles of an n x n block, tile (0, 0) skipped -- its workgroup
// factors the pivot block `df` describes instead (round-3 driver, sp_cholesky.hip)
int sp_launch_syrk_diag(const double *X, long ld, long stride, double *T, int n, int kd, int batch,
                        hipStream_t st, const LazyCov *lazy, const DiagFuse *df) {
  if (n <= 0 || batch <= 0) return SP_OK;
  if ((n % GT) || (kd % 16) || kd <= 0 || (ld & 1) || (stride & 1) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(T)) & 15))
    return SP_ERR_INVALID;
  return mm_launch<MM2<64, 64, 8, 6, 4>>(X, ld, stride, X, ld, stride, T, ld, stride, n, n, kd, -1.0, 1, 1,
                                         batch, st, 1, (lazy && lazy->theta) ? lazy : nullptr, df);
}

int sp_launch_gemm_nt(const double *A, long lda, long strideA, const double *B, long ldb,
                      long strideB, double *C, long ldc, long strideC, int Mrows, int Nrows,
                      int Kd, double alpha, int beta, int lower_only, int batch,
                      hipStream_t st, int skip_tile00, const LazyCov *lazy) {
  return launch_gemm(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd,
                     alpha, beta, lower_only, batch, 0, 0, nullptr, 0, nullptr, st, skip_tile00, lazy);
}

// Update (beta = 1) whose tile (0, 0) is the next diagonal block: that tile's
// workgroup also factors it (nact active columns) and writes L_d^-1 / info.
int sp_launch_gemm_nt_diag(const double *A, long lda, long strideA, const double *B, long ldb,
                           long strideB, double *C, long ldc, long strideC, int Mrows,
                           int Nrows, int Kd, double alpha, int lower_only, int batch,
                           int nact, double *invL, long lts, int32_t *info, hipStream_t st,
                           int skip00) {
  return launch_gemm(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd,
                     alpha, 1, lower_only, batch, 1, nact, invL, lts, info, st, skip00);
}

