// Batched fp64 "NT" matrix products on the CDNA4 matrix cores:
//
//     C[b] = beta * C[b] + alpha * A[b] . B[b]^T          (beta in {0, 1})
//
// The GEMM-shaped steps of the hot path outside the panel kernel (sp_panel.hip):
//   - the rank-64w trailing update of the blocked Cholesky, C -= X X^T   
//   - the conditional covariance A Sigma_y A^T as two products           
//   - the products of predict / the reverse-mode ops                     (sp_gemm_nt)
//
// Two kernels:
//   mm_nt_kernel<MM2<...>> (sp_mm.h)  full tiles, aligned operands: DMA into an XOR-swizzled LDS ring,
//       16-byte fragment reads feeding two MFMAs, the next slice's fragments read behind this
//       slice's MFMAs.  64 x 64 tiles for lower-triangular updates (no wasted half tiles; 0.62 of
//       the fp64 peak), 128 x 128 for rectangular products (0.79-0.81).
//   gemm_nt_kernel                     everything else (ragged edges, odd leading dimensions):
//       global -> registers -> LDS, 32-deep slices, rows padded to 33 doubles.
//
// Workgroup -> tile mapping is XCD-aware: blockIdx.x % 8 selects the XCD (round-robin dispatch), and
// each XCD is given a contiguous run of (matrix, tile) items -- whole matrices when the batch is a
// multiple of 8 -- so that the row panels neighbouring tiles re-read stay in that XCD's 4 MiB L2
// (sp_xcd_decode, sp_tile.h).
#include <cstdlib>

#include <type_traits>

#include "sp_internal.h"
#include "sp_tile.h"
#include "sp_cov.h"
#include "sp_mm.h"
#include "sp_stage.h"
#include "sp_paneldiag.h"

typedef double d2 __attribute__((ext_vector_type(2)));

#define GT SP_GT   // tile edge

namespace {

// (A and C may alias row tile by row tile: each workgroup reads its A rows before it stores C.)
__global__ __launch_bounds__(256) void gemm_nt_kernel(
    const double *A, long lda, long strideA, const double *__restrict__ B, long ldb, long strideB,
    double *C, long ldc, long strideC, int Mrows, int Nrows, int Kd, double alpha, int beta,
    int lower_only, int batch, int ntn, int ntiles) {
  constexpr int BK = 32, LDW = BK + 1;
  __shared__ __attribute__((aligned(16))) double smem[2 * GT * LDW];
  double *sA = smem, *sB = smem + GT * LDW;
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
  const double *Ab = A + (size_t)mtx * strideA;
  const double *Bb = B + (size_t)mtx * strideB;
  double *Cb = C + (size_t)mtx * strideC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  d4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = d4{0.0, 0.0, 0.0, 0.0};
  const bool full = row0 + GT <= Mrows && col0 + GT <= Nrows;
  if (beta) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = row0 + 16 * wave + fk + 4 * r, gj = col0 + 16 * n + fr;
        if (full || (gi < Mrows && gj < Nrows)) acc[n][r] = Cb[(size_t)gi * ldc + gj];
      }
  }
  const bool vecA = ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(Ab) & 15) == 0);
  const bool vecB = ((ldb & 1) == 0) && ((reinterpret_cast<uintptr_t>(Bb) & 15) == 0);
  PanelRegs<BK> ra, rb;
  if (Kd > 0) {
    stage_load<BK>(Ab, lda, row0, Mrows, 0, Kd, vecA, ra);
    stage_load<BK>(Bb, ldb, col0, Nrows, 0, Kd, vecB, rb);
  }
  for (int k0 = 0; k0 < Kd; k0 += BK) {
    stage_store<BK>(ra, alpha, sA);
    stage_store<BK>(rb, 1.0, sB);
    __syncthreads();
    if (k0 + BK < Kd) {   // next slice: its loads fly while this slice is multiplied
      stage_load<BK>(Ab, lda, row0, Mrows, k0 + BK, Kd, vecA, ra);
      stage_load<BK>(Bb, ldb, col0, Nrows, k0 + BK, Kd, vecB, rb);
    }
    const double *pa = sA + (16 * wave + fr) * LDW + fk;
    const double *pb = sB + fr * LDW + fk;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      const double a = pa[kk];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[kk], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[16 * LDW + kk], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[32 * LDW + kk], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[48 * LDW + kk], acc[3], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gi = row0 + 16 * wave + fk + 4 * r, gj = col0 + 16 * n + fr;
      if (full || (gi < Mrows && gj < Nrows)) Cb[(size_t)gi * ldc + gj] = acc[n][r];
    }
}

}  // namespace

// ---- pipelined kernel (sp_mm.h): full tiles, Kd a multiple of BK, aligned operands ------------
// C[b] = beta C[b] + alpha A[b] B[b]^T on TM x TN tiles; the C tile is fetched up front and
// added after the loop (its latency hides behind the product).
namespace {
// SGN: alpha is +1 or -1 and the accumulators start from +-C (no second register tile, exact);
// otherwise the C tile waits in registers and is combined after the loop.
#ifdef SP_MM_STAMPS
__device__ long long sp_mm_dbg[8 * 4096];
#endif
#ifndef SP_MM_WAVES
#define SP_MM_WAVES 1      // (probe: 4 = a register budget for four workgroups per CU, with SP_MM_SYRK_NS = 5)
#endif
#ifndef SP_MM_SYRK_NS
#define SP_MM_SYRK_NS 6
#endif
template <class Core, bool SGN>
__global__ __launch_bounds__(256, SP_MM_WAVES) void mm_nt_kernel(
    const double *__restrict__ A, long lda, long strideA, const double *__restrict__ B, long ldb,
    long strideB, double *__restrict__ C, long ldc, long strideC, int Kd, double alpha, int beta,
    int lower_only, int batch, int ntn, int ntiles, int skip00, LazyCov lz, DiagFuse df) {
  constexpr int TM = Core::TM_, TN = Core::TN_;
  static_assert(Core::LDS_DOUBLES >= SP_DIAG_LDS_DOUBLES, "the tile-(0,0) workgroup factors a pivot block in this LDS");
  __shared__ __attribute__((aligned(16))) double lds[Core::LDS_DOUBLES];
#ifdef SP_MM_STAMPS
  const long long wall_begin = wall_clock64(), cyc_begin = __builtin_readcyclecounter();
#endif
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
  // skip00: bit 0 -- tile (0, 0) is not this launch's; bit 1 -- B = A is UPPER triangular by 64 x 64 blocks (an
  // inverse factor, sp_spd_inverse_batched): the product of tile (ti, tj), ti >= tj, starts at column TM ti;
  // bits 8.. -- tile columns from this one on are not wanted (the rows below a factorisation that carry no
  // pivot: their columns are never read)
  const int tj_limit = skip00 >> 8;
  int k_first = (skip00 & 2) ? ti * TM : 0;
  if (df.tri0 >= 0 && TM == 64) {
    // the trailing update with an identity riding along: the rows of tile df.j + ti are zero left of column block
    // (64 (df.j + ti) - tri0) / 64 -- the product over the super-panel's blocks s0 .. starts there
    const int r = 64 * (df.j + ti) - df.tri0, first = r > 0 ? r / 64 - df.s0 : 0;
    if (first > 0) k_first = 64 * first < Kd ? 64 * first : Kd;
  }
  if (tj_limit && tj >= tj_limit) return;
  if ((skip00 & 1) && lower_only && ti == 0 && tj == 0) {
    // tile (0, 0) -- the next pivot block -- carries every update already (the panel kernels keep it
    // up to date); its workgroup factors it beside the products of this launch (sp_paneldiag.h)
    if (df.sys) {
      __builtin_amdgcn_s_setprio(3);
      panel_diag_item(df.sys + (size_t)mtx * df.stride, df.ld, df.j, df.nact,
                      df.img + (size_t)mtx * df.lts + sp_img_off(df.j),
                      df.info ? df.info + mtx : nullptr, lds, threadIdx.x);
    }
    return;
  }
  const double *Ab = A + (size_t)mtx * strideA + (size_t)ti * TM * lda;
  const double *Bb = B + (size_t)mtx * strideB + (size_t)tj * TN * ldb;
  double *Cb = C + (size_t)mtx * strideC + (size_t)ti * TM * ldc + (size_t)tj * TN;
  Core mm;
  mm.init(Ab, lda, Bb, ldb);
  // A diagonal tile of a SYMMETRIC update (skip00 bit 2: B = A, sp_launch_syrk_diag): only its ten blocks on and
  // below the diagonal, re-dealt over the four wavefronts three / three / two / two (sp_mm.h, SymDeal) -- the plain
  // loop multiplies all sixteen, of which nobody reads the upper six (the eager updates, the diagonal block and the
  // next trailing update all stay below the diagonal of a diagonal tile).  Same bits for the ten.
  if constexpr (SGN && Core::MA == 1 && Core::NA == 4 && TM == 64 && TN == 64) {
    if ((skip00 & 4) && lower_only && ti == tj) {
      // (the planned step: a diagonal tile of the FIRST trailing update is formed here, not loaded -- LazyCov.dlazy;
      //  the star's table passes through the LDS stages before the product claims them, as for the tiles below)
      const bool dform = beta && lz.theta && (lz.dlazy & 2);
      const int lane = threadIdx.x & 63;
      sp_star dst{};
      int dnobs = 0;
      if (dform) {
        dst = lz.stars[mtx];
        dnobs = star_nobs(dst, lz.K);
        spline_table_to_lds(lz.ptab + (size_t)mtx * 4 * (lz.covpts + 4), lds, lz.covpts + 4, threadIdx.x);
        __syncthreads();
      }
      const SplineGen dg{lds, 2 * (lz.covpts + 4), 6.283185307179586 / lz.covpts, 1.0 / (6.283185307179586 / lz.covpts),
                         lz.covpts};
      if (!dform) mm.prologue(lds, k_first, Kd);
      auto run = [&](auto wtag) {
        constexpr int W = decltype(wtag)::value;
        using D = typename Core::template SymDeal<W>;
        mm_d4 a3[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          mm_d4 c = mm_d4{0.0, 0.0, 0.0, 0.0};
          if (b < D::NB && dform) {
            int ri[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ri[r] = 64 * (lz.tr0 + ti) + 16 * D::brow(b) + (lane >> 4) + 4 * r;
            lazy_diag_block(lz, mtx, dg, dst, dnobs, ri, 64 * (lz.tc0 + tj) + 16 * D::bcol(b) + (lane & 15), c);
          } else if (b < D::NB && beta) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              c[r] = Cb[(size_t)(16 * D::brow(b) + (lane >> 4) + 4 * r) * ldc + 16 * D::bcol(b) + (lane & 15)];
          }
          a3[b] = alpha < 0.0 ? -c : c;
        }
        if (dform) {
          __syncthreads();                  // (every wavefront has read the table: the stages are the product's now)
          mm.prologue(lds, k_first, Kd);
        }
        mm.template sym_loop<W>(lds, k_first, Kd, a3);
#pragma unroll
        for (int b = 0; b < D::NB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Cb[(size_t)(16 * D::brow(b) + (lane >> 4) + 4 * r) * ldc + 16 * D::bcol(b) + (lane & 15)] =
                alpha < 0.0 ? -a3[b][r] : a3[b][r];
      };
      switch (threadIdx.x >> 6) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        default: run(std::integral_constant<int, 3>{}); break;
      }
      return;
    }
  }
  mm_d4 acc[Core::MA][Core::NA], cin[SGN ? 1 : Core::MA][SGN ? 1 : Core::NA];
  // (first trailing update of a factorisation whose assembly left the tiles below the diagonal
  //  to their first touch: a tile of covariance rows is evaluated, not loaded -- sp_cov.h; the
  //  star's table passes through the LDS stages before the product claims them)
  constexpr bool CAN_LAZY = Core::MA == 1 && Core::NA == 4 && TM == 64 && TN == 64;
  const bool lazy = CAN_LAZY && beta && lz.theta && lz.tr0 + ti > lz.tc0 + tj && lz.tc0 + tj > 0 &&
                    lz.tr0 + ti < lz.nfull;
  mm_d4 cz1[1][4];
  mm_d4 (&cz)[4] = cz1[0];
  if (!lazy) mm.prologue(lds, k_first, Kd);   // the first slices are on their way while the C tile is fetched
  if (lazy) {
    int ri[1][4], cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ri[0][k] = 64 * (lz.tr0 + ti) + mm.acc_row(0, k);
      cj[k] = 64 * (lz.tc0 + tj) + mm.acc_col(k < Core::NA ? k : 0);
    }
    lazy_cov_tiles<1>(lz, mtx, ri, cj, cz1, lds);
    mm.prologue(lds, k_first, Kd);
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
#ifdef SP_MM_STAMPS
  const long long cyc_loop = __builtin_readcyclecounter();
#endif
  mm.loop(lds, k_first, Kd, acc);
#ifdef SP_MM_STAMPS
  const long long cyc_store = __builtin_readcyclecounter();
#endif
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
#ifdef SP_MM_STAMPS
  if (threadIdx.x == 0 && lower_only && df.sys && blockIdx.x < 4096) {
    long long *o = sp_mm_dbg + 8 * blockIdx.x;
    o[0] = (long long)mtx * 4096 + tile; o[1] = lazy; o[2] = wall_begin; o[3] = wall_clock64();
    o[4] = cyc_loop - cyc_begin; o[5] = cyc_store - cyc_loop; o[6] = __builtin_readcyclecounter() - cyc_store;
    o[7] = ((__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 15u) << 8) | ((__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) >> 8) & 255u);
  }
#endif
}


// The symmetric trailing update of a LARGE remainder on 128 x 64 tiles:  T -= X X^T  on the blocks on and below the
// diagonal of an n x n block, n = nb 64.  Row tiles pair the block rows from the bottom up (block row 0 stays
// out when nb is odd: it holds nothing but the pivot block); row tile I (block rows R0 = o + 2 I, R0 + 1, o = nb & 1)
// takes the column blocks J = 0 .. R0 + 1; a wavefront owns 32 rows (MA = 2), i.e. one of the tile's two blocks: a
// block above the diagonal -- (R0, R0 + 1) -- and the pivot block (0, 0) are computed and not stored (2.4 % of the
// blocks at nb = 39).  One more workgroup factors the pivot block (DiagFuse), as in mm_nt_kernel.  Twice the
// flops per operand byte and per barrier of the 64 x 64 tiles: 0.777 against 0.720 of the fp64 peak on a full
// product of the size (tools/attic/mm_tile_bench.py); for small remainders the wasted blocks weigh more than that
// (nb = 8: 40 blocks executed for 35) and the 64 x 64 kernel stays (sp_launch_syrk_diag).
using Syrk128Core = MM2<128, 64, 8, 4, 4>;
#ifndef SP_SYRK128_WGS
#define SP_SYRK128_WGS 1     // (probe: 3 = a register budget for three workgroups per CU)
#endif
__global__ __launch_bounds__(256, SP_SYRK128_WGS) void syrk128_kernel(const double *__restrict__ X, long ld, long stride,
                                                      double *__restrict__ T, int Kd, int batch, int nb, int ntiles,
                                                      LazyCov lz, DiagFuse df) {
  using Core = Syrk128Core;
  static_assert(Core::LDS_DOUBLES >= SP_DIAG_LDS_DOUBLES, "the last workgroup of a star factors a pivot block in this LDS");
  static_assert(Core::MA == 2 && Core::NA == 4, "a wavefront = 32 rows of one block row");
  __shared__ __attribute__((aligned(16))) double lds[Core::LDS_DOUBLES];
  int mtx, tile;
  if (!sp_xcd_decode(blockIdx.x, batch, ntiles, mtx, tile)) return;
  if (tile == 0) {
    // the next pivot block carries every update already (the panel kernels keep it up to date)
    if (df.sys) {
      __builtin_amdgcn_s_setprio(3);
      panel_diag_item(df.sys + (size_t)mtx * df.stride, df.ld, df.j, df.nact,
                      df.img + (size_t)mtx * df.lts + sp_img_off(df.j), df.info ? df.info + mtx : nullptr, lds, threadIdx.x);
    }
    return;
  }
  tile -= 1;
  const int o = nb & 1;
  // tiles before row tile I: I^2 + I (o + 1)
  int I = (int)((sqrt((double)((o + 1) * (o + 1) + 4 * tile)) - (o + 1)) * 0.5);
  while (I * I + I * (o + 1) > tile) --I;
  while ((I + 1) * (I + 1) + (I + 1) * (o + 1) <= tile) ++I;
  const int J = tile - (I * I + I * (o + 1)), R0 = o + 2 * I;
  const double *Xb = X + (size_t)mtx * stride;
  const double *Ab = Xb + (size_t)R0 * 64 * ld, *Bb = Xb + (size_t)J * 64 * ld;
  double *Cb = T + (size_t)mtx * stride + (size_t)R0 * 64 * ld + (size_t)J * 64;
  Core mm;
  mm.init(Ab, ld, Bb, ld);
  // this wavefront's block row and whether its block is wanted
  const int R = R0 + (mm.acc_row(0, 0) >> 6);
  const bool want = R >= J && !(R == 0 && J == 0);
  // (first trailing update of a factorisation whose assembly left the tiles below the diagonal to their first
  //  touch, sp_cov.h: a block of covariance rows is evaluated, not loaded -- per block, so per wavefront; the
  //  evaluation is the workgroup's, the star's table passes through the LDS stages before the product claims them)
  const auto is_lazy = [&](int Rb) {
    return lz.theta && lz.tr0 + Rb > lz.tc0 + J && lz.tc0 + J > 0 && lz.tr0 + Rb < lz.nfull;
  };
  const bool any_lazy = is_lazy(R0) || is_lazy(R0 + 1), my_lazy = is_lazy(R);
  mm_d4 acc[Core::MA][Core::NA];
  if (any_lazy) {
    int ri[Core::MA][4], cj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cj[k] = 64 * (lz.tc0 + J) + mm.acc_col(k);
#pragma unroll
      for (int m = 0; m < Core::MA; ++m) ri[m][k] = 64 * (lz.tr0 + R0) + mm.acc_row(m, k);
    }
    lazy_cov_tiles<Core::MA>(lz, mtx, ri, cj, acc, lds);
  }
  mm.prologue(lds, 0, Kd);
#pragma unroll
  for (int m = 0; m < Core::MA; ++m)
#pragma unroll
    for (int n = 0; n < Core::NA; ++n) {
      mm_d4 c = mm_d4{0.0, 0.0, 0.0, 0.0};
      if (any_lazy && my_lazy) {
        c = acc[m][n];
      } else if (want) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = Cb[(size_t)mm.acc_row(m, r) * ld + mm.acc_col(n)];
      }
      acc[m][n] = -c;
    }
  mm.loop(lds, 0, Kd, acc);
  if (!want) return;
#pragma unroll
  for (int m = 0; m < Core::MA; ++m)
#pragma unroll
    for (int n = 0; n < Core::NA; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cb[(size_t)mm.acc_row(m, r) * ld + mm.acc_col(n)] = -acc[m][n][r];
}

template <class Core>
int mm_launch(const double *A, long lda, long strideA, const double *B, long ldb, long strideB,
              double *C, long ldc, long strideC, int Mrows, int Nrows, int Kd, double alpha, int beta,
              int lower_only, int batch, hipStream_t st, int skip00, const LazyCov *lazy = nullptr,
              const DiagFuse *dfp = nullptr) {
  const LazyCov lz = lazy ? *lazy : LazyCov{};
  const DiagFuse df = dfp ? *dfp : DiagFuse{nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, -1, 0};
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

#ifdef SP_MM_STAMPS
extern "C" int sp_debug_mm_stamps(long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_mm_dbg), sizeof(long long) * n);
}
#endif

static int launch_gemm(const double *A, long lda, long strideA, const double *B, long ldb,
                       long strideB, double *C, long ldc, long strideC, int Mrows, int Nrows,
                       int Kd, double alpha, int beta, int lower_only, int batch, hipStream_t st,
                       int skip00 = 0, const LazyCov *lazy = nullptr) {
  if (Mrows <= 0 || Nrows <= 0 || batch <= 0) return SP_OK;
  if (Kd < 0) return SP_ERR_INVALID;
  const int ntm = (Mrows + GT - 1) / GT, ntn = (Nrows + GT - 1) / GT;
  if (lower_only && ntm != ntn) return SP_ERR_INVALID;
  const int ntiles = lower_only ? ntm * (ntm + 1) / 2 : ntm * ntn;
  const long nblk = sp_xcd_grid(batch, ntiles);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  const bool fast = (Mrows % GT) == 0 && (Nrows % GT) == 0 && (Kd % 32) == 0 && Kd > 0 &&
                    ((lda | ldb) & 1) == 0 && (strideA & 1) == 0 && (strideB & 1) == 0 &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (lazy && lazy->theta) {
    // tiles formed at first touch: only the 64 x 64 pipelined kernel knows how
    if (!fast) return SP_ERR_INVALID;
    return mm_launch<MM2<64, 64, 8, 6, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows,
                                           Kd, alpha, beta, lower_only, batch, st, skip00, lazy);
  }
  if (fast) {
    // 128 x 128 tiles for full products of that granularity (0.80 of peak against 0.73), 64 x 64
    // tiles otherwise (lower-triangular updates: no wasted half tiles)
#ifdef SP_MM_TILE_PROBE
    // (tools/attic/mm_tile_bench.py: the engine's tile shapes against each other on one full product)
    static const int probe = getenv("SP_MM_TILE") ? atoi(getenv("SP_MM_TILE")) : 0;
    if (!lower_only && probe == 1 && (Mrows % 128) == 0)
      return mm_launch<MM2<128, 64, 8, 4, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd, alpha,
                                              beta, lower_only, batch, st, skip00);
    if (!lower_only && probe == 2 && (Mrows % 128) == 0)
      return mm_launch<MM2<128, 64, 8, 5, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd, alpha,
                                              beta, lower_only, batch, st, skip00);
    if (!lower_only && probe == 3 && (Mrows % 128) == 0)
      return mm_launch<MM2<128, 64, 16, 3, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd, alpha,
                                               beta, lower_only, batch, st, skip00);
    if (!lower_only && probe == 4)
      return mm_launch<MM2<64, 64, 8, 6, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd, alpha,
                                             beta, lower_only, batch, st, skip00);
#endif
    if (!lower_only && (Mrows % 128) == 0 && (Nrows % 128) == 0)
      return mm_launch<MM2<128, 128, 8, 4, 2>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows,
                                               Nrows, Kd, alpha, beta, lower_only, batch, st, skip00);
    return mm_launch<MM2<64, 64, 8, 6, 4>>(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows,
                                           Kd, alpha, beta, lower_only, batch, st, skip00);
  }
  if (skip00) return SP_ERR_INVALID;   // (only the pipelined kernel leaves tiles out)
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((unsigned)nblk), dim3(256), 0, st, A, lda, strideA, B, ldb,
                     strideB, C, ldc, strideC, Mrows, Nrows, Kd, alpha, beta, lower_only, batch, ntn, ntiles);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// from how many 64-blocks on a remainder takes the 128 x 64 tiles (environment SP_SYRK128_FROM; 0 = never)
static int g_syrk128_from = -1;
static int syrk128_from() {
  if (g_syrk128_from < 0) {
    const char *e = getenv("SP_SYRK128_FROM");
    g_syrk128_from = e ? atoi(e) : 17;
    if (g_syrk128_from < 0) g_syrk128_from = 0;
  }
  return g_syrk128_from;
}
extern "C" int sp_debug_set_syrk128_from(int blocks) {
  g_syrk128_from = blocks < 0 ? -1 : blocks;      // (-1: back to the environment / default)
  return SP_OK;
}

static int g_syrk_symdiag = -1;
static int syrk_symdiag() {
  if (g_syrk_symdiag < 0) {
    const char *e = getenv("SP_SYRK_SYMDIAG");
    g_syrk_symdiag = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_syrk_symdiag;
}
int sp_syrk_can_form_diag(int nb) {
  const int big_from = syrk128_from();
  return (syrk_symdiag() && !(big_from > 0 && nb >= big_from)) ? 1 : 0;
}
extern "C" int sp_debug_set_syrk_symdiag(int on) {
  g_syrk_symdiag = on < 0 ? -1 : (on ? 1 : 0);    // (-1: back to the environment / default)
  return SP_OK;
}

// C -= X X^T on the lower 64 x 64 tiles of an n x n block, tile (0, 0) skipped -- its workgroup
// factors the pivot block `df` describes instead (sp_cholesky.hip)
int sp_launch_syrk_diag(const double *X, long ld, long stride, double *T, int n, int kd, int batch,
                        hipStream_t st, const LazyCov *lazy, const DiagFuse *df, int tj_limit) {
  if (n <= 0 || batch <= 0) return SP_OK;
  if ((n % GT) || (kd % 16) || kd <= 0 || (ld & 1) || (stride & 1) ||
      ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(T)) & 15))
    return SP_ERR_INVALID;
  // large remainders without an identity riding along: 128 x 64 tiles (syrk128_kernel)
  const int big_from = syrk128_from();
  const int nb = n / GT;
  if (big_from > 0 && nb >= big_from && tj_limit == 0 && (!df || df->tri0 < 0) && (kd % 16) == 0) {
    const int o = nb & 1, nrt = (nb - o) / 2, ntiles = nrt * nrt + nrt * (o + 1) + 1;
    const long nblk = sp_xcd_grid(batch, ntiles);
    if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
    const DiagFuse d = df ? *df : DiagFuse{nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, -1, 0};
    const LazyCov lzv = (lazy && lazy->theta) ? *lazy : LazyCov{};
    hipLaunchKernelGGL(syrk128_kernel, dim3((unsigned)nblk), dim3(256), 0, st, X, ld, stride, T, kd, batch, nb, ntiles, lzv, d);
    SP_LAUNCH_CHECK();
    return SP_OK;
  }
  // (SP_SYRK_SYMDIAG=0 / sp_debug_set_syrk_symdiag(0): the diagonal tiles on the plain loop, all sixteen blocks -- the
  //  same bits below the diagonal)
  const int symdiag = syrk_symdiag() ? 4 : 0;
  return mm_launch<MM2<64, 64, 8, SP_MM_SYRK_NS, 4>>(X, ld, stride, X, ld, stride, T, ld, stride, n, n, kd, -1.0, 1, 1,
                                         batch, st, 1 | symdiag | (tj_limit > 0 ? tj_limit << 8 : 0),
                                         (lazy && lazy->theta) ? lazy : nullptr, df);
}

int sp_launch_gemm_nt(const double *A, long lda, long strideA, const double *B, long ldb,
                      long strideB, double *C, long ldc, long strideC, int Mrows, int Nrows,
                      int Kd, double alpha, int beta, int lower_only, int batch,
                      hipStream_t st, int skip_tile00, const LazyCov *lazy) {
  return launch_gemm(A, lda, strideA, B, ldb, strideB, C, ldc, strideC, Mrows, Nrows, Kd,
                     alpha, beta, lower_only, batch, st, skip_tile00, lazy);
}
