// Strip solve of the recursive Cholesky driver (sp_cholesky.hip):
//
//     X = A L^-T     A: the rows [r0, r0 + 64 nrt) x the columns [c0, c0 + 64 nb) of every system,
//                    L: the factored nb-block lower triangle on those columns (in the same system)
//
// One long-lived workgroup per (star, 64-row strip).  It walks the column blocks j = 0 .. nb - 1:
//
//     T   = A_j - sum_{k < j} X_k L_jk^T      pipelined product (sp_mm.h): the strip's own solved
//                                             columns are the A operand, the row panel j of L the B
//     X_j = T L_jj^-T                         one more 64-deep product on the matrix cores, against
//                                             the block's L_d^-T (written beside the panel solves of
//                                             the chain, trsm_quad_kernel's identity tile), from LDS
//
// What this replaces: nb launches of a narrow block-column product plus nb launches of a panel
// solve, each with its own prologue, its own tail and a C tile that travels to memory and back
// between the two.  Here a tile is read once and written once, and the next block's tile and
// inverse are already in registers when its product ends.
//
// Why the block solve is a product and not a substitution here: the strips resident on a CU do
// the same work in step, so a substitution phase (vector ALU, 64 dependent steps) in one never
// met a product phase (matrix pipe) in the other -- both pipes idled in turn.  As a product the
// block solve is 40 more MFMAs per wavefront in the same stream (the zero blocks of the
// triangular operand are skipped).  Only the 64 x 64 diagonal blocks are ever inverted
// (condition of L_d ~ 1e3 at the north-star sizes).
//
// The solved columns are read back by the same workgroup as DMA operands.  Only the LAST 64
// columns of the next product depend on the block just stored, so the stores are not waited for
// at the block boundary: the product's DMA queue is drained once, just ahead of the first slice
// that reads them (MM2::loop's fence; one CU, one L1, `s_waitcnt vmcnt(0)` + the slice barrier).
#include "sp_internal.h"
#include "sp_mm.h"
#include "sp_tile.h"

namespace {

using StripCore = MM2<64, 64, 8, 6, 4>;
constexpr int STRIP_XW = 65;                              // padded row of the T tile in LDS
constexpr int STRIP_LDS = 64 * STRIP_XW + 64 * 64;        // T tile + L_d^-T
static_assert(STRIP_LDS >= StripCore::LDS_DOUBLES, "the block solve reuses the product's LDS stages");

struct StripTile {
  mm_d4 c[4];   // the raw tile, accumulator layout
};

__device__ __forceinline__ void strip_fetch(StripTile &t, const double *Ct, long ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) t.c[n][r] = Ct[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * n + fr];
}

__global__ __launch_bounds__(256, 2) void strip_kernel(double *__restrict__ sys, long ld, long stride,
                                                    int batch, int r0, int nrt, int c0, int nb,
                                                    const double *__restrict__ inv_first, long lts,
                                                    int flags) {
  __shared__ __attribute__((aligned(16))) double lds[STRIP_LDS];
  int mtx, rt;
  if (!sp_xcd_decode(blockIdx.x, batch, nrt, mtx, rt)) return;
  double *M = sys + (size_t)mtx * stride;
  const int R = r0 + 64 * rt;
  const double *Arow = M + (size_t)R * ld + c0;       // this strip, from the first column of the triangle
  const double *inv_star = inv_first + (size_t)mtx * lts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  double *sT = lds, *sX = lds + 64 * STRIP_XW;
  if (flags & 8) {
    // (experiment: strips that share a CU start a few microseconds apart, so that their block
    //  boundaries do not coincide)
    const int slot = blockIdx.x >> 3;
    const int k = (slot & 1) + 2 * ((slot >> 5) & 1);
    for (int i = 0; i < 12 * k; ++i) __builtin_amdgcn_s_sleep(32);
  }
  StripTile cur;
  strip_fetch(cur, M + (size_t)R * ld + c0, ld);
  for (int j = 0; j < nb; ++j) {
    const int cj = c0 + 64 * j;
    double *Ct = M + (size_t)R * ld + cj;
    StripCore mm;
    mm.init(Arow, ld, M + (size_t)cj * ld + c0, ld);
    const int kd = (flags & 4) ? 0 : 64 * j;          // (ablation: no product)
    mm_d4 acc[1][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[0][n] = mm_d4{0.0, 0.0, 0.0, 0.0};
    // slices from 8 (j - 1) on read the block stored at the end of the previous iteration
    const int fence = (flags & 16) ? (1 << 30) : 8 * (j - 1);
    if ((flags & 16) || (j >= 1 && fence < StripCore::LDS_DOUBLES / StripCore::STAGE - 1)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    mm.prologue(lds, 0, kd);
    // the next block's tile (memory latency) and this block's L_d^-T (upper triangular like the
    // L_d^T image: the same sparse fetch; L2) travel while the product runs
    StripTile nxt;
    if (j + 1 < nb) strip_fetch(nxt, Ct + 64, ld);
    LtRegs iv;
    lt_load(iv, inv_star + (size_t)j * 2 * SP_LT_IMG);
    mm.loop(lds, 0, kd, acc, fence);
    // T = A_j - (products), accumulator layout -> LDS as the A operand of the block solve
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sT[(16 * wave + fk + 4 * r) * STRIP_XW + 16 * n + fr] = cur.c[n][r] - acc[0][n][r];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      *reinterpret_cast<d2v *>(sX + 2 * (tid + 256 * i)) = iv.v[i];
    __syncthreads();
    // X_j[i][n] = sum_k T[i][k] (L_d^-T)[k][n]; (L_d^-T)[k][n] = 0 for n < k: block (kb, n) with
    // n < kb contributes nothing
    mm_d4 xo[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) xo[n] = mm_d4{0.0, 0.0, 0.0, 0.0};
    if (!(flags & 2)) {
      const double *pa = sT + (16 * wave + fr) * STRIP_XW + fk;
      const double *pb = sX + fk * 64 + fr;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kk = 16 * kb + 4 * s;
          const double a = pa[kk];
#pragma unroll
          for (int n = 0; n < 4; ++n)
            if (n >= kb) xo[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[kk * 64 + 16 * n], xo[n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * n + fr] = xo[n][r];
    // the operands of the block solve must have been read before the next product's DMA overwrites them
    __syncthreads();
    if (j + 1 < nb) cur = nxt;
  }
}

}  // namespace

static int g_strip_flags = 16;   // ablations / experiments for tools/strip_bench.py
void sp_set_strip_flags(int f) { g_strip_flags = f; }

int sp_launch_strip(double *sys, long ld, long stride, int batch, int r0, int nrt, int c0, int nb,
                    const double *inv_first, long lts, hipStream_t st) {
  if (batch <= 0 || nrt <= 0 || nb <= 0) return SP_OK;
  if ((ld & 1) || (stride & 1) || (reinterpret_cast<uintptr_t>(sys) & 15) || (c0 & 1))
    return SP_ERR_INVALID;
  const long nblk = sp_xcd_grid(batch, nrt);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  hipLaunchKernelGGL(strip_kernel, dim3((unsigned)nblk), dim3(256), 0, st, sys, ld, stride, batch, r0,
                     nrt, c0, nb, inv_first, lts, g_strip_flags);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
