// Strip solve of the recursive Cholesky driver (sp_cholesky.hip):
//
//     X = A L^-T     A: the rows [r0, r0 + 64 nrt) x the columns [c0, c0 + 64 nb) of every system,
//                    L: the factored nb-block lower triangle on those columns (in the same system)
//
// One long-lived workgroup per (star, 64-row strip).  It walks the column blocks j = 0 .. nb - 1:
//
//     T   = A_j - sum_{k < j} X_k L_jk^T      pipelined product (sp_mm.h): the strip's own solved
//                                             columns are the A operand, the row panel j of L the B
//     X_j = T L_jj^-T                         substitution against the block's L_d^T image, four
//                                             lanes per row (sp_tile.h), stored over A_j
//
// What this replaces: nb launches of a narrow block-column product plus nb launches of a panel
// solve, each with its own prologue, its own tail and a C tile that travels to memory and back
// between the two.  Here a tile is read once and written once, the product of block j + 1 starts
// the moment X_j is stored, and the two or three strips resident on a CU are at different phases:
// one multiplies (matrix pipe) while another substitutes (vector ALU).
//
// The solved columns are read back by the same workgroup as DMA operands a few microseconds after
// they were stored: `s_waitcnt vmcnt(0)` + the workgroup barrier order the two (one CU, one L1).
#include "sp_internal.h"
#include "sp_mm.h"
#include "sp_tile.h"

namespace {

using StripCore = MM<64, 64, 16, 3, 4>;
static_assert(StripCore::LDS_DOUBLES >= 64 * 65, "the solve reuses the product's LDS stages");

__global__ __launch_bounds__(256) void strip_kernel(double *__restrict__ sys, long ld, long stride,
                                                    int batch, int r0, int nrt, int c0, int nb,
                                                    const double *__restrict__ lt_first, long lts) {
  __shared__ __attribute__((aligned(16))) double lds[StripCore::LDS_DOUBLES];
  int mtx, rt;
  if (!sp_xcd_decode(blockIdx.x, batch, nrt, mtx, rt)) return;
  double *M = sys + (size_t)mtx * stride;
  const int R = r0 + 64 * rt;
  const double *Arow = M + (size_t)R * ld + c0;       // this strip, from the first column of the triangle
  const double *lt_star = lt_first + (size_t)mtx * lts;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int q = tid & 3, lrow = tid >> 2;
  constexpr int XW = 65;
  double *sT = lds;
  for (int j = 0; j < nb; ++j) {
    const int cj = c0 + 64 * j;
    double *Ct = M + (size_t)R * ld + cj;
    StripCore mm;
    mm.init(Arow, ld, M + (size_t)cj * ld + c0, ld);
    mm_d4 acc[1][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[0][n] = mm_d4{0.0, 0.0, 0.0, 0.0};
    mm.prologue(lds, 0, 64 * j);
    // the tile itself and the diagonal block's image travel while the product runs
    mm_d4 cin[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) cin[n][r] = Ct[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * n + fr];
    LtRegs lt;
    lt_load(lt, lt_star + (size_t)j * SP_LT_IMG);
    mm.loop(lds, 0, 64 * j, acc);
    // accumulator layout -> LDS -> four lanes per row
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sT[(16 * wave + fk + 4 * r) * XW + 16 * n + fr] = cin[n][r] - acc[0][n][r];
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
    quad_solve_store(x, sT, sT + 4096, Ct + (size_t)lrow * ld + 2 * q, true);
    // the stores must have landed before this workgroup's next product reads them, and the
    // image must have been read before the next product's DMA overwrites it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

}  // namespace

int sp_launch_strip(double *sys, long ld, long stride, int batch, int r0, int nrt, int c0, int nb,
                    const double *lt_first, long lts, hipStream_t st) {
  if (batch <= 0 || nrt <= 0 || nb <= 0) return SP_OK;
  if ((ld & 1) || (stride & 1) || (reinterpret_cast<uintptr_t>(sys) & 15) || (c0 & 1))
    return SP_ERR_INVALID;
  const long nblk = sp_xcd_grid(batch, nrt);
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  hipLaunchKernelGGL(strip_kernel, dim3((unsigned)nblk), dim3(256), 0, st, sys, ld, stride, batch, r0,
                     nrt, c0, nb, lt_first, lts);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
