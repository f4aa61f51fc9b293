// Work decode shared by the tile kernels, and the per-star scratch of the factorisation.
#ifndef SP_TILE_H
#define SP_TILE_H

#include "sp_diag.h"

// XCD-aware work decode shared by the tile kernels.  Work items are (matrix, tile) pairs,
// matrix-major.  The hardware hands consecutive workgroups to the 8 XCDs round-robin, so
// XCD x = blockIdx.x % 8 is given the CONTIGUOUS items [x c, (x + 1) c), c = ceil(items / 8):
// whole matrices when the batch is a multiple of 8 (every tile of a star re-reads that star's
// row panels from one 4 MiB L2), runs of neighbouring tiles of one matrix when it is not -- a
// single large matrix is spread over all 8 XCDs instead of one.  Launch 8 c workgroups.
__device__ __forceinline__ bool sp_xcd_decode(int b, int batch, int ntiles, int &mtx, int &tile) {
  const int items = batch * ntiles;
  const int c = (items + 7) >> 3;
  const int slot = b >> 3;
  const int item = (b & 7) * c + slot;
  if (slot >= c || item >= items) return false;
  mtx = item / ntiles;
  tile = item - mtx * ntiles;
  return true;
}
static inline long sp_xcd_grid(long batch, long ntiles) { return 8L * ((batch * ntiles + 7) / 8); }


// per-star scratch of the factorisation: two image slots of SP_LT_IMG doubles (L_d^-1 of the pivot
// block in the fragment order of the panel kernel's solve, sp_paneldiag.h: 2560 doubles), used in
// turn -- the launch of panel j reads slot j & 1 while the workgroup that factors block j + 1 in
// its tail writes the other.  `lts` doubles apart from star to star (sp_lt_stride, sp_internal.h).
#define SP_LT_IMG 4096

#endif
