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


// per-star scratch of the factorisation (`lts` = sp_lt_stride doubles apart from star to star, sp_internal.h):
// THREE image slots of SP_IMG_SLOT doubles (L_d^-1 of a pivot block in the fragment order of the panel kernel's
// solve, sp_paneldiag.h: 2560 doubles), slot j mod 3 for pivot block j -- a launch that handles the columns j and
// j + 1 reads the images of blocks j and j + 1 while block j + 2's is being written -- and, behind them, the words
// the chain workgroups of a launch publish (sp_panel.hip).
#define SP_IMG_SLOT 2688
#define SP_IMG_WORDS (3 * SP_IMG_SLOT)
__host__ __device__ __forceinline__ size_t sp_img_off(int j) { return (size_t)(j % 3) * SP_IMG_SLOT; }
// (sp_lt_stride, sp_internal.h, gives a star at least 2 x 4096 doubles: the three slots and the two chain words
//  must fit, or a star's words land in its neighbour's first image)
static_assert(SP_IMG_WORDS + 2 <= 2 * 4096, "per-star scratch: three image slots + the chain words");

#endif
