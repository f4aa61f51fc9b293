// Dataflow panel chain of the blocked Cholesky factorisation (math.py:75-91):
// ONE launch per super-panel instead of one per panel.
//
// A super-panel is the pivot blocks s0 .. s0 + wq - 1 (64 columns each).  Every 64-row strip R of
// a star below the first pivot block is ONE long-lived workgroup that walks the panels j it has to
// see, left-looking:
//
//     T   = A[R][s0 + j] - sum_{k < j} X[R][k] L[s0 + j][k]^T     pipelined product (sp_mm.h)
//     X_j = T L_d(s0 + j)^-T                                      substitution, four lanes per row
//     (pivot strips)  D -= X_j X_j^T                              the strip's own diagonal tile,
//                                                                 kept in registers from start to end
// and a pivot strip then factors its diagonal tile (diag_block, sp_diag.h) and publishes it.
//
// What orders the workgroups is data, not kernel boundaries: two flags per (star, row tile) in
// global memory,
//     rows_done[R]    the solved panels of row tile R that are in memory (the B operand of the later strips)
//     diag_ready[R]   the L_d^T image of pivot block R is in memory
// set with release semantics (device scope) by the producer and polled by ONE lane of the consumer.
// With a kernel boundary per panel every launch ended on the 64 workgroups that factor the next
// diagonal block (11 us, a latency chain) while 190 CUs idled, and the next launch began with a
// product none of them had been allowed to start.  Here the product for panel j + 1 runs under
// the factorisation of block j + 1, and the chain from one diagonal block to the next is: image
// -> substitution of ONE tile -> rank-64 update from registers -> diag_block.
//
// Forward progress: work items are handed out by a ticket counter (one per XCD group), strips in
// ascending order, so whatever a workgroup waits for belongs to a workgroup that has already
// started -- nothing depends on all workgroups being resident or on the order the hardware
// dispatches them in.  Every wait is bounded (wall clock): past the limit the kernel raises a
// sticky abort flag, every waiter drains, and the stars involved are reported as failed.
#include "sp_internal.h"
#include "sp_mm.h"
#include "sp_tile.h"

namespace {

using ChainCore = MM2<64, 64, 8, 6, 4>;
constexpr int CH_XW = 65;                        // padded row of the T / X tile in LDS
// one region, three tenants in turn: the product's stages, the T / X tile (64 x 65 doubles),
// the image of the solve (L_d^T, 64 x 64, and its reciprocal diagonal, 64: exactly the tile's
// 4160 doubles).  48 KB: three workgroups per CU.
constexpr int CH_RD = 4096;
constexpr int CH_LDS = ChainCore::LDS_DOUBLES;
static_assert(CH_LDS >= 64 * CH_XW && CH_RD + 64 <= 64 * CH_XW, "tile and image share the stages");
static_assert(CH_LDS >= SP_DIAG_LDS_DOUBLES, "diag_block works in the same LDS");
constexpr long long CH_WAIT_LIMIT = 300000000LL;   // wall_clock64 ticks (100 MHz): 3 s

struct ChainArgs {
  double *sys;
  long ld, stride;
  int S, ntile;          // stars; 64-row tiles per padded system
  int s0, wq;            // first pivot block of the super-panel, pivot blocks in it
  int nsteps, nact_last; // pivot blocks of the factorisation, active columns of the last one
  int r_first, nstrips;  // row tiles r_first .. r_first + nstrips - 1 are worked on
  int ngrp;              // 8: star s belongs to XCD s % 8 (queue per XCD), 1: one queue for all
  double *img;           // per star `lts` doubles: block j's L_d^T image at j * 2 * SP_LT_IMG
  long lts;
  int *flags;            // per star 2 * ntile: rows_done, diag_ready
  int *tickets;          // ngrp counters, zero at launch
  int *abort_flag;
  int32_t *info;
  long long *dbg;         // (debug) timestamps of star 0's strips: [strip][16 rows][8], wall clock
};

// XCD this wavefront runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ int chain_xcc_id() {
  return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u);
}

// `local`: producer and consumers run on the same XCD (ngrp == 8: a star's strips are only ever
// taken by workgroups that READ their XCD id and draw from that XCD's queue), so they share an
// L2 and a store is visible to them once it is acknowledged (vmcnt) -- no write-back of the L2's
// dirty lines (buffer_wbl2, microseconds with megabytes of solved tiles in flight).  Otherwise a
// device-scope release.
__device__ __forceinline__ void chain_publish(int *flag, int value, bool local) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (local)
      __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// returns false when the wait was abandoned (abort raised here or elsewhere)
__device__ __forceinline__ bool chain_wait(int *flag, int need, int *abort_flag, int *s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    int ok = 1, n = 0;
    long long t0 = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if ((++n & 255) == 0) {
        const long long t = wall_clock64();
        if (t0 == 0) t0 = t;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
            t - t0 > CH_WAIT_LIMIT) {
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    *s_ok = ok;
  }
  __syncthreads();
  // (the L1 of this CU may hold lines of the tiles from before they were solved)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return *s_ok != 0;
}

#define CH_STAMP(row, k)                                                            \
  do {                                                                              \
    if (dbg && tid == 0) dbg[((size_t)strip * 16 + (row)) * 8 + (k)] = wall_clock64(); \
  } while (0)

struct ChainTile {
  mm_d4 c[4];   // a 64 x 64 tile, accumulator layout of the 4 x (16 x 64) wavefront split
};
// element offsets of a lane's 16 entries of a tile, relative to the tile's first element (32-bit:
// 64 rows of a system; one set serves every tile the workgroup touches -- per-tile 64-bit
// addresses hoisted out of the panel loop cost more registers than the tiles themselves)
struct ChainOffs {
  int o[4];     // row part, per accumulator register r; the column part 16 n is an immediate
};
__device__ __forceinline__ void chain_fetch(ChainTile &t, const double *Ct, const ChainOffs &f) {
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) t.c[n][r] = Ct[f.o[r] + 16 * n];
}
__device__ __forceinline__ void chain_put(const ChainTile &t, double *Ct, const ChainOffs &f) {
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[f.o[r] + 16 * n] = t.c[n][r];
}

__global__ __launch_bounds__(256, 3) void chain_kernel(ChainArgs a) {
  __shared__ __attribute__((aligned(16))) double lds[CH_LDS];
  __shared__ int s_tk, s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const bool local = a.ngrp == 8;
  const int grp = local ? chain_xcc_id() : 0;
  const int nst = local ? (a.S >> 3) + (grp < (a.S & 7) ? 1 : 0) : a.S;
  double *sT = lds, *sLT = lds, *sRd = lds + CH_RD;
  const int cS = 64 * a.s0;
  ChainOffs offs;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) offs.o[rr] = (16 * wave + fk + 4 * rr) * (int)a.ld + fr;

  // strips are drawn from this XCD's queue until it is empty (with the usual round-robin
  // placement of workgroups every workgroup draws exactly one)
  for (;;) {
    __syncthreads();
    if (tid == 0) s_tk = atomicAdd(&a.tickets[grp], 1);
    __syncthreads();
    const int tk = s_tk;
    if (tk >= nst * a.nstrips) return;
    const int strip = tk / nst;
    const int mtx = local ? grp + 8 * (tk - strip * nst) : tk - strip * nst;
    const int R = a.r_first + strip;                 // row tile of this workgroup
    const int r = R - a.s0;                          // ... counted from the super-panel's first block
    const int nsolve = r < a.wq ? r : a.wq;          // panels this strip is solved against
    // the strip's own diagonal tile is a pivot block that is factored here: a block of this
    // super-panel, or the first one of the next (complete once this super-panel's panels are in)
    const bool pivot = R < a.nsteps && r <= a.wq;
    double *M = a.sys + (size_t)mtx * a.stride;
    int *rows_done = a.flags + (size_t)mtx * 2 * a.ntile, *diag_ready = rows_done + a.ntile;
    double *img = a.img + (size_t)mtx * a.lts;
    const double *Arow = M + (size_t)(64 * R) * a.ld + cS;
    double *D = M + (size_t)(64 * R) * a.ld + 64 * R;
    bool ok = true;
    long long *dbg = (a.dbg && mtx == 0) ? a.dbg : nullptr;
    CH_STAMP(15, 0);
    // the critical path of the whole factorisation runs through the pivot strips
    if (pivot) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    ChainTile dac;     // the diagonal tile (pivot strips): in registers for the last panel only
    if (pivot && nsolve == 0) chain_fetch(dac, D, offs);

    for (int j = 0; j < nsolve; ++j) {
      const int Jb = a.s0 + j, kd = 64 * j;
      double *Ct = M + (size_t)(64 * R) * a.ld + 64 * Jb;
      CH_STAMP(j, 0);
      // (lane-derived indices are recomputed per panel: hoisted out of the loops they would
      //  occupy a hundred registers for its whole length)
      int t = tid;
      asm volatile("" : "+v"(t));
      const int lane = t & 63, wave = t >> 6, fr = lane & 15, fk = lane >> 4, q = t & 3, lrow = t >> 2;
      ChainTile cur;
      chain_fetch(cur, Ct, offs);      // consumed after the product: its latency hides behind it
      if (j > 0) {
        // row tile Jb complete (and this workgroup's own stores of the previous panel drained)
        ok = chain_wait(rows_done + Jb, j, a.abort_flag, &s_ok) && ok;
        CH_STAMP(j, 1);
        mm_d4 acc[1][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[0][n] = mm_d4{0.0, 0.0, 0.0, 0.0};
        ChainCore mm;
        mm.init(Arow, a.ld, M + (size_t)(64 * Jb) * a.ld + cS, a.ld);
        mm.prologue(lds, 0, kd);
        mm.loop(lds, 0, kd, acc);
#pragma unroll
        for (int n = 0; n < 4; ++n) cur.c[n] -= acc[0][n];
      }
      CH_STAMP(j, 2);
      // T = A - (products): accumulator layout -> LDS
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          sT[(16 * wave + fk + 4 * rr) * CH_XW + 16 * n + fr] = cur.c[n][rr];
      ok = chain_wait(diag_ready + Jb, 1, a.abort_flag, &s_ok) && ok;
      CH_STAMP(j, 3);
      LtRegs lt;
      lt_load(lt, img + (size_t)Jb * 2 * SP_LT_IMG, t);
      double x[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[2 * i] = sT[lrow * CH_XW + 8 * i + 2 * q];
        x[2 * i + 1] = sT[lrow * CH_XW + 8 * i + 2 * q + 1];
      }
      __syncthreads();
      lt_store(lt, sLT, sRd, t);        // (over the tile: every lane holds its part of T)
      __syncthreads();
      quad_solve_store(x, sLT, sRd, Ct + (size_t)lrow * a.ld + 2 * q, true, t);
      CH_STAMP(j, 4);
      if (j == nsolve - 1 && r < a.wq && R < a.nsteps) {
        // last panel of a pivot row tile of this super-panel: the later strips may start the
        // products that read it while this workgroup goes on to its diagonal block
        chain_publish(rows_done + R, j + 1, local);
      }
      if (pivot) {
        // D -= X X^T: the tile just solved goes back to LDS (row layout) as both operands;
        // the diagonal tile itself lives in memory between panels (L2) and stays in
        // registers after the last one
        chain_fetch(dac, D, offs);
        __syncthreads();                // the image has been read
        double *row = sT + lrow * CH_XW + 2 * q;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          row[8 * i] = x[2 * i];
          row[8 * i + 1] = x[2 * i + 1];
        }
        __syncthreads();
        const double *pa = sT + (16 * wave + fr) * CH_XW + fk;
        const double *pb = sT + fr * CH_XW + fk;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 4) {
          const double av = -pa[kk];
          dac.c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[kk], dac.c[0], 0, 0, 0);
          dac.c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[16 * CH_XW + kk], dac.c[1], 0, 0, 0);
          dac.c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[32 * CH_XW + kk], dac.c[2], 0, 0, 0);
          dac.c[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[48 * CH_XW + kk], dac.c[3], 0, 0, 0);
        }
        if (j + 1 < nsolve) chain_put(dac, D, offs);
      }
      // the tile, the image and (pivot strips) the X operand have been read: the next product's
      // DMA may overwrite them
      __syncthreads();
      CH_STAMP(j, 5);
    }

    if (!pivot) {
      if (!ok && a.info && tid == 0) a.info[mtx] = 1;
      continue;
    }
    // ---- the diagonal tile: complete now -- factor it
    int t = tid;
    asm volatile("" : "+v"(t));
    const int lane = t & 63, wave = t >> 6, fr = lane & 15, fk = lane >> 4, q = t & 3, lrow = t >> 2;
    const int nact = (R == a.nsteps - 1) ? a.nact_last : 64;
    double *sD = lds, *sDr = lds + 64 * BLD;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int li = 16 * wave + fk + 4 * rr, lj = 16 * n + fr;
        double v = (li < nact && lj < nact) ? dac.c[n][rr] : (li == lj ? 1.0 : 0.0);
        if (lj > li) v = 0.0;
        sD[li * BLD + lj] = v;
      }
    __syncthreads();
    double *lt_out = img + (size_t)R * 2 * SP_LT_IMG;
    CH_STAMP(15, 1);
    const int notpd = diag_block(sD, sDr, lt_out, nullptr, t);
    CH_STAMP(15, 2);
    if ((notpd || !ok) && a.info) a.info[mtx] = 1;
    {
      const int cj = (t & 15) * 4, ri = t >> 4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int rw = ri + 16 * pass;
        double *dst = D + (size_t)rw * a.ld + cj;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (rw < nact && cj + e <= rw) dst[e] = sD[rw * BLD + cj + e];
      }
    }
    if (nact < 64) {
      // partial last block: the rows of the tile below the active ones (residual rows, padding)
      // carry every update already and are solved against the block just factored (identity
      // padding: their columns >= nact stay as they are)
      __syncthreads();          // sD has been read
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          sT[(16 * wave + fk + 4 * rr) * CH_XW + 16 * n + fr] = dac.c[n][rr];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // diag_block's image stores
      __syncthreads();
      LtRegs lt;
      lt_load(lt, lt_out, t);
      double x[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[2 * i] = sT[lrow * CH_XW + 8 * i + 2 * q];
        x[2 * i + 1] = sT[lrow * CH_XW + 8 * i + 2 * q + 1];
      }
      __syncthreads();
      lt_store(lt, sLT, sRd, t);
      __syncthreads();
      quad_solve_store(x, sLT, sRd, D + (size_t)lrow * a.ld + 2 * q, lrow >= nact, t);
    }
    chain_publish(diag_ready + R, 1, local);
    CH_STAMP(15, 3);
  }
}

}  // namespace

// flags (per star 2 * ntile) + tickets (8 per launch) + the abort word, in h->chain_mem
size_t sp_chain_mem_ints(int S, int ntile, int nlaunch) {
  return (size_t)S * 2 * ntile + 8 * (size_t)nlaunch + 8;
}

int sp_launch_chain(double *sys, long ld, long stride, int S, int ntile, int s0, int wq, int nsteps,
                    int nact_last, double *img, long lts, int *flags, int *tickets, int *abort_flag,
                    int32_t *info, long long *dbg, hipStream_t st) {
  ChainArgs a;
  a.dbg = dbg;
  a.sys = sys; a.ld = ld; a.stride = stride; a.S = S; a.ntile = ntile; a.s0 = s0; a.wq = wq;
  a.nsteps = nsteps; a.nact_last = nact_last;
  a.r_first = s0 == 0 ? 0 : s0 + 1;
  a.nstrips = ntile - a.r_first;
  a.ngrp = S >= 8 ? 8 : 1;
  a.img = img; a.lts = lts; a.flags = flags; a.tickets = tickets; a.abort_flag = abort_flag;
  a.info = info;
  if (S <= 0 || a.nstrips <= 0) return SP_OK;
  if ((ld & 1) || (stride & 1) || (reinterpret_cast<uintptr_t>(sys) & 15)) return SP_ERR_INVALID;
  const long per = a.ngrp == 8 ? (long)((S + 7) / 8) * a.nstrips : (long)S * a.nstrips;
  const long nblk = a.ngrp * per;
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  hipLaunchKernelGGL(chain_kernel, dim3((unsigned)nblk), dim3(256), 0, st, a);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
