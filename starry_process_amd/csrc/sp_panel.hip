// Panel kernel of the blocked Cholesky factorisation (math.py:75-91), round 3:
// ONE launch per 64-column panel j.
//
// Work items of a launch, per star (dealt by workgroup index, a star's items on one XCD when the
// batch allows, sp_xcd_decode):
//
//   T item  (one per row tile i > j, or per PAIR of row tiles where single ones would not fit the
//            CUs in one round)
//        G   = (A_ij - sum_k L_ik L_jk^T)^T        left-looking product over the panels of this
//                                                 super-panel, TRANSPOSED: the accumulators hold T^T
//        X^T = L_d^-1 G                            the triangular solve as 40 MFMAs: the accumulators
//                                                 of the product ARE the B fragments of this product,
//                                                 the blocks of the image L_d^-1 its A fragments --
//                                                 no LDS round trip for the tile, no barrier
//        store X; row tiles that are pivot blocks still to come (i <= last) subtract X X^T from their
//        own diagonal tile (A fragments = the accumulators of X^T, B through LDS).
//   the FIRST T item (row tile j + 1, the next pivot block) then holds that block complete in
//        registers and factors it on the spot (P_TAILD; panel_diag_core, sp_paneldiag.h): L_d to the
//        system, L_d^-1 in fragment order to the star's other image slot, for launch j + 1.
//   look-ahead item (P_LOOKAHEAD): tile (j + 2, j + 1) -- the first tile of launch j + 1 -- is brought
//        up to date with every column block but the one this launch solves.  Launch j + 1's first
//        item is then a rank-64 update, a solve and an eager update (10 us), and the diagonal block
//        behind it runs under the other items' products instead of behind them: the critical chain
//        of a panel no longer contains a product that grows with the panel's position.
//   D item  (P_DITEMS: a launch of D items only) pivot block 0, which nobody precedes.
//
// Round 2's kernel differed in three ways: its pivot workgroup multiplied the whole left-looking
// product before it could factor (the chain grew with q), every workgroup solved by a 64-step
// substitution on the vector ALU (9-13 us against 2), and 960 workgroups on 768 slots ran in 1.25
// rounds (pairs: one round).
//
// Transposed accumulators with a row permutation.  v_mfma_f64_16x16x4_f64 leaves D[i][j] in lane
// (j = lane & 15, fk = lane >> 4), register r, i = fk + 4 r.  The product is formed as
// G = C^T - B A^T with the pivot row tile (B) as the MFMA's A operand, its LDS rows read through
// PI(i) = 4 (i mod 4) + i div 4, so that register r of lane (fr, fk) of accumulator m holds
//     T[16 w + fr][16 m + 4 fk + r]           (w = wavefront): FOUR CONSECUTIVE COLUMNS of one row,
// i.e. 32-byte global loads / stores of the tile, AND exactly the B fragment (k = 16 m + 4 fk + s,
// column 16 w + fr) of the solve's MFMA.  The image is stored by the D code in the matching order.
//
// Measured and dropped (round 3): D items INSIDE launch j ordered by flags in memory
// (tickets per XCD, L2-local atomics, CU gating).  A diagonal block that shares its CU with
// workgroups issuing MFMAs runs 2-3 times slower (fp64 VALU and fp64 MFMA share the pipes), its CU
// mates become the stragglers of the launch, and workgroups waiting for a flag hold CU slots that
// other steps in flight could use: equal alone, -3 % with three steps in flight.
#include "sp_internal.h"
#include "sp_tile.h"
#include "sp_cov.h"
#include "sp_stage.h"
#include "sp_paneldiag.h"
#include "sp_reduce.h"
#include <atomic>

#ifdef SP_PANEL_TRACE
// (variant build only, tools/ab_build.sh trace -DSP_PANEL_TRACE: wall-clock stamps of star 0's work
//  items; rows = pivot blocks, roles: 0 the D item, 1 the first T item, 2 the last; sp_debug_panel2_trace)
__device__ long long g_p2trace[64 * 3 * 16];
// ... and of EVERY star's tail block per launch: [j < 16][star < 64] x (first item start, block start, block end, CU key)
__device__ long long g_p2chain[16 * 64 * 4];
// ... and where the workgroups of a launch run: [j < 16][blockIdx < 1024] -> CU key
__device__ int g_p2cu[16 * 1024];
// ... and when they start and end: [j < 16][blockIdx < 1024][2] (wall clock)
__device__ long long g_p2wg[16 * 1024 * 2];
struct P2WgStamp {
  int j, on;
  __device__ P2WgStamp(int j_, bool on_) : j(j_), on(on_ && j_ < 16 && blockIdx.x < 1024 && threadIdx.x == 0) {
    if (on) g_p2wg[(j * 1024 + blockIdx.x) * 2] = wall_clock64();
  }
  __device__ ~P2WgStamp() {
    if (on) g_p2wg[(j * 1024 + blockIdx.x) * 2 + 1] = wall_clock64();
  }
};
#define P2_CHAIN(k, v)                                                                          \
  do {                                                                                          \
    if (tid == 0 && a.j < 16 && mtx < 64) g_p2chain[((a.j * 64) + mtx) * 4 + (k)] = (v);        \
  } while (0)
#define P2_STAMP(role, k)                                                                       \
  do {                                                                                          \
    if (mtx == 0 && tid == 0 && a.j < 64 && (role) >= 0) g_p2trace[(a.j * 3 + (role)) * 16 + (k)] = wall_clock64(); \
  } while (0)
#else
#define P2_STAMP(role, k) do { } while (0)
#define P2_CHAIN(k, v) do { } while (0)
#endif

#ifndef P_WGS
#define P_WGS 3             // workgroups per CU the register budget is cut for (168 registers; per instantiation
                            // 124-168 are used and nothing spills, vector or scalar: DESIGN.md 4.3 -- with a
                            // budget of 128 the scratch traffic of the D items cost them 20 us per launch)
#endif

namespace {

constexpr int PK = 32;                 // depth of an operand slice in LDS
constexpr int PLDW = PK + 1;
constexpr int XLD = 66;                // LDS row of the solved tile (eager update): 16-byte aligned rows, 33 slots of 16 B
// where the four columns 16 nb + 4 fk .. + 3 of the solved tile sit in its LDS row: the slot index of a lane's 16-byte
// read must not depend on fk & 1 modulo 16 (the row length adds one slot per row)
__device__ __forceinline__ constexpr int sx_col(int nb, int fk) { return 32 * (fk & 1) + 8 * nb + 4 * (fk >> 1); }
#ifndef P_PAIRS
#define P_PAIRS 1           // 128-row items where 64-row ones would not fit the CUs in one round
#endif
constexpr int P_LDS = P_PAIRS ? 3 * 64 * PLDW   // own rows (up to 128) + the pivot rows: 50.7 KB, three workgroups per CU
                              : SP_DIAG_LDS_DOUBLES;
static_assert(P_LDS >= SP_DIAG_LDS_DOUBLES && P_LDS >= 64 * XLD && P_LDS >= SP_IMG_DOUBLES,
              "one LDS region, four tenants in turn");

// what a launch contains (PanelArgs.mode)
enum { P_DITEMS = 1,    // D items: pivot block j is factored by this launch (never together with T items)
       P_TITEMS = 2,    // T items
       P_TAILD = 4,     // the first T item (row tile j + 1, the next pivot block) goes on to factor that block
       P_LOOKAHEAD = 8, // one more item per star: tile (j + 2, j + 1), the FIRST tile of the next launch, is brought
                        // up to date with the columns before block j (everything but the column this launch solves)
       P_FIRSTLA = 16 };// ... and this launch's first tile was treated that way by the launch before it: only the
                        // rank-64 update with column block j - 1 is left of its product

struct PanelArgs {
  double *sys;
  long ld, stride;
  int S, ntile;        // stars; 64-row tiles per padded system
  int j, s0;           // pivot block of the launch; first pivot block of its super-panel
  int nact;            // active columns of block j
  int next_nact;       // ... of block j + 1 (P_TAILD)
  int last;            // row tiles i <= last keep their own diagonal tile up to date
  int mode;
  int pair;            // row tiles below the next pivot row tile are dealt in pairs (128-row items)
  int lay;             // 1: chain-aware layout of the launch (see panel_kernel)
  int seq;             // launch number (tags the chain workgroups' words in the stars' scratch)
  int tri0;            // >= 0: the rows from this one on are an identity riding along (sp_spd_inverse_batched): row tile T
                       // is zero left of column block (64 T - tri0) / 64, and the product of its item starts there
  double *img;         // per star `lts` doubles: three image slots + the chain words (sp_tile.h)
  long lts;
  int32_t *info;
  LazyCov lz;
  SpReduceArgs red;    // red.lnlike != null: the tail that factors the last pivot block reduces its star (sp_reduce.h)
};

typedef double pd4 __attribute__((ext_vector_type(4)));

// the CU a workgroup runs on: HW_REG_XCC_ID[2:0] | HW_REG_HW_ID[15:8] (CU, SH, SE) -- 256 distinct values on
// this part (tools/micro/hwid.hip)
__device__ __forceinline__ long long cu_key() {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
  return (long long)(xcc * 256 + hw);
}

// items of a star in a launch: [D item] then the row tiles below the pivot block -- the next pivot
// row tile alone (it is on the critical path: 64 rows), then PAIRS of row tiles (128 rows per
// workgroup: the pivot rows' fragments feed eight MFMAs instead of four, half the workgroups per
// launch -- one round at three per CU), a single one at the end when their number is odd
__host__ __device__ __forceinline__ int panel_titems(int ntile, int j, int pair) {
  const int n = ntile - j - 1;
  return n <= 0 ? 0 : (pair ? 1 + n / 2 : n);        // 1 + ceil((n - 1) / 2)
}

// ---- T item: NR row tiles (64 NR rows) from row tile i0 -----------------------------------------
// cb: column block of the tiles (the launch's pivot block j; j + 1 for the look-ahead item); the product
// runs over the column blocks kb0 .. kb1 - 1; la: look-ahead item -- the tile goes back updated, unsolved
// LAZY: the launch may form tiles at first touch (first super-panel of a marginal system, sp_cov.h);
// RED: the tail that factors the last pivot block reduces its star (sp_reduce.h) -- compile-time, so that an
// instantiation carries only the state of what its launches contain
// CHAIN: the call site may be handed the launch's first item (row tile j + 1), which goes on to factor the next
// pivot block (P_TAILD): only those copies carry the diagonal block's code
template <int NR, bool LAZY, bool RED, bool CHAIN>
__device__ __forceinline__ void panel_tile_item(const PanelArgs &a, double *M, int mtx, int i0, int cb,
                                                int kb0, int kb1, bool la, bool may_lazy,
                                                double *img_star, double *smem, int tid) {
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
  const long ld = a.ld;
#ifdef P_NO_PRODUCT
  const int Kd = 0 * (kb1 - kb0);       // (timing probe: results are garbage)
#else
  const int Kd = 64 * (kb1 - kb0);
#endif
  const double *img = img_star + sp_img_off(a.j);
  const bool chain = CHAIN && NR == 1 && !la && i0 == a.j + 1 && (a.mode & P_TAILD);   // this item factors the next block
#ifdef SP_PANEL_TRACE
  const int role = i0 == a.j + 1 ? 1 : (i0 + NR == a.ntile ? 2 : -1);
#endif
  P2_STAMP(role, 0);
  if (NR == 1 && !la && i0 == a.j + 1) P2_CHAIN(0, wall_clock64());
#ifndef P_NO_CHAIN_PRIO
  // the next pivot row tile is the launch's critical chain from its first instruction, not only from
  // its diagonal block on: its wavefronts go first wherever they share a SIMD
  if (chain) __builtin_amdgcn_s_setprio(3);
#endif
  double *Ct = M + (size_t)(64 * i0) * ld + 64 * cb;           // tiles (i0 .., cb)
  const double *Ab = M + (size_t)(64 * i0) * ld + 64 * kb0;    // own rows, the product's columns
  const double *Bb = M + (size_t)(64 * cb) * ld + 64 * kb0;    // the pivot row tile
  double *sA = smem, *sB = smem + NR * 64 * PLDW;

  // accumulators of G = T^T: g[h][m][r] = T[64 h + 16 wave + fr][16 m + 4 fk + r]
  pd4 g[NR][4];
  double *crow = Ct + (size_t)(16 * wave + fr) * ld + 4 * fk;
#ifndef P_IMG_PREFETCH
#define P_IMG_PREFETCH 1
#endif
  PanelRegs<PK> ra[NR], rb;
  // (tiles evaluated at first touch: the first operand slices are requested after the evaluation --
  //  48 registers of loads in flight across it made the compiler spill the spline's constants and
  //  reload them for every entry)
  const bool any_lazy = LAZY && may_lazy && a.lz.theta && (cb > 0 || a.lz.c0lazy) && i0 < a.lz.nfull;
  auto first_loads = [&]() {
#pragma unroll
    for (int h = 0; h < NR; ++h) stage_load_fast<PK>(Ab, ld, 64 * h, 0, ra[h], tid);
    stage_load_fast<PK>(Bb, ld, 0, 0, rb, tid);
  };
  if (Kd > 0 && !any_lazy) first_loads();
#pragma unroll
  for (int h = 0; h < NR; ++h) {
    // (first super-panel of a system whose assembly left the tile to its first touch, sp_cov.h:
    //  evaluated while the first operand slices are on their way)
    const bool lazy = LAZY && may_lazy && a.lz.theta && (cb > 0 || a.lz.c0lazy) && i0 + h < a.lz.nfull;
    if (!lazy) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const d2v lo = *reinterpret_cast<const d2v *>(crow + (size_t)(64 * h) * ld + 16 * m);
        const d2v hi = *reinterpret_cast<const d2v *>(crow + (size_t)(64 * h) * ld + 16 * m + 2);
        g[h][m] = pd4{lo.x, lo.y, hi.x, hi.y};
      }
    } else {
#ifdef P_NO_LAZYEVAL
#pragma unroll
      for (int m = 0; m < 4; ++m) g[h][m] = pd4{0.0, 0.0, 0.0, 1.0e-3 * (m + fk)};     // (timing probe)
#else
      lazy_cov_row<(NR == 1 ? SP_LAZY_ROW_BATCH : 0)>(a.lz, mtx, 64 * (i0 + h) + 16 * wave + fr, 64 * cb + 4 * fk, g[h], smem, tid);
#endif
    }
  }
  if (Kd > 0 && any_lazy) first_loads();
  P2_STAMP(role, 1);

  // the image of the pivot block's inverse (the solve's A fragments): requested behind the LAST slice's
  // loads, so that its round trip runs under that slice's MFMAs instead of after them
  d2v im[5];
  auto image_loads = [&](bool now = false) {
    if (la || (!(P_IMG_PREFETCH && NR == 1) && !now)) return;
#pragma unroll
    for (int c = 0; c < 5; ++c) im[c] = *reinterpret_cast<const d2v *>(img + 2 * (tid + 256 * c));
  };
  if (Kd == 0) image_loads(true);

  // G -= B_panel A_panel^T: the pivot rows are the MFMA's A operand (rows through PI), the own rows its B
  {
    const int pfr = sp_pi16(fr);
    auto mfma_slice = [&]() {
      const double *pa = sB + pfr * PLDW + fk;               // pivot rows 16 m + PI(fr)
      const double *pb = sA + (16 * wave + fr) * PLDW + fk;   // own rows 64 h + 16 wave + fr
      // (NR = 2: unrolled by two k-steps only -- fully unrolled the scheduler hoists the fragment
      //  reads of all eight steps above the MFMAs and spills the accumulators)
#pragma clang loop unroll_count(NR == 2 ? 2 : 8)
      for (int kk = 0; kk < PK; kk += 4) {
        const double a0 = pa[kk], a1 = pa[16 * PLDW + kk], a2 = pa[32 * PLDW + kk], a3 = pa[48 * PLDW + kk];
#pragma unroll
        for (int h = 0; h < NR; ++h) {
          const double b = pb[h * 64 * PLDW + kk];
          g[h][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, g[h][0], 0, 0, 0);
          g[h][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, g[h][1], 0, 0, 0);
          g[h][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, g[h][2], 0, 0, 0);
          g[h][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b, g[h][3], 0, 0, 0);
        }
      }
    };
    // (the last slice is peeled off the loop: the image's registers are live from there on only)
    int k0 = 0;
    for (; k0 + PK < Kd; k0 += PK) {
#pragma unroll
      for (int h = 0; h < NR; ++h) stage_store<PK>(ra[h], -1.0, sA + h * 64 * PLDW, tid);
      stage_store<PK>(rb, 1.0, sB, tid);
      __syncthreads();
#pragma unroll
      for (int h = 0; h < NR; ++h) stage_load_fast<PK>(Ab, ld, 64 * h, k0 + PK, ra[h], tid);
      stage_load_fast<PK>(Bb, ld, 0, k0 + PK, rb, tid);
      mfma_slice();
      __syncthreads();
    }
    if (k0 < Kd) {
#pragma unroll
      for (int h = 0; h < NR; ++h) stage_store<PK>(ra[h], -1.0, sA + h * 64 * PLDW, tid);
      stage_store<PK>(rb, 1.0, sB, tid);
      __syncthreads();
      image_loads();
      mfma_slice();
      __syncthreads();
    }
  }

  P2_STAMP(role, 2);
  if (la) {
    // look-ahead item: the tile goes back as it is -- the next launch's first item finishes it
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double *dst = crow + (size_t)(64 * h) * ld + 16 * m;
        *reinterpret_cast<d2v *>(dst) = d2v{g[h][m][0], g[h][m][1]};
        *reinterpret_cast<d2v *>(dst + 2) = d2v{g[h][m][2], g[h][m][3]};
      }
    return;
  }
  P2_STAMP(role, 3);
  // the image: one copy per workgroup through LDS (each wavefront needs all of it)
  if (!(P_IMG_PREFETCH && NR == 1)) image_loads(true);
#pragma unroll
  for (int c = 0; c < 5; ++c) *reinterpret_cast<d2v *>(smem + 2 * (tid + 256 * c)) = im[c];
  __syncthreads();
  // (the diagonal tile of the first row tile's eager update: requested now, it arrives under the solve)
  pd4 dacp[4];
#ifndef P_DAC_PREFETCH
#define P_DAC_PREFETCH 1
#endif
  if (P_DAC_PREFETCH && NR == 1 && i0 <= a.last) {
    const double *Dt0 = M + (size_t)(64 * i0) * ld + 64 * i0;
    // (only the blocks on and below the diagonal: wavefront w owns rows 16 w .., column blocks m <= w -- nobody
    //  reads a diagonal tile's upper blocks, neither the factorisation nor the trailing update's lower tiles)
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (m <= wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dacp[m][r] = Dt0[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * m + fr];
      }
  }
  // Y = X^T = L_d^-1 G, block rows nb = 3 .. 0: y[nb] = sum_{kb <= nb} Linv(nb, kb) G(kb);
  // y[h][nb][r] = X[64 h + 16 wave + fr][16 nb + 4 fk + r]
  pd4 y[NR][4];
  {
    const double *fimg = smem + 2 * lane;
    // (one row tile after the other: both at once would keep 128 accumulator registers alive)
#pragma unroll
    for (int h = 0; h < NR; ++h) {
#pragma unroll
      for (int nb = 3; nb >= 0; --nb) {
        pd4 acc = pd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb <= nb; ++kb) {
          const double *f = fimg + sp_img_block(nb, kb) * 256;
          const d2v lo = *reinterpret_cast<const d2v *>(f);
          const d2v hi = *reinterpret_cast<const d2v *>(f + 128);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.x, g[h][kb][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lo.y, g[h][kb][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.x, g[h][kb][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(hi.y, g[h][kb][3], acc, 0, 0, 0);
        }
        y[h][nb] = acc;
      }
      if (h == 0) P2_STAMP(role, 4);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        double *dst = crow + (size_t)(64 * h) * ld + 16 * nb;
        *reinterpret_cast<d2v *>(dst) = d2v{y[h][nb][0], y[h][nb][1]};
        *reinterpret_cast<d2v *>(dst + 2) = d2v{y[h][nb][2], y[h][nb][3]};
      }
    }
  }
  P2_STAMP(role, 5);
#ifdef P_NO_EAGER
  return;
#endif
  // Row tiles that are pivot blocks still to come: their diagonal tile -= X X^T.  A fragments: the
  // accumulators of X^T (lane (fr, fk), step (nb, s): X[16 wave + fr][16 nb + 4 fk + s]); B
  // fragments: X through LDS.
#pragma unroll
  for (int h = 0; h < NR; ++h) {
    const int i = i0 + h;
    if (i > a.last) break;
    double *sX = smem;
      __syncthreads();               // (the image / the previous half's X has been read by every wavefront)
    // (columns permuted inside a row, sx_col: the 16-byte fragment reads below are served in the lane groups
    //  {0-3, 12-15, 20-27}, ... -- fk = 0 for eight of a group's fragment rows, fk = 1 for the other eight -- and with
    //  the columns in their natural order the two halves of a group overlapped on two of its sixteen 16-byte slots:
    //  a 2-way conflict on every read, tools/lds_bank_model.py)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      double *dst = sX + (16 * wave + fr) * XLD + sx_col(nb, fk);
      *reinterpret_cast<d2v *>(dst) = d2v{y[h][nb][0], y[h][nb][1]};
      *reinterpret_cast<d2v *>(dst + 2) = d2v{y[h][nb][2], y[h][nb][3]};
    }
    double *Dt = M + (size_t)(64 * i) * ld + 64 * i;
    pd4 dac[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (P_DAC_PREFETCH && NR == 1 && h == 0) {
        dac[m] = dacp[m];
      } else if (m <= wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dac[m][r] = Dt[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * m + fr];
      }
    }
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m > wave) continue;      // (10 of the tile's 16 blocks)
        const double *f = sX + (16 * m + fr) * XLD + sx_col(nb, fk);
        const d2v lo = *reinterpret_cast<const d2v *>(f);
        const d2v hi = *reinterpret_cast<const d2v *>(f + 2);
        dac[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-y[h][nb][0], lo.x, dac[m], 0, 0, 0);
        dac[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-y[h][nb][1], lo.y, dac[m], 0, 0, 0);
        dac[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-y[h][nb][2], hi.x, dac[m], 0, 0, 0);
        dac[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-y[h][nb][3], hi.y, dac[m], 0, 0, 0);
      }
    }
    // (a full next pivot block goes from these registers straight into its factorisation, which writes
    //  the tile itself; a partial one reads its rows below the active part back from memory)
    if (!(chain && a.next_nact == 64)) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
        if (m <= wave) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Dt[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * m + fr] = dac[m][r];
        }
    }
    P2_STAMP(role, 6);
    if (CHAIN && NR == 1 && (a.mode & P_TAILD) && i == a.j + 1 && a.next_nact > 0) {
      // the next pivot block, complete now and still in registers: factored here, in the shadow of
      // the launch's other items (its image goes to the other slot: this launch still reads its own)
      __syncthreads();
      double *sD = smem;
      const int nact = a.next_nact;
      const int nlive = (RED && a.red.live_rows > 0) ? a.red.live_rows - 64 * (a.j + 1) : 64;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int li = 16 * wave + fk + 4 * r, lj = 16 * m + fr;
          double v = (li < nact && lj < nact && m <= wave) ? dac[m][r] : (li == lj ? 1.0 : 0.0);
          if (lj > li) v = 0.0;
          sD[li * BLD + lj] = v;
        }
      __builtin_amdgcn_s_setprio(3);
      P2_STAMP(0, 0);
      P2_CHAIN(1, wall_clock64());
      P2_CHAIN(3, cu_key());
#ifdef SP_PANEL_TRACE
      panel_diag_core(Dt, ld, nact, img_star + sp_img_off(a.j + 1),
                      a.info ? a.info + mtx : nullptr, smem, tid,
                      (mtx == 0 && a.j < 64) ? &g_p2trace[(a.j * 3) * 16 + 8] : nullptr, nlive);
#else
      panel_diag_core(Dt, ld, nact, img_star + sp_img_off(a.j + 1),
                      a.info ? a.info + mtx : nullptr, smem, tid, nullptr, nlive);
#endif
      if (RED && a.red.lnlike) {
        // the last pivot block of the system: everything the reduction reads is final (the earlier
        // columns by earlier launches, the last ones by this workgroup just now)
        // (same workgroup, same CU: the barrier's workgroup-scope ordering is all the visibility it takes -- an
        //  agent-scope release here would write the XCD's whole L2 back, 2-9 us)
        __syncthreads();
        lnlike_reduce_body<false>(M, ld, a.red.K, a.red.M, a.info ? a.info + mtx : nullptr, a.red.lnlike + mtx,
                                 a.red.status ? a.red.status + mtx : nullptr,
                                 a.red.status_out ? a.red.status_out + mtx : nullptr,
                                 a.red.stars ? a.red.stars + mtx : nullptr,
                                 a.red.coef ? static_cast<const RedCoef *>(a.red.coef) + mtx : nullptr,
                                 a.red.coef ? a.red.rscal + (size_t)mtx * (SP_RSCAL_HEAD + a.red.M) : nullptr, a.red.dvec,
                                 smem, tid);
      }
      __builtin_amdgcn_s_setprio(0);
      P2_CHAIN(2, wall_clock64());
      P2_STAMP(0, 2);
    }
  }
}

// Chain-aware layout of a launch (PanelArgs.lay).  On an idle GPU the hardware deals the workgroups of a launch
// to the CUs of an XCD in a fixed order with period 32 (tools/panel2_trace.py <S> <K> <j>: XCD-local dispatch
// indices k, k + 32 and k + 64 share a CU), and the critical chain of a launch -- the first item and the
// diagonal block in its tail, 24 us of mostly dependent fp64 vector instructions -- takes 42-46 us when its
// CU also runs two items' worth of fp64 MFMAs (same pipes; the launches of the first super-panel all ended on
// it with the other CUs idle).  So the launch is laid out by CU:
//   k = 0 .. spx - 1 (spx stars per XCD)     the chain items, one CU each;
//   k = 32 + s, 64 + s                       SLEEPERS: a workgroup that holds the chain CU's other slots and does
//                                            nothing.  It first checks that it really shares the CU of star s's
//                                            chain workgroup (both read the hardware's CU identifiers; with other
//                                            kernels on the GPU the deal is irregular and it leaves at once), then
//                                            sleeps until that workgroup is done (or 100 us);
//   the other k                              the launch's other items, one row tile each (and the look-ahead items),
//                                            on the other 32 - spx CUs; what does not fit there takes sleepers'
//                                            places, then k >= 96.  (Tried here: 128-row pair items to make
//                                            everything fit -- a pair gets a third of its CU's MFMA issue like any
//                                            workgroup and ends at twice the others' time; the look-ahead item as
//                                            the chain's fixed CU mate -- the blocks of the second super-panel,
//                                            which have their CU to themselves otherwise, go from 14 to 17-20 us.)
// Nobody waits for a sleeper and the chain waits for nobody: only time is at stake.
// first column block of a row tile's left-looking product: the super-panel's first, or -- an identity riding along
// -- the first one in which the tile holds anything (what lies left of it is zero: nothing to multiply)
__device__ __forceinline__ int tri_first_block(const PanelArgs &a, int row_tile) {
  if (a.tri0 < 0) return a.s0;
  const int first = 64 * row_tile > a.tri0 ? (64 * row_tile - a.tri0) / 64 : 0;
  return first > a.s0 ? first : a.s0;
}

__device__ __forceinline__ unsigned long long *chain_words(double *img_star) {
  return reinterpret_cast<unsigned long long *>(img_star + SP_IMG_WORDS);   // [0] seq << 32 | CU key, [1] seq when done
}

// One instantiation per kind of launch (round 4; round 3's single kernel carried every role behind run-time
// flags: 363 spilled SGPRs, the reloads in the diagonal block's loops -- on the critical chain):
//   PK_D      D items only (pivot block 0)
//   PK_LAY    the chain-aware layout: T items of 64 rows, the chain's tail, sleepers, look-ahead items
//   PK_PLAIN  T items dealt by workgroup index, 128-row pairs where the launcher asks for them
enum { PK_D = 0, PK_LAY = 1, PK_PLAIN = 2 };

template <int KIND, bool LAZY, bool RED>
__global__ __launch_bounds__(256, P_WGS) void panel_kernel(PanelArgs a) {
  __shared__ __attribute__((aligned(16))) double smem[P_LDS];
  const int tid = threadIdx.x;
#ifdef SP_PANEL_TRACE
  if (tid == 0 && a.j < 16 && blockIdx.x < 1024 && (a.mode & P_TITEMS)) g_p2cu[a.j * 1024 + blockIdx.x] = (int)cu_key() + 1;
  P2WgStamp wg_stamp(a.j, (a.mode & P_TITEMS) != 0);
#endif
  if constexpr (KIND == PK_LAY) {
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3, spx = a.S >> 3;
    const int c = k & 31, rnd = k >> 5, nb = 32 - spx;
    const int nsingle = a.ntile - a.j - 2;                      // row tiles below the chain's: one item each
    const int nl = (a.mode & P_LOOKAHEAD) ? 1 : 0;
    const int reg = 3 * nb;                                     // places on the CUs the chains leave
    const int extra = spx * (nsingle + nl) - reg;               // items beyond them ...
    const int nslots = 2 * spx;                                 // ... go to the chain CUs' other slots first
    int pos;
    if (k < 96 && c < spx) {
      const int mtx = x * spx + c;
      double *img_star = a.img + (size_t)mtx * a.lts;
      unsigned long long *F = chain_words(img_star);
      if (rnd == 0) {
        // the chain: row tile j + 1, then pivot block j + 1
        if (tid == 0)
          __hip_atomic_store(F, ((unsigned long long)(unsigned)a.seq << 32) | (unsigned long long)cu_key(),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool fla = a.mode & P_FIRSTLA;
        panel_tile_item<1, LAZY, RED, true>(a, a.sys + (size_t)mtx * a.stride, mtx, a.j + 1, a.j, fla ? a.j - 1 : a.s0,
                                            a.j, false, !fla, img_star, smem, tid);
        if (tid == 0) __hip_atomic_store(F + 1, (unsigned long long)(unsigned)a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      const int so = rnd == 2 ? c : spx + c;                    // this slot among the chain CUs' spare ones
      if (so >= extra) {
        if (tid == 0) {
          // sleeper (the other wavefronts leave; this one keeps the slot)
          const unsigned long long mine = ((unsigned long long)(unsigned)a.seq << 32) | (unsigned long long)cu_key();
          const long long t0 = wall_clock64();
          unsigned long long v;
          while (((v = __hip_atomic_load(F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (unsigned)a.seq &&
                 wall_clock64() - t0 < 250)
            __builtin_amdgcn_s_sleep(4);
          if (v == mine)
            while (__hip_atomic_load(F + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)a.seq &&
                   wall_clock64() - t0 < 10000)
              __builtin_amdgcn_s_sleep(32);
        }
        return;
      }
      pos = reg + so;
    } else {
      pos = k < 96 ? rnd * nb + (c - spx) : reg + nslots + (k - 96);
    }
    const int sl = pos % spx, idx = pos / spx;                   // star of the XCD, item of the star
    if (idx >= nsingle + nl) return;
    const int mtx = x * spx + sl;
    double *M = a.sys + (size_t)mtx * a.stride;
    double *img_star = a.img + (size_t)mtx * a.lts;
    // (one call site for the launch's other items: what differs between a row tile's item and the look-ahead
    //  item is data -- three inlined copies of the item cost the instantiation its registers)
    const bool la = idx >= nsingle;
    const int it = la ? a.j + 2 : a.j + 2 + idx;
    panel_tile_item<1, LAZY, false, false>(a, M, mtx, it, la ? a.j + 1 : a.j, tri_first_block(a, it), a.j, la,
                                           true, img_star, smem, tid);
    return;
  }
  constexpr int nd = KIND == PK_D ? 1 : 0;                                         // D items per star
  const int nt = KIND == PK_D ? 0 : panel_titems(a.ntile, a.j, a.pair);          // T items per star
  const int nl = (KIND != PK_D && (a.mode & P_LOOKAHEAD)) ? 1 : 0;                 // look-ahead items per star
  int mtx, strip;
  if (!sp_xcd_decode(blockIdx.x, a.S, nd + nt + nl, mtx, strip)) return;
  double *M = a.sys + (size_t)mtx * a.stride;
  double *img_star = a.img + (size_t)mtx * a.lts;
  if constexpr (KIND == PK_D) {
    P2_STAMP(0, 0);
    double *img = img_star + sp_img_off(a.j);
#ifdef SP_PANEL_TRACE
    panel_diag_item(M, a.ld, a.j, a.nact, img, a.info ? a.info + mtx : nullptr, smem, tid,
                    (mtx == 0 && a.j < 64) ? &g_p2trace[(a.j * 3) * 16 + 8] : nullptr);
#else
    panel_diag_item(M, a.ld, a.j, a.nact, img, a.info ? a.info + mtx : nullptr, smem, tid);
#endif
    P2_STAMP(0, 2);
  } else if (strip < nt) {
    // T item t: row tile j + 1 + t; with pairs: 0 = row tile j + 1 alone (it is on the critical path),
    // t >= 1 = row tiles j + 2 t, j + 2 t + 1 (the last one may be single)
    const int t = strip;
    const int i0 = a.pair ? (t == 0 ? a.j + 1 : a.j + 2 * t) : a.j + 1 + t;
    // (the first tile after a look-ahead: all but the last column block of its product is in it already)
    const bool fla = t == 0 && (a.mode & P_FIRSTLA);
#if P_PAIRS
    if (a.pair && t > 0 && i0 + 1 < a.ntile)
      panel_tile_item<2, LAZY, false, false>(a, M, mtx, i0, a.j, a.s0, a.j, false, true, img_star, smem, tid);
    else
#endif
    if (t == 0)
      panel_tile_item<1, LAZY, RED, true>(a, M, mtx, i0, a.j, fla ? a.j - 1 : a.s0, a.j, false, !fla, img_star, smem, tid);
    else
      panel_tile_item<1, LAZY, false, false>(a, M, mtx, i0, a.j, tri_first_block(a, i0), a.j, false, true, img_star, smem, tid);
  } else {
    // look-ahead: tile (j + 2, j + 1) with the column blocks s0 .. j - 1
    panel_tile_item<1, LAZY, false, false>(a, M, mtx, a.j + 2, a.j + 1, a.s0, a.j, true, true, img_star, smem, tid);
  }
}

}  // namespace

// One launch of the panel kernel (see PanelArgs): pivot block j of the super-panel that starts at s0.
//   what: SP_PANEL_D (D items only: pivot block j), or SP_PANEL_T (T items) with any of
//   SP_PANEL_TAILD (the first T item factors block j + 1, next_nact active columns), SP_PANEL_LA (a
//   look-ahead item for tile (j + 2, j + 1)), SP_PANEL_FIRSTLA (launch j - 1 had one).
int sp_launch_panel2(int layout, const SpReduceArgs *red, double *sys, long ld, long stride, int S, int ntile, int j, int s0, int nact,
                     int next_nact, int last, int what, int ncu, double *img, long lts, int32_t *info,
                     hipStream_t st, const LazyCov *lazy) {
  if (S <= 0) return SP_OK;
  PanelArgs a;
  a.sys = sys; a.ld = ld; a.stride = stride; a.S = S; a.ntile = ntile; a.j = j; a.s0 = s0;
  a.nact = nact; a.next_nact = next_nact; a.last = last;
  a.mode = ((what & SP_PANEL_D) ? P_DITEMS : 0) | ((what & SP_PANEL_T) ? P_TITEMS : 0) |
           ((what & SP_PANEL_TAILD) ? P_TAILD : 0) | ((what & SP_PANEL_LA) ? P_LOOKAHEAD : 0) |
           ((what & SP_PANEL_FIRSTLA) ? P_FIRSTLA : 0);
  a.img = img; a.lts = lts; a.info = info;
  a.tri0 = (layout & 2) ? (layout >> 8) : -1;     // (layout bits 8..: the identity's first row, see bit 1)
  a.lz = lazy ? *lazy : LazyCov{};
  a.red = red ? *red : SpReduceArgs{};
  if (a.red.lnlike && !(a.mode & P_TAILD)) return SP_ERR_INVALID;
  if ((a.mode & P_DITEMS) && (a.mode & ~P_DITEMS)) return SP_ERR_INVALID;
  if ((a.mode & (P_TAILD | P_LOOKAHEAD | P_FIRSTLA)) && !(a.mode & P_TITEMS)) return SP_ERR_INVALID;
  if ((a.mode & P_LOOKAHEAD) && (j + 2 >= ntile || j <= s0)) return SP_ERR_INVALID;
  if ((a.mode & P_FIRSTLA) && j < s0 + 2) return SP_ERR_INVALID;
  if ((ld & 1) || (stride & 1) || (reinterpret_cast<uintptr_t>(sys) & 15)) return SP_ERR_INVALID;
  if (j < s0 || j >= ntile) return SP_ERR_INVALID;
  // 128-row items where 64-row ones would not fit the CUs in one round (three workgroups per CU): a
  // second round only starts when the first one's items end.  Fewer, fatter workgroups are slower one
  // by one (a CU hides the stalls of one workgroup behind another's MFMAs), so they are not used
  // where there is room.
  {
    const long n64 = (long)S * ((ntile - j - 1) + ((a.mode & P_LOOKAHEAD) ? 1 : 0));
    // (up to an eighth over: the stragglers of a short second round cost less than the pairs' slower products)
    // (layout bit 1: never -- the identity riding along in sp_spd_inverse_batched keeps every launch at 17 row tiles
    //  per star, 1.4 rounds of 64-row items, where pairs measured 114 us against 83-90)
    a.pair = P_PAIRS && !(layout & 2) && (a.mode & P_TITEMS) && 8 * n64 > 9L * P_WGS * ncu;
  }
  const int per_star = ((a.mode & P_DITEMS) ? 1 : 0) + ((a.mode & P_TITEMS) ? panel_titems(ntile, j, a.pair) : 0) +
                       ((a.mode & P_LOOKAHEAD) ? 1 : 0);
  if (per_star <= 0) return SP_OK;
  long nblk = sp_xcd_grid(S, per_star);
  // chain-aware layout (panel_kernel): launches with a diagonal block in their tail, whole stars per XCD
  a.lay = 0; a.seq = 0;
  {
    const int lay_on = layout & 1;
    static std::atomic<int> seq{0};
    const int spx = S / 8;
    if (lay_on && P_PAIRS && (a.mode & P_TAILD) && S % 8 == 0 && spx >= 1 && spx <= 16 && ntile - j - 2 >= 0) {
      const int nrows = ntile - j - 2, la = (a.mode & P_LOOKAHEAD) ? 1 : 0;
      a.lay = 1;
      int sq = ++seq;
      if (sq == 0) sq = ++seq;
      a.seq = sq;
      const long extra = (long)spx * (nrows + la) - 3L * (32 - spx), nslots = 2 * spx;
      nblk = 8 * (96 + (extra > nslots ? extra - nslots : 0));
    }
  }
  if (nblk > 0x7fffffffL) return SP_ERR_INVALID;
  const bool lz = a.lz.theta != nullptr, rd = a.red.lnlike != nullptr;
#define SP_PANEL_GO(KIND)                                                                                    \
  do {                                                                                                       \
    if (lz && rd) hipLaunchKernelGGL((panel_kernel<KIND, true, true>), dim3((unsigned)nblk), dim3(256), 0, st, a);        \
    else if (lz) hipLaunchKernelGGL((panel_kernel<KIND, true, false>), dim3((unsigned)nblk), dim3(256), 0, st, a);        \
    else if (rd) hipLaunchKernelGGL((panel_kernel<KIND, false, true>), dim3((unsigned)nblk), dim3(256), 0, st, a);        \
    else hipLaunchKernelGGL((panel_kernel<KIND, false, false>), dim3((unsigned)nblk), dim3(256), 0, st, a);               \
  } while (0)
  if (a.mode & P_DITEMS)
    hipLaunchKernelGGL((panel_kernel<PK_D, false, false>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  else if (a.lay)
    SP_PANEL_GO(PK_LAY);
  else
    SP_PANEL_GO(PK_PLAIN);
#undef SP_PANEL_GO
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// (debug) stamps of the panel kernel, variant builds with -DSP_PANEL_TRACE only: reset (out == null)
// or copy out 64 x 3 x 16 int64
extern "C" int sp_debug_panel2_trace(long long *out) {
#ifdef SP_PANEL_TRACE
  if (!out) {
    static long long zeros[64 * 3 * 16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_p2trace), zeros, sizeof(zeros)) != hipSuccess) return SP_ERR_HIP;
    return SP_OK;
  }
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2trace), sizeof(long long) * 64 * 3 * 16) != hipSuccess) return SP_ERR_HIP;
  return SP_OK;
#else
  (void)out;
  return SP_ERR_INVALID;
#endif
}

// (debug, variant builds) every star's tail block per launch: 16 x 64 x 4 int64, then the CU of every workgroup:
// 16 x 1024 int32 (0: none) packed behind it
extern "C" int sp_debug_panel2_chain(long long *out) {
#ifdef SP_PANEL_TRACE
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2chain), sizeof(long long) * 16 * 64 * 4) != hipSuccess) return SP_ERR_HIP;
  if (hipMemcpyFromSymbol(out + 16 * 64 * 4, HIP_SYMBOL(g_p2cu), sizeof(int) * 16 * 1024) != hipSuccess) return SP_ERR_HIP;
  if (hipMemcpyFromSymbol(out + 16 * 64 * 4 + 16 * 512, HIP_SYMBOL(g_p2wg), sizeof(long long) * 16 * 1024 * 2) != hipSuccess)
    return SP_ERR_HIP;
  return SP_OK;
#else
  (void)out;
  return SP_ERR_INVALID;
#endif
}
