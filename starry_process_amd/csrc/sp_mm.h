// Pipelined fp64 tile product for gfx950: the inner engine of the GEMM-shaped kernels
// (trailing updates, strip solves, conditional covariance products).
//
//     acc[TM x TN] += A[TM x Kd] . B[TN x Kd]^T            (row-major A, B; "NT")
//
// One 256-thread workgroup (4 wavefronts, one per SIMD) per tile; wavefront (wr, wc) of a
// WR x WC arrangement owns a (TM / WR) x (TN / WC) sub-tile as MA x NA accumulators of
// v_mfma_f64_16x16x4_f64.  That instruction holds a SIMD's matrix pipe for 64 cycles, so a
// BK-deep slice keeps a wavefront busy for 16 BK MA NA cycles while the slice itself is only
// (TM + TN) BK 8 bytes: the product is bound by the LATENCY of the operand stream, not by its
// width, LDS bandwidth or issue slots.  Hence:
//   * operand slices go global -> LDS directly (global_load_lds_dwordx4, no registers, no
//     ds_write pass), NS stages deep: NS - 1 slices are in flight while one is multiplied;
//   * ONE workgroup barrier per slice: a wavefront waits for its own share of slice s with a
//     COUNTED s_waitcnt vmcnt (the younger slices stay in flight), the barrier makes every
//     wavefront's share visible and proves that the stage about to be refilled has been read;
//   * the LDS image is lane-linear (that is what the DMA writes) and un-padded; bank conflicts
//     of the fragment reads (16 rows at one k) are removed by an XOR swizzle of the 16-byte
//     granule index with the row, applied on the SOURCE address of the DMA and on the read.
//     The swizzle is cut for the lane groups gfx950 serves a ds_read_b128 in (MM2::swz).
//
// Requirements (checked by the launchers, which fall back to gemm_nt_kernel otherwise):
//   full tiles, Kd a multiple of BK, lda / ldb even, 16-byte aligned bases.
#ifndef SP_MM_H
#define SP_MM_H

#include <hip/hip_runtime.h>

typedef double mm_d4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void mm_wait_vmcnt() {
  // (inline asm: the compiler's own bookkeeping would drain the queue to 0)
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else static_assert(N < 0, "add the immediate to mm_wait_vmcnt");
}

// TM, TN: tile edges (multiples of 16 WR / 16 WC); BK: slice depth (8, 16 or 32);
// NS: LDS stages (>= 3); WR: wavefronts along the rows (1, 2 or 4).
// (Round 2 measured a first form, MM, that read 8-byte fragments from the same LDS image just in time
//  -- 0.58 / 0.70 of peak where this one reaches 0.62 / 0.81 -- and a wave-tile form without LDS, WT:
//  both are in the git history, tools/mm_bench.py of round 2 compared them.)  On the consuming side:
//   * the MFMA sums over its 4 k-entries whatever their order as long as A and B agree, so lane
//     (r = lane & 15, q = lane >> 4) takes the k-PAIR {8 c + 2 q, 8 c + 2 q + 1} of each 8-column
//     chunk c with ONE ds_read_b128 and feeds two MFMAs with it (.x then .y): half the LDS
//     instructions, and the read is conflict-free on the same swizzled image;
//   * the fragments of slice s + 1 are read into a second register set while the MFMAs of slice s
//     run from the first: the matrix pipe never waits for an LDS read.  The barrier of iteration s
//     therefore guarantees slice s + 1 (not s), and the slice issued after it is s + NS.
template <int TM, int TN, int BK, int NS, int WR>
struct MM2 {
  static constexpr int TM_ = TM, TN_ = TN;
  static constexpr int WC = 4 / WR;
  static constexpr int MA = TM / (16 * WR), NA = TN / (16 * WC);
  static constexpr int G = BK / 2, RPI = 64 / G, RPB = 32 / BK;
  static constexpr int IA = TM / RPI, IB = TN / RPI, LA = IA / 4, LB = IB / 4, LPW = LA + LB;
  static constexpr int STAGE_A = TM * BK, STAGE_B = TN * BK, STAGE = STAGE_A + STAGE_B;
  static constexpr int LDS_DOUBLES = NS * STAGE;
  static constexpr int NC = BK / 8;          // 8-column chunks per slice
  static_assert(BK == 8 || BK == 16 || BK == 32, "slice depth");
  static_assert(IA % 4 == 0 && IB % 4 == 0, "a slice must split evenly over four wavefronts");
  static_assert(NS >= 3, "one slice in registers, one landing, one being issued");
  typedef double v2 __attribute__((ext_vector_type(2)));
  struct Frags {
    v2 a[NC][MA], b[NC][NA];
  };
  // gfx950 serves a ds_read_b128 in four groups of 16 lanes that are NOT contiguous:
  //   {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}
  // (MI355X_MICROARCH.md, LDS): with lane = 16 q + fr a group holds every fragment row fr once, rows
  // 0-3 and 12-15 at one k-granule q and rows 4-11 at q ^ 1.  Conflict-free = the 16 lanes of a group on
  // the 16 distinct 16-byte slots of a 256-byte bank row.  Row fr sits in slot block (fr mod RPB) G of
  // its bank row, so the rows h = fr / RPB of one block must land on distinct granules: granule
  // (4 c + q) ^ h ^ t(fr), t = 1 for rows 4-11 (undoes the group's q ^ 1).  Round 3 swizzled with h alone
  // -- right for contiguous groups of 16 lanes, a 2-way conflict on every read for the real ones
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.49, profiles/r03_pmc_sq.txt).
  __device__ static __forceinline__ int swz(int row) {
    const int fr = row & 15;
#ifdef SP_MM_OLD_SWZ
    return (fr / RPB) & (G - 1);
#endif
    return ((fr / RPB) ^ (((fr >> 2) ^ (fr >> 3)) & 1)) & (G - 1);
  }

  const double *srcA[LA > 0 ? LA : 1];
  const double *srcB[LB > 0 ? LB : 1];
  int rdA, rdB, fsw, q, wave;

  __device__ __forceinline__ void init(const double *A, long lda, const double *B, long ldb) {
    const int lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    const int r = lane / G, p = lane % G;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int row = (wave + 4 * i) * RPI + r;
      srcA[i] = A + (size_t)row * lda + 2 * (p ^ swz(row));
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int row = (wave + 4 * i) * RPI + r;
      srcB[i] = B + (size_t)row * ldb + 2 * (p ^ swz(row));
    }
    const int wr = wave / WC, wc = wave % WC, fr = lane & 15;
    rdA = (wr * (TM / WR) + fr) * BK;
    rdB = STAGE_A + (wc * (TN / WC) + fr) * BK;
    fsw = swz(fr);
    q = lane >> 4;
  }
  __device__ __forceinline__ void issue(double *lds, int st, int k0) const {
    double *base = lds + st * STAGE;
#pragma unroll
    for (int i = 0; i < LA; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(srcA[i] + k0),
          (__attribute__((address_space(3))) void *)(base + (wave + 4 * i) * 128), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < LB; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(srcB[i] + k0),
          (__attribute__((address_space(3))) void *)(base + STAGE_A + (wave + 4 * i) * 128), 16, 0, 0);
  }
  __device__ __forceinline__ void fetch(const double *lds, int st, Frags &f) const {
    const double *base = lds + st * STAGE;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int goff = 2 * ((4 * c + q) ^ fsw);
#pragma unroll
      for (int m = 0; m < MA; ++m) f.a[c][m] = *reinterpret_cast<const v2 *>(base + rdA + m * 16 * BK + goff);
#pragma unroll
      for (int n = 0; n < NA; ++n) f.b[c][n] = *reinterpret_cast<const v2 *>(base + rdB + n * 16 * BK + goff);
    }
  }
  __device__ __forceinline__ void mul(const Frags &f, mm_d4 (&acc)[MA][NA]) const {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
          for (int n = 0; n < NA; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? f.a[c][m].y : f.a[c][m].x,
                                                            h ? f.b[c][n].y : f.b[c][n].x, acc[m][n], 0, 0, 0);
  }
  // multiply the slice held in `cur` while the fragments of the next one are read into `nxt`:
  // the reads sit BEHIND the first MFMAs in program order (one read per MFMA), so the wait that
  // precedes the MFMAs covers only `cur`'s own, older reads
  __device__ __forceinline__ void mul_fetch(const Frags &cur, Frags &nxt, const double *lds, int st,
                                            mm_d4 (&acc)[MA][NA]) const {
    const double *base = lds + st * STAGE;
    constexpr int NR = NC * (MA + NA);
    int issued = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = 0; m < MA; ++m)
#pragma unroll
          for (int n = 0; n < NA; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? cur.a[c][m].y : cur.a[c][m].x,
                                                            h ? cur.b[c][n].y : cur.b[c][n].x, acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (issued < NR) {
              const int rc = issued / (MA + NA), ri = issued % (MA + NA);
              const int goff = 2 * ((4 * rc + q) ^ fsw);
              if (ri < MA)
                nxt.a[rc][ri] = *reinterpret_cast<const v2 *>(base + rdA + ri * 16 * BK + goff);
              else
                nxt.b[rc][ri - MA] = *reinterpret_cast<const v2 *>(base + rdB + (ri - MA) * 16 * BK + goff);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              ++issued;
            }
          }
    static_assert(NR <= NC * 2 * MA * NA, "more fragment reads than MFMAs to hide them behind");
  }
  // slices 0 .. NS - 2 on their way
  __device__ __forceinline__ void prologue(double *lds, int k_begin, int k_end) const {
    const int nsl = (k_end - k_begin) / BK;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
      if (p < nsl) issue(lds, p, k_begin + p * BK);
  }
  // fence: the slices from index `fence` on read memory that this workgroup has just stored; the
  // whole vector-memory queue (those stores included) is drained once, ahead of the barrier that
  // precedes the issue of slice `fence`.  Slices issued by prologue() are not covered:
  // fence < NS - 1 needs a drain + barrier before prologue().
  __device__ __forceinline__ void loop(double *lds, int k_begin, int k_end, mm_d4 (&acc)[MA][NA],
                                       int fence = 1 << 30) const {
    const int nsl = (k_end - k_begin) / BK;
    if (nsl <= 0) return;
    Frags f0, f1;
    // slice 0 into the first register set
    if (nsl >= NS - 1) mm_wait_vmcnt<(NS - 2) * LPW>(); else mm_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    fetch(lds, 0, f0);
    int st1 = 1 % NS;            // stage of slice s + 1
    int stn = NS - 1;            // stage the next issue goes to
    // (the number of slices is even: callers give a depth that is a multiple of 2 BK)
    for (int s = 0; s < nsl; s += 2) {
      // ---- even half: multiply slice s from f0 while slice s + 1 is read into f1
      if (s + NS - 1 < nsl && s + NS - 1 != fence) mm_wait_vmcnt<(NS - 3) * LPW>(); else mm_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();   // slice s + 1 landed everywhere; slice s - 1's stage is free
      if (s + NS - 1 < nsl) issue(lds, stn, k_begin + (s + NS - 1) * BK);
      mul_fetch(f0, f1, lds, st1, acc);
      st1 = st1 + 1 == NS ? 0 : st1 + 1;
      stn = stn + 1 == NS ? 0 : stn + 1;
      // ---- odd half: multiply slice s + 1 from f1 while slice s + 2 is read into f0
      if (s + 2 < nsl) {
        if (s + NS < nsl && s + NS != fence) mm_wait_vmcnt<(NS - 3) * LPW>(); else mm_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + NS < nsl) issue(lds, stn, k_begin + (s + NS) * BK);
      }
      mul_fetch(f1, f0, lds, st1, acc);   // (past the last slice the reads fetch stale bytes: unused)
      st1 = st1 + 1 == NS ? 0 : st1 + 1;
      stn = stn + 1 == NS ? 0 : stn + 1;
    }
    __builtin_amdgcn_s_barrier();
  }
  // ---- a DIAGONAL tile of a symmetric product X X^T (TM = TN = 64, WR = 4, B = A) ------------------------------
  // Only the ten 16 x 16 blocks on and below the tile's diagonal are wanted.  With a wavefront per block row the
  // last one still multiplies four blocks while the first multiplies one: nothing is gained in time (round 3 built
  // that: +- 0).  Here the ten blocks are RE-DEALT, three / three / two / two:
  //     wavefront 0: (0,0) (3,0) (3,1)     1: (1,0) (1,1) (3,2)     2: (2,0) (2,1)     3: (2,2) (3,3)
  // a wavefront reads the fragments of the two to four block rows its blocks need (all from the A image: the tile's
  // rows are its columns) and issues 3 or 2 MFMAs per k-step where the plain loop issues 4.  Same slices, same k-order
  // per block: the blocks are the plain loop's bits.
  template <int W>
  struct SymDeal {
    static constexpr int NB = W < 2 ? 3 : 2;                                     // blocks of this wavefront
    static constexpr int NF = W == 1 ? 4 : (W == 3 ? 2 : 3);                     // block rows whose fragments it reads
    __device__ static constexpr int frow(int f) {                                // ... which ones
      return W == 0 ? (f == 0 ? 0 : f == 1 ? 1 : 3) : W == 1 ? f : W == 2 ? f : (f == 0 ? 2 : 3);
    }
    __device__ static constexpr int brow(int b) {                                // block b: its row block ...
      return W == 0 ? (b == 0 ? 0 : 3) : W == 1 ? (b < 2 ? 1 : 3) : W == 2 ? 2 : (b == 0 ? 2 : 3);
    }
    __device__ static constexpr int bcol(int b) {                                // ... and column block
      return W == 0 ? (b == 0 ? 0 : b - 1) : W == 1 ? b : W == 2 ? b : (b == 0 ? 2 : 3);
    }
    __device__ static constexpr int fidx(int row) {                              // position of a block row in the fragment set
      return W == 0 ? (row == 3 ? 2 : row) : W == 1 ? row : W == 2 ? row : row - 2;
    }
  };
  template <int W>
  struct SymFrags {
    v2 f[NC][SymDeal<W>::NF];
  };
  template <int W>
  __device__ __forceinline__ void sym_fetch(const double *lds, int st, SymFrags<W> &fr) const {
    const double *base = lds + st * STAGE + (threadIdx.x & 15) * BK;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int goff = 2 * ((4 * c + q) ^ fsw);
#pragma unroll
      for (int i = 0; i < SymDeal<W>::NF; ++i)
        fr.f[c][i] = *reinterpret_cast<const v2 *>(base + SymDeal<W>::frow(i) * 16 * BK + goff);
    }
  }
  template <int W>
  __device__ __forceinline__ void sym_mul_fetch(const SymFrags<W> &cur, SymFrags<W> &nxt, const double *lds, int st,
                                                mm_d4 (&acc)[3]) const {
    using D = SymDeal<W>;
    const double *base = lds + st * STAGE + (threadIdx.x & 15) * BK;
    constexpr int NR = NC * D::NF;
    int issued = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < D::NB; ++b) {
          const v2 a = cur.f[c][D::fidx(D::brow(b))], bb = cur.f[c][D::fidx(D::bcol(b))];
          acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? a.y : a.x, h ? bb.y : bb.x, acc[b], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (issued < NR) {
            const int rc = issued / D::NF, ri = issued % D::NF;
            const int goff = 2 * ((4 * rc + q) ^ fsw);
            nxt.f[rc][ri] = *reinterpret_cast<const v2 *>(base + D::frow(ri) * 16 * BK + goff);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            ++issued;
          }
        }
    static_assert(NR <= NC * 2 * D::NB, "more fragment reads than MFMAs to hide them behind");
  }
  // the control flow of loop(), with this wavefront's share of the ten blocks
  template <int W>
  __device__ __forceinline__ void sym_loop(double *lds, int k_begin, int k_end, mm_d4 (&acc)[3]) const {
    const int nsl = (k_end - k_begin) / BK;
    if (nsl <= 0) return;
    SymFrags<W> f0, f1;
    if (nsl >= NS - 1) mm_wait_vmcnt<(NS - 2) * LPW>(); else mm_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    sym_fetch<W>(lds, 0, f0);
    int st1 = 1 % NS, stn = NS - 1;
    for (int s = 0; s < nsl; s += 2) {
      if (s + NS - 1 < nsl) mm_wait_vmcnt<(NS - 3) * LPW>(); else mm_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (s + NS - 1 < nsl) issue(lds, stn, k_begin + (s + NS - 1) * BK);
      sym_mul_fetch<W>(f0, f1, lds, st1, acc);
      st1 = st1 + 1 == NS ? 0 : st1 + 1;
      stn = stn + 1 == NS ? 0 : stn + 1;
      if (s + 2 < nsl) {
        if (s + NS < nsl) mm_wait_vmcnt<(NS - 3) * LPW>(); else mm_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + NS < nsl) issue(lds, stn, k_begin + (s + NS) * BK);
      }
      sym_mul_fetch<W>(f1, f0, lds, st1, acc);
      st1 = st1 + 1 == NS ? 0 : st1 + 1;
      stn = stn + 1 == NS ? 0 : stn + 1;
    }
    __builtin_amdgcn_s_barrier();
  }

  __device__ __forceinline__ int acc_row(int m, int r) const {
    const int lane = threadIdx.x & 63;
    return (wave / WC) * (TM / WR) + 16 * m + (lane >> 4) + 4 * r;
  }
  __device__ __forceinline__ int acc_col(int n) const {
    const int lane = threadIdx.x & 63;
    return (wave % WC) * (TN / WC) + 16 * n + (lane & 15);
  }
};

#endif
