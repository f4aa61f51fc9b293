// Internal declarations shared by the translation units of libsp_hip.so.
// gfx950 (MI355X, CDNA4) only; wavefront = 64.
#ifndef SP_INTERNAL_H
#define SP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "../../include/starry_process_amd.h"

#define SP_WAVE 64
#define SP_MAX_YDEG 30
#define SP_MAX_UDEG 4
#define SP_NORM_MAXORDER 64

// Cholesky blocking: panels of SP_NB columns, systems padded to a multiple of
// SP_NB rows (the right-hand sides ride along as extra rows).
#define SP_NB 64

struct sp_handle {
  int ydeg, udeg, N, NWIG, device;
  // host constants
  std::vector<int32_t> l_of, m_of, mirror, m0, blk;
  std::vector<double> rT;       // phase-curve solution vector, degree ydeg+udeg
  std::vector<double> A1;       // dense change of basis, (NLU x NLU), row-major
  std::vector<double> U1;       // ((udeg+1)^2 x (udeg+1)) limb-darkening basis
  std::vector<double> rta1;     // rT . A1 at degree ydeg (N)
  // device constants
  int32_t *d_l_of, *d_m_of, *d_mirror, *d_blk;
  double *d_Rx90;               // packed Rx(pi/2)
  double *d_Rxm90;              // packed Rx(-pi/2) (sp_upstream.hip; first use)
  double *d_lamcs;              // cos / sin (m lam_q) of the lamcs_Q equispaced longitudes (sp_upstream.hip)
  int lamcs_Q;
  double *d_size_basis;         // spot-size basis Bp [ydeg + 1][spts], then the colatitude grid [spts] (sp_set_size_basis)
  int size_spts;
  double size_sfac;
  double *d_wnp, *d_Wnp;        // marginalisation constants (flux.py:121-179)
  bool have_marginal;
  double *d_xp;                 // lag grid of the last kernel table
  int xp_covpts;
  std::vector<double> xp_host;  // its host copy (re-upload only on change)
  // device state: Ylm moments
  double *d_mean_ylm, *d_cov_ylm, *d_ez, *d_Ez, *d_tmpNN;
  bool have_moments;
  // small device scratch owned by the handle
  double *d_scratch;
  size_t scratch_bytes;
  double *d_tab_scratch;        // [ntab][2][N] row reductions of the kernel table
  size_t tab_scratch_bytes;
  bool table_attr_done;         // dynamic-LDS opt-in of table_finish_kernel made on this handle's device
  // grow-only device scratch of the non-fused ops (sp_cov_*_batched, sp_cho_factor, ...): owned
  // by the handle, so two handles on one GPU never share it
  void *big_ptr;
  size_t big_bytes;
  // cos / sin staging of sp_Rx: a ring of pinned host + device buffer pairs, each guarded by the
  // event of its last use (no allocation, no stream synchronisation in the launch path)
  struct CsSlot {
    double *host;
    double *dev;
    size_t cap;                 // doubles
    hipEvent_t done;
    bool used;
  };
  std::vector<CsSlot> cs_ring;  // (sp_stage_acquire: grows while every slot is still in flight, up to SP_STAGE_MAX)
  int cs_next;
  int superpanel;               // panels per super-panel (SP_SUPER; 0 = chosen from K)
  int groups;                   // concurrent star groups (SP_GROUPS, default 1)
  int ncu;                      // compute units of the device
  int look_ahead;               // panel launches carry a look-ahead item (sp_cholesky.hip; SP_PANEL_LA, default 1)
  int panel_layout;             // panel launches laid out by CU (sp_panel.hip; SP_PANEL_LAYOUT, default 1)
  int fuse_reduce;              // the reduction rides in the last panel launch's tail where it can (SP_FUSE_REDUCE, default 1)
  std::vector<hipStream_t> gstream;
  std::vector<hipEvent_t> gdone;
  hipEvent_t gfork;
  int defer_norm;               // likelihood path: deferred normalisation (SP_DEFER_NORM, default 1)
  int lazy_cov;                 // ... with covariance tiles formed at first touch (SP_LAZY_COV, default 1)
  // optional per-launch timing of the factorisation's launches by kind (bench roofline)
  bool prof_on;
  unsigned prof_mask;                // kinds that are bracketed (bit k = kind k)
  std::vector<hipEvent_t> prof_ev;   // pairs (start, stop)
  std::vector<int> prof_kind;        // kind of pair i
  std::vector<double> prof_fl;       // algorithmic flops of pair i, counted on the K cadences (+ the M riding residual rows)
  std::vector<double> prof_flp;      // the same count on the PADDED system (rows up to roundup(K + M + 2, 64)): what runs
  std::vector<int> prof_n;           // launches bracketed by pair i
  size_t prof_used;                  // events handed out so far
};

// kinds of timed launches (sp_profile_kind; 1 and 3 were round 2's strip solves and assembly)
// SP_PROF_PANELS: the panel kernels of a whole super-panel under ONE pair of events (cheap enough
// for a timed region: 2 pairs per K = 1000 factorisation); SP_PROF_CHAIN and SP_PROF_PANEL_LAUNCH:
// every panel launch under its own pair (comparable with rocprofv3's durations)
enum {
  SP_PROF_SYRK = 0, SP_PROF_CHAIN = 2, SP_PROF_PANELS = 4, SP_PROF_PANEL_LAUNCH = 5, SP_PROF_NKINDS = 6
};

// brackets the launches issued during its lifetime with a pair of events on `st`.  Scopes nest (the
// panel driver holds a PANELS scope around per-launch CHAIN / PANEL_LAUNCH scopes): a scope RESERVES
// its pair of events when it is constructed and keeps the index, so an inner scope never touches
// the outer one's slot.
struct SpProfScope {
  sp_handle *h;
  hipStream_t st;
  bool on;
  size_t idx;     // first event of this scope's pair
  SpProfScope(sp_handle *h_, hipStream_t st_, int kind, double flops, int launches = 1, double flops_padded = -1.0)
      : h(h_), st(st_), on(false), idx(0) {
    if (!h || !h->prof_on || !((h->prof_mask >> kind) & 1u) || h->prof_used + 2 > h->prof_ev.size()) return;
    idx = h->prof_used;
    h->prof_used += 2;
    h->prof_kind[idx / 2] = kind;
    h->prof_fl[idx / 2] = flops;
    h->prof_flp[idx / 2] = flops_padded < 0.0 ? flops : flops_padded;
    h->prof_n[idx / 2] = launches;
    // (a pair whose start could not be recorded stays reserved with zero launches: read as empty)
    if (hipEventRecord(h->prof_ev[idx], st) != hipSuccess) {
      h->prof_n[idx / 2] = 0;
      h->prof_fl[idx / 2] = 0.0;
      h->prof_flp[idx / 2] = 0.0;
      h->prof_kind[idx / 2] = -1;
      return;
    }
    on = true;
  }
  // (a scope around several launches: add each one's algorithmic flops as it is issued)
  void add(double flops, int launches = 1, double flops_padded = -1.0) {
    if (!on) return;
    h->prof_fl[idx / 2] += flops;
    h->prof_flp[idx / 2] += flops_padded < 0.0 ? flops : flops_padded;
    h->prof_n[idx / 2] += launches;
  }
  ~SpProfScope() {
    if (!on) return;
    if (hipEventRecord(h->prof_ev[idx + 1], st) != hipSuccess) h->prof_kind[idx / 2] = -1;
  }
  SpProfScope(const SpProfScope &) = delete;
  SpProfScope &operator=(const SpProfScope &) = delete;
};

const char *sp_set_hip_error(hipError_t e, const char *what);

// A staging slot of at least `doubles` doubles (pinned host + device buffer) that no copy in flight still reads: the
// oldest slot whose event has completed (hipEventQuery -- never a host wait: hipEventSynchronize on an event recorded
// behind a kernel launch waits until the stream has DRAINED, the runtime gives kernels no completion signal of their
// own: 0.9 ms per call with a step's launches queued, round 6), a new slot while all are busy, and only with
// SP_STAGE_MAX slots in flight a wait for the oldest.  The caller fills c->host, enqueues its copy and records c->done
// behind the last launch that reads c->dev, then sets c->used.
#define SP_STAGE_MAX 64
int sp_stage_acquire(sp_handle *h, size_t doubles, sp_handle::CsSlot **out);

#define SP_HIP(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) {                            \
      sp_set_hip_error(e_, #call);                     \
      return SP_ERR_HIP;                               \
    }                                                  \
  } while (0)

#define SP_LAUNCH_CHECK()                              \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) {                            \
      sp_set_hip_error(e_, "kernel launch");           \
      return SP_ERR_HIP;                               \
    }                                                  \
  } while (0)

static inline int sp_nwig_of(int l) {
  return ((l + 1) * (2 * l + 1) * (2 * l + 3)) / 3;
}
static inline int sp_roundup(int x, int m) { return ((x + m - 1) / m) * m; }

// covariance tiles formed at first touch (sp_cov.h); theta == null: every tile comes from memory.
// (the star's table is staged in the LDS of the kernel that forms a tile: 4 (covpts + 4) doubles must
//  fit the smallest of those scratch areas, the one-launch panel kernel's 4544 doubles)
#define SP_TILE_LDS_MIN 4544
struct LazyCov {
  const double *theta;     // [S][K] phases
  const double *t;         // [S][K] cadence times (temporal kernels)
  const sp_star *stars;
  const double *ptab;      // [S][4 np]: the star's table as SplineGen reads it ({a0, a1} pairs, then {a2, a3})
  int K, covpts, temporal;
  int nfull;               // row tiles 0 .. nfull - 1 hold covariance rows only
  int tr0, tc0;            // system tile coordinates of the launch's tile (0, 0)
  int c0lazy;              // the tiles of block column 0 are left to their first touch too (the planned step: its
                           // assembly does not write them; panel launch 0 forms them)
  int no_panels;           // the panel launches form nothing (temporal kernels: an exponential per entry has no place
                           // in the panel kernel): the first super-panel's block columns come from memory, only the
                           // first trailing update forms its tiles
  // The rows BELOW the cadences of a tile left of the diagonal (rid != null: the planned step, which then leaves the
  // row tiles that hold them to their first touch as well -- nfull = every row tile): rid[star][m][col], m < nrid, is
  // row K + m of the star's system as the assembly would have written it (the residuals flux[m] - baseline_mean, the
  // row of ones, the variances / c1; zero beyond the star's cadences); rows from K + nrid on are zero left of the
  // diagonal.  One pointer and one count: the panel kernel's lazy instantiations have no scalar registers to spare.
  // Behind a star's riding rows the block holds one more row: dd[col] = D_col / c1, what the DIAGONAL gets on top of
  // the covariance (sp_reduce.h: B = Sigma + D / c1) -- for the kernels that form diagonal tiles (dlazy).
  const double *rid;       // [S][nrid + 1][K]
  int nrid;
  int dlazy;               // bit 1: the first trailing update forms its diagonal tiles (all but its tile (0, 0), which the
                           // eager updates of the first super-panel keep in memory) instead of loading them
  const double *inorder;   // [S] (or null) 1.0: the star's cadences are in non-decreasing order (the data plan knows) -- the
                           // Matern-3/2 factor of a tile below the diagonal then separates into row and column factors
};

// does the symmetric trailing update of a remainder of nb 64-column blocks run on the 64 x 64 kernel whose diagonal
// tiles can be formed at first touch (sp_gemm.hip: not the 128 x 64 tiles of large remainders, not with SP_SYRK_SYMDIAG=0)?
int sp_syrk_can_form_diag(int nb);

// Per-star normalisation coefficients: 8 doubles per star in the workspace (`coef`), written by
// norm_coef_kernel (direct form) or defer_finish_kernel (deferred form) of sp_assemble.hip, read
// by the assembly kernels, cond_system_kernel (sp_cond.hip) and the reduction (sp_reduce.h).  ONE definition:
// rounds 2-3 carried three hand-made mirrors of it with different field names for the same slots.
struct SpCoef {
  double c1;       // alpha / mu^2                       (1 when not normalised)
  double zab;      // direct: alpha + beta               deferred: d_p = z (alpha + beta) / c1
  double za;       // direct: alpha                      deferred: d_q = -z alpha / c1
  double z;        // m / mu^2
  double gpmean;   // mean of the flux GP (0 when normalised, sp.py:669-670)
  double m;        // mean(Sigma)
  double mu;       // 1 + flux mean
  double d1;       // direct: unused                     deferred: d_1 = baseline_var / c1
};
static_assert(sizeof(SpCoef) == 64, "8 doubles per star (Layout::coef, sub_layout)");

// The data plan of the likelihood step (sp_plan.hip; include/starry_process_amd.h: sp_plan_data): device pointers
struct PlanDev {
  const double *theta;     // [S][K] phases 2 pi mod(t / p, 1)
  const double *wbar;      // [S][covpts + 4] weight of every kernel-table entry in the sum of the covariance
  const double *sflux;     // [S][M] sums of the light curves over the valid cadences
  const double *sdv;       // [S] sum of the per-cadence variances over the valid cadences (0 without them)
  const double *key;       // [S][3] period, tau, nobs as planned
  const double *inorder;   // [S] 1.0: cadences in non-decreasing order over the valid ones, else 0.0
};
struct sp_plan {
  int device, S, K, M, covpts, temporal, has_diag;
  void *buf;               // one device allocation behind the pointers of `dev`
  size_t bytes;
  PlanDev dev;
  // the data the plan was made from: the caller's arrays (sp_plan_data) or the plan's own copies (sp_plan_replicate;
  // then part of `buf`).  sp_lnlike_ensemble_planned reads THESE when it is given no data pointers and refuses others.
  const double *t, *flux, *diag;
  int nrep;                // sp_plan_replicate: S = nrep x the source plan's stars (0: not a replica)
};

// What the workgroup that factors a group's LAST pivot block does behind it (sp_panel.hip): the
// log-likelihood reduction of its star (sp_reduce.h), when the residual / normalisation rows live in
// that block's row tile and the block is factored in a panel launch's tail (sp_panel_fuses_reduce).
// per-star scalars of the deferred normalisation's reduction (sp_reduce.h): {K m, sum(d), delta, sum(r_0), ...}
#define SP_RSCAL_HEAD 3
struct SpReduceArgs {
  double *lnlike;               // null: no reduction (plain factorisations)
  uint32_t *status, *status_out;
  const sp_star *stars;
  const void *coef;             // SpCoef per star (deferred normalisation) or null
  const double *rscal;          // [S][SP_RSCAL_HEAD + M] (deferred normalisation)
  int dvec;                     // per-cadence data variances: L^-1 d rides in row K + M + 1
  int K, M;
  int live_rows;                // rows of the padded system that carry data (0: all): the rest is identity padding
};

// one group of stars factored on its own stream
struct sp_chol_group {
  double *sys;
  int32_t *info;
  double *invL;      // S x sp_lt_stride(Kp) doubles
  int S;
  hipStream_t st;
  LazyCov lazy;
  SpReduceArgs red;
  int tri0 = -1;     // >= 0: the rows from this one on are an IDENTITY riding along (sp_spd_inverse_batched): row
                     // tri0 + m is zero left of column m until the factorisation reaches it, so a launch only
                     // takes the row tiles that hold something, and the columns beyond the matrix are never formed
  bool block0_done = false;   // pivot block 0 is factored already (the planned step's assembly does it, sp_planasm.hip)
};

// does the factorisation of a (K, Kp) system by this handle end in a panel launch's tail (which can carry the reduction)?
bool sp_panel_fuses_reduce(const sp_handle *h, int K, int Kp);
// panels per super-panel of a K-cadence factorisation by this handle (sp_cholesky.hip)
int sp_superpanel_width(const sp_handle *h, int K);

// host-side constant builders (sp_host.cpp)
void sp_build_index_tables(int ydeg, int32_t *l_of, int32_t *m_of,
                           int32_t *mirror, int32_t *m0, int32_t *blk);
void sp_build_flux_constants(int ydeg, int udeg, std::vector<double> &rT,
                             std::vector<double> &A1, std::vector<double> &U1,
                             std::vector<double> &rta1);
void sp_host_rTA1L(const sp_handle *h, const double *u, double *out);
void sp_host_rTA1L_rev(const sp_handle *h, const double *u, const double *bf, double *bu);

// ---- kernel launchers (one per .hip file) -----------------------------------
int sp_launch_Rx(sp_handle *h, const double *cs_dev /* [n,2] cos,sin */, int n,
                 double *R, double *dR, hipStream_t st);
int sp_launch_dotRx(sp_handle *h, const double *M, long strideM, long rs,
                    long cs, int rows, const double *R, long strideR,
                    double *out, int batch, hipStream_t st, int transposeR = 0);

// ez, Ez (and the resident copies of mu, Sigma) from device pointers, one launch
int sp_launch_polar_moments(sp_handle *h, const double *mu_src, const double *cov_src,
                            hipStream_t st);

// C[b] (+)= alpha * A[b] . B[b]^T  on the matrix cores (sp_gemm.hip): A: Mrows x Kd (lda), B: Nrows x Kd
// (ldb), C: Mrows x Nrows (ldc); beta is 0 or 1; lower_only: only tiles with tile_i >= tile_j.
int sp_launch_tri_solve(const double *L, int K, long ldl, long strideL, double *B, long strideB,
                        long rs, long cs, int nrhs, int batch, int mode, hipStream_t st);
int sp_launch_transpose(const double *in, long ldi, long stridei, double *out, int K, int batch,
                        hipStream_t st);
int sp_launch_tri_mask(double *A, int K, int batch, int upper, double dscale, hipStream_t st);
int sp_launch_chol_rev_finish(const double *S, const double *L, long ldl, long strideL, double *out,
                              int K, int batch, hipStream_t st);
int sp_launch_gemm_nt(const double *A, long lda, long strideA, const double *B,
                      long ldb, long strideB, double *C, long ldc, long strideC,
                      int Mrows, int Nrows, int Kd, double alpha, int beta,
                      int lower_only, int batch, hipStream_t st, int skip_tile00 = 0,
                      const LazyCov *lazy = nullptr);

// round-3 panel kernel (sp_panel.hip)
struct DiagFuse;
enum { SP_PANEL_D = 1, SP_PANEL_T = 2, SP_PANEL_TAILD = 4, SP_PANEL_LA = 8, SP_PANEL_FIRSTLA = 16 };
int sp_launch_panel2(int layout, const SpReduceArgs *red, double *sys, long ld, long stride, int S, int ntile, int j, int s0, int nact,
                     int next_nact, int last, int what, int ncu, double *img, long lts, int32_t *info,
                     hipStream_t st, const LazyCov *lazy);
// symmetric trailing update C -= X X^T (lower 64 x 64 tiles, tile (0, 0) skipped) whose tile-(0, 0)
// workgroup factors the pivot block described by `df` (sp_paneldiag.h)
int sp_launch_syrk_diag(const double *X, long ld, long stride, double *T, int n, int kd, int batch,
                        hipStream_t st, const LazyCov *lazy, const DiagFuse *df, int tj_limit = 0);

// reverse sweep of the marginal-branch likelihood (sp_grad.hip): C^-1 (lower tiles in, full out) -> lnL, the table's
// adjoint ybar [S, covpts + 4], the flux mean's adjoint [S]
// LDS the hot form of the assembly needs (sp_assemble.hip, assemble_sums_kernel) and the most it may ask for
#define SP_ASM_LDS_MAX (80 * 1024)
size_t sp_assemble_sums_lds(int Kp, int covpts, int temporal);
int sp_launch_grad_sweep(int S, int K, int Kr, int M, double *Cinv, const double *theta, const double *t,
                         const double *flux, const sp_star *stars, const void *coef, const double *qv,
                         const double *diag, const double *logdet, const int32_t *info, int covpts, int temporal,
                         int normalized, int order, double zmax, double *vec, double *dots, double *hcoef,
                         double *partial, double *lnlike, double *ybar, double *meanbar, uint32_t *status,
                         hipStream_t st);

// per-star scratch of the factorisation: three image slots + the chain words (sp_tile.h).  Doubles.
static inline long sp_lt_stride(int) { return 2 * 4096L; }

#endif
