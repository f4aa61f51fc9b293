/*
 * starry_process_amd -- C ABI of the MI355X (gfx950) log-likelihood hot path.
 *
 * This is the drop-in boundary.  Each entry point
 * replaces one Theano Op (or a fixed chain of them) of the reference
 * rodluger/starry_process; the reference interface it stands in for is cited
 * as  <file>:<line>  relative to the reference checkout.
 *
 * Conventions
 *   - plain C, no Python.h, no torch types; never throws, never aborts;
 *   - every function returns SP_OK (0) or a negative sp_status;
 *   - the CALLER owns every buffer.  Pointers named *_dev are device (HBM)
 *     pointers on the handle's GPU, pointers named *_host are host pointers;
 *   - all floating point is IEEE fp64, matrices are C-contiguous row-major
 *     (reference ops/include/utils.h:33-34, theano_helpers.h:55-87);
 *   - Ylm flat index n(l,m) = l*l + l + m, N = (ydeg+1)^2; packed Wigner
 *     arrays hold block l (a (2l+1)x(2l+1) row-major matrix) at offset
 *     nwig(l-1), total NWIG = nwig(ydeg) (reference ops/include/wigner.h:22-30);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     calls are asynchronous with respect to the host unless stated;
 *   - numerical failure is data, not an error: a non positive definite
 *     covariance or z > zmax yields -inf in the log-likelihood output and a
 *     bit in the per-star status word, exactly like the reference's NaN ->
 *     -inf rule (reference math.py:82-91, sp.py:1178-1188).
 */
#ifndef STARRY_PROCESS_AMD_H
#define STARRY_PROCESS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sp_handle sp_handle;

typedef enum {
  SP_OK = 0,
  SP_ERR_INVALID = -1,  /* bad argument (shape, NULL, unsupported degree)     */
  SP_ERR_HIP = -2,      /* a HIP runtime call failed (see sp_last_hip_error)  */
  SP_ERR_NO_DEVICE = -3,/* no usable gfx950 device                            */
  SP_ERR_STATE = -4,    /* required constants / moments were not set first    */
  SP_ERR_ALLOC = -5,    /* host or device allocation failed                   */
  SP_ERR_COMM = -6      /* no RCCL in the process, or the collective failed   */
} sp_status;

/* per-star status bits written by the likelihood kernels */
#define SP_STAR_NOT_PD 1u   /* Cholesky pivot <= 0 or NaN (math.py:82-91)     */
#define SP_STAR_ZMAX 2u     /* z > normalization_zmax (sp.py:1178-1183)       */
#define SP_STAR_NAN 4u      /* NaN reached the final value (sp.py:1186-1188)  */
#define SP_STAR_STALE_PLAN 8u /* sp_lnlike_ensemble_planned: the star's period, tau
                               or nobs differ from the planned ones; value = NaN   */

/* temporal kernels (reference temporal.py:8-16) */
#define SP_TEMPORAL_NONE 0
#define SP_TEMPORAL_MATERN32 1
#define SP_TEMPORAL_EXPSQUARED 2

/* ---- library / handle --------------------------------------------------- */

int sp_version(void);
const char *sp_strerror(int status);
/* text of the last failing HIP call on this thread ("" if none) */
const char *sp_last_hip_error(void);
/* number of visible HIP devices; does not initialise a device context */
int sp_device_count(void);

/* One handle per GPU.  ydeg >= 1 (the reference asserts ydeg >= 5,
 * sp.py:236), 0 <= udeg <= 4.  Computes the per-degree constants the
 * reference bakes in at compile time through -DSP__LMAX/-DSP__UMAX
 * (ops/base_op.py:81-90): index tables, Rx(pi/2), rTA1, the LimbDark
 * constructor (ops/include/flux.h:483-494). */
int sp_create(int ydeg, int udeg, int device, sp_handle **out);
void sp_destroy(sp_handle *h);
int sp_ydeg(const sp_handle *h);
int sp_udeg(const sp_handle *h);
int sp_nylm(const sp_handle *h);  /* N    */
int sp_nwig(const sp_handle *h);  /* NWIG */
/* blocks until everything queued on `stream` has finished */
int sp_stream_synchronize(sp_handle *h, void *stream);

/* ---- a1: integer layout tables (host outputs, no GPU needed) ------------- */
/* l_of,m_of,mirror: N entries; m0: ydeg+1; blk: ydeg+2 (blk[l] = nwig(l-1)).
 * Replaces the index arithmetic of wigner.h:22-30,319-336 and flux.py:77,200. */
int sp_index_tables(int ydeg, int32_t *l_of, int32_t *m_of, int32_t *mirror,
                    int32_t *m0, int32_t *blk);
/* the integer cos/sin(k*pi/2) factors of the x-rotation, ydeg+1 entries each,
 * entry 0 unused (wigner.h:232-270) */
int sp_wigner_int_tables(int ydeg, int32_t *cosmal, int32_t *sinmal,
                         int32_t *sgn, int32_t *cosmga, int32_t *sinmga);

/* ---- a2: RxOp (ops/wigner/Rx.py:8-43, Rx.cc:10-49, wigner.h:145-284) ------ */
/* nangles rotation angles (host, radians) -> packed R [nangles, NWIG] and, if
 * dR_dev != NULL, dR/dtheta of the same shape.  One workgroup per angle. */
int sp_Rx(sp_handle *h, const double *theta_host, int nangles, double *R_dev,
          double *dR_dev, void *stream);

/* ---- a3: FluxIntegral._dotRx (flux.py:74-86) ------------------------------ */
/* out[b] = M[b] . blockdiag(R^l[b]) for b < batch.  M[b] is rows x N with row
 * stride ldm and element (r, c) at M + b*strideM + r*rs + c*cs (rs/cs let the
 * caller pass a transposed view, flux.py:61); Rpacked[b] at R + b*strideR
 * (strideR = 0 shares one rotation).  out is rows x N contiguous per batch. */
int sp_dotRx(sp_handle *h, const double *M_dev, long strideM, long rs, long cs,
             int rows, const double *Rpacked_dev, long strideR, double *out_dev,
             int batch, void *stream);

/* ---- a12: tensordotRzOp (ops/wigner/tensordotRz.py:9-37, wigner.h:289-339) */
/* f[k, :] = M[k, :] . Rz(theta[k]); M, f are K x N, theta has K entries. */
int sp_tensordotRz(sp_handle *h, const double *M_dev, const double *theta_dev,
                   int K, double *f_dev, void *stream);

/* ---- a9: special_tensordotRzOp (ops/wigner/special_tensordotRz.py:9-37,
 *          wigner.h:409-459) ------------------------------------------------ */
/* f[k] = sum_j [cosmt.(T o M) + sinmt.(T o mirror(M))]_{kj};  T, M are N x N. */
int sp_special_tensordotRz(sp_handle *h, const double *T_dev,
                           const double *M_dev, const double *theta_dev, int K,
                           double *f_dev, void *stream);

/* ---- a5: rTA1Op / rTA1LOp (ops/flux/rTA1.py:8-21, rTA1L.py:8-43,
 *          flux.h:302-309, 500-523).  Host in, host out: per-star scalars. --- */
int sp_rTA1(sp_handle *h, double *rta1_host /* N */);
int sp_rTA1L(sp_handle *h, const double *u_host /* nsets x udeg */, int nsets,
             double *rta1l_host /* nsets x N */);

/* ---- a15: AlphaBetaOp (ops/norm/norm.py:8-44).  Host scalar function. ----- */
int sp_alpha_beta(double z, int order, double *alpha, double *beta,
                  double *dalpha_dz, double *dbeta_dz);

/* ---- a6: constants of FluxIntegral._precompute (flux.py:121-179) ---------- */
/* The reference computes these in Python at graph-build time; the host side
 * (starry_process_amd/flux.py) does the same and hands them over once.
 * wnp_packed: NWIG doubles (block l = wnp[l]); Wnp: N x N. */
int sp_set_marginal_constants(sp_handle *h, const double *wnp_packed_host,
                              const double *Wnp_host);

/* ---- a4: moments of the Ylm process in the polar frame (flux.py:54-62) ---- */
/* mean_ylm (N) and cov_ylm (N x N) are HOST arrays (the output of the
 * out-of-scope upstream integrals, sp.py:264-266).  Uploads them and computes
 * ez = R^T mu, Ez = R^T (Sigma + mu mu^T) R on the device; both stay resident
 * in the handle until the next call.  Synchronous. */
int sp_set_ylm_moments(sp_handle *h, const double *mean_ylm_host,
                       const double *cov_ylm_host);
/* Same, with mean_ylm / cov_ylm already resident in HBM; asynchronous on
 * `stream` (what an MCMC loop with a device-side upstream would call). */
int sp_set_ylm_moments_dev(sp_handle *h, const double *mean_ylm_dev,
                           const double *cov_ylm_dev, void *stream);
/* copies of the resident moments back to the host (any pointer may be NULL) */
int sp_get_polar_moments(sp_handle *h, double *ez_host, double *Ez_host);

/* ---- a7-a10: inclination integrals + kernel table ------------------------- */
/* For each of ntab flux operators rta1[i] (N doubles each, DEVICE):
 *   w, W (flux.py:181-231), mean & var (flux.py:297-308), second moment on the
 *   lag grid xp = arange(-dx, 2pi+2.5dx, dx), dx = 2pi/covpts (flux.py:310-317),
 *   yp = mom2 - mean^2 and the cubic coefficients a0..a3 (flux.py:320-330).
 * xp_host: the covpts+4 grid values (computed by the host exactly like the
 * reference does with arange so that interpolation indices agree bit for bit).
 * tab_dev  : [ntab, 5, covpts+4]   rows = yp, a0, a1, a2, a3 (a* use the first
 *            covpts+1 entries; the rest is zero)
 * meanvar_dev : [ntab, 2] = (mean, var).                                     */
int sp_kernel_table(sp_handle *h, const double *rta1_dev, int ntab, int covpts,
                    const double *xp_host, double *tab_dev,
                    double *meanvar_dev, void *stream);

/* ---- hyperparameter samples in batches (round 6) ----------------------------
 * A sampler evaluates ONE data set at many hyperparameter vectors -- one light curve per log_likelihood call
 * (sp.py:1052-1062), called 10^4-10^5 times (calibrate/sample.py:95-107, interfaces.py:142-166).  These two entry
 * points take B samples per call, so that a single light curve fills the GPU like a 64-star ensemble does:
 *   sp_polar_moments_samples   (r, alpha, beta, c, n)[B] -> ez [B][N], Ez [B][N][N]: the polar-frame moments of
 *       flux.py:54-62 (row a4) of the Ylm process of sp.py:257-266 with ONE spot radius (dr = None) -- the chain
 *       size.py:49-101 -> latitude.py:170-212 -> longitude.py:8-78 -> contrast.py:18-33 by exact quadrature of
 *       rotations like sp_ylm_moments_quadrature, carried out in the polar frame where the longitude average is a
 *       projection (csrc/sp_samples.hip): 2 (ydeg + 2) rotations per sample, one upload of 5 B numbers, six launches,
 *       no host arithmetic (the Gauss-Jacobi rule is found on the device).  Equal to sp_ylm_moments_quadrature +
 *       sp_set_ylm_moments_dev to rounding; B samples in one call give the bits of B calls with one sample each.
 *       samples_host [B][5]: r in RADIANS, alpha, beta (the Beta law's shape parameters, latitude.py:176-197), c, n.
 *       Needs sp_set_size_basis first (the spot profile's basis, size.py:9-47: theta [spts] = the colatitude grid,
 *       Bp [ydeg + 1][spts] = the smoothed pseudo-inverse of the Legendre basis, sfac = the sigmoid's steepness).
 *   sp_kernel_table_samples    sp_kernel_table for B sets of polar moments: table b ntab + i (tab_dev
 *       [B ntab][5][covpts + 4], meanvar_dev [B ntab][2]) from sample b and flux operator i.
 * The likelihood of the (sample, star) pairs is then ONE sp_lnlike_ensemble_planned call on a replicated plan
 * (sp_plan_replicate below) whose stars carry table = b ntab + table_s.                                       */
int sp_set_size_basis(sp_handle *h, const double *theta_host, const double *Bp_host, int spts, double sfac);
int sp_polar_moments_samples(sp_handle *h, int B, const double *samples_host, double epsy, double epsy15,
                             double *ez_dev, double *Ez_dev, void *stream);
int sp_kernel_table_samples(sp_handle *h, int B, const double *ez_dev, const double *Ez_dev, const double *rta1_dev,
                            int ntab, int covpts, const double *xp_host, double *tab_dev, double *meanvar_dev,
                            void *stream);

/* ---- per-star parameter block -------------------------------------------- */
/* All batched entry points below take `S` stars with a common row length K.  A
 * RAGGED ensemble (light curves of different lengths) is padded to the longest:
 * star s uses its first `nobs` cadences (0 = all K), the rest of its row of
 * t / flux / diag is ignored; its covariance is nobs x nobs, the padding rows of
 * the factored system are identity rows.  The array of sp_star lives in DEVICE
 * memory like every other batched input.                                       */
typedef struct {
  double period;        /* p  > 0                         (sp.py:1108-1109)   */
  double inc;           /* inclination in RADIANS, conditional path only       */
  double tau;           /* temporal timescale; ignored if temporal == NONE     */
  double baseline_var;  /* added to every entry           (sp.py:1146-1151)   */
  double baseline_mean; /* subtracted from the flux       (sp.py:1157)        */
  double data_var;      /* scalar data variance (used when diag_dev == NULL)   */
  int32_t table;        /* which kernel table / flux operator this star uses   */
  int32_t nobs;         /* valid cadences of this star, 0 = K (ragged ensembles) */
} sp_star;

/* ---- a11, a14-a16: marginal-path covariance ------------------------------- */
/* cov[s] (K x K, leading dimension ldc >= K, star stride stridec) =
 *   spline(|theta_i - theta_j|)                (flux.py:256-276)
 *   [* temporal kernel]                        (sp.py:697-698)
 *   [-> _normalize(1 + mean, .)]               (sp.py:699-727)
 * t_dev: [S, K]; tab/meanvar: from sp_kernel_table; z_dev (S, may be NULL)
 * receives the normalisation expansion parameter z.  K == 1 returns the
 * variance (flux.py:274-275).  Full matrices are written (both triangles).   */
int sp_cov_marginal_batched(sp_handle *h, int S, int K, const double *t_dev,
                            const sp_star *stars_dev, int covpts,
                            const double *tab_dev, const double *meanvar_dev,
                            int temporal, int normalized, int norm_order,
                            double *cov_dev, long ldc, long stridec,
                            double *z_dev, void *stream);

/* ---- a12-a13: conditional path -------------------------------------------- */
/* A[s] = ((1_K x rTA1[table_s]) . Rx(-inc_s)) Rz(theta_s) Rx(pi/2)
 * (flux.py:278-281, 88-105).  A_dev: [S, K, N]. */
int sp_design_matrix(sp_handle *h, int S, int K, const double *t_dev,
                     const sp_star *stars_dev, const double *rta1_dev,
                     double *A_dev, void *stream);
/* mean[s] = (A mu_y)[0], cov[s] = A Sigma_y A^T (flux.py:337-343), then the
 * same temporal / normalisation steps as the marginal path.                  */
int sp_cov_conditional_batched(sp_handle *h, int S, int K, const double *t_dev,
                               const sp_star *stars_dev,
                               const double *rta1_dev, int temporal,
                               int normalized, int norm_order, double *cov_dev,
                               long ldc, long stridec, double *mean_dev,
                               double *z_dev, void *stream);

/* ---- a17-a18: cho_factor / cho_solve (math.py:75-100) --------------------- */
/* In-place lower Cholesky of `batch` K x K matrices (row-major, leading
 * dimension lda, stride strideA).  The strict upper triangle is zeroed like
 * scipy.linalg.cholesky(lower=True).  A non positive definite matrix is
 * filled with NaN (math.py:88-91) and info_dev[b] (may be NULL) set to 1.    */
int sp_cho_factor(sp_handle *h, double *A_dev, int K, long lda, long strideA,
                  int batch, int32_t *info_dev, void *stream);
/* x = (L L^T)^{-1} b: b_dev is [batch, K, nrhs] row-major (like the
 * reference's (K, M) right-hand sides), overwritten with the solution.
 * NaN in -> NaN out (math.py:27-31).                                         */
int sp_cho_solve(sp_handle *h, const double *L_dev, int K, long ldl,
                 long strideL, double *b_dev, int nrhs, int batch,
                 void *stream);

/* One triangular sweep: trans = 0 solves L x = b (the reference's
 * Solve(A_structure="lower_triangular")(L, b), math.py:98), trans = 1 solves
 * L^T x = b (Solve("upper_triangular")(L.T, .), math.py:99).  Only the lower
 * triangle of L is read.  b_dev as in sp_cho_solve, overwritten.             */
int sp_tri_solve(sp_handle *h, const double *L_dev, int K, long ldl,
                 long strideL, double *b_dev, int nrhs, int batch, int trans,
                 void *stream);
/* Reverse mode of one triangular solve c = A^-1 b (Solve.L_op, math.py:40-72),
 * A = L (trans = 0) or A = L^T (trans = 1):
 *   b_bar = A^-T c_bar,   A_bar = -tril(b_bar c^T)  (triu for trans = 1).
 * c_dev, cbar_dev, bbar_dev: [batch, K, nrhs]; Abar_dev: [batch, K, K].        */
int sp_solve_rev(sp_handle *h, const double *L_dev, int K, long ldl,
                 long strideL, const double *c_dev, const double *cbar_dev,
                 int nrhs, int batch, int trans, double *Abar_dev,
                 double *bbar_dev, void *stream);
/* Reverse mode of the lower Cholesky factorisation (L_op of the Theano/Aesara
 * slinalg.Cholesky the reference's math.py:75 subclasses; Murray 2016):
 *   Phi = tril(L^T L_bar), diagonal halved;  S = L^-T Phi L^-1;
 *   C_bar = tril(S + S^T) - diag(S).
 * L_dev must carry zeros above the diagonal (as sp_cho_factor leaves it);
 * Lbar_dev, Cbar_dev: [batch, K, K] contiguous.  A NaN factor (not positive
 * definite, on_error="nan") gives an all-NaN C_bar.                           */
int sp_cholesky_rev(sp_handle *h, const double *L_dev, int K, long ldl,
                    long strideL, const double *Lbar_dev, int batch,
                    double *Cbar_dev, void *stream);

/* ---- a16-a19 + fused driver: log-likelihood of an ensemble ---------------- */
/* Size in bytes of the device workspace sp_lnlike_ensemble needs.            */
long sp_lnlike_workspace_bytes(sp_handle *h, int S, int K, int M);

/* lnlike[s] for S independent stars, each with M light curves that share the
 * star's covariance (sp.py:1087-1099):
 *   C_s = cov_s + diag(data var) + baseline_var                (sp.py:1135-1151)
 *   lnlike_s = -1/2 sum_m r^T C^-1 r - M sum log diag L - KM/2 log 2pi
 *                                                             (sp.py:1154-1173)
 *   z > zmax -> -inf (normalized only), NaN -> -inf            (sp.py:1178-1188)
 * conditional == 0: marginal path (uses tab/meanvar from sp_kernel_table);
 * conditional != 0: conditional path (uses rta1_dev, stars[s].inc).
 * t_dev [S,K]; flux_dev [S,M,K]; diag_dev [S,K] per-cadence data variances or
 * NULL (then stars[s].data_var is used).  lnlike_dev [S]; status_dev [S] (may
 * be NULL).  workspace_dev must hold sp_lnlike_workspace_bytes(S,K,M) bytes.
 * Ragged ensembles: stars[s].nobs (see sp_star).  Limits: S and K are bounded by
 * the workspace only (checked against the oracle up to K = 20,000,
 * tools/big_k.py).  All launches go to `stream`; nothing is synchronised.       */
int sp_lnlike_ensemble(sp_handle *h, int S, int K, int M, const double *t_dev,
                       const double *flux_dev, const double *diag_dev,
                       const sp_star *stars_dev, int conditional, int covpts,
                       const double *tab_dev, const double *meanvar_dev,
                       const double *rta1_dev, int temporal, int normalized,
                       int norm_order, double zmax, void *workspace_dev,
                       double *lnlike_dev, uint32_t *status_dev, void *stream);

/* ---- the planned step (round 5): what depends on the DATA alone, taken out of the per-sample call -------------
 * A sampler evaluates the likelihood of ONE data set (t, flux, data variances, periods) at many hyperparameter
 * samples -- the reference fixes the data when the log-probability is built, calibrate/log_prob.py:7-55.  The
 * normalisation (sp.py:705-727) needs m = mean(Sigma) before the factorisation, and rounds 2-4 took it -- with the
 * row sums of Sigma -- from a pass over all K^2 entries per evaluation.  But the spline is linear in the kernel
 * table, cov_ij = sum_k yp[s_ij + k] b_k(x0_ij) (flux.py:256-276, 322-330), so
 *     m = sum_n yp[n] wbar[n] / K^2,   wbar[n] = sum_ij [s_ij + k = n] b_k(x0_ij) T_ij
 * with wbar a function of (t, period, covpts[, tau]) only, and the row sums are not needed at all (DESIGN.md 4.7:
 * Sigma 1 = B 1 - d).  sp_plan_data computes, once per data set: the cadences' phases theta = 2 pi mod(t / p, 1),
 * wbar per star, the sums of each light curve's flux and of the per-cadence variances.  sp_lnlike_ensemble_planned
 * is then sp_lnlike_ensemble(conditional = 0, normalized = 1) with the same results to rounding: its assembly
 * evaluates only the tiles the factorisation wants in memory (every other tile is formed once, at first touch).
 * The plan owns its device memory (about (K + covpts + M + 8) doubles per star); it is read-only afterwards and may
 * be shared by any number of handles / streams of the same GPU.
 *   t_dev [S,K], flux_dev [S,M,K], diag_dev [S,K] or NULL, stars_dev [S]: as for sp_lnlike_ensemble; what the plan
 *   fixes of a star is its period, nobs and (temporal != NONE) tau -- table, baseline_mean, baseline_var, data_var
 *   may change from call to call.  A call whose stars differ in a planned field returns NaN for that star and sets
 *   SP_STAR_STALE_PLAN.  workspace_dev: sp_lnlike_workspace_bytes(S, K, M) bytes, used as scratch.
 *   Synchronises `stream` (one-off: ~0.2 ms + the allocation).                                             */
typedef struct sp_plan sp_plan;
int sp_plan_data(sp_handle *h, int S, int K, int M, const double *t_dev, const double *flux_dev,
                 const double *diag_dev, const sp_star *stars_dev, int covpts, int temporal,
                 void *workspace_dev, void *stream, sp_plan **out);
void sp_plan_destroy(sp_plan *plan);
/* wbar of the plan, [S, covpts + 4] (host; for tests) */
int sp_plan_get_wbar(const sp_plan *plan, double *wbar_host);
/* B copies of a planned data set as ONE batch of B S systems (round 6): system b S + s is star s of `src`, to be
 * evaluated under hyperparameter sample b (its sp_star.table = b ntab + table_s).  The replica owns copies of
 * everything the step reads by system index -- the plan's arrays and the data (t, flux, variances): B (K (M + 2) +
 * covpts + M + 9) doubles, 1.6 MB for 64 samples of one K = 1000 light curve.  Call the planned step on it WITHOUT
 * data pointers.  Synchronises `stream`.                                                                  */
int sp_plan_replicate(sp_handle *h, const sp_plan *src, int B, void *stream, sp_plan **out);
/* systems of a plan (S of sp_plan_data; B S of a replica) */
int sp_plan_systems(const sp_plan *plan);
/* The per-sample call on planned data: marginal branch, normalised (sp.py:1129-1188 with sp.py:705-727).
 * t_dev / flux_dev / diag_dev: all NULL = the planned arrays (required for a replica), otherwise they must BE the
 * pointers given to sp_plan_data (SP_ERR_INVALID if not: phases, weights and sums come from plan time, residual rows
 * and variances are read again from these -- other arrays of the same shape would give a finite, wrong value; edits
 * IN PLACE remain the caller's responsibility).  Shapes, covpts and the temporal kernel come from the plan.  */
int sp_lnlike_ensemble_planned(sp_handle *h, const sp_plan *plan, const double *t_dev, const double *flux_dev,
                               const double *diag_dev, const sp_star *stars_dev, const double *tab_dev,
                               const double *meanvar_dev, int norm_order, double zmax, void *workspace_dev,
                               double *lnlike_dev, uint32_t *status_dev, void *stream);

/* The factorisation stage alone: C_dev holds S assembled (K+M padded) systems
 * as produced internally; exposed for testing and for callers that assemble
 * their own covariance.  cov_dev: [S, K, K] (ld = K) full symmetric matrices
 * ALREADY including noise and baseline terms; resid_dev: [S, M, K] residuals.
 */
int sp_cholesky_lnlike_batched(sp_handle *h, int S, int K, int M,
                               const double *cov_dev, const double *resid_dev,
                               void *workspace_dev, double *lnlike_dev,
                               uint32_t *status_dev, void *stream);

/* ---- batched inverse and log-determinant of SPD matrices (round 4; what the reverse sweep of the
 * likelihood needs: d lnL / dC = (alpha alpha^T - C^-1) / 2, math.py:40-72 composed) ------------------------
 * C_dev [S] K x K (leading dimension ldc, stride strideC; symmetric positive definite, the lower triangle is
 * read) -> Cinv_dev [S, Kr, Kr], Kr = roundup(K, 64): the LOWER 64 x 64 tiles of C^-1 (rows / columns >= K zero;
 * the tiles above the diagonal are not written), logdet_dev [S] = log det C (NaN: not positive definite),
 * info_dev [S] (may be NULL).  The identity rides through the blocked factorisation as rows below the matrix
 *  and comes out as L^-T; C^-1 = L^-T L^-1 is one product on the matrix cores.
 * workspace_dev: sp_spd_inverse_workspace_bytes(h, S, K) bytes.                                            */
size_t sp_spd_inverse_workspace_bytes(sp_handle *h, int S, int K);
int sp_spd_inverse_batched(sp_handle *h, int S, int K, const double *C_dev, long ldc, long strideC,
                           double *Cinv_dev, double *logdet_dev, int32_t *info_dev, void *workspace_dev,
                           void *stream);

/* ---- the ensemble gradient's device half (round 4; tests/test_lnlike.py:100-136 for a whole batch) ------------
 * Marginal branch, one light curve per star (M = 1), every cadence valid (sp_star.nobs = 0 or K: a ragged star gets
 * NaN and SP_STAR_NAN, never a silently wrong value).  For every star:
 *   lnlike_dev [S]            the log-likelihood (sp.py:1129-1188; -inf as sp_lnlike_ensemble)
 *   ybar_dev [S, covpts + 4]  d lnL_s / d yp, yp = the star's kernel table (tab_dev[table_s][0, :], the second
 *                             moment on the lag grid minus mean^2, flux.py:310-320), everything else held fixed
 *   meanbar_dev [S]           d lnL_s / d (flux mean, meanvar_dev[2 table_s]) at fixed table
 * by one reverse sweep: C assembled, C^-1 by sp_spd_inverse_batched, d lnL / dC = (alpha alpha^T - C^-1) / 2
 * pulled back through the normalisation (sp.py:705-727) and the cubic interpolation (flux.py:256-276; the spline
 * is linear in yp).  The caller chains (ybar, meanbar) to the hyperparameters: d lnL / d theta = sum_s ybar_s .
 * d yp / d theta + meanbar_s d mean / d theta (starry_process_amd/grad.py: ensemble_gradient).
 * workspace_dev: sp_lnlike_grad_workspace_bytes(h, S, K, covpts) bytes -- about S (2 K)^2 doubles for the system that
 * carries the identity plus S K^2 for the inverse: 1.9 GB at cfg3's shape (64 x 1000), 8.5 GB at cfg5's (32 x 3000).  */
size_t sp_lnlike_grad_workspace_bytes(sp_handle *h, int S, int K, int covpts);
int sp_lnlike_grad_marginal(sp_handle *h, int S, int K, const double *t_dev, const double *flux_dev,
                            const double *diag_dev, const sp_star *stars_dev, int covpts, const double *tab_dev,
                            const double *meanvar_dev, int temporal, int normalized, int norm_order, double zmax,
                            void *workspace_dev, double *lnlike_dev, double *ybar_dev, double *meanbar_dev,
                            uint32_t *status_dev, void *stream);
/* The same for M light curves per star on ONE covariance (flux_dev [S, M, K]; the shared-covariance multi-RHS form of
 * sp.py:1162-1171, what calibrate.get_log_prob evaluates: calibrate/log_prob.py:7-106):
 *   lnL_s = sum_m -1/2 r_m^T C^-1 r_m - M/2 log det C - M K/2 log 2 pi,   d lnL / dC = (sum_m alpha_m alpha_m^T - M C^-1) / 2.
 * One factorisation and one inverse per star whatever M; workspace: sp_lnlike_grad_workspace_bytes_multi.        */
size_t sp_lnlike_grad_workspace_bytes_multi(sp_handle *h, int S, int K, int M, int covpts);
int sp_lnlike_grad_marginal_multi(sp_handle *h, int S, int K, int M, const double *t_dev, const double *flux_dev,
                                  const double *diag_dev, const sp_star *stars_dev, int covpts, const double *tab_dev,
                                  const double *meanvar_dev, int temporal, int normalized, int norm_order, double zmax,
                                  void *workspace_dev, double *lnlike_dev, double *ybar_dev, double *meanbar_dev,
                                  uint32_t *status_dev, void *stream);

/* ---- fp64 NT product on the matrix cores (the kernel behind a13 / a17, exposed) -------
 *   C[b] = beta * C[b] + alpha * A[b] . B[b]^T,   beta in {0, 1}
 * A: M x K (lda), B: N x K (ldb), C: M x N (ldc), row-major, `batch` matrices strideA /
 * strideB / strideC doubles apart; lower_only != 0 (M == N): only the 64 x 64 tiles on
 * or below the diagonal are touched.                                                 */
int sp_gemm_nt(sp_handle *h, const double *A_dev, long lda, long strideA, const double *B_dev,
               long ldb, long strideB, double *C_dev, long ldc, long strideC, int M, int N,
               int K, double alpha, int beta, int lower_only, int batch, void *stream);

/* ---- reverse-mode ops ------------------------------------------
 * The native gradient kernels of the reference, one entry point each:
 *   tensordotRzRevOp          ops/wigner/tensordotRz_rev.py:9-29, wigner.h:344-404
 *       bf [K, N]  ->  bM [K, N], btheta [K]
 *   special_tensordotRzRevOp  ops/wigner/special_tensordotRz_rev.py, wigner.h:464-531
 *       bf [K]     ->  bM [N, N] (gradient w.r.t. M; the op returns zeros for T,
 *                      special_tensordotRz.py:30), btheta [K]
 *   rTA1LRevOp                ops/flux/rTA1L.py:31-48, flux.h:529-557  (host -> host)
 *       bf [N]     ->  bu [udeg]                                                    */
int sp_tensordotRz_rev(sp_handle *h, const double *M_dev, const double *theta_dev, int K,
                       const double *bf_dev, double *bM_dev, double *btheta_dev, void *stream);
int sp_special_tensordotRz_rev(sp_handle *h, const double *T_dev, const double *M_dev,
                               const double *theta_dev, int K, const double *bf_dev,
                               double *bM_dev, double *btheta_dev, void *stream);
int sp_rTA1L_rev(sp_handle *h, const double *u_host, const double *bf_host, double *bu_host);

/* ---- Gaussian conditioning (sp.py:767-903 `predict`, 905-1002
 * `sample_conditional`) -----------------------------------------------------------
 * Ktt_dev [K, K]: covariance at the observed times INCLUDING data covariance and
 * baseline variance (full symmetric, not modified); Kst_dev [Ks, K]: cross covariance
 * (sample times x observed times); Kss_dev [Ks, Ks]: prior covariance at the sample
 * times, overwritten with the posterior  K_ss - K_st K_tt^-1 K_st^T;  r_dev [K]:
 * observed flux minus mean;  mu_dev [Ks] receives  K_st K_tt^-1 r  (add the process
 * mean on the host).  One factorisation of K_tt with K_st and r riding along as extra
 * rows (Y = K_st L^-T, w = L^-1 r), then mu = Y w and K_ss -= Y Y^T on
 * the matrix cores.  info_dev[0] != 0: K_tt was not positive definite (outputs are
 * then meaningless; math.py:82-91 returns NaN in that case).                       */
int sp_gp_condition(sp_handle *h, int K, int Ks, const double *Ktt_dev, const double *Kst_dev,
                    double *Kss_dev, const double *r_dev, double *mu_dev, int32_t *info_dev,
                    void *stream);

/* ---- upstream of the hot path, host only ---------------- */
/* LatitudeIntegralOp values (ops/latitude/latitude.py, ops/include/latitude.h:
 * 21-173): q [N], Q [N x N] for Beta shape parameters alpha, beta.  The
 * remaining upstream integrals are NumPy in the reference and in
 * starry_process_amd/upstream.py.                                             */
int sp_latitude_integrals(int ydeg, double alpha, double beta, double *q_host,
                          double *Q_host);
/* n-point Gauss-Jacobi rule for the weight (1 - t)^a (1 + t)^b on (-1, 1), a, b > -1; weights
 * normalised to sum 1.  The latitude expectation of the device upstream: cos(phi) ~ Beta(alpha,
 * beta) over the WHOLE prior box of latitude.py:176-197 (alpha <= exp(5), beta <= exp(10)) --
 * Golub-Welsch on the Jacobi matrix, so the zeroth moment that overflows in the textbook
 * normalisation never appears.  Host only, no handle.                                       */
int sp_gauss_jacobi(int n, double a, double b, double *nodes_host, double *weights_host);
/* The same rule with the derivatives of its nodes and weights with respect to a and b: the rule is exact for
 * the polynomials it integrates whatever (a, b), so sum_k dw_k F(t_k) + w_k F'(t_k) dt_k is the exact derivative
 * of the expectation -- the quadrature's counterpart of the analytic d/d alpha, d/d beta of the reference's
 * latitude integrals (ops/include/latitude.h:21-173, tests/test_latitude.py:90-129).  Host only, no handle. */
int sp_gauss_jacobi_grad(int n, double a, double b, double *nodes_host, double *weights_host,
                         double *dnodes_da_host, double *dweights_da_host, double *dnodes_db_host,
                         double *dweights_db_host);

/* ---- upstream of the path, on the device ---------------------------
 * (mu_y [N], Sigma_y [N, N]) on the device from the host-side pieces of the hyperparameters, by
 * exact quadrature of rotations (starry_process_amd/upstream_device.py documents the method; it
 * replaces the chain size.py:49-134 -> latitude.py:170-212 -> longitude.py:8-78 ->
 * contrast.py:18-33 of the reference, whose eigen-square-root route is ill conditioned):
 *   vecs_host [mv, N]  the vectors to rotate: the size first moment, then (unless first_is_col:
 *                      dr = None, one vector serves as both) the columns of the second-moment factor;
 *   phi_host, w_host [P]  latitude angles (both signs of the Gauss-Jacobi nodes) and their weights
 *                      (sum 1); Q equispaced longitudes lam_q = 2 pi q / Q;
 *   g = pi c sqrt(n), sqrt_n, epsy, epsy15 (contrast.py:21-33).
 * One staged upload and nine launches on `stream`; scratch from the handle.                     */
int sp_ylm_moments_quadrature(sp_handle *h, const double *vecs_host, int mv, int first_is_col,
                              const double *phi_host, const double *w_host, int P, int Q, double g,
                              double sqrt_n, double epsy, double epsy15, double *mean_dev,
                              double *cov_dev, void *stream);
/* The same moments with their EXACT derivatives with respect to the spot radius and the Beta shape parameters
 * (one radius: dr = None) -- what the reference differentiates analytically (ops/include/latitude.h:21-173 returns
 * d/d alpha, d/d beta; size.py:92-101 through the graph): the tangents ride through the same rotations as three more
 * rows per latitude, three cross products beside the value's (csrc/sp_upstream.hip).
 *   s_host, ds_dr_host [N]       the size vector and its derivative with respect to r;
 *   phi_host, w_host [P]         as above, the second half the mirror image of the first (-phi_k, the same w_k);
 *   dphi_host, dw_host [2][P]    derivatives of the angles and weights with respect to alpha and beta
 *                                (sp_gauss_jacobi_grad through x = cos(phi));
 *   dmean_dev [3][N], dcov_dev [3][N][N]   d/dr, d/dalpha, d/dbeta of (mu_y, Sigma_y).                         */
int sp_ylm_moments_quadrature_grad(sp_handle *h, const double *s_host, const double *ds_dr_host,
                                   const double *phi_host, const double *w_host, const double *dphi_host,
                                   const double *dw_host, int P, int Q, double g, double sqrt_n, double epsy,
                                   double epsy15, double *mean_dev, double *cov_dev, double *dmean_dev,
                                   double *dcov_dev, void *stream);

/* ---- measurement hooks (bench.py) ------------------------------------------ */
/* Between begin and end every launch of the trailing-update kernel (the
 * dominant kernel of the factorisation) is bracketed by HIP events on the
 * stream it is launched on.  end() waits for them and returns the number of
 * launches, their summed duration and their summed ALGORITHMIC flops
 * (n (n + 1) / 2 x 64 multiply-adds x 2 per star and launch).                */
int sp_profile_begin(sp_handle *h, int max_launches);
int sp_profile_end(sp_handle *h, long *launches, double *total_ms, double *flops);
/* The same for one KIND of launch of the factorisation (every launch is bracketed while
 * profiling is on): 0 symmetric trailing updates (what sp_profile_end reports), 2 every panel
 * launch under its own pair of events (5: the same, kept apart so that both may be armed),
 * 4 the panel launches of a whole super-panel under ONE pair (cheap enough for a timed region).
 * Kinds 1 and 3 are not produced any more.  Scopes nest.  Stops the profile like sp_profile_end;
 * may be called for several kinds in a row.                                                  */
int sp_profile_kind(sp_handle *h, int kind, long *launches, double *total_ms, double *flops);
/* The same with both flop counts (round 6): `flops` is the ALGORITHMIC count -- the K cadences' rows and the M residual
 * rows, the last pivot block as wide as it is --, `flops_padded` the count on the padded system the launches execute
 * (rows up to roundup(K + M + 2, 64), every block 64 wide: 7 % more at K = 1000), which rounds 1-5 reported as the
 * former.  sp_profile_kind / sp_profile_end return the algorithmic count.                                      */
int sp_profile_kind_ex(sp_handle *h, int kind, long *launches, double *total_ms, double *flops, double *flops_padded);
/* sp_profile_begin for a chosen set of kinds (bit k of kind_mask = kind k; sp_profile_begin
 * = kind 0 only).  An event pair costs a few microseconds of stream time: bracket every panel
 * launch (17 per K = 1000 factorisation) only outside timed regions.                      */
int sp_profile_begin_kinds(sp_handle *h, int max_launches, unsigned kind_mask);

/* Normalised likelihoods (sp.py:705-727: C = c1 Sigma + z ((alpha + beta) p p^T - alpha q q^T),
 * q = row sums / (K m)), per handle:
 *   1 (default): deferred -- the assembly writes the raw covariance ONCE and takes its sum in
 *                the same pass; the factorisation is that of Sigma + N / c1 and the rank-2 (and
 *                baseline) part is applied to the result by the matrix-determinant / Sherman-
 *                Morrison identities from one extra row of the system (L^-1 1; a second one, L^-1 d,
 *                with per-cadence variances) and sums of the data.  A workspace sized for this mode
 *                also serves the other one.
 *   0: the separate row-sum pass, then the normalised matrix assembled and factored as such.
 * The two agree to rounding (1e-12 relative on the BASELINE configurations); -inf for a matrix
 * that is not positive definite or z > zmax either way.  Set before sizing the workspace
 * (sp_lnlike_workspace_bytes).  Environment: SP_DEFER_NORM.                                 */
int sp_set_defer_norm(sp_handle *h, int on);

/* Covariance tiles formed at first touch (default on; environment SP_LAZY_COV).  Under the deferred
 * normalisation a tile of the system below the diagonal is a pure function of the cadences'
 * phases (flux.py:256-276) until the factorisation first touches it: with this switch on and no
 * temporal kernel, the assembly takes those tiles' row / column sums but does not write them, and the kernel that touches a tile first evaluates
 * it instead of loading it -- the same code, the same bits, a write and a read of 3/4 of the
 * matrix less.  Diagonal tiles and the rows holding residuals are always written.            */
int sp_set_lazy_cov(sp_handle *h, int on);

/* ---- multi-GPU ------------------------------------------------------
 * The only exchange of the path: every rank contributes the log-likelihoods of
 * its `count` stars and receives all `count * nranks` of them, in rank order
 * (ncclAllGather over RCCL / xGMI, fp64, on `stream`).  `nccl_comm` is the
 * caller's ncclComm_t (one process per GPU); the RCCL that created it is looked
 * up in the running process (dlsym), so the library itself does not link RCCL
 * and single-GPU users never load it.  Replaces nothing in the reference, which
 * has no multi-GPU path (joss/paper.md:160-172 describes the ensemble use case);
 * the Python host side (ensemble.py) does the same through torch.distributed.    */
int sp_allgather_lnlike(sp_handle *h, void *nccl_comm, const double *local_dev, int count,
                        double *all_dev, void *stream);

/* (debug) the look-ahead items of the panel launches (csrc/sp_cholesky.hip) on (default; environment
 * SP_PANEL_LA) or off: the factor is the same to rounding, the critical path of a panel is not.  */
int sp_debug_set_look_ahead(sp_handle *h, int on);
/* (debug) the panel launches' layout by CU (chain items first, sleepers on their CUs' other slots; default on,
 * environment SP_PANEL_LAYOUT) and the reduction in the tail of the last panel launch (default on where the
 * system's shape allows, SP_FUSE_REDUCE): where work runs and who reduces, never what is computed.     */
int sp_debug_set_panel_layout(sp_handle *h, int layout, int fuse_reduce);
/* (debug) wall-clock stamps of the panel kernel (csrc/sp_panel.hip); only in a library built with
 * -DSP_PANEL_TRACE (tools/ab_build.sh), SP_ERR_INVALID otherwise.  out == NULL resets; else
 * 64 x 3 x 16 int64 (pivot block, work item of star 0 {diagonal block, first tile, last tile},
 * stamp).  tools/panel2_trace.py                                                             */
int sp_debug_panel2_trace(long long *out);
/* (debug, same builds) the block every star's first item factors in its tail, per launch:
 * 16 x 64 x 4 int64 (pivot block j, star, {first item start, block start, block end, CU key}), then
 * 16 x 1024 int32: the CU key + 1 of every workgroup of the launch (0: none).                     */
int sp_debug_panel2_chain(long long *out);
/* (debug, process-wide) the symmetric trailing update of a remainder of at least `blocks` 64-column blocks runs on
 * 128 x 64 tiles (csrc/sp_gemm.hip, syrk128_kernel; default 17, environment SP_SYRK128_FROM; 0 = never, -1 = back to
 * the default): which kernel multiplies, never what is computed -- the results are bit-identical.        */
int sp_debug_set_syrk128_from(int blocks);
/* (debug, process-wide) the diagonal tiles of the symmetric trailing update on their ten lower 16 x 16 blocks, re-dealt
 * over the workgroup's four wavefronts three / three / two / two (csrc/sp_mm.h, SymDeal; default on, environment
 * SP_SYRK_SYMDIAG; 0 = the plain loop, all sixteen blocks; -1 = back to the default): which blocks are multiplied,
 * never what the wanted ones hold -- identical bits below the diagonal.                                       */
int sp_debug_set_syrk_symdiag(int on);
/* (debug, process-wide) sp_lnlike_ensemble_planned evaluates light curves of K <= 128 cadences (M + 2 <= 4 riding rows, a
 * lag grid that fits beside the tiles) in ONE kernel, a workgroup per star, nothing of the system in memory
 * (csrc/sp_small.hip; default on, environment SP_SMALL_K; 0 = the blocked path at every size, -1 = back to the
 * default): which kernels run, the same values to rounding.                                                        */
int sp_debug_set_small_k(int on);
/* (debug, host only) how the hot assembly kernel (csrc/sp_assemble.hip, assemble_sums_kernel) cuts a star's
 * ntr (ntr + 1) / 2 lower tiles (column-strip order) into nchunk chunks of equal COST: start_host[c] = first tile
 * of chunk c, c = 0 .. nchunk (start_host[nchunk] = the number of tiles).  A function of the shape alone.   */
int sp_debug_asm_chunks(int ntr, int nchunk, int *start_host);

#ifdef __cplusplus
}
#endif
#endif /* STARRY_PROCESS_AMD_H */
