// extern "C" entry points (include/starry_process_amd.h) and the fused
// log-likelihood driver.  Host logic only: argument checks, workspace layout,
// kernel sequencing on the caller's stream.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <new>

#include "sp_internal.h"
#include "sp_cov.h"

// launchers defined in the other translation units
int sp_launch_kernel_table(sp_handle *h, const double *rta1_dev, int ntab,
                           int covpts, const double *xp_dev, double *tab_dev,
                           double *meanvar_dev, hipStream_t st, int nsets = 0, const double *ez_dev = nullptr,
                           const double *Ez_dev = nullptr);
int sp_launch_theta(int S, int K, const double *t, const sp_star *stars,
                    double *theta, hipStream_t st, int32_t *info = nullptr,
                    uint32_t *status = nullptr, const double *tab = nullptr, int covpts = 0,
                    double *ptab = nullptr);
int sp_launch_spline_index(int K, const double *theta, double dx, long long *out,
                           hipStream_t st);
int sp_launch_rowsum(int S, int K, const double *theta, const double *t,
                     const sp_star *stars, int covpts, const double *tab,
                     const double *meanvar, const double *xp, int temporal,
                     const double *raw, double *rowsum, hipStream_t st);
int sp_launch_norm_coef(int S, int K, const sp_star *stars, const double *meanvar,
                        const double *condmean, int normalized, int order,
                        double zmax, const double *rowsum, double *qv, void *coef,
                        uint32_t *status, hipStream_t st);
int sp_launch_assemble(int S, int K, int M, int Kp, int system,
                       const double *theta, const double *t, const sp_star *stars,
                       int covpts, const double *tab, const double *meanvar,
                       const double *xp, int temporal, const double *raw,
                       int normalized, const double *qv, const void *coef,
                       const double *diag, int add_noise, const double *flux,
                       double *out, long ldo, long strideo, hipStream_t st, double *part = nullptr,
                       int lazy_nfull = 0);
int sp_launch_assemble_sums(int S, int K, int M, int Kp, const double *theta, const double *t,
                            const sp_star *stars, int covpts, const double *ptab, const double *meanvar,
                            int temporal, const double *flux, double *sys, hipStream_t st, double *part,
                            int lazy_nfull, int *nflat);
int sp_launch_defer_finish(int S, int K, int M, int Kp, const sp_star *stars, const double *meanvar,
                           const double *condmean, int order, double zmax, const double *part,
                           const double *diag, const double *flux, double *sys, void *coef, double *rscal,
                           uint32_t *status, hipStream_t st, int nflat = 0);
int sp_launch_assemble_planned(int S, int K, int M, int Kp, const PlanDev &plan, const double *t,
                               const sp_star *stars, int covpts, const double *tab, const double *meanvar,
                               int temporal, const double *flux, const double *diag, double *sys, int nfull,
                               int ncolw, int order, double zmax, void *coef, double *rscal, double *ptab, int32_t *info,
                               uint32_t *status, hipStream_t st, double *img, long lts, int fuse0, double *rid, int dfrom);
int sp_launch_cholesky_systems(sp_handle *h, double *sys, int S, int K, int Kp,
                               int32_t *info, double *invL, hipStream_t st);
bool sp_small_k_serves(int K, int M, int covpts, bool has_diag);
static int g_small_k = -1;       // sp_debug_set_small_k: -1 = the environment's SP_SMALL_K (default on)
static bool sp_small_k_on() {
  if (g_small_k >= 0) return g_small_k != 0;
  static const bool env = !(getenv("SP_SMALL_K") && atoi(getenv("SP_SMALL_K")) == 0);
  return env;
}
int sp_launch_small_lnlike(int S, int K, int M, const PlanDev &plan, const double *t, const sp_star *stars, int covpts,
                           const double *tab, const double *meanvar, int temporal, const double *flux, const double *diag,
                           int order, double zmax, double *lnlike, uint32_t *status_out, hipStream_t st);
int sp_launch_cond_system(const double *B1, const double *A, int N, int Kr, int S, int K, int M, int Kp,
                          const double *t, const sp_star *stars, int temporal, const void *coef,
                          const double *diag, const double *flux, double *sys, double *part,
                          hipStream_t st);
int sp_launch_cholesky_groups(sp_handle *h, int ngroups, const sp_chol_group *grp, int K,
                              int Kp);
int sp_launch_lnlike_reduce(const double *sys, int S, int K, int M, int Kp,
                            const int32_t *info, double *lnlike, uint32_t *status,
                            hipStream_t st, uint32_t *status_out = nullptr,
                            const sp_star *stars = nullptr, const void *defer_coef = nullptr,
                            const double *rscal = nullptr, int dvec = 0);
int sp_launch_pad_in(const double *A, int K, long lda, long strideA, double *sys,
                     int Kp, int M, const double *resid, int S, hipStream_t st, int ident = 0);
int sp_launch_pad_out(const double *sys, int Kp, double *A, int K, long lda,
                      long strideA, const int32_t *info, int S, hipStream_t st);
int sp_launch_cho_solve(const double *L, int K, long ldl, long strideL, double *B,
                        int nrhs, int batch, hipStream_t st);

static thread_local char g_hip_err[256] = "";

const char *sp_set_hip_error(hipError_t e, const char *what) {
  snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
  return g_hip_err;
}

namespace {

// ---- small kernels used only by the driver -----------------------------------

// per star: cos / sin of -inc  (the angle of the first rotation, flux.py:97)
__global__ void inc_cs_kernel(int S, const sp_star *__restrict__ stars,
                              double *__restrict__ cs) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  double sn, cn;
  sincos(-stars[s].inc, &sn, &cn);
  cs[2 * s] = cn;
  cs[2 * s + 1] = sn;
}

// v[s] = rTA1[table_s] . blockdiag(R(-inc_s))   (all K rows of the tiled
// operator are identical before the phase rotation, flux.py:280,97)
__global__ __launch_bounds__(256) void cond_prep_kernel(
    int N, int nwig, const int32_t *__restrict__ l_of, const int32_t *__restrict__ blk,
    const sp_star *__restrict__ stars, const double *__restrict__ rta1,
    const double *__restrict__ Rinc, double *__restrict__ v) {
  const int s = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int l = l_of[n], w = 2 * l + 1, base = l * l;
  const double *row = rta1 + (size_t)stars[s].table * N;
  const double *B = Rinc + (size_t)s * nwig + blk[l] + (n - base);
  double acc = 0.0;
  for (int i = 0; i < w; ++i) acc += row[base + i] * B[i * w];
  v[(size_t)s * N + n] = acc;
}

// The design matrix in ONE kernel, on the matrix cores (flux.py:278-281, 88-105; round 4): row k of star s is
//     A[s][k] = (v[s] o Rz(theta_k)) . blockdiag(Rx(pi/2)),     v[s] = rTA1 . Rx(-inc_s)  (cond_prep_kernel),
// a function of theta_k alone, so the K x N intermediate of rounds 1-3 (a phase-rotation kernel's output: written,
// padded, read back by two 190 us launches of dotrx_kernel with one 256-thread workgroup PER ROW) never exists.
// A workgroup takes 64 rows of a star, a wavefront 16 of them: per degree l the (16 x w)(w x w) product, w = 2 l + 1,
// as ceil(w / 16) x ceil(w / 4) v_mfma_f64_16x16x4_f64 -- the A fragments are formed on the fly from v and the
// row's cos / sin table (the Chebyshev recurrence of wigner.h:305-316, one thread per row), the B fragments come
// from the packed rotation in LDS (RLDS; from L2 for degrees whose NWIG doubles do not fit).  A first form on the
// vector ALU (thread = output column, 8 rows per workgroup) spent 178 us on its LDS reads -- w x 9 of them for
// 8 w multiply-adds; this one needs two per 64.  Rows K .. Kr - 1 are zero.
typedef double cd_d4 __attribute__((ext_vector_type(4)));
template <bool RLDS>
__global__ __launch_bounds__(256) void cond_design_kernel(
    int ydeg, int N, int nwig, int K, int Kr, const double *__restrict__ v, const double *__restrict__ theta,
    const double *__restrict__ Rpk, double *__restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) double cd_lds[];
  const int nc = ydeg + 1;
  double *sV = cd_lds;                       // N
  double *sC = sV + N;                       // 64 x nc: cos(m theta_row)
  double *sS = sC + 64 * nc;                 // 64 x nc: sin
  double *sR = sS + 64 * nc;                 // nwig (RLDS)
  const int s = blockIdx.y, r0 = blockIdx.x * 64, tid = threadIdx.x;
  if (RLDS) {
    // (batches of eight loads per thread, all in flight before the first store)
    for (int e0 = 0; e0 < nwig; e0 += 8 * 256) {
      double tmp[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u + tid;
        tmp[u] = Rpk[e < nwig ? e : 0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u + tid;
        if (e < nwig) sR[e] = tmp[u];
      }
    }
  }
  for (int n = tid; n < N; n += 256) sV[n] = v[(size_t)s * N + n];
  if (tid < 64) {
    double *cn = sC + tid * nc, *sn = sS + tid * nc;
    if (r0 + tid < K) {
      double s1, c1;
      sincos(theta[(size_t)s * K + r0 + tid], &s1, &c1);
      cn[0] = 1.0;
      sn[0] = 0.0;
      if (ydeg >= 1) {
        cn[1] = c1;
        sn[1] = s1;
      }
      for (int n = 2; n <= ydeg; ++n) {
        cn[n] = 2.0 * cn[n - 1] * c1 - cn[n - 2];
        sn[n] = 2.0 * sn[n - 1] * c1 - sn[n - 2];
      }
    } else {
      for (int n = 0; n <= ydeg; ++n) cn[n] = sn[n] = 0.0;   // (a padding row: zeros)
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
  const double *crow = sC + (16 * wave + fr) * nc, *srow = sS + (16 * wave + fr) * nc;
  const double *Rsrc = RLDS ? sR : Rpk;
  for (int l = 0; l <= ydeg; ++l) {
    const int w = 2 * l + 1, base = l * l, boff = l * (4 * l * l - 1) / 3;   // sum_{k < l} (2 k + 1)^2
    const int nk = (w + 3) / 4, nct = (w + 15) / 16;
    for (int ct = 0; ct < nct; ++ct) {
      cd_d4 acc = {0.0, 0.0, 0.0, 0.0};
      const int col = 16 * ct + fr;
      for (int kk = 0; kk < nk; ++kk) {
        const int k = 4 * kk + fk;
        double a = 0.0, b = 0.0;
        if (k < w) {
          // entry n = base + k of the phase-rotated vector: m = k - l, its mirror is base + 2 l - k (wigner.h:289-339)
          const int m = k - l, am = m < 0 ? -m : m;
          const double sm = m < 0 ? -srow[am] : srow[am];
          a = sV[base + k] * crow[am] + sV[base + 2 * l - k] * sm;
          if (col < w) b = Rsrc[boff + k * w + col];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      if (col < w) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int orow = r0 + 16 * wave + fk + 4 * r;
          if (orow < Kr) A[((size_t)s * Kr + orow) * N + base + col] = acc[r];
        }
      }
    }
  }
}

// mean[s] = (A mu_y)[0]   (flux.py:340)
__global__ __launch_bounds__(256) void cond_mean_kernel(int N, int K /* rows per star in A */,
                                                        const double *__restrict__ A,
                                                        const double *__restrict__ mu,
                                                        double *__restrict__ mean) {
  __shared__ double red[4];
  const int s = blockIdx.x;
  const double *row = A + (size_t)s * K * N;
  double part = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) part += row[n] * mu[n];
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) mean[s] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void get_z_kernel(int S, const double *__restrict__ coef,
                             double *__restrict__ z) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S) z[s] = coef[8 * s + 3];
}

__global__ void get_gpmean_kernel(int S, const double *__restrict__ coef,
                                  double *__restrict__ out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S) out[s] = coef[8 * s + 6] - 1.0;  // mu - 1 = flux mean
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  int S, K, M, Kp, N, NWIG;
  int Kr;   // rows per star of the design-matrix buffers A, B1: roundup(K, 64), the rows beyond K zero
  size_t theta, rowsum, qv, coef, rscal, info, status, condmean, cs, vrow, Rinc, invL, A,
      B1, raw, part, sys, total;
};

// rows below the matrix that the deferred normalisation adds (L^-1 1 and, with per-cadence variances, L^-1 d;
// sp_reduce.h -- rounds 2-4: p, q, 1)
#define SP_DEFER_ROWS 2

// lean: no design-matrix buffers and no raw covariance (the SPD inverse and the gradient's sweep touch neither: at cfg3's
// shape they were 1.5 GB of the 3.5 GB those calls asked for -- ADVICE r04)
Layout make_layout(const sp_handle *h, int S, int K, int M, bool with_sys, bool lean = false) {
  Layout L;
  L.S = S;
  L.K = K;
  L.M = M;
  // (sized for the deferred normalisation whatever the handle's current setting: it is the superset,
  //  so a workspace sized before sp_set_defer_norm changes is never too small)
  L.Kp = sp_roundup(K + M + (with_sys ? SP_DEFER_ROWS : 0), SP_NB);
  L.N = h->N;
  L.NWIG = h->NWIG;
  L.Kr = sp_roundup(K, SP_NB);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align_up(bytes);
    return o;
  };
  const size_t d = sizeof(double);
  L.theta = take(d * S * K);
  L.rowsum = take(d * S * K);
  L.qv = take(d * S * K);
  L.coef = take(d * S * 8);
  L.rscal = take(d * S * (SP_RSCAL_HEAD + (size_t)M));
  L.info = take(sizeof(int32_t) * S);
  L.status = take(sizeof(uint32_t) * S);
  L.condmean = take(d * S);
  L.cs = take(d * S * 2);
  L.vrow = take(d * S * L.N);
  L.Rinc = take(d * S * L.NWIG);
  L.invL = take(d * (size_t)S * sp_lt_stride(L.Kp));
  L.A = take(lean ? 0 : d * (size_t)S * L.Kr * L.N);
  L.B1 = take(lean ? 0 : d * (size_t)S * L.Kr * L.N);
  L.raw = take(lean ? 0 : d * (size_t)S * K * K);
  L.part = with_sys ? take(d * (size_t)S * (L.Kp / SP_NB) * K) : off;
  L.sys = with_sys ? take(d * (size_t)S * L.Kp * L.Kp) : off;
  L.total = off;
  return L;
}

template <typename T>
T *at(void *base, size_t off) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + off);
}

// grow-only device scratch owned by the handle (non-fused ops only)
int ensure_big(sp_handle *h, size_t bytes, void **out);

int check_handle(const sp_handle *h) { return h ? SP_OK : SP_ERR_INVALID; }

}  // namespace

// (the handle's grow-only scratch for the other translation units)
int ensure_big_scratch(sp_handle *h, size_t bytes, void **out) { return ensure_big(h, bytes, out); }

namespace {
// (growing synchronises the device: the old buffer may still be in use by launches of this
//  handle on any stream; steady-state calls never get here)
int ensure_big(sp_handle *h, size_t bytes, void **out) {
  if (h->big_bytes < bytes) {
    SP_HIP(hipDeviceSynchronize());
    if (h->big_ptr) SP_HIP(hipFree(h->big_ptr));
    h->big_ptr = nullptr;
    h->big_bytes = 0;
    hipError_t e = hipMalloc(&h->big_ptr, bytes);
    if (e != hipSuccess) {
      sp_set_hip_error(e, "hipMalloc(scratch)");
      return SP_ERR_ALLOC;
    }
    h->big_bytes = bytes;
  }
  *out = h->big_ptr;
  return SP_OK;
}

// design matrix for S stars into A_out (uses L.theta already filled); Kr rows per star in A_out and
// in the scratch L.B1 (Kr == K: contiguous stars; Kr > K: the rows beyond K are zeroed)
int build_design(sp_handle *h, const Layout &L, void *ws, const sp_star *stars,
                 const double *rta1, double *A_out, hipStream_t st, int Kr) {
  const int S = L.S, K = L.K, N = L.N;
  double *cs = at<double>(ws, L.cs), *vrow = at<double>(ws, L.vrow);
  double *Rinc = at<double>(ws, L.Rinc), *theta = at<double>(ws, L.theta);
  hipLaunchKernelGGL(inc_cs_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S,
                     stars, cs);
  SP_LAUNCH_CHECK();
  int rc = sp_launch_Rx(h, cs, S, Rinc, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(cond_prep_kernel, dim3((N + 255) / 256, S), dim3(256), 0, st,
                     N, h->NWIG, h->d_l_of, h->d_blk, stars, rta1, Rinc, vrow);
  SP_LAUNCH_CHECK();
  const size_t small = sizeof(double) * ((size_t)N + 2 * 64 * (h->ydeg + 1));
  const size_t withR = small + sizeof(double) * (size_t)h->NWIG;
  dim3 grid((Kr + 63) / 64, S);
  if (withR <= 150 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cond_design_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL((cond_design_kernel<true>), grid, dim3(256), withR, st, h->ydeg, N, h->NWIG, K, Kr, vrow, theta,
                       h->d_Rx90, A_out);
  } else {
    hipLaunchKernelGGL((cond_design_kernel<false>), grid, dim3(256), small, st, h->ydeg, N, h->NWIG, K, Kr, vrow, theta,
                       h->d_Rx90, A_out);
  }
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// raw (un-normalised, no temporal factor) conditional covariance into L.raw
int build_conditional_raw(sp_handle *h, const Layout &L, void *ws, hipStream_t st) {
  const int S = L.S, K = L.K, N = L.N;
  double *A = at<double>(ws, L.A), *B1 = at<double>(ws, L.B1);
  double *raw = at<double>(ws, L.raw), *cm = at<double>(ws, L.condmean);
  hipLaunchKernelGGL(cond_mean_kernel, dim3(S), dim3(256), 0, st, N, K, A,
                     h->d_mean_ylm, cm);
  SP_LAUNCH_CHECK();
  // B1 = A Sigma_y  (Sigma_y symmetric: A . Sigma_y^T), then raw = B1 A^T
  int rc = sp_launch_gemm_nt(A, N, (long)K * N, h->d_cov_ylm, N, 0, B1, N,
                             (long)K * N, K, N, N, 1.0, 0, 0, S, st);
  if (rc) return rc;
  return sp_launch_gemm_nt(B1, N, (long)K * N, A, N, (long)K * N, raw, K,
                           (long)K * K, K, K, N, 1.0, 0, 0, S, st);
}

}  // namespace

namespace {

// the same layout seen by the stars s0 .. s0+Sg-1 only
Layout sub_layout(const Layout &L, int s0, int Sg) {
  Layout G = L;
  const size_t d = sizeof(double), z = (size_t)s0;
  G.S = Sg;
  G.theta += z * L.K * d;
  G.rowsum += z * L.K * d;
  G.qv += z * L.K * d;
  G.coef += z * 8 * d;
  G.rscal += z * (SP_RSCAL_HEAD + (size_t)L.M) * d;
  G.info += z * sizeof(int32_t);
  G.status += z * sizeof(uint32_t);
  G.condmean += z * d;
  G.cs += z * 2 * d;
  G.vrow += z * L.N * d;
  G.Rinc += z * L.NWIG * d;
  G.invL += z * sp_lt_stride(L.Kp) * d;
  G.A += z * L.Kr * L.N * d;
  G.B1 += z * L.Kr * L.N * d;
  G.raw += z * (size_t)L.K * L.K * d;
  G.part += z * (size_t)(L.Kp / SP_NB) * L.K * d;
  G.sys += z * (size_t)L.Kp * L.Kp * d;
  return G;
}

// stage A: everything up to the assembled systems, for one group on its stream
int lnlike_assemble(sp_handle *h, const Layout &L, void *ws, int K, int M, const double *t_dev,
                    const double *flux_dev, const double *diag_dev, const sp_star *stars_dev,
                    int conditional, int covpts, const double *tab_dev,
                    const double *meanvar_dev, const double *rta1_dev, int temporal,
                    int normalized, int norm_order, double zmax, hipStream_t st, int lazy_nfull = 0) {
  const int S = L.S;
  double *theta = at<double>(ws, L.theta), *rowsum = at<double>(ws, L.rowsum);
  double *qv = at<double>(ws, L.qv), *coef = at<double>(ws, L.coef);
  double *raw = at<double>(ws, L.raw), *cm = at<double>(ws, L.condmean);
  double *sys = at<double>(ws, L.sys);
  int32_t *info = at<int32_t>(ws, L.info);
  uint32_t *status = at<uint32_t>(ws, L.status);
  int rc;
  // (the stars' tables packed for the spline gathers of the assembly and of the tiles formed at first
  //  touch, in the design-matrix region the marginal path does not use)
  double *ptab = (conditional || 4 * (size_t)(covpts + 4) > (size_t)L.Kr * L.N) ? nullptr : at<double>(ws, L.A);
  if ((rc = sp_launch_theta(S, K, t_dev, stars_dev, theta, st, info, status, tab_dev, covpts, ptab)))
    return rc;
  const double *rawp = nullptr;
  const double *condmean = nullptr;
  if (conditional && (L.N % 64) == 0 && ((normalized && h->defer_norm) || !normalized)) {
    // Conditional branch, fused (sp_cond.hip): B1 = A Sigma_y, then the LOWER tiles of B1 A^T land in
    // the system assembled -- no K x K raw matrix written and read back (round 2: 0.5 GB per 64-star
    // step), 40 % fewer flops in the second product.  The design-matrix buffers hold roundup(K, 64)
    // rows per star (zero beyond K) so that every tile of the products is a full one.
    const int N = L.N, Kr = L.Kr;
    double *A = at<double>(ws, L.A), *B1 = at<double>(ws, L.B1);
    if ((rc = build_design(h, L, ws, stars_dev, rta1_dev, A, st, Kr))) return rc;
    hipLaunchKernelGGL(cond_mean_kernel, dim3(S), dim3(256), 0, st, N, Kr, A, h->d_mean_ylm, cm);
    SP_LAUNCH_CHECK();
    if ((rc = sp_launch_gemm_nt(A, N, (long)Kr * N, h->d_cov_ylm, N, 0, B1, N, (long)Kr * N, Kr, N, N, 1.0,
                                0, 0, S, st)))
      return rc;
    if (normalized) {
      double *part = at<double>(ws, L.part);
      if ((rc = sp_launch_cond_system(B1, A, N, Kr, S, K, M, L.Kp, t_dev, stars_dev, temporal, nullptr,
                                      nullptr, flux_dev, sys, part, st)))
        return rc;
      return sp_launch_defer_finish(S, K, M, L.Kp, stars_dev, meanvar_dev, cm, norm_order, zmax, part,
                                    diag_dev, flux_dev, sys, coef, at<double>(ws, L.rscal), status, st);
    }
    if ((rc = sp_launch_norm_coef(S, K, stars_dev, meanvar_dev, cm, 0, norm_order, zmax, rowsum, qv, coef,
                                  status, st)))
      return rc;
    return sp_launch_cond_system(B1, A, N, Kr, S, K, M, L.Kp, t_dev, stars_dev, temporal, coef, diag_dev,
                                 flux_dev, sys, nullptr, st);
  }
  if (conditional) {
    if ((rc = build_design(h, L, ws, stars_dev, rta1_dev, at<double>(ws, L.A), st, K)))
      return rc;
    if ((rc = build_conditional_raw(h, L, ws, st))) return rc;
    rawp = raw;
    condmean = cm;
  }
  const int cp = conditional ? 1 : covpts;
  if (normalized && h->defer_norm) {
    // deferred normalisation: ONE pass over the K^2 entries (raw tiles + their row / column sums),
    // then the normalisation's vectors as three more rows of the system (sp_assemble.hip)
    double *part = at<double>(ws, L.part);
    int nflat = 0;
    if (!rawp && ptab && sp_assemble_sums_lds(L.Kp, cp, temporal) <= SP_ASM_LDS_MAX)
      rc = sp_launch_assemble_sums(S, K, M, L.Kp, theta, t_dev, stars_dev, cp, ptab, meanvar_dev, temporal,
                                   flux_dev, sys, st, part, lazy_nfull, &nflat);
    else
      rc = sp_launch_assemble(S, K, M, L.Kp, 1, theta, t_dev, stars_dev, cp, tab_dev, meanvar_dev,
                              h->d_xp, temporal, rawp, 1, qv, coef, diag_dev, 1, flux_dev, sys,
                              L.Kp, (long)L.Kp * L.Kp, st, part, lazy_nfull);
    if (rc) return rc;
    return sp_launch_defer_finish(S, K, M, L.Kp, stars_dev, meanvar_dev, condmean, norm_order, zmax,
                                  part, diag_dev, flux_dev, sys, coef, at<double>(ws, L.rscal), status, st, nflat);
  }
  if (normalized)
    if ((rc = sp_launch_rowsum(S, K, theta, t_dev, stars_dev, cp, tab_dev, meanvar_dev,
                               h->d_xp, temporal, rawp, rowsum, st)))
      return rc;
  if ((rc = sp_launch_norm_coef(S, K, stars_dev, meanvar_dev, condmean, normalized,
                                norm_order, zmax, rowsum, qv, coef, status, st)))
    return rc;
  return sp_launch_assemble(S, K, M, L.Kp, 1, theta, t_dev, stars_dev, cp, tab_dev,
                            meanvar_dev, h->d_xp, temporal, rawp, normalized, qv, coef,
                            diag_dev, 1, flux_dev, sys, L.Kp, (long)L.Kp * L.Kp, st);
}

// stage C: reduction of one group's factored systems
int lnlike_finish(const Layout &L, void *ws, int K, int M, double *lnlike_dev,
                  uint32_t *status_dev, hipStream_t st, const sp_star *stars_dev, bool deferred, bool dvec) {
  const int S = L.S;
  int rc;
  uint32_t *status = at<uint32_t>(ws, L.status);
  if ((rc = sp_launch_lnlike_reduce(at<double>(ws, L.sys), S, K, M, L.Kp,
                                    at<int32_t>(ws, L.info), lnlike_dev, status, st,
                                    status_dev, stars_dev,
                                    deferred ? at<double>(ws, L.coef) : nullptr,
                                    deferred ? at<double>(ws, L.rscal) : nullptr, dvec ? 1 : 0)))
    return rc;
  return SP_OK;
}

}  // namespace

namespace {
// log det C = 2 sum_i log L_ii from the factored systems; NaN where the factorisation failed
__global__ __launch_bounds__(256) void logdet_kernel(const double *__restrict__ sys, long ld, long stride, int K,
                                                     const int32_t *__restrict__ info, double *__restrict__ out) {
  __shared__ double red[4];
  const double *M = sys + (size_t)blockIdx.x * stride;
  double a = 0.0;
  for (int i = threadIdx.x; i < K; i += 256) a += log(M[(size_t)i * ld + i]);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0)
    out[blockIdx.x] = (info && info[blockIdx.x]) ? __builtin_nan("") : 2.0 * ((red[0] + red[1]) + (red[2] + red[3]));
}
// what sp_spd_inverse_batched needs around a K x K matrix already in the system's top-left corner: the identity in
// the rows K .. K + Kr - 1 (columns < Kr: row K + m has its one at column m < K) and zeros in the columns K .. Kr - 1
// of the matrix rows -- nothing else of the Kp x Kp system is ever read.  grid (ceil(Kr / 256), K + Kr, S)
__global__ __launch_bounds__(256) void ident_rows_kernel(double *__restrict__ sys, long ld, long stride, int K, int Kr) {
  // one wavefront per row, 16 bytes per lane and pass (Kr is a multiple of 64, the rows 16-byte aligned: ld even)
  typedef double v2 __attribute__((ext_vector_type(2)));
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= K + Kr) return;
  double *row = sys + (size_t)blockIdx.y * stride + (size_t)i * ld;
  if (i >= K) {
    // (the zeros LEFT of a row's one are read too: they are the operands of the left-looking products and of the
    //  trailing updates of the launches that take the row's tile later)
    const int one = i - K < K ? i - K : -1;
    for (int j = 2 * lane; j < Kr; j += 128)
      *reinterpret_cast<v2 *>(row + j) = v2{j == one ? 1.0 : 0.0, j + 1 == one ? 1.0 : 0.0};
  } else {
    for (int j = K + lane; j < Kr; j += 64) row[j] = 0.0;
  }
}
// K x K matrices into the top-left corners of the systems.  grid (ceil(K / 256), K, S)
__global__ __launch_bounds__(256) void corner_copy_kernel(const double *__restrict__ A, long lda, long strideA,
                                                          double *__restrict__ sys, long ld, long stride, int K) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < K)
    sys[(size_t)blockIdx.z * stride + (size_t)blockIdx.y * ld + j] = A[(size_t)blockIdx.z * strideA + (size_t)blockIdx.y * lda + j];
}
// columns c0 .. c1 - 1 of `rows` rows from row r0 on: zero (the columns of the last, partial pivot block beyond
// the matrix, which the panel solve leaves undefined in the rows below)
__global__ __launch_bounds__(256) void zero_cols_kernel(double *__restrict__ sys, long ld, long stride, int r0,
                                                        int rows, int c0, int c1) {
  const int w = c1 - c0;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)rows * w) return;
  sys[(size_t)blockIdx.y * stride + (size_t)(r0 + e / w) * ld + c0 + e % w] = 0.0;
}
}  // namespace


extern "C" {

const char *sp_last_hip_error(void) { return g_hip_err; }

int sp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int sp_create(int ydeg, int udeg, int device, sp_handle **out) {
  if (!out || ydeg < 1 || ydeg > SP_MAX_YDEG || udeg < 0 || udeg > SP_MAX_UDEG)
    return SP_ERR_INVALID;
  *out = nullptr;
  // device == -1: host-only handle.  It serves the host entry points
  // (sp_rTA1, sp_rTA1L, sp_ydeg ...) and nothing else: every device entry
  // point answers SP_ERR_NO_DEVICE.  There is no CPU compute path.
  const bool host_only = device == -1;
  if (!host_only) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 ||
        device >= ndev)
      return SP_ERR_NO_DEVICE;
    SP_HIP(hipSetDevice(device));
  }
  sp_handle *h = new (std::nothrow) sp_handle();
  if (!h) return SP_ERR_ALLOC;
  h->ydeg = ydeg;
  h->udeg = udeg;
  h->N = (ydeg + 1) * (ydeg + 1);
  h->NWIG = sp_nwig_of(ydeg);
  h->device = device;
  h->have_marginal = false;
  h->have_moments = false;
  h->xp_covpts = -1;
  h->d_l_of = h->d_m_of = h->d_mirror = h->d_blk = nullptr;
  h->d_Rx90 = h->d_wnp = h->d_Wnp = h->d_xp = nullptr;
  h->d_mean_ylm = h->d_cov_ylm = h->d_ez = h->d_Ez = h->d_tmpNN = nullptr;
  h->d_scratch = nullptr;
  h->scratch_bytes = 0;
  h->d_tab_scratch = nullptr;
  h->tab_scratch_bytes = 0;
  h->table_attr_done = false;
  h->big_ptr = nullptr;
  h->big_bytes = 0;
  h->cs_ring.assign(4, sp_handle::CsSlot{nullptr, nullptr, 0, nullptr, false});
  h->cs_next = 0;
  h->superpanel = 0;
  h->groups = 1;
  h->gfork = nullptr;
  h->prof_on = false;
  h->prof_mask = 1u;
  h->prof_used = 0;
  h->defer_norm = 1;
  h->lazy_cov = 1;
  h->ncu = 256;
  h->look_ahead = 1;
  h->panel_layout = 1;
  h->fuse_reduce = 1;
  h->d_Rxm90 = nullptr;
  h->d_lamcs = nullptr;
  h->lamcs_Q = 0;
  h->d_size_basis = nullptr;
  h->size_spts = 0;
  h->size_sfac = 0.0;
  const int N = h->N;
  h->l_of.resize(N);
  h->m_of.resize(N);
  h->mirror.resize(N);
  h->m0.resize(ydeg + 1);
  h->blk.resize(ydeg + 2);
  sp_build_index_tables(ydeg, h->l_of.data(), h->m_of.data(), h->mirror.data(),
                        h->m0.data(), h->blk.data());
  sp_build_flux_constants(ydeg, udeg, h->rT, h->A1, h->U1, h->rta1);
  if (host_only) {
    *out = h;
    return SP_OK;
  }

  const size_t d = sizeof(double);
  SP_HIP(hipMalloc((void **)&h->d_l_of, sizeof(int32_t) * N));
  SP_HIP(hipMalloc((void **)&h->d_m_of, sizeof(int32_t) * N));
  SP_HIP(hipMalloc((void **)&h->d_mirror, sizeof(int32_t) * N));
  SP_HIP(hipMalloc((void **)&h->d_blk, sizeof(int32_t) * (ydeg + 2)));
  SP_HIP(hipMalloc((void **)&h->d_Rx90, d * h->NWIG));
  SP_HIP(hipMalloc((void **)&h->d_wnp, d * h->NWIG));
  SP_HIP(hipMalloc((void **)&h->d_Wnp, d * N * N));
  SP_HIP(hipMalloc((void **)&h->d_mean_ylm, d * N));
  SP_HIP(hipMalloc((void **)&h->d_cov_ylm, d * N * N));
  SP_HIP(hipMalloc((void **)&h->d_ez, d * N));
  SP_HIP(hipMalloc((void **)&h->d_Ez, d * N * N));
  SP_HIP(hipMalloc((void **)&h->d_tmpNN, d * N * N));
  h->scratch_bytes = d * (4 * (size_t)N + 64);
  SP_HIP(hipMalloc((void **)&h->d_scratch, h->scratch_bytes));
  h->d_xp = nullptr;
  SP_HIP(hipMemcpy(h->d_l_of, h->l_of.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(h->d_m_of, h->m_of.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(h->d_mirror, h->mirror.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(h->d_blk, h->blk.data(), sizeof(int32_t) * (ydeg + 2), hipMemcpyHostToDevice));

  {
    const char *e3 = getenv("SP_GROUPS");
    h->groups = e3 ? atoi(e3) : 1;
    const char *e10 = getenv("SP_DEFER_NORM");
    h->defer_norm = e10 ? atoi(e10) : 1;
    const char *e11 = getenv("SP_LAZY_COV");
    h->lazy_cov = e11 ? atoi(e11) : 1;
    {
      hipDeviceProp_t prop;
      h->ncu = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    const char *e12 = getenv("SP_PANEL_LA");
    h->look_ahead = e12 ? atoi(e12) : 1;
    const char *e13 = getenv("SP_PANEL_LAYOUT");
    h->panel_layout = e13 ? atoi(e13) : 1;
    const char *e14 = getenv("SP_FUSE_REDUCE");
    h->fuse_reduce = e14 ? atoi(e14) : 1;
    const char *e2 = getenv("SP_SUPER");
    h->superpanel = e2 ? atoi(e2) : 0;   // 0: chosen from K (sp_launch_cholesky_groups)
    if (h->superpanel < 0) h->superpanel = 0;
  }
  // Rx(pi/2): the polar-frame rotation every path uses (flux.py:56,61,62,103)
  const double th = 0.5 * M_PI;
  int rc = sp_Rx(h, &th, 1, h->d_Rx90, nullptr, nullptr);
  if (rc != SP_OK) {
    sp_destroy(h);
    return rc;
  }
  SP_HIP(hipDeviceSynchronize());
  *out = h;
  return SP_OK;
}

void sp_destroy(sp_handle *h) {
  if (!h) return;
  if (h->device < 0) {
    delete h;
    return;
  }
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  void *ptrs[] = {h->d_l_of, h->d_m_of,   h->d_mirror, h->d_blk,   h->d_Rx90,
                  h->d_wnp,  h->d_Wnp,    h->d_mean_ylm, h->d_cov_ylm, h->d_ez,
                  h->d_Ez,   h->d_tmpNN,  h->d_scratch, h->d_xp, h->d_tab_scratch, h->d_Rxm90, h->d_lamcs,
                  h->d_size_basis};
  for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->gdone) (void)hipEventDestroy(e);
  for (hipStream_t s2 : h->gstream) (void)hipStreamDestroy(s2);
  if (h->gfork) (void)hipEventDestroy(h->gfork);
  if (h->big_ptr) (void)hipFree(h->big_ptr);
  for (auto &c : h->cs_ring) {
    if (c.host) (void)hipHostFree(c.host);
    if (c.dev) (void)hipFree(c.dev);
    if (c.done) (void)hipEventDestroy(c.done);
  }
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete h;
}

int sp_ydeg(const sp_handle *h) { return h ? h->ydeg : SP_ERR_INVALID; }
int sp_udeg(const sp_handle *h) { return h ? h->udeg : SP_ERR_INVALID; }
int sp_nylm(const sp_handle *h) { return h ? h->N : SP_ERR_INVALID; }
int sp_nwig(const sp_handle *h) { return h ? h->NWIG : SP_ERR_INVALID; }

}  // extern "C"

int sp_stage_acquire(sp_handle *h, size_t doubles, sp_handle::CsSlot **out) {
  const size_t n = h->cs_ring.size();
  int pick = -1;
  for (size_t k = 0; k < n && pick < 0; ++k) {
    const size_t i = ((size_t)h->cs_next + k) % n;
    sp_handle::CsSlot &c = h->cs_ring[i];
    if (!c.used) {
      pick = (int)i;
    } else {
      const hipError_t q = hipEventQuery(c.done);
      if (q == hipSuccess) pick = (int)i;
      else (void)hipGetLastError();     // (hipErrorNotReady is an answer, not a failure to report later)
    }
  }
  if (pick < 0 && n < SP_STAGE_MAX) {
    h->cs_ring.push_back(sp_handle::CsSlot{nullptr, nullptr, 0, nullptr, false});
    pick = (int)n;
  }
  if (pick < 0) {
    pick = h->cs_next % (int)n;
    SP_HIP(hipEventSynchronize(h->cs_ring[pick].done));
  }
  sp_handle::CsSlot &c = h->cs_ring[pick];
  c.used = false;
  if (c.cap < doubles) {
    if (c.host) SP_HIP(hipHostFree(c.host));
    if (c.dev) SP_HIP(hipFree(c.dev));
    c.host = c.dev = nullptr;
    c.cap = 0;
    const size_t cap = doubles < 512 ? 512 : doubles;
    SP_HIP(hipHostMalloc((void **)&c.host, sizeof(double) * cap, hipHostMallocDefault));
    SP_HIP(hipMalloc((void **)&c.dev, sizeof(double) * cap));
    c.cap = cap;
  }
  if (!c.done) SP_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
  h->cs_next = (pick + 1) % (int)h->cs_ring.size();
  *out = &c;
  return SP_OK;
}

extern "C" {

int sp_stream_synchronize(sp_handle *h, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h) return SP_ERR_INVALID;
  SP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return SP_OK;
}

int sp_Rx(sp_handle *h, const double *theta_host, int nangles, double *R_dev,
          double *dR_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !theta_host || !R_dev || nangles < 0) return SP_ERR_INVALID;
  if (nangles == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  // cos/sin on the host with libm, like the reference (wigner.h:153-154), staged through the
  // handle's ring: the slot's previous use is waited for on the host (long finished in
  // practice), nothing is allocated, freed or synchronised once the ring has grown
  const size_t need = 2 * (size_t)nangles;
  sp_handle::CsSlot *cp = nullptr;
  {
    int rc = sp_stage_acquire(h, need, &cp);
    if (rc) return rc;
  }
  sp_handle::CsSlot &c = *cp;
  for (int i = 0; i < nangles; ++i) {
    c.host[2 * i] = std::cos(theta_host[i]);
    c.host[2 * i + 1] = std::sin(theta_host[i]);
  }
  SP_HIP(hipMemcpyAsync(c.dev, c.host, sizeof(double) * need, hipMemcpyHostToDevice, st));
  int rc = sp_launch_Rx(h, c.dev, nangles, R_dev, dR_dev, st);
  SP_HIP(hipEventRecord(c.done, st));
  c.used = true;
  return rc;
}

int sp_rTA1(sp_handle *h, double *rta1_host) {
  if (!h || !rta1_host) return SP_ERR_INVALID;
  memcpy(rta1_host, h->rta1.data(), sizeof(double) * h->N);
  return SP_OK;
}

int sp_rTA1L(sp_handle *h, const double *u_host, int nsets, double *out) {
  if (!h || !out || nsets < 0 || (h->udeg > 0 && !u_host)) return SP_ERR_INVALID;
  for (int i = 0; i < nsets; ++i)
    sp_host_rTA1L(h, u_host ? u_host + (size_t)i * h->udeg : nullptr,
                  out + (size_t)i * h->N);
  return SP_OK;
}

int sp_rTA1L_rev(sp_handle *h, const double *u_host, const double *bf_host, double *bu_host) {
  if (!h || h->udeg < 1 || !u_host || !bf_host || !bu_host) return SP_ERR_INVALID;
  sp_host_rTA1L_rev(h, u_host, bf_host, bu_host);
  return SP_OK;
}

int sp_set_marginal_constants(sp_handle *h, const double *wnp, const double *Wnp) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !wnp || !Wnp) return SP_ERR_INVALID;
  SP_HIP(hipSetDevice(h->device));
  SP_HIP(hipMemcpy(h->d_wnp, wnp, sizeof(double) * h->NWIG, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(h->d_Wnp, Wnp, sizeof(double) * h->N * h->N, hipMemcpyHostToDevice));
  h->have_marginal = true;
  return SP_OK;
}

int sp_set_ylm_moments(sp_handle *h, const double *mean_ylm, const double *cov_ylm) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !mean_ylm || !cov_ylm) return SP_ERR_INVALID;
  const int N = h->N;
  SP_HIP(hipSetDevice(h->device));
  SP_HIP(hipMemcpy(h->d_mean_ylm, mean_ylm, sizeof(double) * N, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(h->d_cov_ylm, cov_ylm, sizeof(double) * N * N, hipMemcpyHostToDevice));
  hipStream_t st = nullptr;
  // ez = R^T mu, Ez = R^T (Sigma + mu mu^T) R  (flux.py:55-62)
  int rc = sp_launch_polar_moments(h, h->d_mean_ylm, h->d_cov_ylm, st);
  if (rc) return rc;
  SP_HIP(hipStreamSynchronize(st));
  h->have_moments = true;
  return SP_OK;
}

int sp_set_ylm_moments_dev(sp_handle *h, const double *mean_ylm_dev,
                           const double *cov_ylm_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !mean_ylm_dev || !cov_ylm_dev) return SP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  // one launch: resident copies of mu / Sigma, ez and Ez
  int rc = sp_launch_polar_moments(h, mean_ylm_dev, cov_ylm_dev, st);
  if (rc) return rc;
  h->have_moments = true;
  return SP_OK;
}

int sp_profile_begin(sp_handle *h, int max_launches) {
  return sp_profile_begin_kinds(h, max_launches, 1u << SP_PROF_SYRK);
}

int sp_profile_begin_kinds(sp_handle *h, int max_launches, unsigned kind_mask) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || max_launches < 0) return SP_ERR_INVALID;
  h->prof_mask = kind_mask;
  while (h->prof_ev.size() < 2 * (size_t)max_launches) {
    hipEvent_t e;
    SP_HIP(hipEventCreate(&e));
    h->prof_ev.push_back(e);
  }
  h->prof_kind.assign(h->prof_ev.size() / 2, 0);
  h->prof_fl.assign(h->prof_ev.size() / 2, 0.0);
  h->prof_flp.assign(h->prof_ev.size() / 2, 0.0);
  h->prof_n.assign(h->prof_ev.size() / 2, 0);
  h->prof_used = 0;
  h->prof_on = true;
  return SP_OK;
}

int sp_profile_kind(sp_handle *h, int kind, long *launches, double *total_ms, double *flops) {
  return sp_profile_kind_ex(h, kind, launches, total_ms, flops, nullptr);
}

int sp_profile_kind_ex(sp_handle *h, int kind, long *launches, double *total_ms, double *flops, double *flops_padded) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || kind < 0 || kind >= SP_PROF_NKINDS) return SP_ERR_INVALID;
  h->prof_on = false;
  double ms = 0.0, fl = 0.0, flp = 0.0;
  long n = 0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    if (h->prof_kind[i / 2] != kind) continue;
    SP_HIP(hipEventSynchronize(h->prof_ev[i + 1]));
    float dt = 0.f;
    SP_HIP(hipEventElapsedTime(&dt, h->prof_ev[i], h->prof_ev[i + 1]));
    ms += dt;
    fl += h->prof_fl[i / 2];
    flp += h->prof_flp[i / 2];
    n += h->prof_n[i / 2];
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = fl;
  if (flops_padded) *flops_padded = flp;
  return SP_OK;
}

int sp_profile_end(sp_handle *h, long *launches, double *total_ms, double *flops) {
  return sp_profile_kind(h, SP_PROF_SYRK, launches, total_ms, flops);
}

int sp_set_lazy_cov(sp_handle *h, int on) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h) return SP_ERR_INVALID;
  h->lazy_cov = on ? 1 : 0;
  return SP_OK;
}

int sp_set_defer_norm(sp_handle *h, int on) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || (on != 0 && on != 1)) return SP_ERR_INVALID;
  h->defer_norm = on;
  return SP_OK;
}

// (debug, process-wide) the one-kernel path of short light curves on / off / back to the environment's setting
int sp_debug_set_small_k(int on) {
  g_small_k = on < 0 ? -1 : (on ? 1 : 0);
  return SP_OK;
}

// (debug) look-ahead items of the panel launches on / off (sp_cholesky.hip); results agree to rounding
int sp_debug_set_look_ahead(sp_handle *h, int on) {
  if (!h) return SP_ERR_INVALID;
  h->look_ahead = on ? 1 : 0;
  return SP_OK;
}

// (debug) the panel launches' layout by CU and the reduction in the last launch's tail, on / off: same bits
int sp_debug_set_panel_layout(sp_handle *h, int layout, int fuse_reduce) {
  if (!h) return SP_ERR_INVALID;
  h->panel_layout = layout ? 1 : 0;
  h->fuse_reduce = fuse_reduce ? 1 : 0;
  return SP_OK;
}

int sp_get_polar_moments(sp_handle *h, double *ez, double *Ez) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h) return SP_ERR_INVALID;
  if (!h->have_moments) return SP_ERR_STATE;
  SP_HIP(hipSetDevice(h->device));
  if (ez) SP_HIP(hipMemcpy(ez, h->d_ez, sizeof(double) * h->N, hipMemcpyDeviceToHost));
  if (Ez)
    SP_HIP(hipMemcpy(Ez, h->d_Ez, sizeof(double) * h->N * h->N, hipMemcpyDeviceToHost));
  return SP_OK;
}

// the lag grid of the kernel table on the device: uploaded when it changes (sp_kernel_table, sp_kernel_table_samples)
int sp_ensure_lag_grid(sp_handle *h, int covpts, const double *xp_host) {
  const int np = covpts + 4;
  const bool same = h->xp_covpts == covpts && (int)h->xp_host.size() == np &&
                    memcmp(h->xp_host.data(), xp_host, sizeof(double) * np) == 0;
  if (same) return SP_OK;
  // new lag grid: (re)allocate and upload once; later calls with the same
  // grid are fully asynchronous
  SP_HIP(hipSetDevice(h->device));
  SP_HIP(hipDeviceSynchronize());
  if (h->d_xp) SP_HIP(hipFree(h->d_xp));
  h->d_xp = nullptr;
  h->xp_covpts = -1;
  SP_HIP(hipMalloc((void **)&h->d_xp, sizeof(double) * np));
  SP_HIP(hipMemcpy(h->d_xp, xp_host, sizeof(double) * np, hipMemcpyHostToDevice));
  h->xp_host.assign(xp_host, xp_host + np);
  h->xp_covpts = covpts;
  return SP_OK;
}

int sp_kernel_table(sp_handle *h, const double *rta1_dev, int ntab, int covpts,
                    const double *xp_host, double *tab_dev, double *meanvar_dev,
                    void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !rta1_dev || !xp_host || !tab_dev || !meanvar_dev || ntab < 0 ||
      covpts < 1)
    return SP_ERR_INVALID;
  if (!h->have_marginal || !h->have_moments) return SP_ERR_STATE;
  if (ntab == 0) return SP_OK;
  int rc = sp_ensure_lag_grid(h, covpts, xp_host);
  if (rc) return rc;
  return sp_launch_kernel_table(h, rta1_dev, ntab, covpts, h->d_xp, tab_dev,
                                meanvar_dev, (hipStream_t)stream);
}

// The kernel tables of B hyperparameter samples in one call (round 6): polar-frame moments ez [B][N], Ez [B][N][N]
// given (sp_polar_moments_samples), table b ntab + i from sample b's moments and flux operator i.
int sp_kernel_table_samples(sp_handle *h, int B, const double *ez_dev, const double *Ez_dev, const double *rta1_dev,
                            int ntab, int covpts, const double *xp_host, double *tab_dev, double *meanvar_dev,
                            void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !ez_dev || !Ez_dev || !rta1_dev || !xp_host || !tab_dev || !meanvar_dev || ntab < 0 || covpts < 1 || B < 0)
    return SP_ERR_INVALID;
  if (!h->have_marginal) return SP_ERR_STATE;
  if (ntab == 0 || B == 0) return SP_OK;
  int rc = sp_ensure_lag_grid(h, covpts, xp_host);
  if (rc) return rc;
  return sp_launch_kernel_table(h, rta1_dev, ntab, covpts, h->d_xp, tab_dev, meanvar_dev, (hipStream_t)stream, B,
                                ez_dev, Ez_dev);
}

int sp_cov_marginal_batched(sp_handle *h, int S, int K, const double *t_dev,
                            const sp_star *stars_dev, int covpts,
                            const double *tab_dev, const double *meanvar_dev,
                            int temporal, int normalized, int norm_order,
                            double *cov_dev, long ldc, long stridec,
                            double *z_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !t_dev || !stars_dev || !tab_dev || !meanvar_dev || !cov_dev || S < 0 ||
      K < 1 || ldc < K || norm_order < 0 || norm_order > SP_NORM_MAXORDER)
    return SP_ERR_INVALID;
  if (h->xp_covpts != covpts) return SP_ERR_STATE;
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(h, S, K, 0, false);
  // only the small per-star arrays are needed here
  void *ws = nullptr;
  int rc = ensure_big(h, L.condmean + 256, &ws);
  if (rc) return rc;
  double *theta = at<double>(ws, L.theta), *rowsum = at<double>(ws, L.rowsum);
  double *qv = at<double>(ws, L.qv), *coef = at<double>(ws, L.coef);
  if ((rc = sp_launch_theta(S, K, t_dev, stars_dev, theta, st))) return rc;
  if (normalized)
    if ((rc = sp_launch_rowsum(S, K, theta, t_dev, stars_dev, covpts, tab_dev,
                               meanvar_dev, h->d_xp, temporal, nullptr, rowsum, st)))
      return rc;
  if ((rc = sp_launch_norm_coef(S, K, stars_dev, meanvar_dev, nullptr, normalized,
                                norm_order, INFINITY, rowsum, qv, coef, nullptr, st)))
    return rc;
  if ((rc = sp_launch_assemble(S, K, 0, K, 0, theta, t_dev, stars_dev, covpts,
                               tab_dev, meanvar_dev, h->d_xp, temporal, nullptr,
                               normalized, qv, coef, nullptr, 0, nullptr, cov_dev,
                               ldc, stridec, st)))
    return rc;
  if (z_dev) {
    hipLaunchKernelGGL(get_z_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, coef,
                       z_dev);
    SP_LAUNCH_CHECK();
  }
  return SP_OK;
}

int sp_design_matrix(sp_handle *h, int S, int K, const double *t_dev,
                     const sp_star *stars_dev, const double *rta1_dev,
                     double *A_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !t_dev || !stars_dev || !rta1_dev || !A_dev || S < 0 || K < 1)
    return SP_ERR_INVALID;
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(h, S, K, 0, false);
  void *ws = nullptr;
  int rc = ensure_big(h, L.raw, &ws);  // up to and including B1
  if (rc) return rc;
  if ((rc = sp_launch_theta(S, K, t_dev, stars_dev, at<double>(ws, L.theta), st)))
    return rc;
  return build_design(h, L, ws, stars_dev, rta1_dev, A_dev, st, K);
}

int sp_cov_conditional_batched(sp_handle *h, int S, int K, const double *t_dev,
                               const sp_star *stars_dev, const double *rta1_dev,
                               int temporal, int normalized, int norm_order,
                               double *cov_dev, long ldc, long stridec,
                               double *mean_dev, double *z_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !t_dev || !stars_dev || !rta1_dev || !cov_dev || S < 0 || K < 1 ||
      ldc < K || norm_order < 0 || norm_order > SP_NORM_MAXORDER)
    return SP_ERR_INVALID;
  if (!h->have_moments) return SP_ERR_STATE;
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(h, S, K, 0, false);
  void *ws = nullptr;
  int rc = ensure_big(h, L.total, &ws);
  if (rc) return rc;
  double *theta = at<double>(ws, L.theta), *rowsum = at<double>(ws, L.rowsum);
  double *qv = at<double>(ws, L.qv), *coef = at<double>(ws, L.coef);
  double *raw = at<double>(ws, L.raw), *cm = at<double>(ws, L.condmean);
  if ((rc = sp_launch_theta(S, K, t_dev, stars_dev, theta, st))) return rc;
  if ((rc = build_design(h, L, ws, stars_dev, rta1_dev, at<double>(ws, L.A), st, K)))
    return rc;
  if ((rc = build_conditional_raw(h, L, ws, st))) return rc;
  if (normalized)
    if ((rc = sp_launch_rowsum(S, K, theta, t_dev, stars_dev, 1, nullptr, nullptr,
                               nullptr, temporal, raw, rowsum, st)))
      return rc;
  if ((rc = sp_launch_norm_coef(S, K, stars_dev, nullptr, cm, normalized, norm_order,
                                INFINITY, rowsum, qv, coef, nullptr, st)))
    return rc;
  if ((rc = sp_launch_assemble(S, K, 0, K, 0, theta, t_dev, stars_dev, 1, nullptr,
                               nullptr, nullptr, temporal, raw, normalized, qv, coef,
                               nullptr, 0, nullptr, cov_dev, ldc, stridec, st)))
    return rc;
  if (mean_dev) {
    hipLaunchKernelGGL(get_gpmean_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S,
                       coef, mean_dev);
    SP_LAUNCH_CHECK();
  }
  if (z_dev) {
    hipLaunchKernelGGL(get_z_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, coef,
                       z_dev);
    SP_LAUNCH_CHECK();
  }
  return SP_OK;
}

int sp_cho_factor(sp_handle *h, double *A_dev, int K, long lda, long strideA,
                  int batch, int32_t *info_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !A_dev || K < 1 || lda < K || batch < 0) return SP_ERR_INVALID;
  if (batch == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  const int Kp = sp_roundup(K, SP_NB);
  const size_t sysb = align_up(sizeof(double) * (size_t)batch * Kp * Kp);
  const size_t invb = align_up(sizeof(double) * (size_t)batch * sp_lt_stride(Kp));
  void *ws = nullptr;
  int rc = ensure_big(h, sysb + invb + align_up(sizeof(int32_t) * batch), &ws);
  if (rc) return rc;
  double *sys = at<double>(ws, 0);
  double *invL = at<double>(ws, sysb);
  int32_t *info = at<int32_t>(ws, sysb + invb);
  SP_HIP(hipMemsetAsync(info, 0, sizeof(int32_t) * batch, st));
  if ((rc = sp_launch_pad_in(A_dev, K, lda, strideA, sys, Kp, 0, nullptr, batch, st)))
    return rc;
  if ((rc = sp_launch_cholesky_systems(h, sys, batch, K, Kp, info, invL, st))) return rc;
  if ((rc = sp_launch_pad_out(sys, Kp, A_dev, K, lda, strideA, info, batch, st)))
    return rc;
  if (info_dev)
    SP_HIP(hipMemcpyAsync(info_dev, info, sizeof(int32_t) * batch,
                          hipMemcpyDeviceToDevice, st));
  return SP_OK;
}

int sp_cho_solve(sp_handle *h, const double *L_dev, int K, long ldl, long strideL,
                 double *b_dev, int nrhs, int batch, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !L_dev || !b_dev || K < 1 || ldl < K || nrhs < 0 || batch < 0)
    return SP_ERR_INVALID;
  if (nrhs == 0 || batch == 0) return SP_OK;
  if (nrhs > 65535 || batch > 65535) return SP_ERR_INVALID;
  return sp_launch_cho_solve(L_dev, K, ldl, strideL, b_dev, nrhs, batch,
                             (hipStream_t)stream);
}

int sp_tri_solve(sp_handle *h, const double *L_dev, int K, long ldl, long strideL, double *b_dev,
                 int nrhs, int batch, int trans, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !L_dev || !b_dev || K < 1 || ldl < K || nrhs < 0 || batch < 0) return SP_ERR_INVALID;
  if (nrhs == 0 || batch == 0) return SP_OK;
  if (batch > 65535) return SP_ERR_INVALID;
  return sp_launch_tri_solve(L_dev, K, ldl, strideL, b_dev, (long)K * nrhs, nrhs, 1, nrhs, batch,
                             trans ? 2 : 1, (hipStream_t)stream);
}

int sp_solve_rev(sp_handle *h, const double *L_dev, int K, long ldl, long strideL,
                 const double *c_dev, const double *cbar_dev, int nrhs, int batch, int trans,
                 double *Abar_dev, double *bbar_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !L_dev || !c_dev || !cbar_dev || !Abar_dev || !bbar_dev || K < 1 || ldl < K ||
      nrhs < 1 || batch < 0 || batch > 65535)
    return SP_ERR_INVALID;
  if (batch == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  const long sb = (long)K * nrhs;
  int rc;
  // b_bar = A^-T c_bar: the transposed system (math.py:55-63)
  SP_HIP(hipMemcpyAsync(bbar_dev, cbar_dev, sizeof(double) * (size_t)batch * sb,
                        hipMemcpyDeviceToDevice, st));
  if ((rc = sp_launch_tri_solve(L_dev, K, ldl, strideL, bbar_dev, sb, nrhs, 1, nrhs, batch,
                                trans ? 1 : 2, st)))
    return rc;
  // A_bar = -b_bar c^T, restricted to the triangle A lives on (math.py:65-69)
  if ((rc = sp_launch_gemm_nt(bbar_dev, nrhs, sb, c_dev, nrhs, sb, Abar_dev, K, (long)K * K, K, K,
                              nrhs, -1.0, 0, 0, batch, st)))
    return rc;
  return sp_launch_tri_mask(Abar_dev, K, batch, trans ? 1 : 0, 1.0, st);
}

int sp_cholesky_rev(sp_handle *h, const double *L_dev, int K, long ldl, long strideL,
                    const double *Lbar_dev, int batch, double *Cbar_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !L_dev || !Lbar_dev || !Cbar_dev || K < 1 || ldl < K || batch < 0 || batch > 65535)
    return SP_ERR_INVALID;
  if (batch == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  const long kk = (long)K * K;
  const size_t mb = align_up(sizeof(double) * (size_t)batch * kk);
  void *ws = nullptr;
  int rc = ensure_big(h, 3 * mb, &ws);
  if (rc) return rc;
  double *Lt = at<double>(ws, 0), *Lbt = at<double>(ws, mb), *P = at<double>(ws, 2 * mb);
  // P = L^T L_bar
  if ((rc = sp_launch_transpose(L_dev, ldl, strideL, Lt, K, batch, st))) return rc;
  if ((rc = sp_launch_transpose(Lbar_dev, K, kk, Lbt, K, batch, st))) return rc;
  if ((rc = sp_launch_gemm_nt(Lt, K, kk, Lbt, K, kk, P, K, kk, K, K, K, 1.0, 0, 0, batch, st)))
    return rc;
  // Phi = tril(P) with the diagonal halved
  if ((rc = sp_launch_tri_mask(P, K, batch, 0, 0.5, st))) return rc;
  // S = L^-T Phi L^-1: solve L^T X = Phi^T with P read as its own transpose (X^T = Phi L^-1
  // lands in P row-major), then L^T S = X^T
  if ((rc = sp_launch_tri_solve(L_dev, K, ldl, strideL, P, kk, 1, K, K, batch, 2, st))) return rc;
  if ((rc = sp_launch_tri_solve(L_dev, K, ldl, strideL, P, kk, K, 1, K, batch, 2, st))) return rc;
  return sp_launch_chol_rev_finish(P, L_dev, ldl, strideL, Cbar_dev, K, batch, st);
}

long sp_lnlike_workspace_bytes(sp_handle *h, int S, int K, int M) {
  if (!h || S < 0 || K < 1 || M < 1) return SP_ERR_INVALID;
  return (long)make_layout(h, S, K, M, true).total;
}

int sp_lnlike_ensemble(sp_handle *h, int S, int K, int M, const double *t_dev,
                       const double *flux_dev, const double *diag_dev,
                       const sp_star *stars_dev, int conditional, int covpts,
                       const double *tab_dev, const double *meanvar_dev,
                       const double *rta1_dev, int temporal, int normalized,
                       int norm_order, double zmax, void *workspace_dev,
                       double *lnlike_dev, uint32_t *status_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !t_dev || !flux_dev || !stars_dev || !workspace_dev || !lnlike_dev ||
      S < 0 || K < 1 || M < 1 || norm_order < 0 || norm_order > SP_NORM_MAXORDER)
    return SP_ERR_INVALID;
  if (conditional) {
    if (!rta1_dev) return SP_ERR_INVALID;
    if (!h->have_moments) return SP_ERR_STATE;
  } else {
    if (!tab_dev || !meanvar_dev || covpts < 1) return SP_ERR_INVALID;
    if (h->xp_covpts != covpts) return SP_ERR_STATE;
  }
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(h, S, K, M, true);
  void *ws = workspace_dev;
  // Star groups on concurrent streams: the diagonal-block kernel
  // is a latency-bound chain that occupies 1/4 of the CUs with one wavefront
  // each; with G groups in flight one group's GEMMs fill the machine while
  // another group sits in its chain.  No event traffic inside the loop: one fork
  // and one join per call.
  int G = h->groups;
  if (G > S / 8) G = S / 8;  // keep groups large enough to fill the matrix cores
  if (G < 1) G = 1;
  if (G > 1) {
    while ((int)h->gstream.size() < G - 1) {
      hipStream_t s2;
      hipEvent_t e2;
      SP_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      SP_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      h->gstream.push_back(s2);
      h->gdone.push_back(e2);
    }
    if (!h->gfork) SP_HIP(hipEventCreateWithFlags(&h->gfork, hipEventDisableTiming));
    SP_HIP(hipEventRecord(h->gfork, st));
  }
  const bool fused_reduce = sp_panel_fuses_reduce(h, K, L.Kp);
  std::vector<Layout> LG(G, L);
  std::vector<sp_chol_group> CG(G);
  std::vector<int> first(G);
  for (int g = 0; g < G; ++g) {
    const int s0 = (int)((long)S * g / G), s1 = (int)((long)S * (g + 1) / G);
    first[g] = s0;
    LG[g] = sub_layout(L, s0, s1 - s0);
    hipStream_t sg = g == 0 ? st : h->gstream[g - 1];
    if (g > 0) SP_HIP(hipStreamWaitEvent(sg, h->gfork, 0));
    CG[g] = sp_chol_group{at<double>(ws, LG[g].sys), at<int32_t>(ws, LG[g].info),
                          at<double>(ws, LG[g].invL), s1 - s0, sg, LazyCov{}, SpReduceArgs{}};
    // the reduction rides in the tail of the last panel launch when the system's shape allows
    if (fused_reduce)
      CG[g].red = SpReduceArgs{lnlike_dev + s0, at<uint32_t>(ws, LG[g].status),
                               status_dev ? status_dev + s0 : nullptr, stars_dev + s0,
                               (normalized && h->defer_norm) ? (const void *)at<double>(ws, LG[g].coef) : nullptr,
                               at<double>(ws, LG[g].rscal), diag_dev ? 1 : 0, K, M,
                               K + M + ((normalized && h->defer_norm) ? (diag_dev ? 2 : 1) : 0)};
  }
  // Tiles formed at first touch (LazyCov, sp_cov.h): the marginal path under the deferred
  // normalisation.
  int lazy_nfull = 0;
  // (not with a temporal kernel: its exp per entry, evaluated twice, costs more than the traffic
  //  it saves -- cfg5 shape: -2.5 %)
  if (h->lazy_cov && !conditional && temporal == SP_TEMPORAL_NONE && normalized && h->defer_norm && G == 1 &&
      K / SP_NB >= 2 &&
      (size_t)K * L.N >= 4 * (size_t)(covpts + 4) && 4 * (covpts + 4) + 64 <= SP_TILE_LDS_MIN) {   // (+ a tile's column phases)
    lazy_nfull = K / SP_NB;
    CG[0].lazy = LazyCov{at<double>(ws, L.theta), t_dev, stars_dev, at<double>(ws, L.A), K, covpts,
                         temporal, lazy_nfull, 0, 0};
  }
  for (int g = 0; g < G; ++g) {
    const int s0 = first[g];
    int rc = lnlike_assemble(h, LG[g], ws, K, M, t_dev + (size_t)s0 * K,
                             flux_dev + (size_t)s0 * M * K,
                             diag_dev ? diag_dev + (size_t)s0 * K : nullptr, stars_dev + s0,
                             conditional, covpts, tab_dev, meanvar_dev, rta1_dev, temporal,
                             normalized, norm_order, zmax, CG[g].st, lazy_nfull);
    if (rc) return rc;
  }
  {
    int rc = sp_launch_cholesky_groups(h, G, CG.data(), K, L.Kp);
    if (rc) return rc;
  }
  for (int g = 0; g < G; ++g) {
    const int s0 = first[g];
    int rc = fused_reduce ? SP_OK
                          : lnlike_finish(LG[g], ws, K, M, lnlike_dev + s0,
                                          status_dev ? status_dev + s0 : nullptr, CG[g].st, stars_dev + s0,
                                          normalized && h->defer_norm, diag_dev != nullptr);
    if (rc) return rc;
    if (g > 0) {
      SP_HIP(hipEventRecord(h->gdone[g - 1], CG[g].st));
      SP_HIP(hipStreamWaitEvent(st, h->gdone[g - 1], 0));
    }
  }
  return SP_OK;
}

// The per-sample call on planned data (sp_plan.hip): sp_lnlike_ensemble's marginal, normalised branch with the
// pre-pass over the covariance's entries gone -- one assembly launch (only the tiles the factorisation wants in
// memory, the normalisation's coefficients from the table and the plan's weights), then the factorisation.
int sp_lnlike_ensemble_planned(sp_handle *h, const sp_plan *plan, const double *t_dev, const double *flux_dev,
                               const double *diag_dev, const sp_star *stars_dev, const double *tab_dev,
                               const double *meanvar_dev, int norm_order, double zmax, void *workspace_dev,
                               double *lnlike_dev, uint32_t *status_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !plan || !stars_dev || !tab_dev || !meanvar_dev || !workspace_dev || !lnlike_dev ||
      norm_order < 0 || norm_order > SP_NORM_MAXORDER)
    return SP_ERR_INVALID;
  if (plan->device != h->device) return SP_ERR_INVALID;
  // The data are the plan's: no data pointers at all = the planned ones (a replica's own copies, sp_plan_replicate);
  // pointers given must BE the planned ones -- the plan's phases, weights and sums come from the arrays of plan time,
  // residual rows and variances from these: another tensor of the same shape would give a finite, wrong value.
  if (!t_dev && !flux_dev && !diag_dev) {
    t_dev = plan->t;
    flux_dev = plan->flux;
    diag_dev = plan->diag;
  } else if (t_dev != plan->t || flux_dev != plan->flux || diag_dev != plan->diag) {
    return SP_ERR_INVALID;
  }
  const int S = plan->S, K = plan->K, M = plan->M, covpts = plan->covpts, temporal = plan->temporal;
  if (h->xp_covpts != covpts) return SP_ERR_STATE;
  hipStream_t st = (hipStream_t)stream;
  // Short light curves: the whole evaluation of a star in one workgroup's LDS (sp_small.hip) -- no system in memory, no
  // panel launches.  SP_SMALL_K=0: the blocked path at every size.
  if (sp_small_k_on() && sp_small_k_serves(K, M, covpts, diag_dev != nullptr))
    return sp_launch_small_lnlike(S, K, M, plan->dev, t_dev, stars_dev, covpts, tab_dev, meanvar_dev, temporal, flux_dev,
                                  diag_dev, norm_order, zmax, lnlike_dev, status_dev, st);
  Layout L = make_layout(h, S, K, M, true);
  void *ws = workspace_dev;
  // (the stars' packed tables for the kernels that form tiles at first touch: the design-matrix region)
  double *ptab = 4 * (size_t)(covpts + 4) > (size_t)L.Kr * L.N ? nullptr : at<double>(ws, L.A);
  // Tiles formed at first touch.  Without a temporal kernel: everything below the diagonal (the panel launches form
  // the first super-panel's block columns, the first trailing update the rest).  With one: the trailing update's
  // tiles only -- an exponential per entry has no place in the panel kernel's register budget, so the assembly writes
  // the first super-panel's block columns; every entry is still evaluated once.
  int lazy_nfull = 0, ncolw = 0, no_panels = 0;
  // (the row tiles that hold riding rows -- residuals, ones, variances -- are formed at first touch as well: every row
  //  tile is formable, the assembly writes the diagonal tiles only; SP_PLAN_RIDING_LAZY=0: it writes those row tiles)
  static const bool riding_env = !(getenv("SP_PLAN_RIDING_LAZY") && atoi(getenv("SP_PLAN_RIDING_LAZY")) == 0);
  bool riding = false;
  const int nrid = M + (diag_dev ? 2 : 1);
  double *rid = at<double>(ws, L.B1);
  if (h->lazy_cov && ptab && K / SP_NB >= 2 && 4 * (covpts + 4) + 64 <= SP_TILE_LDS_MIN) {
    // (the riding rows as the assembly would write them, [S][nrid][K], in the second design-matrix buffer)
    riding = riding_env && ((size_t)nrid + 1) * K <= (size_t)L.Kr * L.N;
    lazy_nfull = riding ? L.Kp / SP_NB : K / SP_NB;
    // (SP_PLAN_PANEL_LAZY=0: without a temporal kernel too, the panel launches load their tiles and only the first
    //  trailing update forms its own -- measured, not the default: DESIGN.md 4.11)
    static const bool panel_lazy = !(getenv("SP_PLAN_PANEL_LAZY") && atoi(getenv("SP_PLAN_PANEL_LAZY")) == 0);
    if (temporal != SP_TEMPORAL_NONE || !panel_lazy) {
      static const bool tl = !(getenv("SP_PLAN_TEMPORAL_LAZY") && atoi(getenv("SP_PLAN_TEMPORAL_LAZY")) == 0);
      ncolw = sp_superpanel_width(h, K);
      no_panels = 1;
      if ((!tl && temporal != SP_TEMPORAL_NONE) || ncolw * SP_NB >= K) lazy_nfull = ncolw = no_panels = 0;   // (one super-panel: no trailing update)
    }
  }
  // The diagonal tiles beyond the first super-panel's reach (those the eager updates of its launches do not touch) are
  // formed by the first trailing update too (LazyCov.dlazy bit 1), when that update runs on the kernel that can
  // (sp_syrk_can_form_diag) and everything else of their strips is left to its first touch; SP_PLAN_DIAG_LAZY=0: the
  // assembly writes them.
  static const bool dlazy_env = !(getenv("SP_PLAN_DIAG_LAZY") && atoi(getenv("SP_PLAN_DIAG_LAZY")) == 0);
  const int ntr = L.Kp / SP_NB, wsp = sp_superpanel_width(h, K), nsteps = (K + SP_NB - 1) / SP_NB;
  int dlazy = 0, dfrom = ntr;
  if (dlazy_env && riding && lazy_nfull == ntr && wsp * SP_NB < K && sp_syrk_can_form_diag(ntr - wsp) &&
      temporal == SP_TEMPORAL_NONE) {
    const int last = wsp < nsteps - 1 ? wsp : nsteps - 1;     // (cholesky_panel2: row tiles i <= last keep their diagonal tile up to date)
    dlazy = 2;
    dfrom = last + 1;
  }
  // (pivot block 0 is factored by the assembly's workgroup of tile (0, 0): no launch of its own; SP_PLAN_FUSE0=0 for
  //  the separate launch)
  static const bool fuse0_env = !(getenv("SP_PLAN_FUSE0") && atoi(getenv("SP_PLAN_FUSE0")) == 0);
  const int fuse0 = (fuse0_env && K >= SP_NB) ? 1 : 0;
  int rc = sp_launch_assemble_planned(S, K, M, L.Kp, plan->dev, t_dev, stars_dev, covpts, tab_dev, meanvar_dev, temporal,
                                      flux_dev, diag_dev, at<double>(ws, L.sys), lazy_nfull, ncolw, norm_order, zmax,
                                      at<double>(ws, L.coef), at<double>(ws, L.rscal), ptab, at<int32_t>(ws, L.info),
                                      at<uint32_t>(ws, L.status), st, at<double>(ws, L.invL), sp_lt_stride(L.Kp), fuse0,
                                      riding ? rid : nullptr, dfrom);
  if (rc) return rc;
  const bool fused_reduce = sp_panel_fuses_reduce(h, K, L.Kp);
  sp_chol_group G{at<double>(ws, L.sys), at<int32_t>(ws, L.info), at<double>(ws, L.invL), S, st, LazyCov{}, SpReduceArgs{}};
  if (fused_reduce)
    G.red = SpReduceArgs{lnlike_dev, at<uint32_t>(ws, L.status), status_dev, stars_dev, (const void *)at<double>(ws, L.coef),
                         at<double>(ws, L.rscal), diag_dev ? 1 : 0, K, M, K + M + (diag_dev ? 2 : 1)};
  G.block0_done = fuse0 != 0;
  (void)dlazy;
  if (lazy_nfull)
    G.lazy = LazyCov{plan->dev.theta, t_dev, stars_dev, ptab, K, covpts, temporal, lazy_nfull, 0, 0, no_panels ? 0 : 1,
                     no_panels, riding ? rid : nullptr, riding ? nrid : 0, dlazy, plan->dev.inorder};
  if ((rc = sp_launch_cholesky_groups(h, 1, &G, K, L.Kp))) return rc;
  if (!fused_reduce) return lnlike_finish(L, ws, K, M, lnlike_dev, status_dev, st, stars_dev, true, diag_dev != nullptr);
  return SP_OK;
}

int sp_gemm_nt(sp_handle *h, const double *A_dev, long lda, long strideA, const double *B_dev,
               long ldb, long strideB, double *C_dev, long ldc, long strideC, int M, int N,
               int K, double alpha, int beta, int lower_only, int batch, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !A_dev || !B_dev || !C_dev || M < 0 || N < 0 || K < 0 || batch < 0 || lda < K ||
      ldb < K || ldc < N || (beta != 0 && beta != 1))
    return SP_ERR_INVALID;
  return sp_launch_gemm_nt(A_dev, lda, strideA, B_dev, ldb, strideB, C_dev, ldc, strideC, M, N, K,
                           alpha, beta, lower_only, batch, (hipStream_t)stream);
}

int sp_gp_condition(sp_handle *h, int K, int Ks, const double *Ktt_dev, const double *Kst_dev,
                    double *Kss_dev, const double *r_dev, double *mu_dev, int32_t *info_dev,
                    void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || K < 1 || Ks < 1 || !Ktt_dev || !Kst_dev || !Kss_dev || !r_dev || !mu_dev)
    return SP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int M = Ks + 1, Kp = sp_roundup(K + M, SP_NB);
  const size_t sysb = align_up(sizeof(double) * (size_t)Kp * Kp);
  const size_t resb = align_up(sizeof(double) * (size_t)M * K);
  const size_t invb = align_up(sizeof(double) * sp_lt_stride(Kp));
  void *ws = nullptr;
  int rc = ensure_big(h, sysb + resb + invb + 256, &ws);
  if (rc) return rc;
  double *sys = at<double>(ws, 0), *res = at<double>(ws, sysb);
  double *lt = at<double>(ws, sysb + resb);
  int32_t *info = at<int32_t>(ws, sysb + resb + invb);
  SP_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), st));
  SP_HIP(hipMemcpyAsync(res, Kst_dev, sizeof(double) * (size_t)Ks * K, hipMemcpyDeviceToDevice, st));
  SP_HIP(hipMemcpyAsync(res + (size_t)Ks * K, r_dev, sizeof(double) * K, hipMemcpyDeviceToDevice,
                        st));
  if ((rc = sp_launch_pad_in(Ktt_dev, K, K, (long)K * K, sys, Kp, M, res, 1, st))) return rc;
  if ((rc = sp_launch_cholesky_systems(h, sys, 1, K, Kp, info, lt, st))) return rc;
  const double *Y = sys + (size_t)K * Kp;          // [Ks, K], row stride Kp
  const double *w = sys + (size_t)(K + Ks) * Kp;   // [1, K]
  if ((rc = sp_launch_gemm_nt(Y, Kp, 0, w, Kp, 0, mu_dev, 1, 0, Ks, 1, K, 1.0, 0, 0, 1, st)))
    return rc;
  if ((rc = sp_launch_gemm_nt(Y, Kp, 0, Y, Kp, 0, Kss_dev, Ks, 0, Ks, Ks, K, -1.0, 1, 0, 1, st)))
    return rc;
  if (info_dev)
    SP_HIP(hipMemcpyAsync(info_dev, info, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return SP_OK;
}

// The one collective of the path.  RCCL is resolved in the running
// process: the communicator belongs to the caller, so must the library.
int sp_allgather_lnlike(sp_handle *h, void *nccl_comm, const double *local_dev, int count,
                        double *all_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !nccl_comm || !local_dev || !all_dev || count < 0) return SP_ERR_INVALID;
  if (count == 0) return SP_OK;
  typedef int (*allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
  static allgather_fn fn = nullptr;
  if (!fn) fn = reinterpret_cast<allgather_fn>(dlsym(RTLD_DEFAULT, "ncclAllGather"));
  if (!fn) return SP_ERR_COMM;
  const int nccl_float64 = 8;  // ncclFloat64 / ncclDouble
  return fn(local_dev, all_dev, (size_t)count, nccl_float64, nccl_comm, (hipStream_t)stream) == 0
             ? SP_OK
             : SP_ERR_COMM;
}

size_t sp_spd_inverse_workspace_bytes(sp_handle *h, int S, int K) {
  if (!h || S < 0 || K < 1) return 0;
  return make_layout(h, S, K, sp_roundup(K, SP_NB), true, true).total;
}

// C^-1 and log det C of S symmetric positive definite K x K matrices with the factorisation's own machinery:
// the identity rides through the blocked Cholesky as rows below the matrix (a row r below becomes
// (L^-1 r)^T, so the identity becomes Y = L^-T), then C^-1 = Y Y^T on the matrix cores.  Y is upper triangular:
// a launch of the factorisation only takes the identity's row tiles that hold something yet, the trailing updates
// leave the columns without pivots alone, and the product of tile (ti, tj) starts at column 64 ti --
// K^3 (1/3 + 1/2 + 1/3) flops, against K^3 (1/3 + 1 + 1) without the structure.
// the inverse of the matrices ALREADY in the top-left K x K corners of the systems of `ws` (lower triangles)
static int spd_inverse_in_place(sp_handle *h, int S, int K, const Layout &L, void *ws, double *Cinv_dev,
                                double *logdet_dev, hipStream_t st) {
  const int Kr = sp_roundup(K, SP_NB);
  double *sys = at<double>(ws, L.sys);
  int32_t *info = at<int32_t>(ws, L.info);
  const long ld = L.Kp, stride = (long)L.Kp * L.Kp;
  int rc;
  SP_HIP(hipMemsetAsync(info, 0, sizeof(int32_t) * S, st));
  hipLaunchKernelGGL(ident_rows_kernel, dim3((K + Kr + 3) / 4, S), dim3(256), 0, st, sys, ld, stride, K, Kr);
  SP_LAUNCH_CHECK();
  sp_chol_group g{sys, info, at<double>(ws, L.invL), S, st, LazyCov{}, SpReduceArgs{}, K};
  if ((rc = sp_launch_cholesky_groups(h, 1, &g, K, L.Kp))) return rc;
  if (logdet_dev) {
    hipLaunchKernelGGL(logdet_kernel, dim3(S), dim3(256), 0, st, sys, ld, stride, K, info, logdet_dev);
    SP_LAUNCH_CHECK();
  }
  if (Kr > K) {
    const long n = (long)Kr * (Kr - K);
    hipLaunchKernelGGL(zero_cols_kernel, dim3((unsigned)((n + 255) / 256), S), dim3(256), 0, st, sys, ld, stride, K, Kr,
                       K, Kr);
    SP_LAUNCH_CHECK();
  }
  // C^-1 = Y Y^T, lower 64 x 64 tiles, into [S, Kr, Kr]
  const double *Y = sys + (size_t)K * ld;
  return sp_launch_gemm_nt(Y, ld, stride, Y, ld, stride, Cinv_dev, Kr, (long)Kr * Kr, Kr, Kr, Kr, 1.0, 0, 1, S, st, 2,
                           nullptr);
}

int sp_spd_inverse_batched(sp_handle *h, int S, int K, const double *C_dev, long ldc, long strideC,
                           double *Cinv_dev, double *logdet_dev, int32_t *info_dev, void *workspace_dev,
                           void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !C_dev || !Cinv_dev || !workspace_dev || S < 0 || K < 1 || ldc < K) return SP_ERR_INVALID;
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  const int Kr = sp_roundup(K, SP_NB);
  Layout L = make_layout(h, S, K, Kr, true, true);
  void *ws = workspace_dev;
  // the matrices into the systems' corners (nothing else of the systems is touched here)
  hipLaunchKernelGGL(corner_copy_kernel, dim3((K + 255) / 256, K, S), dim3(256), 0, st, C_dev, ldc, strideC,
                     at<double>(ws, L.sys), (long)L.Kp, (long)L.Kp * L.Kp, K);
  SP_LAUNCH_CHECK();
  int rc = spd_inverse_in_place(h, S, K, L, ws, Cinv_dev, logdet_dev, st);
  if (rc) return rc;
  if (info_dev)
    SP_HIP(hipMemcpyAsync(info_dev, at<int32_t>(ws, L.info), sizeof(int32_t) * S, hipMemcpyDeviceToDevice, st));
  return SP_OK;
}

// ---- the ensemble gradient's device half (sp_grad.hip; grad.py chains the table to the hyperparameters) ----
namespace {
struct GradLayout {
  size_t inv, cinv, vec, dots, hcoef, logdet, partial, total;
};
GradLayout grad_layout(sp_handle *h, int S, int K, int M, int covpts) {
  const int Kr = sp_roundup(K, SP_NB);
  GradLayout G;
  size_t off = 0;
  auto take = [&](size_t b) {
    size_t o = off;
    off += align_up(b);
    return o;
  };
  const size_t d = sizeof(double);
  G.inv = take(make_layout(h, S, K, Kr, true, true).total);
  G.cinv = take(d * (size_t)S * Kr * Kr);
  G.vec = take(d * (size_t)S * (M + 3) * K);       // C^-1 [p, q, 1, r_0 .. r_{M-1}]
  G.dots = take(d * (size_t)S * M * 2);
  G.hcoef = take(d * S);
  G.logdet = take(d * S);
  {
    // the scatter's bins per lower tile; before that, the row parts of the products with C^-1 ([S][ntr][4][K])
    const size_t ntr = Kr / SP_NB, bins = ntr * (ntr + 1) / 2 * (covpts + 4), rows = ntr * 4 * (size_t)K;
    G.partial = take(d * (size_t)S * (bins > rows ? bins : rows));
  }
  G.total = off;
  return G;
}
}  // namespace

size_t sp_lnlike_grad_workspace_bytes_multi(sp_handle *h, int S, int K, int M, int covpts) {
  if (!h || S < 0 || K < 2 || M < 1 || covpts < 1) return 0;
  return grad_layout(h, S, K, M, covpts).total;
}
size_t sp_lnlike_grad_workspace_bytes(sp_handle *h, int S, int K, int covpts) {
  return sp_lnlike_grad_workspace_bytes_multi(h, S, K, 1, covpts);
}

int sp_lnlike_grad_marginal(sp_handle *h, int S, int K, const double *t_dev, const double *flux_dev,
                            const double *diag_dev, const sp_star *stars_dev, int covpts, const double *tab_dev,
                            const double *meanvar_dev, int temporal, int normalized, int norm_order, double zmax,
                            void *workspace_dev, double *lnlike_dev, double *ybar_dev, double *meanbar_dev,
                            uint32_t *status_dev, void *stream) {
  return sp_lnlike_grad_marginal_multi(h, S, K, 1, t_dev, flux_dev, diag_dev, stars_dev, covpts, tab_dev, meanvar_dev,
                                       temporal, normalized, norm_order, zmax, workspace_dev, lnlike_dev, ybar_dev,
                                       meanbar_dev, status_dev, stream);
}

int sp_lnlike_grad_marginal_multi(sp_handle *h, int S, int K, int M, const double *t_dev, const double *flux_dev,
                                  const double *diag_dev, const sp_star *stars_dev, int covpts, const double *tab_dev,
                                  const double *meanvar_dev, int temporal, int normalized, int norm_order, double zmax,
                                  void *workspace_dev, double *lnlike_dev, double *ybar_dev, double *meanbar_dev,
                                  uint32_t *status_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !t_dev || !flux_dev || !stars_dev || !tab_dev || !meanvar_dev || !workspace_dev || !lnlike_dev ||
      !ybar_dev || !meanbar_dev || S < 0 || K < 2 || M < 1 || covpts < 1 || norm_order < 0 ||
      norm_order > SP_NORM_MAXORDER)
    return SP_ERR_INVALID;
  if (h->xp_covpts != covpts) return SP_ERR_STATE;
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  const int Kr = sp_roundup(K, SP_NB);
  const GradLayout G = grad_layout(h, S, K, M, covpts);
  char *base = static_cast<char *>(workspace_dev);
  void *ws = base + G.inv;
  Layout L = make_layout(h, S, K, Kr, true, true);
  double *theta = at<double>(ws, L.theta), *rowsum = at<double>(ws, L.rowsum), *qv = at<double>(ws, L.qv);
  double *coef = at<double>(ws, L.coef), *sys = at<double>(ws, L.sys);
  int32_t *info = at<int32_t>(ws, L.info);
  double *Cinv = reinterpret_cast<double *>(base + G.cinv), *vec = reinterpret_cast<double *>(base + G.vec);
  double *hcoef = reinterpret_cast<double *>(base + G.hcoef), *logdet = reinterpret_cast<double *>(base + G.logdet);
  double *partial = reinterpret_cast<double *>(base + G.partial);
  int rc;
  // the covariance as the likelihood sees it, K x K (direct normalisation: sp.py:705-727, 1135-1151)
  if ((rc = sp_launch_theta(S, K, t_dev, stars_dev, theta, st))) return rc;
  if (normalized)
    if ((rc = sp_launch_rowsum(S, K, theta, t_dev, stars_dev, covpts, tab_dev, meanvar_dev, h->d_xp, temporal, nullptr,
                               rowsum, st)))
      return rc;
  if ((rc = sp_launch_norm_coef(S, K, stars_dev, meanvar_dev, nullptr, normalized, norm_order, zmax, rowsum, qv, coef,
                                nullptr, st)))
    return rc;
  // (straight into the corner of the system the inverse factors: leading dimension Kp)
  // (the LOWER tiles of the Kr x Kr corner: the system form of the assembly with no rows below the matrix)
  if ((rc = sp_launch_assemble(S, K, 0, Kr, 1, theta, t_dev, stars_dev, covpts, tab_dev, meanvar_dev, h->d_xp, temporal,
                               nullptr, normalized, qv, coef, diag_dev, 1, nullptr, sys, L.Kp, (long)L.Kp * L.Kp, st)))
    return rc;
  if ((rc = spd_inverse_in_place(h, S, K, L, ws, Cinv, logdet, st))) return rc;
  return sp_launch_grad_sweep(S, K, Kr, M, Cinv, theta, t_dev, flux_dev, stars_dev, coef, qv, diag_dev, logdet, info,
                              covpts, temporal, normalized, norm_order, zmax, vec,
                              reinterpret_cast<double *>(base + G.dots), hcoef, partial, lnlike_dev, ybar_dev,
                              meanbar_dev, status_dev, st);
}

int sp_cholesky_lnlike_batched(sp_handle *h, int S, int K, int M,
                               const double *cov_dev, const double *resid_dev,
                               void *workspace_dev, double *lnlike_dev,
                               uint32_t *status_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !cov_dev || !resid_dev || !workspace_dev || !lnlike_dev || S < 0 ||
      K < 1 || M < 1)
    return SP_ERR_INVALID;
  if (S == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  Layout L = make_layout(h, S, K, M, true);
  void *ws = workspace_dev;
  double *sys = at<double>(workspace_dev, L.sys);
  int32_t *info = at<int32_t>(workspace_dev, L.info);
  uint32_t *status = at<uint32_t>(workspace_dev, L.status);
  int rc;
  SP_HIP(hipMemsetAsync(info, 0, sizeof(int32_t) * S, st));
  SP_HIP(hipMemsetAsync(status, 0, sizeof(uint32_t) * S, st));
  if ((rc = sp_launch_pad_in(cov_dev, K, K, (long)K * K, sys, L.Kp, M, resid_dev, S, st)))
    return rc;
  if ((rc = sp_launch_cholesky_systems(h, sys, S, K, L.Kp, info, at<double>(ws, L.invL), st))) return rc;
  if ((rc = sp_launch_lnlike_reduce(sys, S, K, M, L.Kp, info, lnlike_dev, status, st)))
    return rc;
  if (status_dev)
    SP_HIP(hipMemcpyAsync(status_dev, status, sizeof(uint32_t) * S,
                          hipMemcpyDeviceToDevice, st));
  return SP_OK;
}

}  // extern "C"
