// The data plan of the likelihood step (round 5): everything the per-sample call used to recompute that depends on
// the DATA alone -- what calibrate.get_log_prob(t, flux, ferr, p, ...) fixes when the log-probability is built
// (calibrate/log_prob.py:7-55).
//
//   theta     the cadences' phases 2 pi mod(t / p, 1)                                   flux.py:262
//   wbar      the weight of every kernel-table entry in the SUM of the covariance:
//             cov_ij = spline(|theta_i - theta_j|) T_ij is linear in the table,
//                 cov_ij = sum_{k < 4} yp[s_ij + k] b_k(x0_ij) T_ij                     flux.py:256-276, 322-330
//             (s_ij the int64 segment index, x0_ij the position inside the segment, b_k the cubic's weights of
//             the four grid values, T_ij the temporal factor, temporal.py:8-16), so that
//                 sum_ij cov_ij = sum_n yp[n] wbar[n],   wbar[n] = sum_ij [s_ij + k = n] b_k(x0_ij) T_ij
//             and the normalisation's m = mean(Sigma) (sp.py:705-727) costs covpts + 4 multiply-adds per sample
//             instead of a pass over the K^2 entries;
//   sflux, sdv   sums of each light curve's flux and of the per-cadence variances over the valid cadences (what
//             the reduction needs of q = Sigma 1 / (K m) without forming it, sp_reduce.h);
//   key       period, tau, nobs as planned (a call with other values gets NaN and SP_STAR_STALE_PLAN).
//
// Compiled with -ffp-contract=off: the segment index must be the reference's int64 floor(x / dx) bit for bit.
#include <new>

#include "sp_internal.h"
#include "sp_cov.h"

int sp_launch_theta(int S, int K, const double *t, const sp_star *stars, double *theta, hipStream_t st,
                    int32_t *info = nullptr, uint32_t *status = nullptr, const double *tab = nullptr,
                    int covpts = 0, double *ptab = nullptr);

namespace {

// One workgroup per LOWER 64 x 64 tile of a star (the covariance is symmetric: tiles below the diagonal count
// twice).  Thread (lane, wave): column i = 64 tb + lane, rows j = 64 ta + wave, + 4, ...  The bins live in LDS, one
// set PER WAVEFRONT: a wavefront's adds reach its own bins in program order, and lanes of one instruction that hit
// the same bin are served by the LDS in a fixed order -- the partial table of a tile is the same bits every time
// (two wavefronts racing for one set of bins would not be); the four sets and then the tiles' partial tables are
// added in a fixed order (plan_wbar_reduce_kernel).
template <int TK>
__global__ __launch_bounds__(256) void plan_wbar_kernel(
    int K, const double *__restrict__ theta, const double *__restrict__ t, const sp_star *__restrict__ stars,
    int covpts, double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double bins[];   // [4][np]
  const int s = blockIdx.y, np = covpts + 4, tid = threadIdx.x, wave = tid >> 6;
  for (int k = tid; k < 4 * np; k += 256) bins[k] = 0.0;
  __syncthreads();
  const int tile = blockIdx.x;
  int ta = (int)((sqrtf(8.0f * tile + 1.0f) - 1.0f) * 0.5f);     // row tile (ta >= tb)
  while (ta * (ta + 1) / 2 > tile) --ta;
  while ((ta + 1) * (ta + 2) / 2 <= tile) ++ta;
  const int tb = tile - ta * (ta + 1) / 2;
  const int i = tb * 64 + (tid & 63);
  const sp_star st = stars[s];
  const int nobs = star_nobs(st, K);
  double *mine = bins + wave * np;
  if (i < nobs) {
    const double thi = theta[(size_t)s * K + i];
    const double ti = TK != SP_TEMPORAL_NONE ? t[(size_t)s * K + i] : 0.0;
    const double dx = 6.283185307179586 / covpts, inv_dx = 1.0 / dx;
    const double mult = ta > tb ? 2.0 : 1.0;
    const int jend = ta * 64 + 64 < nobs ? ta * 64 + 64 : nobs;
    for (int j = ta * 64 + wave; j < jend; j += 4) {
      const double T = mult * temporal_factor(TK, ti, TK != SP_TEMPORAL_NONE ? t[(size_t)s * K + j] : 0.0, st.tau);
      // the segment of the lag and the position inside it: SplineGen's index (sp_cov.h; flux.py:262-265)
      int idx;
      double x;
      {
        const double lag = fabs(thi - theta[(size_t)s * K + j]);
        const double qd = lag * inv_dx;
        idx = (int)qd;
        x = qd - (double)idx;
        if (fabs(x - 0.5) > 0.5 - 1.0e-9) {
          idx = (int)floor(lag / dx);
          x = qd - (double)idx;
        }
        idx = idx < 0 ? 0 : (idx > covpts ? covpts : idx);
      }
      // value = sum_k yp[idx + k] b_k(x):  a0 = y1, a1 = -y0/3 - y1/2 + y2 - y3/6, a2 = (y0 + y2)/2 - y1,
      // a3 = ((y1 - y2) + (y3 - y0)/3)/2   (flux.py:322-330)
      const double x2 = x * x, x3 = x2 * x;
      atomicAdd(&mine[idx], T * (-x / 3.0 + 0.5 * x2 - x3 / 6.0));
      atomicAdd(&mine[idx + 1], T * (1.0 - 0.5 * x - x2 + 0.5 * x3));
      atomicAdd(&mine[idx + 2], T * (x + 0.5 * x2 - 0.5 * x3));
      atomicAdd(&mine[idx + 3], T * (-x / 6.0 + x3 / 6.0));
    }
  }
  __syncthreads();
  double *P = partial + ((size_t)s * gridDim.x + blockIdx.x) * np;
  for (int k = tid; k < np; k += 256) P[k] = (bins[k] + bins[np + k]) + (bins[2 * np + k] + bins[3 * np + k]);
}

// wbar[s][k] = sum over the tiles' partial tables, in a fixed order: thread (k, g) adds the partials w = g, g + 4, ...,
// the four groups are added in order.  grid (ceil(np / 64), S)
__global__ __launch_bounds__(256) void plan_wbar_reduce_kernel(int np, int nwg, const double *__restrict__ partial,
                                                               double *__restrict__ wbar) {
  __shared__ double red[4][64];
  const int s = blockIdx.y, k = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  double a = 0.0;
  if (k < np) {
#pragma unroll 4
    for (int w = g; w < nwg; w += 4) a += partial[((size_t)s * nwg + w) * np + k];
  }
  red[g][threadIdx.x & 63] = a;
  __syncthreads();
  if (g == 0 && k < np)
    wbar[(size_t)s * np + k] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// sums of the data and the planned fields of the stars: one workgroup per star
__global__ __launch_bounds__(256) void plan_scalars_kernel(
    int K, int M, const double *__restrict__ flux, const double *__restrict__ diag,
    const sp_star *__restrict__ stars, const double *__restrict__ t, double *__restrict__ sflux,
    double *__restrict__ sdv, double *__restrict__ key, double *__restrict__ inorder) {
  __shared__ double red[4];
  const int s = blockIdx.x, tid = threadIdx.x;
  const sp_star st = stars[s];
  const int nobs = star_nobs(st, K);
  auto block_sum = [&](double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  for (int m = 0; m <= M; ++m) {
    // (m == M: the variances)
    const double *src = m < M ? flux + ((size_t)s * M + m) * K : (diag ? diag + (size_t)s * K : nullptr);
    double a = 0.0;
    if (src)
      for (int i = tid; i < nobs; i += 256) a += src[i];
    const double total = block_sum(a);
    if (tid == 0) {
      if (m < M) sflux[(size_t)s * M + m] = total;
      else sdv[s] = total;
    }
  }
  // are the cadences in order?  (what lets a Matern factor below the diagonal separate: sp_cov.h, lazy_cov_tiles)
  int unsorted = 0;
  for (int i = tid; i + 1 < nobs; i += 256)
    if (!(t[(size_t)s * K + i + 1] >= t[(size_t)s * K + i])) unsorted = 1;
  unsorted = __syncthreads_or(unsorted);
  if (tid == 0) {
    inorder[s] = unsorted ? 0.0 : 1.0;
    key[3 * s] = st.period;
    key[3 * s + 1] = st.tau;
    key[3 * s + 2] = (double)nobs;
  }
}

}  // namespace

extern "C" {

int sp_plan_data(sp_handle *h, int S, int K, int M, const double *t_dev, const double *flux_dev,
                 const double *diag_dev, const sp_star *stars_dev, int covpts, int temporal, void *workspace_dev,
                 void *stream, sp_plan **out) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !out || !t_dev || !flux_dev || !stars_dev || !workspace_dev || S < 1 || K < 2 || M < 1 || covpts < 1)
    return SP_ERR_INVALID;
  if (temporal != SP_TEMPORAL_NONE && temporal != SP_TEMPORAL_MATERN32 && temporal != SP_TEMPORAL_EXPSQUARED)
    return SP_ERR_INVALID;
  *out = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int np = covpts + 4;
  const int ntr = (K + 63) / 64, ntl = ntr * (ntr + 1) / 2;
  // the tiles' partial tables go to the systems' region of the likelihood workspace: [S][ntl][np] doubles
  const long ws_bytes = sp_lnlike_workspace_bytes(h, S, K, M);
  const size_t Kp = (size_t)sp_roundup(K + M, SP_NB);
  if (ws_bytes < 0 || (size_t)ntl * np > Kp * Kp) return SP_ERR_INVALID;
  const size_t lds = sizeof(double) * 4 * (size_t)np;
  if (lds > 150 * 1024) return SP_ERR_INVALID;
  // (the systems are the last region of the workspace: S Kp'^2 doubles with Kp' >= Kp)
  double *partial = reinterpret_cast<double *>(static_cast<char *>(workspace_dev) + (size_t)ws_bytes) - (size_t)S * Kp * Kp;

  // (the plan's buffer lives on the HANDLE's GPU whatever the caller's current device is)
  SP_HIP(hipSetDevice(h->device));
  sp_plan *p = new (std::nothrow) sp_plan();
  if (!p) return SP_ERR_ALLOC;
  p->device = h->device;
  p->S = S; p->K = K; p->M = M; p->covpts = covpts; p->temporal = temporal; p->has_diag = diag_dev != nullptr;
  p->t = t_dev; p->flux = flux_dev; p->diag = diag_dev; p->nrep = 0;
  const size_t d = sizeof(double);
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_theta = 0, o_wbar = o_theta + up(d * S * K), o_sflux = o_wbar + up(d * S * np),
               o_sdv = o_sflux + up(d * S * M), o_key = o_sdv + up(d * S), o_ord = o_key + up(d * S * 3),
               total = o_ord + up(d * S);
  p->bytes = total;
  hipError_t e = hipMalloc(&p->buf, total);
  if (e != hipSuccess) {
    sp_set_hip_error(e, "hipMalloc(plan)");
    delete p;
    return SP_ERR_ALLOC;
  }
  char *base = static_cast<char *>(p->buf);
  double *theta = reinterpret_cast<double *>(base + o_theta), *wbar = reinterpret_cast<double *>(base + o_wbar);
  double *sflux = reinterpret_cast<double *>(base + o_sflux), *sdv = reinterpret_cast<double *>(base + o_sdv);
  double *key = reinterpret_cast<double *>(base + o_key), *inorder = reinterpret_cast<double *>(base + o_ord);
  p->dev = PlanDev{theta, wbar, sflux, sdv, key, inorder};
  auto fail = [&](int rc) {
    (void)hipFree(p->buf);
    delete p;
    return rc;
  };
  int rc = sp_launch_theta(S, K, t_dev, stars_dev, theta, st);
  if (rc) return fail(rc);
  hipLaunchKernelGGL(plan_scalars_kernel, dim3(S), dim3(256), 0, st, K, M, flux_dev, diag_dev, stars_dev, t_dev, sflux,
                     sdv, key, inorder);
  dim3 grid(ntl, S);
#define SP_PLAN_WBAR(TK)                                                                                          \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(plan_wbar_kernel<TK>),                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                            \
    hipLaunchKernelGGL((plan_wbar_kernel<TK>), grid, dim3(256), lds, st, K, theta, t_dev, stars_dev, covpts, partial); \
  } while (0)
  if (temporal == SP_TEMPORAL_NONE) SP_PLAN_WBAR(SP_TEMPORAL_NONE);
  else if (temporal == SP_TEMPORAL_MATERN32) SP_PLAN_WBAR(SP_TEMPORAL_MATERN32);
  else SP_PLAN_WBAR(SP_TEMPORAL_EXPSQUARED);
#undef SP_PLAN_WBAR
  hipLaunchKernelGGL(plan_wbar_reduce_kernel, dim3((np + 63) / 64, S), dim3(256), 0, st, np, ntl, partial, wbar);
  e = hipGetLastError();
  if (e != hipSuccess) {
    sp_set_hip_error(e, "kernel launch (plan)");
    return fail(SP_ERR_HIP);
  }
  // (the partial tables live in the caller's workspace: done with it before the call returns)
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    sp_set_hip_error(e, "hipStreamSynchronize(plan)");
    return fail(SP_ERR_HIP);
  }
  *out = p;
  return SP_OK;
}

// B copies of a planned data set as ONE batch of B S systems (round 6: hyperparameter samples batched into the planned
// call -- system b S + s is star s of the source under sample b).  The replica owns copies of everything the step reads
// by system index: the plan's arrays and the data (t, flux, variances).  A data INDEX in the kernels instead would cost
// the panel kernel's lazy instantiations scalar registers they do not have (106 of 106 SGPRs, sp_panel.hip); the copies
// are B (K (M + 2) + covpts + M + 9) doubles -- 1.6 MB for 64 samples of a K = 1000 light curve, against 0.5 GB of systems.
namespace {
__global__ __launch_bounds__(256) void replicate_kernel(const double *__restrict__ src, long n, double *__restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[(size_t)blockIdx.y * n + i] = src[i];
}
}  // namespace

int sp_plan_replicate(sp_handle *h, const sp_plan *src, int B, void *stream, sp_plan **out) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !src || !out || B < 1 || B > 65535 || src->device != h->device || (long)src->S * B > (1L << 24))
    return SP_ERR_INVALID;
  *out = nullptr;
  hipStream_t st = (hipStream_t)stream;
  SP_HIP(hipSetDevice(h->device));
  sp_plan *p = new (std::nothrow) sp_plan(*src);
  if (!p) return SP_ERR_ALLOC;
  const int S0 = src->S, K = src->K, M = src->M, np = src->covpts + 4;
  p->S = S0 * B;
  p->nrep = B;
  p->buf = nullptr;
  const size_t d = sizeof(double), S = (size_t)p->S;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_theta = 0, o_wbar = o_theta + up(d * S * K), o_sflux = o_wbar + up(d * S * np),
               o_sdv = o_sflux + up(d * S * M), o_key = o_sdv + up(d * S), o_ord = o_key + up(d * S * 3),
               o_t = o_ord + up(d * S), o_flux = o_t + up(d * S * K), o_diag = o_flux + up(d * S * M * K),
               total = o_diag + (src->has_diag ? up(d * S * K) : 0);
  p->bytes = total;
  hipError_t e = hipMalloc(&p->buf, total);
  if (e != hipSuccess) {
    sp_set_hip_error(e, "hipMalloc(plan replica)");
    delete p;
    return SP_ERR_ALLOC;
  }
  char *base = static_cast<char *>(p->buf);
  auto at = [&](size_t o) { return reinterpret_cast<double *>(base + o); };
  p->dev = PlanDev{at(o_theta), at(o_wbar), at(o_sflux), at(o_sdv), at(o_key), at(o_ord)};
  p->t = at(o_t);
  p->flux = at(o_flux);
  p->diag = src->has_diag ? at(o_diag) : nullptr;
  auto rep = [&](const double *from, size_t n, double *to) {
    hipLaunchKernelGGL(replicate_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, from, (long)n, to);
  };
  rep(src->dev.theta, (size_t)S0 * K, at(o_theta));
  rep(src->dev.wbar, (size_t)S0 * np, at(o_wbar));
  rep(src->dev.sflux, (size_t)S0 * M, at(o_sflux));
  rep(src->dev.sdv, (size_t)S0, at(o_sdv));
  rep(src->dev.key, (size_t)S0 * 3, at(o_key));
  rep(src->dev.inorder, (size_t)S0, at(o_ord));
  rep(src->t, (size_t)S0 * K, at(o_t));
  rep(src->flux, (size_t)S0 * M * K, at(o_flux));
  if (src->has_diag) rep(src->diag, (size_t)S0 * K, at(o_diag));
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    sp_set_hip_error(e, "sp_plan_replicate");
    (void)hipFree(p->buf);
    delete p;
    return SP_ERR_HIP;
  }
  *out = p;
  return SP_OK;
}

int sp_plan_systems(const sp_plan *plan) { return plan ? plan->S : SP_ERR_INVALID; }

void sp_plan_destroy(sp_plan *plan) {
  if (!plan) return;
  if (plan->buf) (void)hipFree(plan->buf);
  delete plan;
}

int sp_plan_get_wbar(const sp_plan *plan, double *wbar_host) {
  if (!plan || !wbar_host) return SP_ERR_INVALID;
  SP_HIP(hipMemcpy(wbar_host, plan->dev.wbar, sizeof(double) * (size_t)plan->S * (plan->covpts + 4),
                   hipMemcpyDeviceToHost));
  return SP_OK;
}

}  // extern "C"
