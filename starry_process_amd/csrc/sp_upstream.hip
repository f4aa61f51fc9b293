// Upstream of the hot path on the device, as ONE host call (the algorithm is
// starry_process_amd/upstream_device.py's, which documents it): hyperparameters -> (mu_y, Sigma_y)
// as expectations of rotated spot expansions (latitude.py:199-212, longitude.py:19-24,
// contrast.py:18-33), taken by exact quadrature of actual rotations,
//
//     mu_y    = sqrt(n) m1,        m1 = g sum_kq W_kq  Ry(lam_q) Rx(phi_k) s,      g = pi c sqrt(n)
//     Sigma_y = sum_kqj (g sqrt(W_kq) Ry Rx v_j)(.)^T - m1 m1^T + diag(eps),
//
// with Gauss-Jacobi nodes phi_k (both signs) and equispaced longitudes lam_q.  The host computes the
// size moments and the nodes (a few dozen numbers) and passes them here; everything else is
// enqueued natively -- one staged upload, ten launches -- where the Python composition of the same
// ops (Rx, dotRx, tensordotRz, gemm_nt through ctypes and torch) cost 0.48 ms of host time per
// hyperparameter sample, more than the device needs for the whole likelihood step behind it
// (VERDICT r02 item 7).
//
// Rotations: row vectors v^T R; Ry(lam) = Rx(pi/2) Rz(lam) Rx(-pi/2) (flux.py:88-105 uses the same
// decomposition); Rz by the cos / sin (m lam) table of the handle (wigner.h:289-339).
#include <cmath>
#include <cstring>

#include "sp_internal.h"

namespace {

// M0[k][j][:] = sw[k] vecs[j][:]
__global__ void up_outer_kernel(const double *__restrict__ sw, const double *__restrict__ vecs, int mv,
                                int N, double *__restrict__ out) {
  const int k = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < mv * N) out[(size_t)k * mv * N + e] = sw[k] * vecs[e];
}

// out[q R + r][n] = U[r][n] cos(m lam_q) + U[r][mirror n] sin(m lam_q)      (tensordotRz, wigner.h:289-339,
// on the rows of U repeated for every longitude)
__global__ __launch_bounds__(256) void up_rz_repeat_kernel(int N, int R, int nm,
                                                           const int32_t *__restrict__ m_of,
                                                           const int32_t *__restrict__ mirror,
                                                           const double *__restrict__ lamcs,
                                                           const double *__restrict__ U,
                                                           double *__restrict__ out) {
  const int rr = blockIdx.x, q = rr / R, r = rr - q * R;
  const double *src = U + (size_t)r * N, *cs = lamcs + (size_t)q * 2 * nm;
  double *dst = out + (size_t)rr * N;
  for (int n = threadIdx.x; n < N; n += 256) {
    const int m = m_of[n], am = m < 0 ? -m : m;
    // (cos / sin of |m| lam; the signs as in the reference's f = M cos(m th) + M_mirror sin(m th))
    const double c = cs[am], sn = m < 0 ? -cs[nm + am] : cs[nm + am];
    dst[n] = src[n] * c + src[mirror[n]] * sn;
  }
}

// m1[n] = sum over (q, k) of w[k] A[(q P + k) mv][n]  (the first vector of every rotation), in two
// stages with a fixed order: one partial sum per longitude, then the sum over the longitudes
__global__ __launch_bounds__(256) void up_first_moment_kernel(int N, int P, int mv,
                                                              const double *__restrict__ w,
                                                              const double *__restrict__ A,
                                                              double *__restrict__ part) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (n >= N) return;
  double acc = 0.0;
  for (int k = 0; k < P; ++k) acc += w[k] * A[((size_t)(q * P + k) * mv) * N + n];
  part[(size_t)q * N + n] = acc;
}
__global__ __launch_bounds__(256) void up_first_moment_sum_kernel(int N, int Q, const double *__restrict__ part,
                                                                  double *__restrict__ m1) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double acc = 0.0;
  for (int q = 0; q < Q; ++q) acc += part[(size_t)q * N + n];
  m1[n] = acc;
}

// T[n][c] = A2[row(c)][n] for c < R2 (row(c): the c-th second-moment row of A), zero for R2 <= c < ld
__global__ void up_transpose_kernel(int N, int R2, int ld, int mv, int j0, int m, const double *__restrict__ A,
                                    double *__restrict__ T) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + threadIdx.x;
    double v = 0.0;
    if (c < R2 && n < N) {
      const int rot = c / m, j = c - rot * m;
      v = A[((size_t)rot * mv + j0 + j) * N + n];
    }
    tile[i][threadIdx.x] = v;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + threadIdx.x;
    if (n < N && c < ld) T[(size_t)n * ld + c] = tile[threadIdx.x][i];
  }
}

// cov = sum of the nc partial products (fixed order) - m1 m1^T + diag(eps); mean = sqrt(n) m1
__global__ void up_finish_kernel(int N, int nc, const double *__restrict__ parts, const double *__restrict__ m1,
                                 double sqrt_n, double epsy, double epsy15, double *__restrict__ cov,
                                 double *__restrict__ mean) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)N * N) return;
  const int i = (int)(e / N), j = (int)(e - (long)i * N);
  double acc = 0.0;
  for (int c = 0; c < nc; ++c) acc += parts[(size_t)c * N * N + e];
  double v = acc - m1[i] * m1[j];
  if (i == j) v += i >= 15 * 15 ? epsy15 : epsy;
  cov[e] = v;
  if (j == 0) mean[i] = sqrt_n * m1[i];
}

inline size_t up_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

int ensure_big_scratch(sp_handle *h, size_t bytes, void **out);   // sp_api.hip

extern "C" int sp_ylm_moments_quadrature(sp_handle *h, const double *vecs_host, int mv, int first_is_col,
                                         const double *phi_host, const double *w_host, int P, int Q,
                                         double g, double sqrt_n, double epsy, double epsy15,
                                         double *mean_dev, double *cov_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !vecs_host || !phi_host || !w_host || !mean_dev || !cov_dev || mv < 1 || P < 1 || Q < 1 ||
      (!first_is_col && mv < 2))
    return SP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int N = h->N, NWIG = h->NWIG, nm = h->ydeg + 1;
  bool paired = (P % 2) == 0;
  for (int k = 0; paired && k < P / 2; ++k) paired = phi_host[k + P / 2] == -phi_host[k];
  const int m = first_is_col ? mv : mv - 1, j0 = first_is_col ? 0 : 1;
  const long R = (long)P * mv, RR = (long)Q * R, R2 = (long)Q * P * m;
  if (RR > 65535) return SP_ERR_INVALID;
  // the second-moment product T T^T is cut along its long dimension (R2 rotations x columns: 1 122 to
  // 19 074) into chunks of KC: as ONE product of 256 x 256 x R2 it was 4 workgroups walking 144 slices
  // each (196 us); as a batch of R2 / KC products the chunks run side by side (split-K, summed in a
  // fixed order by up_finish_kernel)
  const int KC = 128;
  const int ld2 = (int)((R2 + KC - 1) / KC * KC), nchunk = ld2 / KC;
  SP_HIP(hipSetDevice(h->device));

  // constants of the handle: Rx(-pi/2) and the cos / sin (m lam_q) table of the Q longitudes
  if (!h->d_Rxm90) {
    SP_HIP(hipMalloc((void **)&h->d_Rxm90, sizeof(double) * NWIG));
    const double th = -0.5 * M_PI;
    int rc = sp_Rx(h, &th, 1, h->d_Rxm90, nullptr, stream);
    if (rc) return rc;
  }
  if (h->lamcs_Q != Q) {
    // (a change of Q while launches that read the old table are in flight: drained first; never in a sampler loop)
    if (h->d_lamcs) {
      SP_HIP(hipStreamSynchronize(st));
      SP_HIP(hipFree(h->d_lamcs));
      h->d_lamcs = nullptr;
    }
    std::vector<double> tab((size_t)Q * 2 * nm);
    for (int q = 0; q < Q; ++q) {
      const double lam = 2.0 * M_PI * q / Q;
      for (int k = 0; k < nm; ++k) {
        tab[(size_t)q * 2 * nm + k] = std::cos(k * lam);
        tab[(size_t)q * 2 * nm + nm + k] = std::sin(k * lam);
      }
    }
    SP_HIP(hipMalloc((void **)&h->d_lamcs, sizeof(double) * tab.size()));
    SP_HIP(hipMemcpy(h->d_lamcs, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    h->lamcs_Q = Q;
  }

  // scratch: Rphi [P, NWIG] | M0, V [P, mv, N] | U [R, N] | U2, A [RR, N] | T [N, ld2] | m1 [N]
  size_t off = 0;
  auto take = [&](size_t doubles) { size_t o = off; off += up_align(sizeof(double) * doubles); return o; };
  const size_t oR = take((size_t)P * NWIG), oM0 = take((size_t)R * N), oV = take((size_t)R * N),
               oU = take((size_t)R * N), oU2 = take((size_t)RR * N), oA = take((size_t)RR * N),
               oT = take((size_t)N * ld2), om1 = take(N), opart = take((size_t)Q * N),
               oC = take((size_t)nchunk * N * N);
  void *ws = nullptr;
  int rc = ensure_big_scratch(h, off, &ws);
  if (rc) return rc;
  auto at = [&](size_t o) { return reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + o); };
  double *Rphi = at(oR), *M0 = at(oM0), *V = at(oV), *U = at(oU), *U2 = at(oU2), *A = at(oA), *T = at(oT),
         *m1 = at(om1), *part = at(opart), *Cp = at(oC);

  // ONE staged upload: cos / sin of the latitudes, sqrt weights, plain weights, the vectors
  const size_t need = 2 * (size_t)P + 2 * (size_t)P + (size_t)mv * N;
  sp_handle::CsSlot *cp = nullptr;
  if ((rc = sp_stage_acquire(h, need, &cp))) return rc;
  sp_handle::CsSlot &c = *cp;
  double *hcs = c.host, *hsw = hcs + 2 * P, *hw = hsw + P, *hv = hw + P;
  for (int k = 0; k < P; ++k) {
    hcs[2 * k] = std::cos(phi_host[k]);
    hcs[2 * k + 1] = std::sin(phi_host[k]);
    hsw[k] = g * std::sqrt(w_host[k] / Q);     // g sqrt(W_kq)
    hw[k] = std::sqrt(w_host[k] / Q);          // sqrt(W_kq):  m1 = sum sqrt(W) (g sqrt(W) row)
  }
  memcpy(hv, vecs_host, sizeof(double) * (size_t)mv * N);
  SP_HIP(hipMemcpyAsync(c.dev, c.host, sizeof(double) * need, hipMemcpyHostToDevice, st));
  const double *dcs = c.dev, *dsw = dcs + 2 * P, *dw = dsw + P, *dv = dw + P;

  // (the latitudes come in pairs +phi, -phi -- phi_host[k + P/2] = -phi_host[k], checked below --: one
  //  Wigner recursion per pair, the second rotation of a pair uses the transposed blocks)
  const int Ph = paired ? P / 2 : P;
  if ((rc = sp_launch_Rx(h, dcs, Ph, Rphi, nullptr, st))) return rc;
  hipLaunchKernelGGL(up_outer_kernel, dim3((mv * N + 255) / 256, P), dim3(256), 0, st, dsw, dv, mv, N, M0);
  SP_LAUNCH_CHECK();
  SP_HIP(hipEventRecord(c.done, st));
  c.used = true;
  // V = M0 Rx(phi_k), rotation by rotation; U = V Rx(pi/2); U2 = Rz(lam_q) on every row; A = U2 Rx(-pi/2)
  if ((rc = sp_launch_dotRx(h, M0, (long)mv * N, N, 1, mv, Rphi, NWIG, V, Ph, st))) return rc;
  if (paired &&
      (rc = sp_launch_dotRx(h, M0 + (size_t)Ph * mv * N, (long)mv * N, N, 1, mv, Rphi, NWIG, V + (size_t)Ph * mv * N,
                            Ph, st, 1)))
    return rc;
  if ((rc = sp_launch_dotRx(h, V, 0, N, 1, (int)R, h->d_Rx90, 0, U, 1, st))) return rc;
  hipLaunchKernelGGL(up_rz_repeat_kernel, dim3((unsigned)RR), dim3(256), 0, st, N, (int)R, nm, h->d_m_of,
                     h->d_mirror, h->d_lamcs, U, U2);
  SP_LAUNCH_CHECK();
  if ((rc = sp_launch_dotRx(h, U2, 0, N, 1, (int)RR, h->d_Rxm90, 0, A, 1, st))) return rc;
  // moments
  hipLaunchKernelGGL(up_first_moment_kernel, dim3((N + 255) / 256, Q), dim3(256), 0, st, N, P, mv, dw, A, part);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(up_first_moment_sum_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, Q, part, m1);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(up_transpose_kernel, dim3(ld2 / 32, (N + 31) / 32), dim3(32, 8), 0, st, N, (int)R2, ld2, mv,
                     j0, m, A, T);
  SP_LAUNCH_CHECK();
  // (batch b = columns b KC .. of T: the "matrix stride" is a column offset)
  if ((rc = sp_launch_gemm_nt(T, ld2, KC, T, ld2, KC, Cp, N, (long)N * N, N, N, KC, 1.0, 0, 0, nchunk, st))) return rc;
  hipLaunchKernelGGL(up_finish_kernel, dim3((unsigned)(((long)N * N + 255) / 256)), dim3(256), 0, st, N, nchunk, Cp,
                     m1, sqrt_n, epsy, epsy15, cov_dev, mean_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}


// ---------------------------------------------------------------------------------------------------
// The same moments WITH their exact derivatives with respect to the spot radius and the two Beta shape parameters
// (one radius, dr = None): the tangents ride through the same rotations as extra rows.  With the rows
//     A_kq = g sqrt(W_kq) Ry(lam_q) Rx(phi_k) s           (m1 = sum sqrt(W) A,   Sigma = sum A A^T - m1 m1^T + eps)
// a parameter x in {r, alpha, beta} moves
//     dA_kq = g sqrt(W) Ry Rx (ds/dx)  +  g (d sqrt(W)/dx) Ry Rx s  +  g sqrt(W) (d phi_k/dx) Ry Rx'(phi_k) s,
// the first term for r (size.py:92-101: the sigmoid profile's own derivative), the other two for alpha and beta
// (sp_gauss_jacobi_grad: the rule is exact whatever the exponents, so this IS the derivative of the expectation --
// the reference's analytic d/d alpha, d/d beta, ops/include/latitude.h:21-173).  Everything behind Rx(phi_k) is
// linear, so the four rows (A, dA_r, dA_alpha, dA_beta) of a latitude go through Rx(pi/2) Rz(lam) Rx(-pi/2)
// together, and
//     dm1 = sum sqrt(W) dA + (d sqrt(W)) A,       dSigma = sum (dA A^T + A dA^T) - dm1 m1^T - m1 dm1^T:
// three cross products on the matrix cores beside the one of the value.
namespace {
constexpr int UPG = 4;     // rows per rotation: value, d/dr, d/dalpha, d/dbeta

// V[k][tau][n] = sum_i (c0[k][tau] s[i] + c1[k][tau] ds[i]) R_k[i][n] + c2[k][tau] s[i] R'_k[i][n]
// (row vector times the l-block of the rotation; the second half of the latitudes, -phi_k, takes the TRANSPOSED
//  blocks of its partner: R(-phi) = R(phi)^T and d/dx R(-phi_k) = R'(phi_k)^T dphi_k/dx)
__global__ __launch_bounds__(256) void up_tangent_rows_kernel(int N, int Ph, const int32_t *__restrict__ l_of,
                                                              const int32_t *__restrict__ blk, int nwig,
                                                              const double *__restrict__ coef /* [P][UPG][3] */,
                                                              const double *__restrict__ sv /* [2][N]: s, ds/dr */,
                                                              const double *__restrict__ Rpk, const double *__restrict__ dRpk,
                                                              double *__restrict__ V) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, tau = blockIdx.y, k = blockIdx.z;
  if (n >= N) return;
  const int l = l_of[n], w = 2 * l + 1, base = l * l;
  const bool rt = k >= Ph;
  const int kr = rt ? k - Ph : k;
  const double *B = Rpk + (size_t)kr * nwig + blk[l] + (rt ? (n - base) * w : (n - base));
  const double *dB = dRpk + (size_t)kr * nwig + blk[l] + (rt ? (n - base) * w : (n - base));
  const int bs = rt ? 1 : w;
  const double c0 = coef[((size_t)k * UPG + tau) * 3], c1 = coef[((size_t)k * UPG + tau) * 3 + 1],
               c2 = coef[((size_t)k * UPG + tau) * 3 + 2];
  double acc = 0.0;
  for (int i = 0; i < w; ++i) {
    const double s0 = sv[base + i], s1 = sv[N + base + i];
    acc += (c0 * s0 + c1 * s1) * B[i * bs] + (c2 * s0) * dB[i * bs];
  }
  V[((size_t)k * UPG + tau) * N + n] = acc;
}

// part[q][tau][n] = sum_k sqw[k] A[(q P + k) UPG + tau][n] + dsqw[tau][k] A[(q P + k) UPG][n]      (dsqw[0] = 0)
__global__ __launch_bounds__(256) void up_first_moment_grad_kernel(int N, int P, const double *__restrict__ sqw,
                                                                   const double *__restrict__ dsqw /* [UPG][P] */,
                                                                   const double *__restrict__ A,
                                                                   double *__restrict__ part) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y, tau = blockIdx.z;
  if (n >= N) return;
  double acc = 0.0;
  for (int k = 0; k < P; ++k) {
    const double *row = A + ((size_t)(q * P + k) * UPG) * N + n;
    acc += sqw[k] * row[(size_t)tau * N];
    if (tau) acc += dsqw[(size_t)tau * P + k] * row[0];
  }
  part[((size_t)q * UPG + tau) * N + n] = acc;
}
__global__ __launch_bounds__(256) void up_first_moment_grad_sum_kernel(int N, int Q, const double *__restrict__ part,
                                                                       double *__restrict__ m1 /* [UPG][N] */) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, tau = blockIdx.y;
  if (n >= N) return;
  double acc = 0.0;
  for (int q = 0; q < Q; ++q) acc += part[((size_t)q * UPG + tau) * N + n];
  m1[(size_t)tau * N + n] = acc;
}

// value: cov = sum_c C0[c] - m1 m1^T + diag(eps), mean = sqrt(n) m1;
// tangents: dcov_x = sum_c (C_x[c] + C_x[c]^T) - dm1 m1^T - m1 dm1^T, dmean_x = sqrt(n) dm1
__global__ void up_finish_grad_kernel(int N, int nc, const double *__restrict__ parts /* [UPG][nc][N][N] */,
                                      const double *__restrict__ m1 /* [UPG][N] */, double sqrt_n, double epsy,
                                      double epsy15, double *__restrict__ cov, double *__restrict__ mean,
                                      double *__restrict__ dcov /* [3][N][N] */, double *__restrict__ dmean /* [3][N] */) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tau = blockIdx.y;
  if (e >= (long)N * N) return;
  const int i = (int)(e / N), j = (int)(e - (long)i * N);
  const double *pt = parts + (size_t)tau * nc * N * N;
  double acc = 0.0;
  if (tau == 0) {
    for (int c = 0; c < nc; ++c) acc += pt[(size_t)c * N * N + e];
    double v = acc - m1[i] * m1[j];
    if (i == j) v += i >= 15 * 15 ? epsy15 : epsy;
    cov[e] = v;
    if (j == 0) mean[i] = sqrt_n * m1[i];
  } else {
    const long et = (long)j * N + i;
    for (int c = 0; c < nc; ++c) acc += pt[(size_t)c * N * N + e] + pt[(size_t)c * N * N + et];
    const double *d1 = m1 + (size_t)tau * N;
    // (symmetric to the bit: the two products in the order of the smaller index, whatever the compiler fuses)
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    dcov[(size_t)(tau - 1) * N * N + e] = acc - (d1[lo] * m1[hi] + m1[lo] * d1[hi]);
    if (j == 0) dmean[(size_t)(tau - 1) * N + i] = sqrt_n * d1[i];
  }
}
}  // namespace

extern "C" int sp_ylm_moments_quadrature_grad(sp_handle *h, const double *s_host, const double *ds_dr_host,
                                              const double *phi_host, const double *w_host,
                                              const double *dphi_host, const double *dw_host, int P, int Q, double g,
                                              double sqrt_n, double epsy, double epsy15, double *mean_dev,
                                              double *cov_dev, double *dmean_dev, double *dcov_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !s_host || !ds_dr_host || !phi_host || !w_host || !dphi_host || !dw_host || !mean_dev || !cov_dev ||
      !dmean_dev || !dcov_dev || P < 2 || (P & 1) || Q < 1)
    return SP_ERR_INVALID;
  const int Ph = P / 2;
  for (int k = 0; k < Ph; ++k)
    if (phi_host[k + Ph] != -phi_host[k] || w_host[k + Ph] != w_host[k]) return SP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int N = h->N, NWIG = h->NWIG, nm = h->ydeg + 1;
  const long R = (long)P * UPG, RR = (long)Q * R, R2 = (long)Q * P;
  if (RR > 65535) return SP_ERR_INVALID;
  const int KC = 128;
  const int ld2 = (int)((R2 + KC - 1) / KC * KC), nchunk = ld2 / KC;
  SP_HIP(hipSetDevice(h->device));
  if (!h->d_Rxm90) {
    SP_HIP(hipMalloc((void **)&h->d_Rxm90, sizeof(double) * NWIG));
    const double th = -0.5 * M_PI;
    int rc = sp_Rx(h, &th, 1, h->d_Rxm90, nullptr, stream);
    if (rc) return rc;
  }
  if (h->lamcs_Q != Q) {
    if (h->d_lamcs) {
      SP_HIP(hipStreamSynchronize(st));
      SP_HIP(hipFree(h->d_lamcs));
      h->d_lamcs = nullptr;
    }
    std::vector<double> tab((size_t)Q * 2 * nm);
    for (int q = 0; q < Q; ++q) {
      const double lam = 2.0 * M_PI * q / Q;
      for (int k = 0; k < nm; ++k) {
        tab[(size_t)q * 2 * nm + k] = std::cos(k * lam);
        tab[(size_t)q * 2 * nm + nm + k] = std::sin(k * lam);
      }
    }
    SP_HIP(hipMalloc((void **)&h->d_lamcs, sizeof(double) * tab.size()));
    SP_HIP(hipMemcpy(h->d_lamcs, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    h->lamcs_Q = Q;
  }
  // scratch: Rphi, dRphi [Ph, NWIG] | V, U [R, N] | U2, A [RR, N] | T [UPG][N, ld2] | m1 [UPG][N] | part | C [UPG][nchunk][N][N]
  size_t off = 0;
  auto take = [&](size_t doubles) { size_t o = off; off += up_align(sizeof(double) * doubles); return o; };
  const size_t oR = take((size_t)Ph * NWIG), odR = take((size_t)Ph * NWIG), oV = take((size_t)R * N),
               oU = take((size_t)R * N), oU2 = take((size_t)RR * N), oA = take((size_t)RR * N),
               oT = take((size_t)UPG * N * ld2), om1 = take((size_t)UPG * N), opart = take((size_t)Q * UPG * N),
               oC = take((size_t)UPG * nchunk * N * N);
  void *ws = nullptr;
  int rc = ensure_big_scratch(h, off, &ws);
  if (rc) return rc;
  auto at = [&](size_t o) { return reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + o); };
  double *Rphi = at(oR), *dRphi = at(odR), *V = at(oV), *U = at(oU), *U2 = at(oU2), *A = at(oA), *T = at(oT),
         *m1 = at(om1), *part = at(opart), *Cp = at(oC);

  // ONE staged upload: cos / sin of the Ph latitudes | coefficients [P][UPG][3] | sqrt weights [P] | their
  // derivatives [UPG][P] | s, ds/dr [2][N]
  const size_t need = 2 * (size_t)Ph + (size_t)P * UPG * 3 + (size_t)P + (size_t)UPG * P + 2 * (size_t)N;
  sp_handle::CsSlot *cp = nullptr;
  if ((rc = sp_stage_acquire(h, need, &cp))) return rc;
  sp_handle::CsSlot &c = *cp;
  double *hcs = c.host, *hco = hcs + 2 * Ph, *hsq = hco + (size_t)P * UPG * 3, *hdsq = hsq + P, *hsv = hdsq + (size_t)UPG * P;
  for (int k = 0; k < Ph; ++k) {
    hcs[2 * k] = std::cos(phi_host[k]);
    hcs[2 * k + 1] = std::sin(phi_host[k]);
  }
  for (int k = 0; k < P; ++k) {
    const int kr = k < Ph ? k : k - Ph;
    const double sq = std::sqrt(w_host[k] / Q);                 // sqrt(W_kq)
    hsq[k] = sq;
    hdsq[k] = hdsq[P + k] = 0.0;
    double *co = hco + (size_t)k * UPG * 3;
    co[0] = g * sq; co[1] = 0.0; co[2] = 0.0;                    // the value
    co[3] = 0.0; co[4] = g * sq; co[5] = 0.0;                    // d/dr
    for (int x = 0; x < 2; ++x) {                                // d/dalpha, d/dbeta
      const double dsq = 0.5 * dw_host[(size_t)x * P + k] / (Q * sq);
      hdsq[(size_t)(2 + x) * P + k] = dsq;
      co[3 * (2 + x)] = g * dsq;
      co[3 * (2 + x) + 1] = 0.0;
      co[3 * (2 + x) + 2] = g * sq * dphi_host[(size_t)x * P + kr];   // (the partner's: see the kernel)
    }
  }
  memcpy(hsv, s_host, sizeof(double) * N);
  memcpy(hsv + N, ds_dr_host, sizeof(double) * N);
  SP_HIP(hipMemcpyAsync(c.dev, c.host, sizeof(double) * need, hipMemcpyHostToDevice, st));
  const double *dcs = c.dev, *dco = dcs + 2 * Ph, *dsq = dco + (size_t)P * UPG * 3, *ddsq = dsq + P,
               *dsv = ddsq + (size_t)UPG * P;
  if ((rc = sp_launch_Rx(h, dcs, Ph, Rphi, dRphi, st))) return rc;
  hipLaunchKernelGGL(up_tangent_rows_kernel, dim3((N + 255) / 256, UPG, P), dim3(256), 0, st, N, Ph, h->d_l_of, h->d_blk,
                     NWIG, dco, dsv, Rphi, dRphi, V);
  SP_LAUNCH_CHECK();
  if ((rc = sp_launch_dotRx(h, V, 0, N, 1, (int)R, h->d_Rx90, 0, U, 1, st))) return rc;
  hipLaunchKernelGGL(up_rz_repeat_kernel, dim3((unsigned)RR), dim3(256), 0, st, N, (int)R, nm, h->d_m_of,
                     h->d_mirror, h->d_lamcs, U, U2);
  SP_LAUNCH_CHECK();
  if ((rc = sp_launch_dotRx(h, U2, 0, N, 1, (int)RR, h->d_Rxm90, 0, A, 1, st))) return rc;
  hipLaunchKernelGGL(up_first_moment_grad_kernel, dim3((N + 255) / 256, Q, UPG), dim3(256), 0, st, N, P, dsq, ddsq, A,
                     part);
  SP_LAUNCH_CHECK();
  SP_HIP(hipEventRecord(c.done, st));
  c.used = true;
  hipLaunchKernelGGL(up_first_moment_grad_sum_kernel, dim3((N + 255) / 256, UPG), dim3(256), 0, st, N, Q, part, m1);
  SP_LAUNCH_CHECK();
  for (int tau = 0; tau < UPG; ++tau) {
    hipLaunchKernelGGL(up_transpose_kernel, dim3(ld2 / 32, (N + 31) / 32), dim3(32, 8), 0, st, N, (int)R2, ld2, UPG,
                       tau, 1, A, T + (size_t)tau * N * ld2);
    SP_LAUNCH_CHECK();
  }
  for (int tau = 0; tau < UPG; ++tau)
    if ((rc = sp_launch_gemm_nt(T + (size_t)tau * N * ld2, ld2, KC, T, ld2, KC, Cp + (size_t)tau * nchunk * N * N, N,
                                (long)N * N, N, N, KC, 1.0, 0, 0, nchunk, st)))
      return rc;
  hipLaunchKernelGGL(up_finish_grad_kernel, dim3((unsigned)(((long)N * N + 255) / 256), UPG), dim3(256), 0, st, N,
                     nchunk, Cp, m1, sqrt_n, epsy, epsy15, cov_dev, mean_dev, dcov_dev, dmean_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
