// B hyperparameter samples -> B sets of polar-frame moments (ez, Ez), in ONE call (round 6).
//
// A sampler evaluates the likelihood of one data set at many (r, a, b, c, n) (calibrate/sample.py:95-107,
// interfaces.py:142-166 of the reference); one sample at a time the upstream of the path was a chain of ten launches
// behind 0.5 ms of host preparation (size integral in NumPy, Gauss-Jacobi rule through ctypes).  Here the whole chain
// hyperparameters -> (ez, Ez) runs on the device for the B samples of a batch: one staged upload of 5 B numbers, five
// launches, no host arithmetic.
//
// What is computed is what sp_ylm_moments_quadrature + sp_set_ylm_moments_dev compute (csrc/sp_upstream.hip documents
// the quadrature of rotations; size.py:49-101, latitude.py:170-212, longitude.py:8-78, contrast.py:18-33 and
// flux.py:54-62 of the reference), in the frame where the marginal branch needs it.  With the rows
//     A_kq = g sqrt(w_k / Q)  s Rx(phi_k) Rx(pi/2) Rz(lam_q) Rx(-pi/2)        g = pi c sqrt(n)
// the moments of the Ylm process are mu = sqrt(n) m1, m1 = sum sqrt(w / Q) A, Sigma = sum A^T A - m1 m1^T + eps, and the
// POLAR-frame moments are ez = mu Rx(pi/2), Ez = Rx(pi/2)^T (Sigma + mu mu^T) Rx(pi/2): the last rotation of every row
// cancels, Rx(-pi/2) Rx(pi/2) = 1, and what is left of the longitude sum is an average of Rz(lam) M Rz(lam)^T over
// Q > 2 ydeg equispaced angles -- EXACTLY the projection of M = sum_k w_k u_k^T u_k, u_k = s Rx(phi_k) Rx(pi/2), onto
// the matrices that commute with every Rz: entries between orders of different |m| vanish, and between (l, +-m) and
// (l', +-m), m > 0, the 2 x 2 block X becomes (X11 + X22) / 2 on its diagonal and +-(X12 - X21) / 2 off it.  So
//     e1 = g sum_k w_k (the m = 0 entries of u_k),        ez = sqrt(n) e1,
//     Ez = g^2 Proj(sum_k w_k u_k^T u_k) + (n - 1) e1 e1^T + diag(eps):
// 2 (ydeg + 2) rotations per sample instead of 2 (ydeg + 2) (2 ydeg + 3), and no rotation back and forth; the rotation
// Rx(phi_k) of the zonal size vector is a row of associated Legendre functions (sm_rows_kernel): no Wigner recursion.  (Checked on
// the CPU against the oracle's quadrature + polar_moments: 1e-15 relative, tests/test_samples_identities.py.)
//
// The Gauss-Jacobi nodes come from multi-section on the Jacobi matrix's Sturm sequence (a group of threads per node; the
// weights from the orthonormal recurrence at the node): the same rule as sp_gauss_jacobi's implicit QL to 1e-15 in the nodes
// and 2e-11 in the weights over the reference's whole prior box (tests/test_gpu_samples.py).
#include <cmath>
#include <cstring>

#include "sp_internal.h"

int ensure_big_scratch(sp_handle *h, size_t bytes, void **out);   // sp_api.hip

namespace {

constexpr int SM_TK = 64;     // columns of a sample's row block T (its 2 (ydeg + 2) <= 64 rotations, zero padded)

__device__ __forceinline__ double sm_wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Per sample: the size vector (size.py:92-101: the sigmoid profile's Legendre coefficients, only the m = 0 entries are
// nonzero), the Gauss-Jacobi rule of the latitude law and the scales of the rotations.  grid B.
//   basis: Bp [nl][spts] (upstream._spot_basis), then the colatitude grid theta [spts]
//   samp [B][5]: r [rad], alpha, beta, c, n
__global__ __launch_bounds__(256) void sm_prepare_kernel(int ydeg, int spts, double sfac, const double *__restrict__ basis,
                                                         const double *__restrict__ samp, double *__restrict__ svec,
                                                         double *__restrict__ cs, double *__restrict__ sc,
                                                         double *__restrict__ scal) {
  extern __shared__ __attribute__((aligned(16))) double sm_lds[];
  const int nl = ydeg + 1, nq = ydeg + 2, b = blockIdx.x, tid = threadIdx.x;
  double *s_b = sm_lds;          // [spts] the profile
  double *s_d = s_b + spts;      // [nq] diagonal of the Jacobi matrix
  double *s_e = s_d + nq;        // [nq] off-diagonal (e[k] couples k and k + 1)
  double *s_e2 = s_e + nq;       // [nq] its squares
  double *s_w = s_e2 + nq;       // [nq] weights before normalisation
  const double r = samp[5 * b], alpha = samp[5 * b + 1], beta = samp[5 * b + 2], c = samp[5 * b + 3], n = samp[5 * b + 4];
  const double *theta = basis + (size_t)nl * spts;
  for (int j = tid; j < spts; j += 256) s_b[j] = 1.0 / (1.0 + exp(-sfac * (theta[j] - r))) - 1.0;
  // weight (1 - t)^(beta - 1) (1 + t)^(alpha - 1): the recurrence coefficients of sp_gauss_jacobi (sp_host.cpp)
  if (tid < nq) {
    const double a = beta - 1.0, bb = alpha - 1.0, ab = a + bb;
    const int k = tid;
    if (k == 0) {
      s_d[0] = (bb - a) / (ab + 2.0);
      s_e[nq - 1] = 0.0;
      s_e2[nq - 1] = 0.0;
    } else {
      const double s = 2.0 * k + ab;
      s_d[k] = (bb - a) * (bb + a) / (s * (s + 2.0));
      const double num = (k == 1) ? 4.0 * (1.0 + a) * (1.0 + bb) / ((s * s) * (s + 1.0))
                                  : 4.0 * k * (k + a) * (k + bb) * (k + ab) / ((s * s) * (s + 1.0) * (s - 1.0));
      s_e2[k - 1] = num;
      s_e[k - 1] = sqrt(num);
    }
  }
  __syncthreads();
  // size vector: wavefront w takes the degrees w, w + 4, ...; fixed summation order
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int l = wave; l < nl; l += 4) {
      const double *row = basis + (size_t)l * spts;
      double acc = 0.0;
      for (int j = lane; j < spts; j += 64) acc += row[j] * s_b[j];
      acc = sm_wave_sum(acc);
      if (lane == 0) svec[(size_t)b * nl + l] = acc;
    }
  }
  // node i: the i-th eigenvalue of the Jacobi matrix, bracketed on the Sturm count (all of them lie in (-1, 1)).  A
  // group of npt threads per node cuts the bracket into npt + 1 parts per round (plain bisection, one thread per node,
  // was 58 dependent rounds of nq divisions: 110 us of latency in front of every batch; 15 rounds now).
  const int npt = 256 / nq < 15 ? 256 / nq : 15, node = tid / npt, pt = tid - node * npt;
  int rounds = 0;
  for (double span = 2.0; span > 3.0e-18; span /= npt + 1) ++rounds;
  int *s_flag = reinterpret_cast<int *>(s_w + nq);        // [256]
  double lo = -1.0, hi = 1.0;
  for (int it = 0; it < rounds; ++it) {
    int above = 0;
    if (node < nq) {
      const double x = lo + (hi - lo) * ((double)(pt + 1) / (double)(npt + 1));
      // (LAPACK's dlaebz: a pivot below pivmin counts as negative and is replaced BEFORE it is counted and used --
      //  alpha = beta makes the diagonal zero and the first midpoint, x = 0, an exact zero pivot)
      double q = s_d[0] - x;
      if (fabs(q) < 1.0e-290) q = -1.0e-290;
      int cnt = q < 0.0 ? 1 : 0;
      for (int k = 1; k < nq; ++k) {
        q = s_d[k] - x - s_e2[k - 1] / q;
        if (fabs(q) < 1.0e-290) q = -1.0e-290;
        cnt += q < 0.0 ? 1 : 0;
      }
      above = cnt > node ? 1 : 0;       // more than `node` eigenvalues below x: node's eigenvalue is below x
    }
    s_flag[tid] = above;
    __syncthreads();
    if (node < nq) {
      int below = 0;                    // trial points at or below the eigenvalue
      for (int j = 0; j < npt; ++j) below += 1 - s_flag[node * npt + j];
      const double w = hi - lo, l0 = lo;
      if (below > 0) lo = l0 + w * ((double)below / (double)(npt + 1));
      if (below < npt) hi = l0 + w * ((double)(below + 1) / (double)(npt + 1));
    }
    __syncthreads();
  }
  double ti = 0.0;
  if (node < nq && pt == 0) {
    ti = 0.5 * (lo + hi);
    // weight: 1 / sum_k p_k(t_i)^2 of the orthonormal polynomials (p_0 = 1)
    double p0 = 0.0, p1 = 1.0, sum = 1.0;
    for (int k = 0; k + 1 < nq; ++k) {
      const double p2 = ((ti - s_d[k]) * p1 - (k > 0 ? s_e[k - 1] : 0.0) * p0) / s_e[k];
      sum += p2 * p2;
      p0 = p1;
      p1 = p2;
    }
    s_w[node] = 1.0 / sum;
    s_e2[node] = ti;                    // (the squares are not needed any more: the nodes, by index)
  }
  __syncthreads();
  if (tid < nq) {
    const double ti = s_e2[tid];
    double tot = 0.0;
    for (int k = 0; k < nq; ++k) tot += s_w[k];
    const double wphi = 0.5 * (s_w[tid] / tot);      // both signs of the latitude share a node's weight
    const double x = 0.5 * (1.0 + ti);               // cos(phi)
    const int P = 2 * nq;
    const double g = 3.141592653589793 * c * sqrt(n);
    cs[((size_t)b * nq + tid) * 2] = x;
    cs[((size_t)b * nq + tid) * 2 + 1] = sqrt((1.0 - x) * (1.0 + x));
    double *sb = sc + (size_t)b * 2 * P;
    const double sq = sqrt(wphi);
    sb[tid] = sb[tid + nq] = g * sq;
    sb[P + tid] = sb[P + tid + nq] = sq;
    if (tid == 0) {
      scal[4 * b] = g;
      scal[4 * b + 1] = sqrt(n);
      scal[4 * b + 2] = n;
      scal[4 * b + 3] = 0.0;
    }
  }
}

// T[b][n][k] = g sqrt(w_k) (s Rx(+-phi_k) Rx(pi/2))[n]: rotation k of sample b.  grid (SM_TK, B); the columns from
// P on are the zero padding of the product's long dimension.
//
// The size vector s has its nonzero entries at m = 0 (size.py:92-101: a spot at the pole is zonal), so s Rx(phi) needs
// row m' = 0 of every degree's block only -- and that row is the Wigner function d^l_{m0}(phi), a normalised
// associated Legendre function: with N_l^m = sqrt((l - m)! / (l + m)!) P_l^m(cos phi) (no Condon-Shortley phase),
//     N_m^m = sqrt((2m - 1) / (2m)) sin(phi) N_{m-1}^{m-1},
//     N_l^m = ((2l - 1) cos(phi) N_{l-1}^m - sqrt((l - 1)^2 - m^2) N_{l-2}^m) / sqrt(l^2 - m^2),
// the real block's row is R_l[0][0] = N_l^0, R_l[0][+m] = (-1)^m sqrt(2) cos(m pi/2) N_l^m, R_l[0][-m] = -(-1)^m sqrt(2)
// sin(m pi/2) N_l^m (the complex -> real step of wigner.h:225-271 applied to that row; equal to sp_Rx's row to 1e-14,
// tests/test_samples_identities.py).  O(ydeg^2) per rotation, one thread per order m -- the full recursion of
// wigner.h:36-139 (rx_kernel: all (2l + 1)^2 entries of every block, 100 us per angle and 23 KB of LDS) is not run here.
__global__ __launch_bounds__(256) void sm_rows_kernel(int ydeg, int N, int P, const int32_t *__restrict__ l_of,
                                                      const int32_t *__restrict__ blk, const double *__restrict__ svec,
                                                      const double *__restrict__ cs, const double *__restrict__ Rx90,
                                                      const double *__restrict__ sc, double *__restrict__ T) {
  extern __shared__ __attribute__((aligned(16))) double sm_v[];   // [N]
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nl = ydeg + 1;
  double *Tb = T + (size_t)b * N * SM_TK;
  if (k >= P) {
    for (int n = tid; n < N; n += 256) Tb[(size_t)n * SM_TK + k] = 0.0;
    return;
  }
  const int Ph = P / 2, kr = k < Ph ? k : k - Ph;
  const double c = cs[((size_t)b * Ph + kr) * 2];
  const double sn = k < Ph ? cs[((size_t)b * Ph + kr) * 2 + 1] : -cs[((size_t)b * Ph + kr) * 2 + 1];   // (-phi: the mirror latitude)
  if (tid < nl) {
    const int m = tid;
    double nmm = 1.0;
    for (int j = 1; j <= m; ++j) nmm *= sqrt((double)(2 * j - 1) / (double)(2 * j)) * sn;
    const int t4 = m & 3;
    const double sg = (m & 1) ? -1.0 : 1.0, r2 = 1.4142135623730951;
    const double fc = m == 0 ? 1.0 : sg * r2 * (t4 == 0 ? 1.0 : (t4 == 2 ? -1.0 : 0.0));     // on N_l^m at order +m
    const double fs = m == 0 ? 0.0 : -sg * r2 * (t4 == 1 ? 1.0 : (t4 == 3 ? -1.0 : 0.0));    // at order -m
    double p2 = 0.0, p1 = nmm;
    for (int l = m; l < nl; ++l) {
      double v = nmm;
      if (l > m) {
        v = ((double)(2 * l - 1) * c * p1 - sqrt((double)((l - 1) * (l - 1) - m * m)) * p2) / sqrt((double)(l * l - m * m));
        p2 = p1;
        p1 = v;
      }
      const double sl = svec[(size_t)b * nl + l];
      sm_v[l * l + l + m] = sl * fc * v;
      if (m > 0) sm_v[l * l + l - m] = sl * fs * v;
    }
  }
  __syncthreads();
  const double scale = sc[(size_t)b * 2 * P + k];
  for (int n = tid; n < N; n += 256) {
    const int l = l_of[n], w = 2 * l + 1, base = l * l;
    const double *B = Rx90 + blk[l] + (n - base);
    double acc = 0.0;
    for (int i = 0; i < w; ++i) acc += sm_v[base + i] * B[i * w];
    Tb[(size_t)n * SM_TK + k] = scale * acc;
  }
}

// e1[b][n] = sum_k sqrt(w_k) T[b][n][k] at the m = 0 entries, zero elsewhere.  grid B
__global__ __launch_bounds__(256) void sm_first_kernel(int N, int P, const int32_t *__restrict__ m_of,
                                                       const double *__restrict__ sc, const double *__restrict__ T,
                                                       double *__restrict__ e1) {
  const int b = blockIdx.x;
  const double *sq = sc + (size_t)b * 2 * P + P;
  for (int n = threadIdx.x; n < N; n += 256) {
    double acc = 0.0;
    if (m_of[n] == 0) {
      const double *row = T + ((size_t)b * N + n) * SM_TK;
      for (int k = 0; k < P; ++k) acc += sq[k] * row[k];
    }
    e1[(size_t)b * N + n] = acc;
  }
}

// Ez[b] = Proj(M[b]) + (n - 1) e1 e1^T + diag(eps), ez[b] = sqrt(n) e1.  grid (ceil(N^2 / 256), B)
__global__ __launch_bounds__(256) void sm_finish_kernel(int N, const int32_t *__restrict__ m_of,
                                                        const int32_t *__restrict__ mirror, const double *__restrict__ M,
                                                        const double *__restrict__ e1, const double *__restrict__ scal,
                                                        double epsy, double epsy15, double *__restrict__ ez,
                                                        double *__restrict__ Ez) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (e >= (long)N * N) return;
  const int i = (int)(e / N), j = (int)(e - (long)i * N);
  const double *Mb = M + (size_t)b * N * N, *eb = e1 + (size_t)b * N;
  const int mi = m_of[i], mj = m_of[j];
  double v = 0.0;
  if (mi == mj || mi == -mj) {
    // (the two entries in the order of the smaller row index: a symmetric M gives a symmetric Ez to the bit)
    const int ii = mirror[i], jj = mirror[j];
    const double x = Mb[(size_t)i * N + j], y = Mb[(size_t)ii * N + jj];
    v = mi == mj ? 0.5 * (x + y) : 0.5 * (x - y);
  }
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  v += (scal[4 * b + 2] - 1.0) * (eb[lo] * eb[hi]);
  if (i == j) v += i >= 15 * 15 ? epsy15 : epsy;
  Ez[(size_t)b * N * N + e] = v;
  if (j == 0) ez[(size_t)b * N + i] = scal[4 * b + 1] * eb[i];
}

inline size_t sm_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int sp_set_size_basis(sp_handle *h, const double *theta_host, const double *Bp_host, int spts, double sfac) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !theta_host || !Bp_host || spts < 2 || spts > 16384 || !(sfac > 0.0)) return SP_ERR_INVALID;
  SP_HIP(hipSetDevice(h->device));
  const size_t nl = (size_t)h->ydeg + 1, n = (nl + 1) * (size_t)spts;
  if (h->d_size_basis) {
    SP_HIP(hipDeviceSynchronize());
    SP_HIP(hipFree(h->d_size_basis));
    h->d_size_basis = nullptr;
    h->size_spts = 0;
  }
  SP_HIP(hipMalloc((void **)&h->d_size_basis, sizeof(double) * n));
  SP_HIP(hipMemcpy(h->d_size_basis, Bp_host, sizeof(double) * nl * spts, hipMemcpyHostToDevice));
  SP_HIP(hipMemcpy(h->d_size_basis + nl * spts, theta_host, sizeof(double) * spts, hipMemcpyHostToDevice));
  h->size_spts = spts;
  h->size_sfac = sfac;
  return SP_OK;
}

int sp_polar_moments_samples(sp_handle *h, int B, const double *samples_host, double epsy, double epsy15,
                             double *ez_dev, double *Ez_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !samples_host || !ez_dev || !Ez_dev || B < 0 || B > 65535) return SP_ERR_INVALID;
  if (!h->d_size_basis) return SP_ERR_STATE;
  if (B == 0) return SP_OK;
  const int N = h->N, nl = h->ydeg + 1, nq = h->ydeg + 2, P = 2 * nq, spts = h->size_spts;
  if (P > SM_TK) return SP_ERR_INVALID;
  for (int b = 0; b < B; ++b) {
    const double *s = samples_host + 5 * (size_t)b;
    // r in [0, pi/2], alpha, beta > 0 (Beta law), c finite, n >= 0 (size.py:68, latitude.py:176-197, contrast.py:21-33)
    if (!(s[0] >= 0.0 && s[0] <= 1.5707963267948966 + 1e-6) || !(s[1] > 0.0) || !(s[2] > 0.0) || !std::isfinite(s[1]) ||
        !std::isfinite(s[2]) || !std::isfinite(s[3]) || !(s[4] >= 0.0) || !std::isfinite(s[4]))
      return SP_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  SP_HIP(hipSetDevice(h->device));
  // scratch: svec [B][nl] | cs [B nq][2] | sc [B][2][P] | scal [B][4] | T [B][N][64] | M [B][N][N] | e1 [B][N]
  size_t off = 0;
  auto take = [&](size_t doubles) { size_t o = off; off += sm_align(sizeof(double) * doubles); return o; };
  const size_t oS = take((size_t)B * nl), oC = take((size_t)B * nq * 2),
               oSc = take((size_t)B * 2 * P), oSl = take((size_t)B * 4), oT = take((size_t)B * N * SM_TK),
               oM = take((size_t)B * N * N), oE = take((size_t)B * N);
  void *ws = nullptr;
  int rc = ensure_big_scratch(h, off, &ws);
  if (rc) return rc;
  auto at = [&](size_t o) { return reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + o); };
  double *svec = at(oS), *cs = at(oC), *sc = at(oSc), *scal = at(oSl), *T = at(oT), *M = at(oM),
         *e1 = at(oE);
  // ONE staged upload: the samples
  const size_t need = 5 * (size_t)B;
  sp_handle::CsSlot *cp = nullptr;
  if ((rc = sp_stage_acquire(h, need, &cp))) return rc;
  sp_handle::CsSlot &c = *cp;
  memcpy(c.host, samples_host, sizeof(double) * need);
  SP_HIP(hipMemcpyAsync(c.dev, c.host, sizeof(double) * need, hipMemcpyHostToDevice, st));
  const size_t lds1 = sizeof(double) * ((size_t)spts + 4 * nq + 128);      // (+ 256 ints of flags)
  if (lds1 > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sm_prepare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds1);
  hipLaunchKernelGGL(sm_prepare_kernel, dim3(B), dim3(256), lds1, st, h->ydeg, spts, h->size_sfac, h->d_size_basis, c.dev,
                     svec, cs, sc, scal);
  SP_LAUNCH_CHECK();
  SP_HIP(hipEventRecord(c.done, st));
  c.used = true;
  hipLaunchKernelGGL(sm_rows_kernel, dim3(SM_TK, B), dim3(256), sizeof(double) * N, st, h->ydeg, N, P, h->d_l_of, h->d_blk,
                     svec, cs, h->d_Rx90, sc, T);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(sm_first_kernel, dim3(B), dim3(256), 0, st, N, P, h->d_m_of, sc, T, e1);
  SP_LAUNCH_CHECK();
  if ((rc = sp_launch_gemm_nt(T, SM_TK, (long)N * SM_TK, T, SM_TK, (long)N * SM_TK, M, N, (long)N * N, N, N, SM_TK, 1.0, 0,
                              0, B, st)))
    return rc;
  hipLaunchKernelGGL(sm_finish_kernel, dim3((unsigned)(((long)N * N + 255) / 256), B), dim3(256), 0, st, N, h->d_m_of,
                     h->d_mirror, M, e1, scal, epsy, epsy15, ez_dev, Ez_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

}  // extern "C"
