// Inclination-marginalised kernel table.
//
// For one flux operator rTA1 (one limb-darkening vector) this produces what
// FluxIntegral._compute builds before it interpolates (flux.py:181-231,
// 295-330): the scalar mean and variance, the second moment on the lag grid
// and the cubic-interpolation coefficients.
//
// The reference evaluates the lag-grid second moment as two dense
// (K' x N)(N x N) products followed by a row sum (wigner.h:409-459, 80 MFLOP).
// Algebraically the row sum commutes with the products, leaving
//     f_k = sum_a cos(a x_k) C_a + sin(a x_k) S_a ,  a = 0..ydeg,
// with C_a, S_a sums of the N row-reductions r1, r2 of W o Ez -- O(N^2 + K' ydeg)
// work.  One workgroup per table; everything but the N x N reads stays in LDS.
#include "sp_internal.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// sum over the 256 threads of the block; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// rows of W o Ez: grid (ceil(N/4), ntab, nsets), one wavefront per row n; set b = blockIdx.z has its own polar moments
// (Ez + b N^2, ez + b N: the hyperparameter samples of a batch, sp_kernel_table_samples) and writes table b ntab + blockIdx.y
//   r1[n] = sum_j W[n,j] Ez[n,j],  r2[n] = sum_j W[n,j] Ez[n,mirror(j)],
//   W[n,j] = Wnp[n,j] * rTA1[m0(l_n)] * rTA1[m0(l_j)]          (flux.py:199-209)
// and the row's term of the first moment (flux.py:196-198, 297-300), which table_finish_kernel used to take
// with one thread per n walking its 2 l + 1 entries of wnp one memory round trip after the other (8 of its 14 us):
//   m1[n] = (sum_r rTA1[l^2 + r] wnp[l][r][n - l^2]) ez[n]
__global__ __launch_bounds__(256) void table_rows_kernel(
    int N, const int32_t *__restrict__ l_of, const int32_t *__restrict__ mirror,
    const double *__restrict__ Wnp, const double *__restrict__ Ez,
    const double *__restrict__ rta1_all, double *__restrict__ rows_all,
    const int32_t *__restrict__ blk, const double *__restrict__ wnp, const double *__restrict__ ez) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const double *rta1 = rta1_all + (size_t)blockIdx.y * N;
  Ez += (size_t)blockIdx.z * N * N;
  ez += (size_t)blockIdx.z * N;
  const int ln = l_of[n];
  const double rn = rta1[ln * ln + ln];
  const double *Wn = Wnp + (size_t)n * N, *En = Ez + (size_t)n * N;
  double a = 0.0, b = 0.0;
  for (int j = lane; j < N; j += 64) {
    const int lj = l_of[j];
    const double wv = Wn[j] * (rn * rta1[lj * lj + lj]);
    a += wv * En[j];
    b += wv * En[mirror[j]];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = 2 * ln + 1;
  double m1 = lane < w ? rta1[ln * ln + lane] * wnp[blk[ln] + lane * w + (n - ln * ln)] : 0.0;   // (2 ydeg + 1 <= 64)
  m1 = wave_sum(m1);
  if (lane == 0) {
    double *rows = rows_all + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * 3 * N;
    rows[n] = a;
    rows[N + n] = b;
    rows[2 * N + n] = m1 * ez[n];
  }
}

// one workgroup per table: mean, variance, harmonics, lag grid, spline.  grid (ntab, nsets): table blockIdx.y ntab + blockIdx.x
__global__ __launch_bounds__(256) void table_finish_kernel(
    int ydeg, int N, const int32_t *__restrict__ l_of, const int32_t *__restrict__ blk,
    const double *__restrict__ wnp, const double *__restrict__ ez,
    const double *__restrict__ rta1_all, const double *__restrict__ rows_all,
    int covpts, const double *__restrict__ xp, double *__restrict__ tab_all,
    double *__restrict__ meanvar_all) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *s_rta1 = lds;             // N
  double *s_r1 = s_rta1 + N;        // N
  double *s_r2 = s_r1 + N;          // N
  double *s_yp = s_r2 + N;          // covpts + 4
  double *s_ca = s_yp + covpts + 4; // ydeg + 1
  double *s_sa = s_ca + ydeg + 1;   // ydeg + 1
  double *s_red = s_sa + ydeg + 1;  // 4
  const int tid = threadIdx.x;
  const int np = covpts + 4;
  const size_t it = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const double *rta1 = rta1_all + (size_t)blockIdx.x * N;
  const double *rows = rows_all + it * 3 * N;
  double *tab = tab_all + it * 5 * np;

  for (int n = tid; n < N; n += 256) {
    s_rta1[n] = rta1[n];
    s_r1[n] = rows[n];
    s_r2[n] = rows[N + n];
  }
  __syncthreads();

  // first moment: w[l] = rTA1[l-block] . wnp[l];  mean = sum_l w[l] . ez[l-block]  (terms by table_rows_kernel)
  double part = 0.0;
  for (int n = tid; n < N; n += 256) part += rows[2 * N + n];
  const double mean = block_sum(part, s_red);

  // variance = <W, Ez> - mean^2 (flux.py:305-308)
  double pv = 0.0;
  for (int n = tid; n < N; n += 256) pv += s_r1[n];
  const double wez = block_sum(pv, s_red);
  if (tid == 0) {
    meanvar_all[2 * it] = mean;
    meanvar_all[2 * it + 1] = wez - mean * mean;
  }

  // harmonic coefficients
  if (tid <= ydeg) {
    const int a = tid;
    double c = 0.0, s = 0.0;
    for (int l = a; l <= ydeg; ++l) {
      const int n0 = l * l + l;
      if (a == 0) {
        c += s_r1[n0];
      } else {
        c += s_r1[n0 - a] + s_r1[n0 + a];
        s += s_r2[n0 + a] - s_r2[n0 - a];
      }
    }
    s_ca[a] = c;
    s_sa[a] = s;
  }
  __syncthreads();

  // second moment on the lag grid minus mean^2 (flux.py:317-320)
  const double mean2 = mean * mean;
  for (int k = tid; k < np; k += 256) {
    double s1, c1;
    sincos(xp[k], &s1, &c1);
    double cm2 = 1.0, sm2 = 0.0, cm1 = c1, sm1 = s1;
    double acc = s_ca[0];
    if (ydeg >= 1) acc += cm1 * s_ca[1] + sm1 * s_sa[1];
    for (int a = 2; a <= ydeg; ++a) {
      const double cn = 2.0 * cm1 * c1 - cm2, sn = 2.0 * sm1 * c1 - sm2;
      acc += cn * s_ca[a] + sn * s_sa[a];
      cm2 = cm1;
      sm2 = sm1;
      cm1 = cn;
      sm1 = sn;
    }
    const double y = acc - mean2;
    s_yp[k] = y;
    tab[k] = y;
  }
  __syncthreads();

  // cubic interpolant (flux.py:322-330)
  for (int i = tid; i < np; i += 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (i <= covpts) {
      const double y0 = s_yp[i], y1 = s_yp[i + 1], y2 = s_yp[i + 2],
                   y3 = s_yp[i + 3];
      a0 = y1;
      a1 = -y0 / 3.0 - 0.5 * y1 + y2 - y3 / 6.0;
      a2 = 0.5 * (y0 + y2) - y1;
      a3 = 0.5 * ((y1 - y2) + (y3 - y0) / 3.0);
    }
    tab[np + i] = a0;
    tab[2 * np + i] = a1;
    tab[3 * np + i] = a2;
    tab[4 * np + i] = a3;
  }
}

}  // namespace

// nsets == 0: the handle's resident polar moments (sp_kernel_table); nsets >= 1: ez [nsets][N], Ez [nsets][N][N] given
// (sp_kernel_table_samples), tables [nsets ntab]
int sp_launch_kernel_table(sp_handle *h, const double *rta1_dev, int ntab,
                           int covpts, const double *xp_dev, double *tab_dev,
                           double *meanvar_dev, hipStream_t st, int nsets, const double *ez_dev,
                           const double *Ez_dev) {
  const double *ez = nsets ? ez_dev : h->d_ez, *Ez = nsets ? Ez_dev : h->d_Ez;
  const int nb = nsets ? nsets : 1;
  if ((long)ntab * nb > 65535 || nb > 65535) return SP_ERR_INVALID;
  const size_t lds =
      sizeof(double) * ((size_t)3 * h->N + covpts + 4 + 2 * (h->ydeg + 1) + 4);
  if (lds > 150 * 1024) return SP_ERR_INVALID;
  // row-reduction scratch [ntab][2][N], grown on demand (rare: new ntab)
  const size_t need = sizeof(double) * (size_t)ntab * nb * 3 * h->N;
  if (h->tab_scratch_bytes < need) {
    SP_HIP(hipDeviceSynchronize());
    if (h->d_tab_scratch) SP_HIP(hipFree(h->d_tab_scratch));
    h->d_tab_scratch = nullptr;
    h->tab_scratch_bytes = 0;
    SP_HIP(hipMalloc((void **)&h->d_tab_scratch, need));
    h->tab_scratch_bytes = need;
  }
  // (the attribute is per device: remembered per handle, a handle lives on one device)
  if (!h->table_attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(table_finish_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    h->table_attr_done = true;
  }
  hipLaunchKernelGGL(table_rows_kernel, dim3((h->N + 3) / 4, ntab, nb), dim3(256), 0, st,
                     h->N, h->d_l_of, h->d_mirror, h->d_Wnp, Ez, rta1_dev,
                     h->d_tab_scratch, h->d_blk, h->d_wnp, ez);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(table_finish_kernel, dim3(ntab, nb), dim3(256), lds, st, h->ydeg, h->N,
                     h->d_l_of, h->d_blk, h->d_wnp, ez, rta1_dev, h->d_tab_scratch,
                     covpts, xp_dev, tab_dev, meanvar_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
