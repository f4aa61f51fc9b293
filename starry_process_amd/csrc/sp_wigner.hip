// Wigner rotation kernels for gfx950:
//   Rx            packed real rotation matrices about x   
//   dotRx         M . blockdiag(R^l)                       (row a3)
//   tensordotRz   M . Rz(theta_k)                          (row a12)
//   special       inclination-marginal second moment       (row a9)
//
// This file is compiled with -ffp-contract=off: the recursion below performs
// the same IEEE operations, in the same order, as the reference's host code
// (which is built for baseline x86-64, no FMA), so R comes out bit-identical
// given the same cos/sin of the angle.
#include "sp_internal.h"

// element (r, c), signed orders r, c in [-l, l], of a degree-l block
#define DEL(b, l, r, c) ((b)[((r) + (l)) * (2 * (l) + 1) + ((c) + (l))])

namespace {

__device__ __forceinline__ int icos4(int k) {  // cos(k pi/2)
  const int t = k & 3;
  return t == 0 ? 1 : (t == 2 ? -1 : 0);
}
__device__ __forceinline__ int isin4(int k) {  // sin(k pi/2)
  const int t = k & 3;
  return t == 1 ? 1 : (t == 3 ? -1 : 0);
}

// One workgroup per angle.  LDS holds three rolling (2*ydeg+1)^2 blocks of the
// complex d-matrix and three of its theta-derivative.
template <bool WITH_DERIV>
__global__ __launch_bounds__(256) void rx_kernel(int ydeg,
                                                 const double *__restrict__ cs,
                                                 const int32_t *__restrict__ blk,
                                                 int nwig, double *__restrict__ Rout,
                                                 double *__restrict__ dRout) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int bs = (2 * ydeg + 1) * (2 * ydeg + 1);
  double *Dbuf[3] = {lds, lds + bs, lds + 2 * bs};
  // (without the derivative the launch asks for the three value buffers only: twice the workgroups per CU)
  double *Pbuf[3] = {WITH_DERIV ? lds + 3 * bs : lds, WITH_DERIV ? lds + 4 * bs : lds, WITH_DERIV ? lds + 5 * bs : lds};
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const double c2 = cs[2 * blockIdx.x], s2 = cs[2 * blockIdx.x + 1];
  const double c2p = -s2, s2p = c2;
  double *R = Rout + (size_t)blockIdx.x * nwig;
  double *Rp = WITH_DERIV ? dRout + (size_t)blockIdx.x * nwig : nullptr;
  const double r2 = sqrt(2.0);

  // degrees 0 and 1 (wigner.h:162-204): buffers 0 and 1
  if (tid == 0) {
    double *D0 = Dbuf[0], *D1 = Dbuf[1], *P0 = Pbuf[0], *P1 = Pbuf[1];
    D0[0] = 1.0;
    D1[8] = 0.5 * (1.0 + c2);
    D1[7] = -s2 / r2;
    D1[6] = 0.5 * (1.0 - c2);
    D1[5] = -D1[7];
    D1[4] = D1[8] - D1[6];
    D1[3] = D1[7];
    D1[2] = D1[6];
    D1[1] = D1[5];
    D1[0] = D1[8];
    if (WITH_DERIV) {      // (without the derivative the P buffers do not exist: rx_kernel's launch)
      P0[0] = 0.0;
      P1[8] = 0.5 * c2p;
      P1[7] = -s2p / r2;
      P1[6] = -0.5 * c2p;
      P1[5] = -P1[7];
      P1[4] = P1[8] - P1[6];
      P1[3] = P1[7];
      P1[2] = P1[6];
      P1[1] = P1[5];
      P1[0] = P1[8];
    }
    R[0] = 1.0;
    if (WITH_DERIV) Rp[0] = 0.0;
    if (ydeg >= 1) {
      double *Q = R + 1;
      Q[0] = D1[8] - D1[6];
      Q[1] = -r2 * D1[5];
      Q[2] = 0;
      Q[3] = -r2 * D1[7];
      Q[4] = D1[4];
      Q[5] = 0;
      Q[6] = 0;
      Q[7] = 0;
      Q[8] = D1[8] + D1[6];
      if (WITH_DERIV) {
        double *Qp = Rp + 1;
        Qp[0] = P1[8] - P1[6];
        Qp[1] = -r2 * P1[5];
        Qp[2] = 0;
        Qp[3] = -r2 * P1[7];
        Qp[4] = P1[4];
        Qp[5] = 0;
        Qp[6] = 0;
        Qp[7] = 0;
        Qp[8] = P1[8] + P1[6];
      }
    }
  }
  __syncthreads();

  double tg;
  if (fabs(s2) < 1.0e-14)  // SP_WIGNER_TOL (constants.h:69-71)
    tg = s2;
  else
    tg = (1.0 - c2) / s2;

  for (int l = 2; l <= ydeg; ++l) {
    const double *A2 = Dbuf[(l - 2) % 3], *A2p = Pbuf[(l - 2) % 3];
    const double *A1 = Dbuf[(l - 1) % 3], *A1p = Pbuf[(l - 1) % 3];
    double *A = Dbuf[l % 3], *Ap = Pbuf[l % 3];

    // (a) row m' = l: corners then the recurrence in m (wigner.h:54-71).
    //     Sequential, done by the last thread while the others do (b).
    if (tid == nthr - 1) {
      const double a11 = DEL(A1, l - 1, l - 1, l - 1);
      const double a10 = DEL(A1, l - 1, l - 1, 1 - l);
      DEL(A, l, l, l) = 0.5 * a11 * (1.0 + c2);
      DEL(A, l, l, -l) = 0.5 * a10 * (1.0 - c2);
      if (WITH_DERIV) {
        const double a11p = DEL(A1p, l - 1, l - 1, l - 1);
        const double a10p = DEL(A1p, l - 1, l - 1, 1 - l);
        DEL(Ap, l, l, l) = 0.5 * (a11p * (1.0 + c2) - a11 * s2);
        DEL(Ap, l, l, -l) = 0.5 * (a10p * (1.0 - c2) + a10 * s2);
      }
      for (int m = l - 1; m >= 1 - l; --m) {
        const double rt = sqrt((double)(l + m + 1) / (l - m));
        const double nxt = DEL(A, l, l, m + 1);
        DEL(A, l, l, m) = -tg * rt * nxt;
        if (WITH_DERIV) {
          const double nxtp = DEL(Ap, l, l, m + 1);
          DEL(Ap, l, l, m) = -rt * (nxt / (1.0 + c2) + tg * nxtp);
        }
      }
    }
    // (b) rows m' = 0..l-1, |m| <= m' (wigner.h:73-106): l^2 independent entries
    {
      const int al = l, al1 = l - 1, tal1 = 2 * l - 1;
      const double ali = 1.0 / al1;
      const double cosaux = c2 * al * al1;
      for (int e = tid; e < l * l; e += nthr) {
        const int mp = (int)sqrt((double)e);
        const int mpf = (mp + 1) * (mp + 1) <= e ? mp + 1 : (mp * mp > e ? mp - 1 : mp);
        const int m = e - mpf * mpf - mpf;
        const int laux = l + mpf, lbux = l - mpf;
        const double aux = ali / sqrt((double)(laux * lbux));
        const double cux = sqrt((double)((laux - 1) * (lbux - 1))) * al;
        const int lauz = l + m, lbuz = l - m;
        const double auz = 1.0 / sqrt((double)(lauz * lbuz));
        const double fact = aux * auz;
        const double p1 = DEL(A1, l - 1, mpf, m);
        const double cm = cosaux - (double)(m * mpf);
        double term = tal1 * cm * p1;
        double termp = 0.0;
        if (WITH_DERIV) {
          const double p1p = DEL(A1p, l - 1, mpf, m);
          termp = tal1 * (-s2 * al * al1 * p1 + cm * p1p);
        }
        if (lbuz != 1 && lbux != 1) {
          const double cuz = sqrt((double)((lauz - 1) * (lbuz - 1)));
          term = term - DEL(A2, l - 2, mpf, m) * cux * cuz;
          if (WITH_DERIV) termp = termp - DEL(A2p, l - 2, mpf, m) * cux * cuz;
        }
        DEL(A, l, mpf, m) = fact * term;
        if (WITH_DERIV) DEL(Ap, l, mpf, m) = fact * termp;
      }
    }
    __syncthreads();
    // (c) reflection (wigner.h:113-125): (m', m) with m = 1..l, -m <= m' < m
    //     takes (-1)^(m+m') d[m, m'].  l(l+1) independent entries.
    for (int e = tid; e < l * (l + 1); e += nthr) {
      // m is the smallest integer with m(m+1) > e
      int m = (int)((sqrt(4.0 * e + 1.0) - 1.0) * 0.5) + 1;
      while (m * (m - 1) > e) --m;
      while (m * (m + 1) <= e) ++m;
      const int mp = e - m * (m - 1) - m;
      const double sg = ((m + mp) & 1) ? -1.0 : 1.0;
      DEL(A, l, mp, m) = sg * DEL(A, l, m, mp);
      if (WITH_DERIV) DEL(Ap, l, mp, m) = sg * DEL(Ap, l, m, mp);
    }
    __syncthreads();
    // (d) inversion (wigner.h:127-138): every (m', m) with m' + m < 0 takes
    //     (-1)^(m+m') d[-m', -m]; sources have m' + m > 0 (set in a-c).
    {
      const int w = 2 * l + 1;
      for (int e = tid; e < w * w; e += nthr) {
        const int mp = -l + e / w;
        const int m = -l + e % w;
        if (mp + m < 0) {
          const double sg = ((m + mp) & 1) ? -1.0 : 1.0;
          DEL(A, l, mp, m) = sg * DEL(A, l, -mp, -m);
          if (WITH_DERIV) DEL(Ap, l, mp, m) = sg * DEL(Ap, l, -mp, -m);
        }
      }
    }
    __syncthreads();
    // (e) complex -> real (wigner.h:225-271), written straight to HBM
    {
      double *Q = R + blk[l];
      double *Qp = WITH_DERIV ? Rp + blk[l] : nullptr;
      if (tid == 0) {
        DEL(Q, l, 0, 0) = DEL(A, l, 0, 0);
        if (WITH_DERIV) DEL(Qp, l, 0, 0) = DEL(Ap, l, 0, 0);
      }
      for (int e = tid; e < l * l; e += nthr) {
        const int mp = 1 + e / l, m = 1 + e % l;
        // alpha = -pi/2: cos/sin(mp * alpha); gamma = +pi/2: cos/sin(m * gamma)
        const int cosmal = icos4(mp), sinmal = -isin4(mp);
        const int cosmga = icos4(m), sinmga = isin4(m);
        const int sign = (mp & 1) ? -1 : 1;
        const int cosag = cosmal * cosmga - sinmal * sinmga;
        const int cosagm = cosmal * cosmga + sinmal * sinmga;
        const int sinag = sinmal * cosmga + cosmal * sinmga;
        const int sinagm = sinmal * cosmga - cosmal * sinmga;
        {
          const double d1 = DEL(A, l, -mp, -m);
          const double d2 = sign * DEL(A, l, mp, -m);
          DEL(Q, l, mp, m) = d1 * cosag + d2 * cosagm;
          DEL(Q, l, mp, -m) = -d1 * sinag + d2 * sinagm;
          DEL(Q, l, -mp, m) = d1 * sinag + d2 * sinagm;
          DEL(Q, l, -mp, -m) = d1 * cosag - d2 * cosagm;
          if (m == 1) {
            DEL(Q, l, mp, 0) = r2 * DEL(A, l, 0, mp) * cosmal;
            DEL(Q, l, -mp, 0) = r2 * DEL(A, l, 0, mp) * sinmal;
          }
          if (mp == 1) {
            DEL(Q, l, 0, m) = r2 * DEL(A, l, m, 0) * cosmga;
            DEL(Q, l, 0, -m) = -r2 * DEL(A, l, m, 0) * sinmga;
          }
        }
        if (WITH_DERIV) {
          const double d1 = DEL(Ap, l, -mp, -m);
          const double d2 = sign * DEL(Ap, l, mp, -m);
          DEL(Qp, l, mp, m) = d1 * cosag + d2 * cosagm;
          DEL(Qp, l, mp, -m) = -d1 * sinag + d2 * sinagm;
          DEL(Qp, l, -mp, m) = d1 * sinag + d2 * sinagm;
          DEL(Qp, l, -mp, -m) = d1 * cosag - d2 * cosagm;
          if (m == 1) {
            DEL(Qp, l, mp, 0) = r2 * DEL(Ap, l, 0, mp) * cosmal;
            DEL(Qp, l, -mp, 0) = r2 * DEL(Ap, l, 0, mp) * sinmal;
          }
          if (mp == 1) {
            DEL(Qp, l, 0, m) = r2 * DEL(Ap, l, m, 0) * cosmga;
            DEL(Qp, l, 0, -m) = -r2 * DEL(Ap, l, m, 0) * sinmga;
          }
        }
      }
    }
    __syncthreads();
  }
}

// out[b][r][n] = sum_i M[b](r, l^2 + i) * R[b][blk[l] + i (2l+1) + (n - l^2)]
// (rt: against the TRANSPOSED blocks, R[b][blk[l] + (n - l^2) (2l+1) + i] -- a rotation by the opposite
//  angle, R(-theta) = R(theta)^T, without a second Wigner recursion)
__global__ __launch_bounds__(256) void dotrx_kernel(
    int N, const int32_t *__restrict__ l_of, const int32_t *__restrict__ blk,
    const double *__restrict__ M, long strideM, long rs, long cs,
    const double *__restrict__ Rpk, long strideR, double *__restrict__ out,
    int rows, int rt) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  const int b = blockIdx.z;
  if (n >= N) return;
  const int l = l_of[n];
  const int w = 2 * l + 1, base = l * l;
  const double *Mr = M + (size_t)b * strideM + (size_t)r * rs;
  const double *B = Rpk + (size_t)b * strideR + blk[l] + (rt ? (n - base) * w : (n - base));
  const int bs = rt ? 1 : w;
  double acc = 0.0;
  for (int i = 0; i < w; ++i) acc += Mr[(size_t)(base + i) * cs] * B[i * bs];
  out[((size_t)b * rows + r) * N + n] = acc;
}

// Polar-frame moments in ONE launch (flux.py:54-62):
//     ez = R^T mu,    Ez = R^T (Sigma + mu mu^T) R,    R = blockdiag(R^l_x(pi/2)).
// One workgroup per pair of degrees (l1, l2): its (2 l1 + 1) x (2 l2 + 1) block of
// the second moment goes through LDS twice (S R_l2, then R_l1^T .), so the
// intermediate never touches HBM.  The same workgroups copy mu / Sigma into the
// handle's resident buffers.  Same summation order as three dotrx_kernel passes.
__global__ __launch_bounds__(1024) void polar_moments_kernel(
    int N, const int32_t *__restrict__ blk, const double *__restrict__ Rpk,
    const double *__restrict__ mu, const double *__restrict__ cov, double *__restrict__ mu_dst,
    double *__restrict__ cov_dst, double *__restrict__ ez, double *__restrict__ Ez) {
  extern __shared__ __attribute__((aligned(16))) double pm_lds[];
  const int nmax = 2 * ((int)gridDim.x - 1) + 1, LD = nmax + 1;
  double *sS = pm_lds, *sT = sS + nmax * LD, *sR1 = sT + nmax * LD, *sR2 = sR1 + nmax * nmax;
  const int l1 = blockIdx.y, l2 = blockIdx.x;
  const int n1 = 2 * l1 + 1, n2 = 2 * l2 + 1, b1 = l1 * l1, b2 = l2 * l2;
  const int tid = threadIdx.x;
  for (int e = tid; e < n1 * n1; e += 1024) sR1[e] = Rpk[blk[l1] + e];
  for (int e = tid; e < n2 * n2; e += 1024) sR2[e] = Rpk[blk[l2] + e];
  for (int e = tid; e < n1 * n2; e += 1024) {
    const int i = e / n2, j = e % n2;
    const double c = cov[(size_t)(b1 + i) * N + b2 + j];
    if (cov_dst != cov) cov_dst[(size_t)(b1 + i) * N + b2 + j] = c;
    sS[i * LD + j] = c + mu[b1 + i] * mu[b2 + j];
  }
  if (l2 == 0 && mu_dst != mu)
    for (int i = tid; i < n1; i += 1024) mu_dst[b1 + i] = mu[b1 + i];
  __syncthreads();
  for (int e = tid; e < n1 * n2; e += 1024) {
    const int i = e / n2, j = e % n2;
    double acc = 0.0;
    for (int k = 0; k < n2; ++k) acc += sS[i * LD + k] * sR2[k * n2 + j];
    sT[i * LD + j] = acc;
  }
  if (l2 == 0)
    for (int i = tid; i < n1; i += 1024) {
      double acc = 0.0;
      for (int k = 0; k < n1; ++k) acc += mu[b1 + k] * sR1[k * n1 + i];
      ez[b1 + i] = acc;
    }
  __syncthreads();
  for (int e = tid; e < n1 * n2; e += 1024) {
    const int i = e % n1, j = e / n1;   // consecutive threads -> consecutive addresses of a row of Ez
    double acc = 0.0;
    for (int k = 0; k < n1; ++k) acc += sT[k * LD + j] * sR1[k * n1 + i];
    Ez[(size_t)(b2 + j) * N + b1 + i] = acc;
  }
}

// Chebyshev recurrence for cos(a th), sin(a th), a = 0..ydeg (wigner.h:305-316)
__device__ __forceinline__ void cheb_fill(int ydeg, double th, double *cn,
                                          double *sn) {
  double s1, c1;
  sincos(th, &s1, &c1);
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
}

// one workgroup per row k
__global__ __launch_bounds__(256) void tensordotrz_kernel(
    int ydeg, int N, const int32_t *__restrict__ m_of,
    const int32_t *__restrict__ mirror, const double *__restrict__ M,
    const double *__restrict__ theta, double *__restrict__ f) {
  __shared__ double cn[SP_MAX_YDEG + 1], sn[SP_MAX_YDEG + 1];
  const int k = blockIdx.x;
  if (threadIdx.x == 0) cheb_fill(ydeg, theta[k], cn, sn);
  __syncthreads();
  const double *Mk = M + (size_t)k * N;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const int m = m_of[n];
    const double cm = cn[m < 0 ? -m : m];
    const double sm = m < 0 ? -sn[-m] : sn[m];
    f[(size_t)k * N + n] = Mk[n] * cm + Mk[mirror[n]] * sm;
  }
}

// Reverse mode of tensordotRz (wigner.h:344-404), one workgroup per row k:
//   bM[k][n]   = bf[k][n] cos(m_n th) + bf[k][mirror(n)] sin(m_mirror(n) th)
//   btheta[k]  = sum_n m_n (M[k][mirror(n)] bf[k][n] cos(m_n th) - M[k][n] bf[k][n] sin(m_n th))
// (the reference scatters tmp_s into column mirror(n); this is the gather form).
__global__ __launch_bounds__(256) void tensordotrz_rev_kernel(
    int ydeg, int N, const int32_t *__restrict__ m_of, const int32_t *__restrict__ mirror,
    const double *__restrict__ M, const double *__restrict__ theta,
    const double *__restrict__ bf, double *__restrict__ bM, double *__restrict__ btheta) {
  __shared__ double cn[SP_MAX_YDEG + 1], sn[SP_MAX_YDEG + 1], red[4];
  const int k = blockIdx.x;
  if (threadIdx.x == 0) cheb_fill(ydeg, theta[k], cn, sn);
  __syncthreads();
  const double *Mk = M + (size_t)k * N, *bk = bf + (size_t)k * N;
  double part = 0.0;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const int m = m_of[n], a = m < 0 ? -m : m, nm = mirror[n];
    const double cm = cn[a];
    const double sm = m < 0 ? -sn[a] : sn[a];
    const double tc = bk[n] * cm, ts = bk[n] * sm;
    // column n also receives tmp_s of its mirror, whose m is -m: sin(-m th) = -sm
    bM[(size_t)k * N + n] = tc + bk[nm] * (-sm);
    part += m * (Mk[nm] * tc - Mk[n] * ts);
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) btheta[k] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Reverse mode of special_tensordotRz (wigner.h:464-531).
//   coef kernel (one workgroup):  Ac[a] = sum_k bf_k cos(a th_k),  As[a] = sum_k bf_k sin(a th_k)
//   bM kernel (one workgroup per row n):
//       bM[n][c] = sign(m_n) As[|m_n|] T[n][mirror(c)] + Ac[|m_n|] T[n][c]
//   btheta kernel: with r1 / r2 the row sums of special_rows_kernel,
//       btheta_k = bf_k sum_n ( -m_n sin(m_n th_k) r1[n] + m_n cos(m_n th_k) r2[n] )
__global__ __launch_bounds__(256) void special_rev_coef_kernel(
    int ydeg, int K, const double *__restrict__ theta, const double *__restrict__ bf,
    double *__restrict__ Ac, double *__restrict__ As) {
  __shared__ double red[4];
  double pc[SP_MAX_YDEG + 1], ps[SP_MAX_YDEG + 1];
  for (int a = 0; a <= ydeg; ++a) pc[a] = ps[a] = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) {
    double cn[SP_MAX_YDEG + 1], sn[SP_MAX_YDEG + 1];
    cheb_fill(ydeg, theta[k], cn, sn);
    const double b = bf[k];
    for (int a = 0; a <= ydeg; ++a) {
      pc[a] += b * cn[a];
      ps[a] += b * sn[a];
    }
  }
  for (int a = 0; a <= ydeg; ++a) {
    for (int pass = 0; pass < 2; ++pass) {
      double v = pass ? ps[a] : pc[a];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) (pass ? As : Ac)[a] = (red[0] + red[1]) + (red[2] + red[3]);
    }
  }
}

__global__ __launch_bounds__(256) void special_rev_bM_kernel(
    int N, const int32_t *__restrict__ m_of, const int32_t *__restrict__ mirror,
    const double *__restrict__ T, const double *__restrict__ Ac, const double *__restrict__ As,
    double *__restrict__ bM) {
  const int n = blockIdx.x;
  const int m = m_of[n], a = m < 0 ? -m : m;
  const double cs = Ac[a], ss = m < 0 ? -As[a] : As[a];
  const double *Tn = T + (size_t)n * N;
  for (int c = threadIdx.x; c < N; c += 256) bM[(size_t)n * N + c] = ss * Tn[mirror[c]] + cs * Tn[c];
}

__global__ __launch_bounds__(256) void special_rev_btheta_kernel(
    int ydeg, int N, const int32_t *__restrict__ m_of, const double *__restrict__ r1,
    const double *__restrict__ r2, const double *__restrict__ theta,
    const double *__restrict__ bf, int K, double *__restrict__ btheta) {
  __shared__ double R1[SP_MAX_YDEG + 1], R2[SP_MAX_YDEG + 1];
  if (threadIdx.x <= ydeg) {
    // harmonic a collects the rows with |m_n| = a, in index order
    const int a = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
      const int m = m_of[n];
      if (m == a || m == -a) {
        s1 += r1[n];
        s2 += m * r2[n];
      }
    }
    R1[a] = s1;
    R2[a] = s2;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double cn[SP_MAX_YDEG + 1], sn[SP_MAX_YDEG + 1];
  cheb_fill(ydeg, theta[k], cn, sn);
  double acc = 0.0;
  for (int a = 0; a <= ydeg; ++a) acc += -a * sn[a] * R1[a] + cn[a] * R2[a];
  btheta[k] = bf[k] * acc;
}

// r1[n] = sum_j T[n,j] M[n,j];  r2[n] = sum_j T[n,j] M[n, mirror(j)]
// one wave per row n, 4 waves per workgroup
__global__ __launch_bounds__(256) void special_rows_kernel(
    int N, const int32_t *__restrict__ mirror, const double *__restrict__ T,
    const double *__restrict__ M, double *__restrict__ r1,
    double *__restrict__ r2) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  const double *Tn = T + (size_t)n * N, *Mn = M + (size_t)n * N;
  double a = 0.0, b = 0.0;
  for (int j = lane; j < N; j += 64) {
    const double t = Tn[j];
    a += t * Mn[j];
    b += t * Mn[mirror[j]];
  }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  if (lane == 0) {
    r1[n] = a;
    r2[n] = b;
  }
}

// f[k] = sum_a cos(a th_k) C_a + sin(a th_k) S_a with
//   C_a = sum_{|m_n| = a} r1[n],  S_a = sum_{m_n = a} r2[n] - sum_{m_n = -a} r2[n]
__global__ __launch_bounds__(256) void special_series_kernel(
    int ydeg, const double *__restrict__ r1, const double *__restrict__ r2,
    const double *__restrict__ theta, int K, double *__restrict__ f) {
  __shared__ double Ca[SP_MAX_YDEG + 1], Sa[SP_MAX_YDEG + 1];
  if ((int)threadIdx.x <= ydeg) {
    const int a = threadIdx.x;
    double c = 0.0, s = 0.0;
    for (int l = a; l <= ydeg; ++l) {
      const int n0 = l * l + l;
      if (a == 0) {
        c += r1[n0];
      } else {
        c += r1[n0 - a] + r1[n0 + a];
        s += r2[n0 + a] - r2[n0 - a];
      }
    }
    Ca[a] = c;
    Sa[a] = s;
  }
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  double s1, c1;
  sincos(theta[k], &s1, &c1);
  double cm2 = 1.0, sm2 = 0.0, cm1 = c1, sm1 = s1;
  double acc = Ca[0];
  if (ydeg >= 1) acc += cm1 * Ca[1] + sm1 * Sa[1];
  for (int a = 2; a <= ydeg; ++a) {
    const double cn = 2.0 * cm1 * c1 - cm2, sn = 2.0 * sm1 * c1 - sm2;
    acc += cn * Ca[a] + sn * Sa[a];
    cm2 = cm1;
    sm2 = sm1;
    cm1 = cn;
    sm1 = sn;
  }
  f[k] = acc;
}

}  // namespace

int sp_launch_Rx(sp_handle *h, const double *cs_dev, int n, double *R,
                 double *dR, hipStream_t st) {
  const int bs = (2 * h->ydeg + 1) * (2 * h->ydeg + 1);
  const size_t lds = (size_t)(dR ? 6 : 3) * bs * sizeof(double);
  if (dR)
    hipLaunchKernelGGL(rx_kernel<true>, dim3(n), dim3(256), lds, st, h->ydeg,
                       cs_dev, h->d_blk, h->NWIG, R, dR);
  else
    hipLaunchKernelGGL(rx_kernel<false>, dim3(n), dim3(256), lds, st, h->ydeg,
                       cs_dev, h->d_blk, h->NWIG, R, dR);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_dotRx(sp_handle *h, const double *M, long strideM, long rs,
                    long cs, int rows, const double *R, long strideR,
                    double *out, int batch, hipStream_t st, int transposeR) {
  if (rows <= 0 || batch <= 0) return SP_OK;
  if (rows > 65535 || batch > 65535) return SP_ERR_INVALID;
  dim3 grid((h->N + 255) / 256, rows, batch);
  hipLaunchKernelGGL(dotrx_kernel, grid, dim3(256), 0, st, h->N, h->d_l_of,
                     h->d_blk, M, strideM, rs, cs, R, strideR, out, rows, transposeR);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_polar_moments(sp_handle *h, const double *mu_src, const double *cov_src,
                            hipStream_t st) {
  const int nmax = 2 * h->ydeg + 1;
  const size_t lds = sizeof(double) * (2 * (size_t)nmax * (nmax + 1) + 2 * (size_t)nmax * nmax);
  if (lds > 160 * 1024) return SP_ERR_INVALID;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(polar_moments_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(polar_moments_kernel, dim3(h->ydeg + 1, h->ydeg + 1), dim3(1024), lds, st, h->N,
                     h->d_blk, h->d_Rx90, mu_src, cov_src, h->d_mean_ylm, h->d_cov_ylm, h->d_ez,
                     h->d_Ez);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

extern "C" {

int sp_dotRx(sp_handle *h, const double *M_dev, long strideM, long rs, long cs,
             int rows, const double *Rpacked_dev, long strideR, double *out_dev,
             int batch, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !M_dev || !Rpacked_dev || !out_dev || rows < 0 || batch < 0)
    return SP_ERR_INVALID;
  return sp_launch_dotRx(h, M_dev, strideM, rs, cs, rows, Rpacked_dev, strideR,
                         out_dev, batch, (hipStream_t)stream);
}

int sp_tensordotRz(sp_handle *h, const double *M_dev, const double *theta_dev,
                   int K, double *f_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !M_dev || !theta_dev || !f_dev || K < 0) return SP_ERR_INVALID;
  if (K == 0) return SP_OK;
  hipLaunchKernelGGL(tensordotrz_kernel, dim3(K), dim3(256), 0,
                     (hipStream_t)stream, h->ydeg, h->N, h->d_m_of, h->d_mirror,
                     M_dev, theta_dev, f_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_special_tensordotRz(sp_handle *h, const double *T_dev,
                           const double *M_dev, const double *theta_dev, int K,
                           double *f_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !T_dev || !M_dev || !theta_dev || !f_dev || K < 0)
    return SP_ERR_INVALID;
  if (K == 0) return SP_OK;
  hipStream_t st = (hipStream_t)stream;
  double *r1 = h->d_scratch, *r2 = h->d_scratch + h->N;
  hipLaunchKernelGGL(special_rows_kernel, dim3((h->N + 3) / 4), dim3(256), 0,
                     st, h->N, h->d_mirror, T_dev, M_dev, r1, r2);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(special_series_kernel, dim3((K + 255) / 256), dim3(256), 0,
                     st, h->ydeg, r1, r2, theta_dev, K, f_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

// ---- reverse mode -----------------------------------------
int sp_tensordotRz_rev(sp_handle *h, const double *M_dev, const double *theta_dev, int K,
                       const double *bf_dev, double *bM_dev, double *btheta_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !M_dev || !theta_dev || !bf_dev || !bM_dev || !btheta_dev || K < 0)
    return SP_ERR_INVALID;
  if (K == 0) return SP_OK;
  hipLaunchKernelGGL(tensordotrz_rev_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, h->ydeg,
                     h->N, h->d_m_of, h->d_mirror, M_dev, theta_dev, bf_dev, bM_dev, btheta_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_special_tensordotRz_rev(sp_handle *h, const double *T_dev, const double *M_dev,
                               const double *theta_dev, int K, const double *bf_dev,
                               double *bM_dev, double *btheta_dev, void *stream) {
  if (h && h->device < 0) return SP_ERR_NO_DEVICE;
  if (!h || !T_dev || !M_dev || !theta_dev || !bf_dev || !bM_dev || !btheta_dev || K < 0)
    return SP_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int N = h->N;
  double *r1 = h->d_scratch, *r2 = r1 + N, *Ac = r2 + N, *As = Ac + (SP_MAX_YDEG + 1);
  hipLaunchKernelGGL(special_rev_coef_kernel, dim3(1), dim3(256), 0, st, h->ydeg, K, theta_dev,
                     bf_dev, Ac, As);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(special_rev_bM_kernel, dim3(N), dim3(256), 0, st, N, h->d_m_of, h->d_mirror,
                     T_dev, Ac, As, bM_dev);
  SP_LAUNCH_CHECK();
  if (K == 0) return SP_OK;
  hipLaunchKernelGGL(special_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, h->d_mirror,
                     T_dev, M_dev, r1, r2);
  SP_LAUNCH_CHECK();
  hipLaunchKernelGGL(special_rev_btheta_kernel, dim3((K + 255) / 256), dim3(256), 0, st, h->ydeg,
                     N, h->d_m_of, r1, r2, theta_dev, bf_dev, K, btheta_dev);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

}  // extern "C"
