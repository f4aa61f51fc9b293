// Batched fp64 Cholesky, triangular solves and the likelihood reduction
//  for gfx950.
//
// Blocked factorisation, panels of SP_NB = 64 columns (driver:
// sp_launch_cholesky_groups below):
//   diag_block (sp_diag.h)  factors the 64x64 diagonal block (and writes L_d^T
//                  with the reciprocal diagonal for the solve);
//   trsm_quad_kernel        the panel solve X = P L_d^-T by substitution, four
//                  lanes per row (fp64 VALU has the MFMA's peak on gfx950 and
//                  substitution needs half the flops of a product with L_d^-1);
//   sp_launch_gemm_nt (sp_gemm.hip) does everything else on the matrix cores:
//                  the left-looking block-column updates and the rank-64w
//                  trailing updates.
// The systems are padded to a multiple of 64 rows and carry the residual
// vectors as EXTRA ROWS below the matrix: factoring
//     [ C   . ]          gives          [ L   . ]
//     [ r^T . ]                         [ y^T . ] ,   y = L^-1 r,
// so r^T C^-1 r = |y|^2 falls out of the factorisation and no separate
// triangular solve is needed for the likelihood.
#include <cstdio>
#include <vector>

#include "sp_internal.h"

#include "sp_tile.h"
#include "sp_paneldiag.h"

#define DLD 65  // padded row length of a diagonal block in cho_solve_kernel

namespace {

// Stand-alone diagonal-block kernel: one workgroup per star (used for the first
// panel of every super-panel; the other panels get their diagonal block from the
// fused tile-(0,0) workgroup of the block-column update, sp_gemm.hip).
template <bool TIMED, bool INV = false>
__global__ __launch_bounds__(256) void diag_kernel(double *__restrict__ sys, long ld,
                                                   long stride, int c0, int nact,
                                                   double *__restrict__ invL_all, long lts,
                                                   int32_t *__restrict__ info,
                                                   long long *__restrict__ dbg) {
  __shared__ __attribute__((aligned(16))) double lds[SP_DIAG_LDS_DOUBLES];
  long long ts[5], tc[5];
  if (TIMED) { ts[0] = wall_clock64(); tc[0] = clock64(); }
  double *sD = lds, *sRd = lds + 64 * BLD;
  double *Mx = sys + (size_t)blockIdx.x * stride;
  const int tid = threadIdx.x;
  // stage the block; outside the active nact x nact part use the identity so a
  // partial last panel factors as diag(L_act, I)
  {
    const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      const double *src = Mx + (size_t)(c0 + r) * ld + c0 + cj;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cj + e;
        double v = (r < nact && c < nact) ? src[e] : (r == c ? 1.0 : 0.0);
        if (c > r) v = 0.0;
        sD[r * BLD + c] = v;
      }
    }
  }
  __syncthreads();
  if (TIMED) { ts[1] = wall_clock64(); tc[1] = clock64(); }
  const int notpd = diag_block(sD, sRd, invL_all + (size_t)blockIdx.x * lts,
                               TIMED ? dbg + 8 + 40 * blockIdx.x + 0 : nullptr, threadIdx.x,
                               (INV && SP_PANEL_MFMA_SOLVE == 1) ? invL_all + (size_t)blockIdx.x * lts + SP_LT_IMG
                                                                 : nullptr);
  if (notpd && info) info[blockIdx.x] = 1;
  if (TIMED) { ts[2] = wall_clock64(); tc[2] = clock64(); }
  {
    const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      double *dst = Mx + (size_t)(c0 + r) * ld + c0 + cj;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cj + e;
        if (c <= r && r < nact) dst[e] = sD[r * BLD + c];
      }
    }
  }
  if (INV) {
    // L_d^-T behind the image: the one-launch-per-panel kernel solves with it (sp_gemm.hip)
#if SP_PANEL_MFMA_SOLVE == 2
    diag_solve_operand(sD, sRd, invL_all + (size_t)blockIdx.x * lts + SP_LT_IMG);
#endif
  }
  if (TIMED) {
    __syncthreads();
    ts[3] = wall_clock64();
    tc[3] = clock64();
    if (threadIdx.x == 0) {
      for (int i = 0; i < 4; ++i) {
        dbg[blockIdx.x * 40 + i] = ts[i];
        dbg[blockIdx.x * 40 + 4 + i] = tc[i];
      }
    }
  }
}

// Panel solve  X = P L_d^-T  (rows below the diagonal block, in place) by forward
// substitution on the vector ALU.  Four lanes share a row: lane q of the quad
// holds the columns 8 i + 2 q + {0, 1}, i = 0..7, so that the quad's 16-byte
// loads cover 64 contiguous bytes of the row.  With the row pre-scaled by
// 1 / L_cc and  lt[k][c] = L_ck / L_cc  (diag_block's output), step k is: a
// quad-permute DPP broadcasts x_k from its owner lane and every lane updates
// its remaining columns,  x_c -= x_k lt[k][c]  -- lt is staged in LDS (zero on
// and left of the diagonal) and read as 16-byte pairs two steps ahead of use.
// 576 fused multiply-adds per lane; the dependent chain per step is
// FMA -> DPP -> FMA.  No barrier after the staging one.
__global__ __launch_bounds__(256) void trsm_quad_kernel(double *sys, long ld, long stride,
                                                        int r1, int c0, int nrows,
                                                        const double *__restrict__ LT_all, long lts,
                                                        int batch, int ntiles, int neager,
                                                        double *__restrict__ inv_all) {
  __shared__ __attribute__((aligned(16))) double sLT[64 * 64 + 64];
  // XCD-aware decode as in sp_gemm.hip (sp_tile.h)
  int mtx, tile;
  if (!sp_xcd_decode(blockIdx.x, batch, ntiles + (inv_all ? 1 : 0), mtx, tile)) return;
  const int tid = threadIdx.x;
  const int lrow = tile * 64 + (tid >> 2), q = tid & 3;
  // inv_all: one more tile per star whose rows are those of the identity -- solved like any
  // other, it is L_d^-T (row k, column n: (L_d^-1)[n][k]), the operand with which the strip
  // solves (sp_strip.hip) apply this block on the matrix cores; off the chain, beside the others
  const bool ident = tile == ntiles;
  const bool valid = ident || lrow < nrows;
  double *prow = ident ? inv_all + (size_t)mtx * lts + (size_t)(tid >> 2) * 64 + 2 * q
                       : sys + (size_t)mtx * stride + (size_t)(r1 + (valid ? lrow : 0)) * ld + c0 + 2 * q;
  LtRegs lt;
  lt_load(lt, LT_all + (size_t)mtx * lts);
  double x[16];
  if (ident) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[2 * i] = (8 * i + 2 * q == (tid >> 2)) ? 1.0 : 0.0;
      x[2 * i + 1] = (8 * i + 2 * q + 1 == (tid >> 2)) ? 1.0 : 0.0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const d2v v = *reinterpret_cast<const d2v *>(prow + 8 * i);
      x[2 * i] = v.x;
      x[2 * i + 1] = v.y;
    }
  }
  lt_store(lt, sLT, sLT + 4096);
  __syncthreads();
  quad_solve_store(x, sLT, sLT + 4096, prow, valid);
  if (ident || tile >= neager) return;
  // Eager update of a coming diagonal block.  The first `neager` row tiles of this panel are
  // the rows of the pivot blocks still to be factored before the next trailing update reaches
  // them; each takes its share D_ii -= X_i X_i^T now, from the rows it has just solved, so
  // that when block i's turn comes its workgroup finds it up to date and factors it at once
  // (skip00 in gemm_nt_kernel).  The rank-64 product costs this workgroup ~2 us inside a
  // launch that is bound by the other tiles' traffic; it used to cost the workgroup on the
  // critical path the whole left-looking product, one operand slice latency after another.
  constexpr int XW = 65;   // padded row: 64 * 65 doubles = the L_d^T image + diagonal, reused
  __syncthreads();
  double *sX = sLT;
  {
    double *row = sX + (tid >> 2) * XW + 2 * q;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      row[8 * i] = x[2 * i];
      row[8 * i + 1] = x[2 * i + 1];
    }
  }
  __syncthreads();
  // (fetching the tile before the solve instead, to hide its latency, costs a wavefront of
  //  occupancy -- 156 VGPRs -- and measures no faster)
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
  const int d0 = r1 + tile * 64;
  double *D = sys + (size_t)mtx * stride + (size_t)d0 * ld + d0;
  d4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      acc[n][r] = D[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * n + fr];
  const double *pa = sX + (16 * wave + fr) * XW + fk;
  const double *pb = sX + fr * XW + fk;
#pragma unroll
  for (int kk = 0; kk < 64; kk += 4) {
    const double a = -pa[kk];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[kk], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[16 * XW + kk], acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[32 * XW + kk], acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[48 * XW + kk], acc[3], 0, 0, 0);
  }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      D[(size_t)(16 * wave + fk + 4 * r) * ld + 16 * n + fr] = acc[n][r];
}

// lnlike = -1/2 sum_m |y_m|^2 - M sum_i log L_ii - K M / 2 log(2 pi)
// (sp.py:1157-1188).  One workgroup per star.
//
// coef != nullptr: deferred normalisation (sp_assemble.hip, defer_finish_kernel).  The factored
// matrix is B'' = Sigma + N / c1 and the true covariance is
//     C = c1 (B'' + d_p p p^T + d_1 1 1^T + d_q q q^T),
// with y_p, y_q, y_1 = L''^-1 p, q, 1 in the three rows below the residuals.  The matrix
// determinant lemma and the Sherman-Morrison formula, one rank at a time (the two non-negative
// terms first), give log det C and r^T C^-1 r from the Gram matrix of those rows and the
// residuals'; a rank-1 step whose pivot 1 + d u^T B^-1 u is not positive means C is not positive
// definite: the same -inf the reference's failed Cholesky gives (math.py:82-91, sp.py:1186-1188).
struct RedCoef {   // = Coef of sp_assemble.hip
  double c1, dp, dq, z, gpmean, m, mu, d1;
};

__global__ __launch_bounds__(256) void lnlike_reduce_kernel(
    const double *__restrict__ sys, long ld, long stride, int K, int M,
    const int32_t *__restrict__ info, double *__restrict__ lnlike,
    uint32_t *__restrict__ status, uint32_t *__restrict__ status_out,
    const sp_star *__restrict__ stars, const RedCoef *__restrict__ coef) {
  __shared__ double red[4][12];
  const int s = blockIdx.x;
  const double *Mx = sys + (size_t)s * stride;
  const int wave = threadIdx.x >> 6;
  // sums of v[0 .. 12) over the workgroup, in every thread (all twelve always: constant indices
  // keep v in registers; the unused ones are zero)
  auto block_sum = [&](double (&v)[12]) {
#pragma unroll
    for (int a = 0; a < 12; ++a)
      for (int off = 32; off > 0; off >>= 1) v[a] += __shfl_down(v[a], off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int a = 0; a < 12; ++a) red[wave][a] = v[a];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] = (red[0][a] + red[1][a]) + (red[2][a] + red[3][a]);
  };
  double v[12];
  for (int a = 0; a < 12; ++a) v[a] = 0.0;
  const double *yp = Mx + (size_t)(K + M) * ld, *yq = yp + ld, *y1 = yq + ld;
  const double *y0 = Mx + (size_t)K * ld;       // the first light curve's residuals ride in the same pass
  for (int i = threadIdx.x; i < K; i += 256) {
    v[0] += log(Mx[(size_t)i * ld + i]);
    const double r = y0[i];
    v[7] += r * r;
    if (coef) {
      const double a = yp[i], b = yq[i], c = y1[i];
      v[1] += a * a; v[2] += a * b; v[3] += a * c; v[4] += b * b; v[5] += b * c; v[6] += c * c;
      v[8] += r * a; v[9] += r * c; v[10] += r * b;
    }
  }
  block_sum(v);
  const double logdet = v[0];
  // rank-1 steps on the 3 x 3 Gram matrix H (0 = p, 1 = 1, 2 = q): factor f_k and old column c_k
  double f[3] = {0.0, 0.0, 0.0}, col[3][3], logs = 0.0;
  bool notpd = false;
  double c1 = 1.0;
  if (coef) {
    const RedCoef rc = coef[s];
    c1 = rc.c1;
    double H[3][3] = {{v[1], v[3], v[2]}, {v[3], v[6], v[5]}, {v[2], v[5], v[4]}};
    const double d[3] = {rc.dp, rc.d1, rc.dq};
    for (int k = 0; k < 3; ++k) {
      for (int a = 0; a < 3; ++a) col[k][a] = H[a][k];
      if (d[k] == 0.0) continue;
      const double piv = 1.0 + d[k] * H[k][k];
      if (!(piv > 0.0)) notpd = true;
      logs += log(piv);
      f[k] = d[k] / piv;
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) H[a][b] -= f[k] * col[k][a] * col[k][b];
    }
  }
  double quad = 0.0;
  for (int m = 0; m < M; ++m) {
    const double *y = Mx + (size_t)(K + m) * ld;
    double w[12];
    for (int a = 0; a < 12; ++a) w[a] = 0.0;
    if (m == 0) {
      w[0] = v[7]; w[1] = v[8]; w[2] = v[9]; w[3] = v[10];
    } else {
      for (int k = threadIdx.x; k < K; k += 256) {
        const double r = y[k];
        w[0] += r * r;
        if (coef) {
          w[1] += r * yp[k];
          w[2] += r * y1[k];
          w[3] += r * yq[k];
        }
      }
      block_sum(w);
    }
    double g = w[0], h[3] = {w[1], w[2], w[3]};
    for (int k = 0; k < 3; ++k) {
      if (f[k] == 0.0) continue;
      const double hk = h[k];
      g -= f[k] * hk * hk;
      for (int a = 0; a < 3; ++a) h[a] -= f[k] * hk * col[k][a];
    }
    quad += g;
  }
  if (threadIdx.x == 0) {
    // (ragged ensembles: the padding rows have unit pivots and zero residuals, only
    //  the constants know the number of valid cadences)
    const int nobs = (stars && stars[s].nobs > 0 && stars[s].nobs < K) ? stars[s].nobs : K;
    double val = -0.5 * quad / c1;
    val -= M * (logdet + 0.5 * nobs * log(c1) + 0.5 * logs);
    val -= 0.5 * nobs * M * 1.8378770664093453;  // log(2 pi)
    uint32_t st = status ? status[s] : 0u;
    if ((info && info[s]) || notpd) st |= SP_STAR_NOT_PD;
    if (val != val) st |= SP_STAR_NAN;
    if (st & (SP_STAR_NOT_PD | SP_STAR_ZMAX | SP_STAR_NAN)) val = -INFINITY;
    lnlike[s] = val;
    if (status) status[s] = st;
    if (status_out) status_out[s] = st;
  }
}

// copy a batch of K x K matrices into zero/identity padded Kp x Kp systems
__global__ __launch_bounds__(256) void pad_in_kernel(const double *__restrict__ A,
                                                     int K, long lda, long strideA,
                                                     double *__restrict__ sys, int Kp,
                                                     long strideS, int M,
                                                     const double *__restrict__ resid) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Kp) return;
  double v = 0.0;
  if (i < K && j < K)
    v = A[(size_t)s * strideA + (size_t)i * lda + j];
  else if (i >= K && i < K + M && j < K && resid)
    v = resid[((size_t)s * M + (i - K)) * K + j];
  else if (i == j)
    v = 1.0;
  sys[(size_t)s * strideS + (size_t)i * Kp + j] = v;
}

// copy the lower factor back, zero the strict upper triangle, NaN on failure
__global__ __launch_bounds__(256) void pad_out_kernel(const double *__restrict__ sys,
                                                      int Kp, long strideS,
                                                      double *__restrict__ A, int K,
                                                      long lda, long strideA,
                                                      const int32_t *__restrict__ info) {
  const int s = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= K) return;
  double v = j <= i ? sys[(size_t)s * strideS + (size_t)i * Kp + j] : 0.0;
  if (info && info[s]) v = __builtin_nan("");
  A[(size_t)s * strideA + (size_t)i * lda + j] = v;
}

// x = (L L^T)^-1 b (math.py:97-100).  grid (nrhs, batch), one right-hand side
// per workgroup; b is [K, nrhs] row-major per matrix.  Blocked substitution:
// the 64x64 diagonal blocks are solved by one wavefront with lane shuffles, the
// off-diagonal updates are one row (forward) / one column (backward) per thread.
//
// mode: 0 both sweeps, 1 forward only (L y = b), 2 backward only (L^T x = b) -- the two
// Solve ops of math.py:97-100 taken one at a time, which their reverse mode needs
// (math.py:40-72).  Element (i, rhs) of a right-hand side lives at B[i * rs + rhs * cs].
__global__ __launch_bounds__(256) void cho_solve_kernel(
    const double *__restrict__ Lall, int K, long ldl, long strideL,
    double *__restrict__ Ball, long strideB, long rs, long cs, int mode) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int Kr = ((K + 63) / 64) * 64;
  double *x = lds;        // Kr
  double *sL = lds + Kr;  // 64 * DLD
  const double *L = Lall + (size_t)blockIdx.y * strideL;
  double *B = Ball + (size_t)blockIdx.y * strideB + (size_t)blockIdx.x * cs;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < Kr; i += 256) x[i] = i < K ? B[(size_t)i * rs] : 0.0;
  const int nb = Kr / 64;
  __syncthreads();
  // forward: L y = b
  for (int blk = 0; blk < nb && mode != 2; ++blk) {
    const int c0 = blk * 64, n = K - c0 < 64 ? K - c0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      sL[r * DLD + c] = (c0 + r < K && c <= r) ? L[(size_t)(c0 + r) * ldl + c0 + c]
                                               : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = x[c0 + lane];
      for (int c = 0; c < n; ++c) {
        const double xc = __shfl(v, c, 64) / sL[c * DLD + c];
        if (lane == c)
          v = xc;
        else if (lane > c)
          v -= sL[lane * DLD + c] * xc;
      }
      x[c0 + lane] = v;
    }
    __syncthreads();
    for (int j = c0 + 64 + tid; j < K; j += 256) {
      const double *row = L + (size_t)j * ldl + c0;
      double acc = x[j];
      for (int k = 0; k < n; ++k) acc -= row[k] * x[c0 + k];
      x[j] = acc;
    }
    __syncthreads();
  }
  // backward: L^T x = y
  for (int blk = nb - 1; blk >= 0 && mode != 1; --blk) {
    const int c0 = blk * 64, n = K - c0 < 64 ? K - c0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      sL[r * DLD + c] = (c0 + r < K && c <= r) ? L[(size_t)(c0 + r) * ldl + c0 + c]
                                               : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = x[c0 + lane];
      for (int c = n - 1; c >= 0; --c) {
        const double xc = __shfl(v, c, 64) / sL[c * DLD + c];
        if (lane == c)
          v = xc;
        else if (lane < c)
          v -= sL[c * DLD + lane] * xc;
      }
      x[c0 + lane] = v;
    }
    __syncthreads();
    for (int j = tid; j < c0; j += 256) {
      double acc = x[j];
      for (int k = 0; k < n; ++k) acc -= L[(size_t)(c0 + k) * ldl + j] * x[c0 + k];
      x[j] = acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < K; i += 256) B[(size_t)i * rs] = x[i];
}

// out = in^T per matrix (K x K, row-major; out has leading dimension K), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ in, long ldi,
                                                        long stridei, double *__restrict__ out,
                                                        int K) {
  __shared__ double tile[64][65];
  const double *A = in + (size_t)blockIdx.z * stridei;
  double *O = out + (size_t)blockIdx.z * K * K;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
  for (int r = r4; r < 64; r += 4)
    tile[r][c] = (i0 + r < K && j0 + c < K) ? A[(size_t)(i0 + r) * ldi + j0 + c] : 0.0;
  __syncthreads();
  for (int r = r4; r < 64; r += 4)
    if (j0 + r < K && i0 + c < K) O[(size_t)(j0 + r) * K + i0 + c] = tile[c][r];
}

// keep one triangle of every K x K matrix, scale its diagonal:  A <- tri(A), diag *= dscale
// (tril / triu of Solve.L_op, math.py:66-69; tril_and_halve_diagonal of Cholesky.L_op)
__global__ __launch_bounds__(256) void tri_mask_kernel(double *__restrict__ A, int K, int upper,
                                                       double dscale) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)K * K) return;
  const int i = e / K, j = e % K;
  double *p = A + (size_t)blockIdx.y * K * K + e;
  if (i == j)
    *p *= dscale;
  else if ((j > i) != (upper != 0))
    *p = 0.0;
}

// C_bar = tril(S + S^T) - diag(S); all NaN when the factor itself is NaN (on_error = "nan")
__global__ __launch_bounds__(256) void chol_rev_finish_kernel(const double *__restrict__ S,
                                                              const double *__restrict__ L, long ldl,
                                                              long strideL, double *__restrict__ out,
                                                              int K) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)K * K) return;
  const int i = e / K, j = e % K;
  const double *Sb = S + (size_t)blockIdx.y * K * K;
  double v = 0.0;
  if (i > j)
    v = Sb[(size_t)i * K + j] + Sb[(size_t)j * K + i];
  else if (i == j)
    v = Sb[e];
  const double l00 = L[(size_t)blockIdx.y * strideL];
  if (l00 != l00) v = __builtin_nan("");
  out[(size_t)blockIdx.y * K * K + e] = v;
}

// Sustained fp64 MFMA rate of the device (debug phase 5): every wavefront issues
// `iters` x 8 independent v_mfma_f64_16x16x4_f64 from registers; clock64() /
// wall_clock64() give the shader clock actually held under that load.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, double *sink, long long *ts, int rnd) {
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  // operands with random mantissas (two alternating sets): data toggling as in a
  // real product, which is what the power management reacts to
  unsigned long long hsh = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xD1B54A32D192ED03ull;
  double av[2], bv[2];
  for (int i = 0; i < 2; ++i) {
    hsh ^= hsh >> 29; hsh *= 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 32;
    av[i] = rnd ? __longlong_as_double(0x3FE0000000000000ull | (hsh & 0xFFFFFFFFFFFFFull)) - 0.75 : 1.0 + threadIdx.x * 1e-9;
    hsh ^= hsh >> 29; hsh *= 0x94D049BB133111EBull; hsh ^= hsh >> 32;
    bv[i] = rnd ? __longlong_as_double(0x3FE0000000000000ull | (hsh & 0xFFFFFFFFFFFFFull)) - 0.75 : 1.0 - threadIdx.x * 1e-9;
  }
  const long long w0 = wall_clock64(), c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i & 1], bv[i & 1], acc[i % NACC], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 123.456) sink[0] = t;
  if (threadIdx.x == 0) {
    ts[2 * blockIdx.x] = w1 - w0;
    ts[2 * blockIdx.x + 1] = c1 - c0;
  }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------

// C[cfrom:rend, cfrom:rend] -= X X^T with X = columns c0..c0+kd-1 of the rows cfrom..rend-1;
// lower-triangle tiles only.  Timed for bench.py when profiling is on (kind SP_PROF_SYRK).
static int bulk_update(sp_handle *h, double *sys, long ld, long stride, int S, int c0,
                       int cfrom, int rend, int kd, hipStream_t st, long lts, int fuse_nact = 0,
                       double *invL = nullptr, int32_t *info = nullptr, int skip00 = 0,
                       int skip_tile00 = 0, const LazyCov *lazy = nullptr) {
  // fuse_nact > 0: tile (0, 0) is the diagonal block of the next panel and its
  // workgroup factors it on the spot (hidden behind the other tiles)
  const int n = rend - cfrom;
  if (n <= 0 || kd <= 0) return SP_OK;
  double *X = sys + (size_t)cfrom * ld + c0;
  double *T = sys + (size_t)cfrom * ld + cfrom;
  // algorithmic work of a symmetric rank-kd update of an n x n block:
  // n (n + 1) / 2 entries x kd multiply-adds
  SpProfScope prof(h, st, SP_PROF_SYRK, (double)S * (double)n * (n + 1) * kd);
  return fuse_nact > 0
             ? sp_launch_gemm_nt_diag(X, ld, stride, X, ld, stride, T, ld, stride, n, n, kd,
                                      -1.0, 1, S, fuse_nact, invL, lts, info, st, skip00)
             : sp_launch_gemm_nt(X, ld, stride, X, ld, stride, T, ld, stride, n, n, kd, -1.0,
                                 1, 1, S, st, skip_tile00, lazy);
}

// rows r1..rend-1 of panel column c0: X = P L_d^-T in place (LT = diag_block's output)
// neager: leading row tiles that also update their own diagonal block (trsm_quad_kernel)
static int launch_trsm(sp_handle *h, double *sys, long ld, long stride, int S, int r1, int c0,
                       int rend, const double *LT, long lts, hipStream_t st, int neager = 0,
                       double *inv_out = nullptr) {
  const int nrows = rend - r1 > 0 ? rend - r1 : 0;
  if (nrows <= 0 && !inv_out) return SP_OK;
  const int ntiles = (nrows + 63) / 64;
  const long nblk = sp_xcd_grid(S, ntiles + (inv_out ? 1 : 0));
  // substitution: 64 x 64 multiply-adds per row; each eager update a 64 x 64 x 64 product
  SpProfScope prof(h, st, SP_PROF_CHAIN,
                   (double)S * (2.0 * nrows * 64 * 64 / 2 + 2.0 * neager * 64 * 64 * 64 / 2));
  hipLaunchKernelGGL(trsm_quad_kernel, dim3((unsigned)nblk), dim3(256), 0, st, sys, ld, stride,
                     r1, c0, nrows, LT, lts, S, ntiles, neager, inv_out);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

static int launch_diag(sp_handle *h, double *sys, long ld, long stride, int S, int c0, int nact,
                       double *invL, long lts, int32_t *info, hipStream_t st, bool inverse = false) {
  SpProfScope prof(h, st, SP_PROF_CHAIN, (double)S * nact * (double)nact * nact / 3.0);
  if (inverse)
    hipLaunchKernelGGL((diag_kernel<false, true>), dim3(S), dim3(256), 0, st, sys, ld, stride, c0, nact,
                       invL, lts, info, nullptr);
  else
    hipLaunchKernelGGL((diag_kernel<false, false>), dim3(S), dim3(256), 0, st, sys, ld, stride, c0, nact,
                       invL, lts, info, nullptr);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

static int diag_and_solve(sp_handle *h, double *sys, long ld, long stride, int S, int K, int rend,
                          int j, int32_t *info, double *invL, long lts, hipStream_t st,
                          bool have_diag = false, int neager = 0) {
  const int c0 = j * SP_NB;
  const int nact = K - c0 < SP_NB ? K - c0 : SP_NB;
  // diagonal block: L_d (and L_d^T for the solve)
  if (!have_diag) {
    int rc = launch_diag(h, sys, ld, stride, S, c0, nact, invL, lts, info, st);
    if (rc != SP_OK) return rc;
  }
  // rows below the active block: X = P L_d^-T, in place
  return launch_trsm(h, sys, ld, stride, S, c0 + nact, c0, rend, invL, lts, st, neager);
}

// left-looking update of block column j (rows c0..rend-1) by the q panels from column cS on; the
// tile-(0,0) workgroup goes on to factor the diagonal block (image -> invL)
static int launch_blockcol_diag(sp_handle *h, const sp_chol_group &G, long ld, long stride, int c0,
                                int cS, int rend, int q, int nact, double *invL, long lts,
                                int skip00) {
  double *A = G.sys + (size_t)c0 * ld + cS;
  double *T = G.sys + (size_t)c0 * ld + c0;
  const double rows = rend - c0;
  SpProfScope prof(h, G.st, SP_PROF_CHAIN,
                   (double)G.S * (2.0 * rows * 64 * (q * 64.0) + 64.0 * 64 * 64 / 3));
  return sp_launch_gemm_nt_diag(A, ld, stride, A, ld, stride, T, ld, stride, rend - c0, SP_NB,
                                q * SP_NB, -1.0, 0, G.S, nact, invL, lts, G.info, G.st, skip00);
}

// ---- recursive driver (h->chol_mode == 2) -----------------------------------------------------
//
//   factor(b0, b1):  the diagonal block of the pivot blocks [b0, b1), rows and columns
//     more than 4 blocks:  factor(b0, bm);  X = A21 L11^-T  (strip kernel, sp_strip.hip);
//                          A22 -= X X^T  (one symmetric update);  factor(bm, b1)
//     else:  the panels one after the other, left-looking (block-column update fused with the
//            diagonal block, substitution solve with eager diagonal updates), rows of the block only
//
// Three quarters of the flops of a factorisation sit in the top-level strip solve and symmetric
// update, each ONE launch of long-lived workgroups; the latency-bound chain (16 diagonal blocks
// at K = 1000) works on 256-row blocks.  Every block keeps its own L_d^T image: the strip
// solves need the images of all the diagonal blocks of their triangle.
namespace {
struct RecCtx {
  sp_handle *h;
  const sp_chol_group *G;
  int K, Kp, nsteps;
  long ld, stride, lts;
};
// block j's slot: its L_d^T image, then L_d^-T
inline double *rec_img(const RecCtx &c, int j) { return c.G->invL + (size_t)j * 2 * SP_LT_IMG; }

int rec_base(const RecCtx &c, int b0, int b1) {
  const sp_chol_group &G = *c.G;
  const int rend = b1 * SP_NB, cS = b0 * SP_NB;
  for (int j = b0; j < b1; ++j) {
    const int q = j - b0, c0 = j * SP_NB;
    const int nact = c.K - c0 < SP_NB ? c.K - c0 : SP_NB;
    const int neager = b1 - 1 - j;   // the pivot blocks of this base block still to come
    int rc;
    if (q > 0)
      rc = launch_blockcol_diag(c.h, G, c.ld, c.stride, c0, cS, rend, q, nact, rec_img(c, j), c.lts, 1);
    else
      rc = launch_diag(c.h, G.sys, c.ld, c.stride, G.S, c0, nact, rec_img(c, j), c.lts, G.info, G.st);
    if (rc != SP_OK) return rc;
    rc = launch_trsm(c.h, G.sys, c.ld, c.stride, G.S, c0 + nact, c0, rend, rec_img(c, j), c.lts, G.st,
                     nact == SP_NB ? neager : 0, rec_img(c, j) + SP_LT_IMG);
    if (rc != SP_OK) return rc;
  }
  return SP_OK;
}

// rows [r0, r1) x column blocks [b0, b1):  X = A L^-T against the factored triangle of those blocks
int rec_trsm(const RecCtx &c, int r0, int r1, int b0, int b1) {
  const sp_chol_group &G = *c.G;
  const int nb = b1 - b0;
  if (r1 <= r0 || nb <= 0) return SP_OK;
  if (nb <= SP_STRIP_MAXB) {
    const double rows = r1 - r0, w = nb * 64.0;
    SpProfScope prof(c.h, G.st, SP_PROF_STRIP, (double)G.S * rows * w * w);
    return sp_launch_strip(G.sys, c.ld, c.stride, G.S, r0, (r1 - r0) / SP_NB, b0 * SP_NB, nb,
                           rec_img(c, b0) + SP_LT_IMG, c.lts, G.st);
  }
  const int bm = b0 + (nb + 1) / 2;
  int rc = rec_trsm(c, r0, r1, b0, bm);
  if (rc != SP_OK) return rc;
  {
    const double *A = G.sys + (size_t)r0 * c.ld + b0 * SP_NB;
    const double *B = G.sys + (size_t)bm * SP_NB * c.ld + b0 * SP_NB;
    double *C = G.sys + (size_t)r0 * c.ld + bm * SP_NB;
    SpProfScope prof(c.h, G.st, SP_PROF_STRIP,
                     (double)G.S * 2.0 * (r1 - r0) * ((b1 - bm) * 64.0) * ((bm - b0) * 64.0));
    rc = sp_launch_gemm_nt(A, c.ld, c.stride, B, c.ld, c.stride, C, c.ld, c.stride, r1 - r0,
                           (b1 - bm) * SP_NB, (bm - b0) * SP_NB, -1.0, 1, 0, G.S, G.st);
    if (rc != SP_OK) return rc;
  }
  return rec_trsm(c, r0, r1, bm, b1);
}

int rec_factor(const RecCtx &c, int b0, int b1) {
  const sp_chol_group &G = *c.G;
  const int nb = b1 - b0;
  if (nb <= 0) return SP_OK;
  if (nb <= c.h->rec_base) return rec_base(c, b0, b1);
  // split on a multiple of the base size where possible (equal halves at K = 1000)
  int bm = b0 + ((nb / 2 + c.h->rec_base - 1) / c.h->rec_base) * c.h->rec_base;
  if (bm >= b1) bm = b0 + nb / 2;
  int rc = rec_factor(c, b0, bm);
  if (rc != SP_OK) return rc;
  const int rend = b1 * SP_NB;
  if ((rc = rec_trsm(c, bm * SP_NB, rend, b0, bm)) != SP_OK) return rc;
  if ((rc = bulk_update(c.h, G.sys, c.ld, c.stride, G.S, b0 * SP_NB, bm * SP_NB, rend,
                        (bm - b0) * SP_NB, G.st, c.lts)) != SP_OK)
    return rc;
  return rec_factor(c, bm, b1);
}
}  // namespace

static int cholesky_recursive(sp_handle *h, int ngroups, const sp_chol_group *grp, int K, int Kp) {
  for (int g = 0; g < ngroups; ++g) {
    RecCtx c{h, &grp[g], K, Kp, (K + SP_NB - 1) / SP_NB, (long)Kp, (long)Kp * Kp, sp_lt_stride(Kp)};
    int rc = rec_factor(c, 0, c.nsteps);
    if (rc != SP_OK) return rc;
    // rows below the last pivot block (residual rows of a wide right-hand side): solved against
    // the whole factor
    if ((rc = rec_trsm(c, c.nsteps * SP_NB, Kp, 0, c.nsteps)) != SP_OK) return rc;
  }
  return SP_OK;
}

// ---- dataflow driver (h->chol_mode == 3): one launch per super-panel (sp_chain.hip) -------------
// super-panels of w pivot blocks; between two of them the rank-64w update of the trailing matrix
// (its tile (0, 0), the next pivot block, was completed and factored by the chain).
static int ensure_chain_mem(sp_handle *h, size_t ints) {
  if (h->chain_ints >= ints) return SP_OK;
  // (growing synchronises the device: launches of this handle may still read the old buffer)
  SP_HIP(hipDeviceSynchronize());
  if (h->chain_mem) SP_HIP(hipFree(h->chain_mem));
  h->chain_mem = nullptr;
  h->chain_ints = 0;
  hipError_t e = hipMalloc((void **)&h->chain_mem, ints * sizeof(int));
  if (e != hipSuccess) {
    sp_set_hip_error(e, "hipMalloc(chain flags)");
    return SP_ERR_ALLOC;
  }
  h->chain_ints = ints;
  return SP_OK;
}

static int cholesky_dataflow(sp_handle *h, const sp_chol_group &G, int K, int Kp, int w) {
  const long ld = Kp, stride = (long)Kp * Kp, lts = sp_lt_stride(Kp);
  const int nsteps = (K + SP_NB - 1) / SP_NB, ntile = Kp / SP_NB;
  const int nact_last = K - (nsteps - 1) * SP_NB;
  const int nlaunch = (nsteps + w - 1) / w;
  const size_t ints = sp_chain_mem_ints(G.S, ntile, nlaunch);
  int rc = ensure_chain_mem(h, ints);
  if (rc != SP_OK) return rc;
  SP_HIP(hipMemsetAsync(h->chain_mem, 0, ints * sizeof(int), G.st));
  int *flags = h->chain_mem, *tickets = flags + (size_t)G.S * 2 * ntile;
  int *abort_flag = tickets + 8 * (size_t)nlaunch;
  int launch = 0;
  for (int s0 = 0; s0 < nsteps; s0 += w, ++launch) {
    const int wq = nsteps - s0 < w ? nsteps - s0 : w;
    {
      // algorithmic work of the launch, panel by panel as the per-panel drivers count it:
      // left-looking product + substitution + eager rank-64 updates + diagonal blocks
      double fl = 0.0;
      int nd = 0;
      for (int q = 0; q < wq; ++q) {
        const double rows = Kp - (s0 + q + 1) * SP_NB;
        int last = s0 + w;
        if (last > nsteps - 1) last = nsteps - 1;
        const int neager = last > s0 + q ? last - (s0 + q) : 0;
        fl += 2.0 * rows * 64 * (q * 64.0) + rows * 64 * 64 + neager * 64.0 * 64 * 64;
      }
      nd = (s0 == 0 ? 1 : 0) + (s0 + wq < nsteps ? wq : wq - 1);
      fl += nd * 64.0 * 64 * 64 / 3;
      SpProfScope prof(h, G.st, SP_PROF_PANELS, (double)G.S * fl);
      rc = sp_launch_chain(G.sys, ld, stride, G.S, ntile, s0, wq, nsteps, nact_last, G.invL, lts,
                           flags, tickets + 8 * launch, abort_flag, G.info,
                           h->chain_dbg ? h->chain_dbg + (size_t)launch * ntile * 128 : nullptr, G.st);
      if (rc != SP_OK) return rc;
    }
    const int cE = (s0 + w) * SP_NB;
    if (cE < K) {
      rc = bulk_update(h, G.sys, ld, stride, G.S, s0 * SP_NB, cE, Kp, w * SP_NB, G.st, lts, 0, nullptr,
                       nullptr, 0, 1);
      if (rc != SP_OK) return rc;
    }
  }
  return SP_OK;
}

// ---- round-3 driver (h->panel2): ONE launch per panel (sp_panel.hip) ---------------------------------
// Super-panels of w pivot blocks as above.  Launch j = T items (left-looking product over the q
// panels of the super-panel before it, solve on the matrix cores, eager update of the coming
// diagonal tiles).  Who factors pivot block j (a 12-17 us latency chain):
//   j = 0                      a launch of its own (64 workgroups);
//   q = 0, later super-panels  the tile-(0, 0) workgroup of the trailing update (sp_launch_syrk_diag);
//   q > 0                      the workgroup that solved row tile j in launch j - 1, at its end (SP_PANEL_TAILD).
// Look-ahead (SP_PANEL_LA, default on): the first tile of launch j + 1 -- tile (j + 2, j + 1), whose solve
// and eager update stand between this launch and pivot block j + 2 -- is brought up to date with the
// column blocks s0 .. j - 1 by an item of THIS launch; launch j + 1 then only adds the rank-64 update
// with column block j: the first item's product leaves the critical path, and the diagonal block
// behind it runs under the products of the launch's other items.
static int cholesky_panel2(sp_handle *h, int ngroups, const sp_chol_group *grp, int K, int Kp, int w) {
  const long ld = Kp, stride = (long)Kp * Kp, lts = sp_lt_stride(Kp);
  const int nsteps = (K + SP_NB - 1) / SP_NB, ntile = Kp / SP_NB;
  static int la_on = -1;
  if (la_on < 0) {
    const char *e2 = getenv("SP_PANEL_LA");
    la_on = e2 ? atoi(e2) : 1;
  }
  auto nact_of = [&](int j) { return K - j * SP_NB < SP_NB ? K - j * SP_NB : SP_NB; };
  for (int s0 = 0; s0 < nsteps; s0 += w) {
    {
      SpProfScope sp_scope(ngroups == 1 ? h : nullptr, grp[0].st, SP_PROF_PANELS, 0.0, 0);
      for (int q = 0; q < w && s0 + q < nsteps; ++q) {
        const int j = s0 + q;
        int last = s0 + w;
        if (last > nsteps - 1) last = nsteps - 1;
        const int neager = last > j ? last - j : 0;
        const double rows = (double)(ntile - j - 1) * SP_NB;
        for (int g = 0; g < ngroups; ++g) {
          const sp_chol_group &G = grp[g];
          const LazyCov *lzp = (G.lazy.theta && s0 == 0) ? &G.lazy : nullptr;
          // algorithmic work as the per-panel drivers have always counted it: left-looking product,
          // triangular solve, eager rank-64 updates, the diagonal block
          const double fl = (double)G.S * (2.0 * rows * 64 * (q * 64.0) + rows * 64 * 64 +
                                           neager * 64.0 * 64 * 64 + 64.0 * 64 * 64 / 3);
          const bool d_alone = j == 0;                         // block 0: nobody before it
          // block j + 1 in the tail of this launch (same super-panel: the next one's first block
          // belongs to the trailing update)
          const bool tail = rows > 0 && j + 1 < nsteps && q + 1 < w;
          const bool la = la_on && q >= 1 && j + 2 < ntile && j + 1 < nsteps && q + 1 < w && rows > 0;
          const bool first_la = la_on && q >= 2 && rows > 0;   // (launch j - 1 qualified: q - 1 >= 1, j + 1 < ntile)
          const int nl = d_alone && rows > 0 ? 2 : 1;
          SpProfScope prof(h, G.st, SP_PROF_CHAIN, fl, nl);
          SpProfScope prof1(h, G.st, SP_PROF_PANEL_LAUNCH, fl, nl);
          sp_scope.add(fl, nl);
          int rc = SP_OK;
          if (d_alone)
            rc = sp_launch_panel2(G.sys, ld, stride, G.S, ntile, j, s0, nact_of(j), 0, last, SP_PANEL_D, h->ncu,
                                  G.invL, lts, G.info, G.st, nullptr);
          const int what = rows > 0 ? (SP_PANEL_T | (tail ? SP_PANEL_TAILD : 0) | (la ? SP_PANEL_LA : 0) |
                                       (first_la ? SP_PANEL_FIRSTLA : 0))
                                    : 0;
          if (rc == SP_OK && what)
            rc = sp_launch_panel2(G.sys, ld, stride, G.S, ntile, j, s0, nact_of(j), tail ? nact_of(j + 1) : 0,
                                  last, what, h->ncu, G.invL, lts, G.info, G.st, lzp);
          if (rc != SP_OK) return rc;
        }
      }
    }   // (sp_scope ends here: the trailing update has its own pair)
    const int jE = s0 + w, cE = jE * SP_NB;
    if (cE < K) {
      const int n = Kp - cE, kd = w * SP_NB, cS = s0 * SP_NB;
      for (int g = 0; g < ngroups; ++g) {
        const sp_chol_group &G = grp[g];
        LazyCov lzv = G.lazy;
        lzv.tr0 = lzv.tc0 = jE;
        DiagFuse df{G.sys, ld, stride, jE, nact_of(jE), G.invL, lts, G.info};
        SpProfScope prof(h, G.st, SP_PROF_SYRK, (double)G.S * (double)n * (n + 1) * kd);
        int rc = sp_launch_syrk_diag(G.sys + (size_t)cE * ld + cS, ld, stride, G.sys + (size_t)cE * ld + cE,
                                     n, kd, G.S, G.st, (G.lazy.theta && s0 == 0) ? &lzv : nullptr, &df);
        if (rc != SP_OK) return rc;
      }
    }
  }
  return SP_OK;
}

// In-place factorisation of S padded systems (Kp x Kp, ld = Kp): the leading
// K x K part is factored, rows K..Kp-1 only receive the triangular solve.
//
// chol_mode 2 (default): the recursive driver above.
// chol_mode 0 / 1: two-level blocking.  Panels (64 columns) are grouped in super-panels of w
// panels.  Inside a super-panel a block column is brought up to date
// left-looking (ONE narrow product over the q previous panels of the group,
// k = 64 q) just before it is factored; the big trailing matrix is touched once
// per super-panel with a rank-64w update instead of w rank-64 updates.  The
// trailing update is HBM-bound at k = 64 (8 flop per byte of C traffic,
// measured 4.0 TB/s, profiles/r01_*); k = 64 w divides that traffic by w.
int sp_launch_cholesky_groups(sp_handle *h, int ngroups, const sp_chol_group *grp, int K,
                              int Kp) {
  const long ld = Kp, stride = (long)Kp * Kp, lts = sp_lt_stride(Kp);
  const int nsteps = (K + SP_NB - 1) / SP_NB;
  if (h && h->chol_mode == 2 && h->fuse_diag && h->eager)
    return cholesky_recursive(h, ngroups, grp, K, Kp);
  // panels per super-panel: wider super-panels raise the arithmetic intensity of the
  // trailing update (k = 64 w) at the price of more left-looking work per block column;
  // measured with the eager diagonal updates: K = 1000 (16 panels) w = 2 / 4 / 6 / 8 /
  // 12 / 16 -> 1.17 / 1.10 / 1.085 / 1.08 / 1.12 / 1.14 ms per step; K = 3000 (47 panels): 8 best as well
  const int w = (h && h->superpanel > 0) ? h->superpanel : (nsteps >= 16 ? 8 : 4);
  if (h && h->chol_mode == 3 && ngroups == 1) return cholesky_dataflow(h, grp[0], K, Kp, w);
  if (h && h->onelaunch && h->panel2 && h->fuse_diag > 1 && h->eager)
    return cholesky_panel2(h, ngroups, grp, K, Kp, w);
  if (h && h->onelaunch && h->fuse_diag > 1 && h->eager) {
    // ONE launch per panel (sp_launch_panel): update + solve + eager diagonal updates + the
    // next diagonal block; the L_d^T images ping-pong between the two slots of a star
    for (int g = 0; g < ngroups; ++g) {
      const sp_chol_group &G = grp[g];
      int rc = launch_diag(h, G.sys, ld, stride, G.S, 0, K < SP_NB ? K : SP_NB, G.invL, lts, G.info,
                           G.st, SP_PANEL_MFMA_SOLVE != 0);
      if (rc != SP_OK) return rc;
    }
    for (int s0 = 0; s0 < nsteps; s0 += w) {
      const int cS = s0 * SP_NB;
      // one group (the usual case): the panel launches of the super-panel follow one another on
      // the stream and can share ONE pair of profiling events (kind SP_PROF_PANELS)
      {
      SpProfScope sp_scope(ngroups == 1 ? h : nullptr, grp[0].st, SP_PROF_PANELS, 0.0, 0);
      for (int q = 0; q < w && s0 + q < nsteps; ++q) {
        const int j = s0 + q, c0 = j * SP_NB;
        const int nact = K - c0 < SP_NB ? K - c0 : SP_NB;
        const int r1 = c0 + nact;
        int last = s0 + w;
        if (last > nsteps - 1) last = nsteps - 1;
        const int neager = last > j ? last - j : 0;
        const int c1 = c0 + SP_NB;
        const int next_nact = (j + 1 < nsteps) ? (K - c1 < SP_NB ? K - c1 : SP_NB) : 0;
        for (int g = 0; g < ngroups; ++g) {
          const sp_chol_group &G = grp[g];
          // (image, L_d^-T) pairs, two per star, used in turn
          const double *lt_in = G.invL + (size_t)(j & 1) * 2 * SP_LT_IMG;
          double *lt_out = G.invL + (size_t)((j + 1) & 1) * 2 * SP_LT_IMG;
          const double rows = Kp - r1;
          // first super-panel of a system whose assembly left the tiles below the diagonal to
          // their first touch (sp_cov.h): this launch's rows start at row tile j + 1 of block
          // column j (partial blocks: the tiles beyond them hold residual rows -- from memory)
          LazyCov lzv = G.lazy;
          lzv.tr0 = j + 1;
          lzv.tc0 = j;
          const LazyCov *lzp = (G.lazy.theta && s0 == 0) ? &lzv : nullptr;
          // left-looking product + substitution + eager rank-64 updates + the next diagonal block
          const double fl = (double)G.S * (2.0 * rows * 64 * (q * 64.0) + rows * 64 * 64 +
                                           neager * 64.0 * 64 * 64 + 64.0 * 64 * 64 / 3);
          SpProfScope prof(h, G.st, SP_PROF_CHAIN, fl);
          SpProfScope prof1(h, G.st, SP_PROF_PANEL_LAUNCH, fl, (nact < SP_NB && (j == 0 || SP_PANEL_MFMA_SOLVE)) ? 2 : 1);
          sp_scope.add(fl, (nact < SP_NB && (j == 0 || SP_PANEL_MFMA_SOLVE)) ? 2 : 1);
          int rc;
          if (nact < SP_NB) {
            // partial last block: the rows of its own diagonal tile below the active ones
            // (residual rows, padding) already carry every update -- the eager updates cover
            // the whole tile -- and are only solved; the rows beyond the tile get the product
            // (a block factored by the previous panel launch had them solved on the spot)
            rc = SP_OK;
            if (j == 0 || SP_PANEL_MFMA_SOLVE)
              rc = sp_launch_panel(G.sys + (size_t)r1 * ld + cS, ld, G.sys + (size_t)c0 * ld + cS, ld,
                                   G.sys + (size_t)r1 * ld + c0, ld, stride, c1 - r1, 0, G.S, lt_in,
                                   lt_out, lts, 0, 0, G.info, G.st);
            if (rc != SP_OK) return rc;
            rc = sp_launch_panel(G.sys + (size_t)c1 * ld + cS, ld, G.sys + (size_t)c0 * ld + cS, ld,
                                 G.sys + (size_t)c1 * ld + c0, ld, stride, Kp - c1, q * SP_NB, G.S,
                                 lt_in, lt_out, lts, 0, 0, G.info, G.st, lzp);
          } else {
            rc = sp_launch_panel(G.sys + (size_t)r1 * ld + cS, ld, G.sys + (size_t)c0 * ld + cS, ld,
                                 G.sys + (size_t)r1 * ld + c0, ld, stride, Kp - r1, q * SP_NB, G.S,
                                 lt_in, lt_out, lts, neager, next_nact, G.info, G.st, lzp);
          }
          if (rc != SP_OK) return rc;
        }
      }
      }   // (sp_scope ends here: the trailing update has its own pair)
      const int cE = (s0 + w) * SP_NB;
      if (cE < K) {
        for (int g = 0; g < ngroups; ++g) {
          const sp_chol_group &G = grp[g];
          LazyCov lzv = G.lazy;
          lzv.tr0 = lzv.tc0 = cE / SP_NB;
          int rc = bulk_update(h, G.sys, ld, stride, G.S, cS, cE, Kp, w * SP_NB, G.st, lts, 0,
                               nullptr, nullptr, 0, 1, (G.lazy.theta && s0 == 0) ? &lzv : nullptr);
          if (rc != SP_OK) return rc;
        }
      }
    }
    return SP_OK;
  }
  // launches are issued breadth-first over the groups so that the groups'
  // streams advance together (the host enqueues ~3-8 us per launch)
  for (int s0 = 0; s0 < nsteps; s0 += w) {
    const int cS = s0 * SP_NB;
    for (int q = 0; q < w && s0 + q < nsteps; ++q) {
      const int j = s0 + q, c0 = j * SP_NB;
      for (int g = 0; g < ngroups; ++g) {
        const sp_chol_group &G = grp[g];
        const int nact = K - c0 < SP_NB ? K - c0 : SP_NB;
        // pivot blocks after j that are factored before a trailing update reaches them
        // (the rest of this super-panel, and the next one's first block when the trailing
        // update factors it): their diagonal tiles are kept up to date by the panel solves
        int neager = 0;
        if (h && h->fuse_diag && h->eager) {
          int last = s0 + w - (h->fuse_diag > 1 ? 0 : 1);
          if (last > nsteps - 1) last = nsteps - 1;
          neager = last > j ? last - j : 0;
        }
        if (q > 0 && h && h->fuse_diag) {
          // left-looking update of block column j by panels s0..j-1; the tile-(0,0)
          // workgroup goes on to factor the diagonal block (fused), so only the
          // panel solve remains as a separate launch
          int rc = launch_blockcol_diag(h, G, ld, stride, c0, cS, Kp, q, nact, G.invL, lts,
                                        h->eager ? 1 : 0);
          if (rc != SP_OK) return rc;
          rc = launch_trsm(h, G.sys, ld, stride, G.S, c0 + nact, c0, Kp, G.invL, lts, G.st, neager);
          if (rc != SP_OK) return rc;
          continue;
        }
        if (q > 0) {  // left-looking update of block column j by panels s0..j-1
          double *A = G.sys + (size_t)c0 * ld + cS;
          double *T = G.sys + (size_t)c0 * ld + c0;
          int rc = sp_launch_gemm_nt(A, ld, stride, A, ld, stride, T, ld, stride, Kp - c0,
                                     SP_NB, q * SP_NB, -1.0, 1, 0, G.S, G.st);
          if (rc != SP_OK) return rc;
        }
        // first panel of a super-panel: its diagonal block was factored by the
        // fused bulk update of the previous super-panel (if fusing is on)
        const bool have_diag = q == 0 && s0 > 0 && h && h->fuse_diag > 1;
        int rc = diag_and_solve(h, G.sys, ld, stride, G.S, K, Kp, j, G.info, G.invL, lts, G.st,
                                have_diag, neager);
        if (rc != SP_OK) return rc;
      }
    }
    const int cE = (s0 + w) * SP_NB;
    if (cE < K) {
      const int nactE = K - cE < SP_NB ? K - cE : SP_NB;
      for (int g = 0; g < ngroups; ++g) {
        int rc = bulk_update(h, grp[g].sys, ld, stride, grp[g].S, cS, cE, Kp, w * SP_NB,
                             grp[g].st, lts, (h && h->fuse_diag > 1) ? nactE : 0, grp[g].invL,
                             grp[g].info, (h && h->fuse_diag > 1 && h->eager) ? 1 : 0);
        if (rc != SP_OK) return rc;
      }
    }
  }
  return SP_OK;
}

int sp_launch_cholesky_systems(sp_handle *h, double *sys, int S, int K, int Kp,
                               int32_t *info, double *invL, hipStream_t st) {
  sp_chol_group g{sys, info, invL, S, st};
  return sp_launch_cholesky_groups(h, 1, &g, K, Kp);
}

// Micro-benchmark hook: launch ONE phase of panel step j on `st`.
// phase 0 = diagonal block, 1 = panel solve, 2 = rank-64 trailing update.
int sp_debug_phase(sp_handle *h, double *sys, int S, int K, int Kp, int32_t *info,
                   double *invL, int phase, int j, hipStream_t st) {
  const long ld = Kp, stride = (long)Kp * Kp;
  const int c0 = j * SP_NB;
  const int nact = K - c0 < SP_NB ? K - c0 : SP_NB;
  const long lts = sp_lt_stride(Kp);
  if (phase == 0) return launch_diag(nullptr, sys, ld, stride, S, c0, nact, invL, lts, info, st);
  if (phase == 5) {
    const int nb = 256 * (j > 0 ? j : 1), iters = 20000;
    long long *ts = nullptr;
    double *sink = nullptr;
    SP_HIP(hipMalloc(&ts, sizeof(long long) * 2 * nb));
    SP_HIP(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    SP_HIP(hipEventCreate(&e0));
    SP_HIP(hipEventCreate(&e1));
    const int rnd = K & 1 ? 0 : 1;  // (debug) odd K: constant operands
    const int nacc = (K >> 1) & 3;  // (debug) accumulators per wave: 0 -> 8, 1 -> 4, 2 -> 2, 3 -> 1
#define SP_PEAK_GO(N, IT) hipLaunchKernelGGL(mfma_peak_kernel<N>, dim3(nb), dim3(256), 0, st, IT, sink, ts, rnd)
    if (nacc == 0) SP_PEAK_GO(8, 100); else if (nacc == 1) SP_PEAK_GO(4, 100); else if (nacc == 2) SP_PEAK_GO(2, 100); else SP_PEAK_GO(1, 100);
    SP_HIP(hipEventRecord(e0, st));
    if (nacc == 0) SP_PEAK_GO(8, iters); else if (nacc == 1) SP_PEAK_GO(4, iters); else if (nacc == 2) SP_PEAK_GO(2, iters); else SP_PEAK_GO(1, iters);
    SP_HIP(hipEventRecord(e1, st));
    SP_HIP(hipStreamSynchronize(st));
    float ms = 0;
    SP_HIP(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> hst(2 * (size_t)nb);
    SP_HIP(hipMemcpy(hst.data(), ts, sizeof(long long) * 2 * nb, hipMemcpyDeviceToHost));
    double mhz = 0;
    for (int b = 0; b < nb; ++b) mhz += (double)hst[2 * b + 1] / ((double)hst[2 * b] * 0.01);
    mhz /= nb;
    const double flops = (double)nb * 4 * iters * 8 * 2048.0;
    fprintf(stderr, "fp64 MFMA peak (%s operands): %d workgroups/CU-slot x 4 waves: %.1f TFLOP/s, shader clock %.0f MHz, %.3f ms\n",
            nacc == 0 ? "8 acc" : nacc == 1 ? "4 acc" : nacc == 2 ? "2 acc" : "1 acc", j > 0 ? j : 1, flops / (ms * 1e-3) * 1e-12, mhz, ms);
    (void)hipFree(ts);
    (void)hipFree(sink);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return SP_OK;
  }
  if (phase == 6 || phase == 7 || phase == 8) {
    // pieces of the recursive driver on whatever the workspace holds (timing only):
    // 6 the top-level strip solve, 7 the top-level symmetric update, 8 base block j (4 panels)
    sp_chol_group G{sys, info, invL, S, st};
    RecCtx c{h, &G, K, Kp, (K + SP_NB - 1) / SP_NB, ld, stride, lts};
    const int nb = c.nsteps, bm = ((nb / 2 + c.h->rec_base - 1) / c.h->rec_base) * c.h->rec_base;
    if (phase == 6) return rec_trsm(c, bm * SP_NB, nb * SP_NB, 0, bm);
    if (phase == 7)
      return bulk_update(h, sys, ld, stride, S, 0, bm * SP_NB, nb * SP_NB, bm * SP_NB, st, lts);
    const int b0 = j * c.h->rec_base, b1 = b0 + c.h->rec_base < nb ? b0 + c.h->rec_base : nb;
    return b0 < nb ? rec_base(c, b0, b1) : SP_ERR_INVALID;
  }
  if (phase == 4)  // the rank-256 trailing update of the first super-panel, not fused
    return bulk_update(nullptr, sys, ld, stride, S, 0, 4 * SP_NB, Kp, 4 * SP_NB, st, lts);
  if (phase == 3) {  // in-kernel timestamps of the diagonal-block kernel, printed to stderr
    long long *dbg = nullptr;
    SP_HIP(hipMalloc(&dbg, sizeof(long long) * 40 * S));
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(diag_kernel<true>, dim3(S), dim3(256), 0, st, sys, ld, stride, c0, nact,
                         invL, lts, info, dbg);
      SP_LAUNCH_CHECK();
    }
    SP_HIP(hipStreamSynchronize(st));
    std::vector<long long> hst(40 * (size_t)S);
    SP_HIP(hipMemcpy(hst.data(), dbg, sizeof(long long) * 40 * S, hipMemcpyDeviceToHost));
    (void)hipFree(dbg);
    long long t0 = hst[0];
    for (int b = 0; b < S; ++b) t0 = hst[40 * b] < t0 ? hst[40 * b] : t0;
    {
      const long long *r = &hst[0], c0k = r[5];
      for (int kb = 0; kb < 4; ++kb) {
        const long long *q = r + 8 + 8 * kb;
        fprintf(stderr,
                "  panel %d: start %6lld | load %5lld | leaf %6lld | publish %5lld | below %5lld | "
                "barrier %5lld | update %5lld\n",
                kb, q[0] - c0k, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3], q[5] - q[4],
                q[6] - q[5]);
      }
    }
    for (int b = 0; b < S; b += (S > 8 ? S / 8 : 1)) {
      const long long *r = &hst[40 * b];
      fprintf(stderr,
              "diag wg %3d: start +%6.2f us | load %6.2f us (%lld clk) | factor %6.2f us (%lld clk) | "
              "store %6.2f us (%lld clk)\n",
              b, (r[0] - t0) * 0.01, (r[1] - r[0]) * 0.01, r[5] - r[4], (r[2] - r[1]) * 0.01,
              r[6] - r[5], (r[3] - r[2]) * 0.01, r[7] - r[6]);
    }
    return SP_OK;
  }
  if (phase == 1) {
    return launch_trsm(nullptr, sys, ld, stride, S, c0 + nact, c0, Kp, invL, lts, st);
  }
  return bulk_update(nullptr, sys, ld, stride, S, c0, c0 + SP_NB, Kp, SP_NB, st, lts);
}

int sp_launch_lnlike_reduce(const double *sys, int S, int K, int M, int Kp,
                            const int32_t *info, double *lnlike, uint32_t *status,
                            hipStream_t st, uint32_t *status_out, const sp_star *stars,
                            const void *defer_coef) {
  hipLaunchKernelGGL(lnlike_reduce_kernel, dim3(S), dim3(256), 0, st, sys,
                     (long)Kp, (long)Kp * Kp, K, M, info, lnlike, status, status_out, stars,
                     (const RedCoef *)defer_coef);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_pad_in(const double *A, int K, long lda, long strideA, double *sys,
                     int Kp, int M, const double *resid, int S, hipStream_t st) {
  hipLaunchKernelGGL(pad_in_kernel, dim3((Kp + 255) / 256, Kp, S), dim3(256), 0,
                     st, A, K, lda, strideA, sys, Kp, (long)Kp * Kp, M, resid);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_pad_out(const double *sys, int Kp, double *A, int K, long lda,
                      long strideA, const int32_t *info, int S, hipStream_t st) {
  hipLaunchKernelGGL(pad_out_kernel, dim3((K + 255) / 256, K, S), dim3(256), 0, st,
                     sys, Kp, (long)Kp * Kp, A, K, lda, strideA, info);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_tri_solve(const double *L, int K, long ldl, long strideL, double *B,
                        long strideB, long rs, long cs, int nrhs, int batch, int mode,
                        hipStream_t st) {
  const int Kr = ((K + 63) / 64) * 64;
  const size_t lds = sizeof(double) * ((size_t)Kr + 64 * DLD);
  if (lds > 150 * 1024) return SP_ERR_INVALID;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cho_solve_kernel),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(cho_solve_kernel, dim3(nrhs, batch), dim3(256), lds, st, L, K,
                     ldl, strideL, B, strideB, rs, cs, mode);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_cho_solve(const double *L, int K, long ldl, long strideL, double *B,
                        int nrhs, int batch, hipStream_t st) {
  return sp_launch_tri_solve(L, K, ldl, strideL, B, (long)K * nrhs, nrhs, 1, nrhs, batch, 0, st);
}

int sp_launch_transpose(const double *in, long ldi, long stridei, double *out, int K, int batch,
                        hipStream_t st) {
  const int nt = (K + 63) / 64;
  hipLaunchKernelGGL(transpose_kernel, dim3(nt, nt, batch), dim3(256), 0, st, in, ldi, stridei, out,
                     K);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_tri_mask(double *A, int K, int batch, int upper, double dscale, hipStream_t st) {
  const long n = (long)K * K;
  hipLaunchKernelGGL(tri_mask_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, st, A,
                     K, upper, dscale);
  SP_LAUNCH_CHECK();
  return SP_OK;
}

int sp_launch_chol_rev_finish(const double *S, const double *L, long ldl, long strideL, double *out,
                              int K, int batch, hipStream_t st) {
  const long n = (long)K * K;
  hipLaunchKernelGGL(chol_rev_finish_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0,
                     st, S, L, ldl, strideL, out, K);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
