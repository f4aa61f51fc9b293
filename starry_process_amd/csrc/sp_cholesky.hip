// Batched fp64 Cholesky, triangular solves and the likelihood reduction
//  for gfx950.
//
// Right-looking blocked factorisation, panels of SP_NB = 64 columns:
//   panel_kernel   factors the 64x64 diagonal block in LDS and solves the rows
//                  below it (X L_d^T = P), 256 rows per workgroup;
//   sp_launch_gemm_nt (sp_gemm.hip) applies the trailing update C -= X X^T on
//                  the matrix cores, lower-triangle tiles only.
// The systems are padded to a multiple of 64 rows and carry the residual
// vectors as EXTRA ROWS below the matrix: factoring
//     [ C   . ]          gives          [ L   . ]
//     [ r^T . ]                         [ y^T . ] ,   y = L^-1 r,
// so r^T C^-1 r = |y|^2 falls out of the factorisation and no separate
// triangular solve is needed for the likelihood.
#include "sp_internal.h"

#define DLD 65  // padded row length of the diagonal block in LDS

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double read_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to ~1 ulp: hardware seed (v_rsq_f64) + two Newton steps
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
  double e = fma(-p * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-p * r, r, 1.0);
  r = fma(0.5 * r, e, r);
  return r;
}

// ---- diagonal-block kernel ----------------------------------------------------
// One workgroup (4 wavefronts) per star factors the 64 x 64 diagonal block
// A = L L^T and also forms L^-1, so that the panel below it becomes a plain
// product X = P L^-T on the matrix cores (sp_gemm.hip).  The block is processed
// as 4 x 4 sub-blocks of 16 x 16:
//   leaf   : wavefront 0, lane = row, 16-column right-looking sweep with
//            v_readlane broadcasts (no LDS, no barriers inside), followed by
//            the 16 x 16 triangular inverse, lane = column;
//   updates: v_mfma_f64_16x16x4_f64 on LDS-resident operands, sub-blocks
//            spread over the four wavefronts.
// LDS rows are padded to 66 doubles (132 dwords = 4 mod 64 banks): the MFMA
// operand reads (16 rows x 2 k per 32-lane half) are conflict free.
#define BLD 66

// a-operand / NT b-operand fragment: M[row0 + (lane & 15)][col0 + 4 s + (lane >> 4)]
__device__ __forceinline__ double frag_rowmajor(const double *M, int row0, int col0,
                                                int s, int lane) {
  return M[(row0 + (lane & 15)) * BLD + col0 + 4 * s + (lane >> 4)];
}
// NN b-operand fragment: M[row0 + 4 s + (lane >> 4)][col0 + (lane & 15)]
__device__ __forceinline__ double frag_kmajor(const double *M, int row0, int col0,
                                              int s, int lane) {
  return M[(row0 + 4 * s + (lane >> 4)) * BLD + col0 + (lane & 15)];
}
// accumulator <-> LDS, C/D map of the fp64 MFMA: col = lane & 15, row = (lane >> 4) + 4 reg
__device__ __forceinline__ d4 acc_load(const double *M, int row0, int col0, int lane) {
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = M[(row0 + (lane >> 4) + 4 * r) * BLD + col0 + (lane & 15)];
  return v;
}
__device__ __forceinline__ void acc_store(double *M, int row0, int col0, int lane, d4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) M[(row0 + (lane >> 4) + 4 * r) * BLD + col0 + (lane & 15)] = v[r];
}

__global__ __launch_bounds__(256) void diag_kernel(double *__restrict__ sys, long ld,
                                                   long stride, int c0, int nact,
                                                   double *__restrict__ invL_all,
                                                   int32_t *__restrict__ info) {
  __shared__ __attribute__((aligned(16))) double sA[64 * BLD];  // block, becomes L
  __shared__ __attribute__((aligned(16))) double sI[64 * BLD];  // L^-1
  double *Mx = sys + (size_t)blockIdx.x * stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        sA[r * BLD + c] = v;
        sI[r * BLD + c] = 0.0;
      }
    }
  }
  __syncthreads();
  int notpd = 0;
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    const int o = 16 * kb;
    if (wave == 0) {
      // Leaf Cholesky AND leaf inverse on the matrix core, one column at a time.
      // The (symmetric, fully stored) 16 x 16 leaf sits in ONE accumulator tile:
      // lane (fk, fr) holds rows fk + 4 q, column fr.  By symmetry row c = column
      // c, and row c is held by the 16 lanes of group fk = c & 3 in register
      // q = c >> 2 -- exactly the operand slot k = fk of v_mfma_f64_16x16x4, so
      //   A <- A - l l^T         (l = column c of L)          needs no lane traffic:
      // lanes of that group pass l, every other lane passes 0.  The inverse rides
      // along: L = L_0 L_1 .. L_15 with L_c = I + (l_c - e_c) e_c^T, hence
      //   Y <- Y - u_c (e_c^T Y),  u_c = (l_c - e_c) / l_cc,   Y_0 = I
      // ends at Y = L^-1; e_c^T Y is again "row c", same operand slot.
      const int fr = lane & 15, fk = lane >> 4;
      d4 Am, Ym;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = fk + 4 * q;
        Am[q] = fr <= row ? sA[(o + row) * BLD + o + fr] : sA[(o + fr) * BLD + o + row];
        Ym[q] = row == fr ? 1.0 : 0.0;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int g = c & 3, q = c >> 2;
        const double piv = read_lane(Am[q], 16 * g + c);
        if (!(piv > 0.0)) notpd = 1;
        const double r = rsqrt_nr(piv);
        const bool act = (fk == g) && (fr >= c);
        const double l = act ? Am[q] * r : 0.0;           // l_{fr,c}; fr == c: sqrt(piv)
        const double u = act ? (fr == c ? 1.0 - r : l * r) : 0.0;
        const double yrow = (fk == g) ? Ym[q] : 0.0;      // row c of Y
        if (act) sA[(o + fr) * BLD + o + c] = l;
        Am = __builtin_amdgcn_mfma_f64_16x16x4f64(-l, l, Am, 0, 0, 0);
        Ym = __builtin_amdgcn_mfma_f64_16x16x4f64(-u, yrow, Ym, 0, 0, 0);
      }
      acc_store(sI, o, o, lane, Ym);
    }
    __syncthreads();
    // sub-diagonal blocks of this block column: L_ik = A_ik . (L_kk^-1)^T
    if (wave > kb) {
      const int ib = wave;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(frag_rowmajor(sA, 16 * ib, o, s, lane),
                                                   frag_rowmajor(sI, o, o, s, lane), acc, 0, 0, 0);
      acc_store(sA, 16 * ib, o, lane, acc);
    }
    __syncthreads();
    // trailing sub-blocks: A_ij -= L_ik L_jk^T, kb < jb <= ib
    {
      int q = 0;
      for (int ib = kb + 1; ib < 4; ++ib)
        for (int jb = kb + 1; jb <= ib; ++jb, ++q) {
          if ((q & 3) != wave) continue;
          d4 acc = acc_load(sA, 16 * ib, 16 * jb, lane);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag_rowmajor(sA, 16 * ib, o, s, lane),
                                                       frag_rowmajor(sA, 16 * jb, o, s, lane), acc, 0, 0, 0);
          acc_store(sA, 16 * ib, 16 * jb, lane, acc);
        }
    }
    __syncthreads();
  }
  // off-diagonal blocks of L^-1: X_ij = -(L_ii^-1) sum_{k=j}^{i-1} L_ik X_kj,
  // block column j on wavefront j, rows i in sequence
#pragma unroll 1
  for (int i = 1; i < 4; ++i) {
    if (wave < i) {
      const int j = wave;
      d4 t = {0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < i; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          t = __builtin_amdgcn_mfma_f64_16x16x4f64(frag_rowmajor(sA, 16 * i, 16 * k, s, lane),
                                                   frag_kmajor(sI, 16 * k, 16 * j, s, lane), t, 0, 0, 0);
      // t[s] = T[4 s + (lane >> 4)][lane & 15] is exactly the k-major operand of step s
      d4 xacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        xacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-frag_rowmajor(sI, 16 * i, 16 * i, s, lane),
                                                    t[s], xacc, 0, 0, 0);
      acc_store(sI, 16 * i, 16 * j, lane, xacc);
    }
    __syncthreads();
  }
  // write back L (active lower part) and the full L^-1 tile
  {
    if (tid == 0 && notpd && info) info[blockIdx.x] = 1;
    double *inv = invL_all + (size_t)blockIdx.x * 4096;
    const int cj = (tid & 15) * 4, ri = tid >> 4;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = ri + 16 * pass;
      double *dst = Mx + (size_t)(c0 + r) * ld + c0 + cj;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cj + e;
        if (c <= r && r < nact) dst[e] = sA[r * BLD + c];
        inv[r * 64 + c] = sI[r * BLD + c];
      }
    }
  }
}

// lnlike = -1/2 sum_m |y_m|^2 - M sum_i log L_ii - K M / 2 log(2 pi)
// (sp.py:1157-1188).  One workgroup per star.
__global__ __launch_bounds__(256) void lnlike_reduce_kernel(
    const double *__restrict__ sys, long ld, long stride, int K, int M,
    const int32_t *__restrict__ info, double *__restrict__ lnlike,
    uint32_t *__restrict__ status) {
  __shared__ double red[8];
  const int s = blockIdx.x;
  const double *Mx = sys + (size_t)s * stride;
  double ld_part = 0.0, q_part = 0.0;
  for (int i = threadIdx.x; i < K; i += 256) ld_part += log(Mx[(size_t)i * ld + i]);
  for (int m = 0; m < M; ++m) {
    const double *y = Mx + (size_t)(K + m) * ld;
    for (int k = threadIdx.x; k < K; k += 256) q_part += y[k] * y[k];
  }
  for (int off = 32; off > 0; off >>= 1) {
    ld_part += __shfl_down(ld_part, off, 64);
    q_part += __shfl_down(q_part, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = ld_part;
    red[4 + (threadIdx.x >> 6)] = q_part;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double logdet = (red[0] + red[1]) + (red[2] + red[3]);
    const double quad = (red[4] + red[5]) + (red[6] + red[7]);
    double v = -0.5 * quad;
    v -= M * logdet;
    v -= 0.5 * K * M * 1.8378770664093453;  // log(2 pi)
    uint32_t st = status ? status[s] : 0u;
    if (info && info[s]) st |= SP_STAR_NOT_PD;
    if (v != v) st |= SP_STAR_NAN;
    if (st & (SP_STAR_NOT_PD | SP_STAR_ZMAX | SP_STAR_NAN)) v = -INFINITY;
    lnlike[s] = v;
    if (status) status[s] = st;
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
__global__ __launch_bounds__(256) void cho_solve_kernel(
    const double *__restrict__ Lall, int K, long ldl, long strideL,
    double *__restrict__ Ball, int nrhs) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int Kr = ((K + 63) / 64) * 64;
  double *x = lds;        // Kr
  double *sL = lds + Kr;  // 64 * DLD
  const double *L = Lall + (size_t)blockIdx.y * strideL;
  double *B = Ball + (size_t)blockIdx.y * K * nrhs + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < Kr; i += 256) x[i] = i < K ? B[(size_t)i * nrhs] : 0.0;
  const int nb = Kr / 64;
  __syncthreads();
  // forward: L y = b
  for (int blk = 0; blk < nb; ++blk) {
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
  for (int blk = nb - 1; blk >= 0; --blk) {
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
  for (int i = tid; i < K; i += 256) B[(size_t)i * nrhs] = x[i];
}

}  // namespace

// ---- launchers ---------------------------------------------------------------

// C[cfrom:, cfrom:] -= X[cfrom:, :] X[cfrom:, :]^T with X = columns c0..c0+kd-1 of the
// same rows; lower-triangle tiles only.  Timed for bench.py when profiling is on.
static int bulk_update(sp_handle *h, double *sys, long ld, long stride, int S, int c0,
                       int cfrom, int Kp, int kd, hipStream_t st) {
  const int n = Kp - cfrom;
  double *X = sys + (size_t)cfrom * ld + c0;
  double *T = sys + (size_t)cfrom * ld + cfrom;
  const bool timed = h && h->prof_on && h->prof_used + 2 <= h->prof_ev.size();
  if (timed) SP_HIP(hipEventRecord(h->prof_ev[h->prof_used], st));
  int rc = sp_launch_gemm_nt(X, ld, stride, X, ld, stride, T, ld, stride, n, n, kd, -1.0, 1,
                             1, S, st);
  if (rc != SP_OK) return rc;
  if (timed) {
    SP_HIP(hipEventRecord(h->prof_ev[h->prof_used + 1], st));
    h->prof_used += 2;
    // algorithmic work of a symmetric rank-kd update of an n x n block:
    // n (n + 1) / 2 entries x kd multiply-adds
    h->prof_flops += (double)S * (double)n * (n + 1) * kd;
    h->prof_launches += 1;
  }
  return SP_OK;
}

static int diag_and_solve(double *sys, long ld, long stride, int S, int K, int Kp, int j,
                          int32_t *info, double *invL, hipStream_t st) {
  const int c0 = j * SP_NB;
  const int nact = K - c0 < SP_NB ? K - c0 : SP_NB;
  // diagonal block: L_d and L_d^-1
  hipLaunchKernelGGL(diag_kernel, dim3(S), dim3(256), 0, st, sys, ld, stride, c0, nact,
                     invL, info);
  SP_LAUNCH_CHECK();
  // rows below the active block: X = P L_d^-T, in place, on the matrix cores
  const int r1 = c0 + nact;
  if (r1 < Kp) {
    double *P = sys + (size_t)r1 * ld + c0;
    return sp_launch_gemm_nt(P, ld, stride, invL, SP_NB, (long)SP_NB * SP_NB, P, ld, stride,
                             Kp - r1, SP_NB, SP_NB, 1.0, 0, 0, S, st);
  }
  return SP_OK;
}

// In-place factorisation of S padded systems (Kp x Kp, ld = Kp): the leading
// K x K part is factored, rows K..Kp-1 only receive the triangular solve.
//
// Two-level blocking.  Panels (64 columns) are grouped in super-panels of w
// panels.  Inside a super-panel a block column is brought up to date
// left-looking (ONE narrow product over the q previous panels of the group,
// k = 64 q) just before it is factored; the big trailing matrix is touched once
// per super-panel with a rank-64w update instead of w rank-64 updates.  The
// trailing update is HBM-bound at k = 64 (8 flop per byte of C traffic,
// measured 4.0 TB/s, profiles/r01_*); k = 64 w divides that traffic by w.
int sp_launch_cholesky_groups(sp_handle *h, int ngroups, const sp_chol_group *grp, int K,
                              int Kp) {
  const long ld = Kp, stride = (long)Kp * Kp;
  const int nsteps = (K + SP_NB - 1) / SP_NB;
  const int w = (h && h->superpanel > 0) ? h->superpanel : 1;
  // launches are issued breadth-first over the groups so that the groups'
  // streams advance together (the host enqueues ~3-8 us per launch)
  for (int s0 = 0; s0 < nsteps; s0 += w) {
    const int cS = s0 * SP_NB;
    for (int q = 0; q < w && s0 + q < nsteps; ++q) {
      const int j = s0 + q, c0 = j * SP_NB;
      for (int g = 0; g < ngroups; ++g) {
        const sp_chol_group &G = grp[g];
        if (q > 0) {  // left-looking update of block column j by panels s0..j-1
          double *A = G.sys + (size_t)c0 * ld + cS;
          double *T = G.sys + (size_t)c0 * ld + c0;
          int rc = sp_launch_gemm_nt(A, ld, stride, A, ld, stride, T, ld, stride, Kp - c0,
                                     SP_NB, q * SP_NB, -1.0, 1, 0, G.S, G.st);
          if (rc != SP_OK) return rc;
        }
        int rc = diag_and_solve(G.sys, ld, stride, G.S, K, Kp, j, G.info, G.invL, G.st);
        if (rc != SP_OK) return rc;
      }
    }
    const int cE = (s0 + w) * SP_NB;
    if (cE < K)
      for (int g = 0; g < ngroups; ++g) {
        int rc = bulk_update(h, grp[g].sys, ld, stride, grp[g].S, cS, cE, Kp, w * SP_NB,
                             grp[g].st);
        if (rc != SP_OK) return rc;
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
  if (phase == 0) {
    hipLaunchKernelGGL(diag_kernel, dim3(S), dim3(256), 0, st, sys, ld, stride, c0, nact,
                       invL, info);
    SP_LAUNCH_CHECK();
    return SP_OK;
  }
  if (phase == 1) {
    const int r1 = c0 + nact;
    double *P = sys + (size_t)r1 * ld + c0;
    return sp_launch_gemm_nt(P, ld, stride, invL, SP_NB, (long)SP_NB * SP_NB, P, ld, stride,
                             Kp - r1, SP_NB, SP_NB, 1.0, 0, 0, S, st);
  }
  return bulk_update(nullptr, sys, ld, stride, S, c0, c0 + SP_NB, Kp, SP_NB, st);
}

int sp_launch_lnlike_reduce(const double *sys, int S, int K, int M, int Kp,
                            const int32_t *info, double *lnlike, uint32_t *status,
                            hipStream_t st) {
  hipLaunchKernelGGL(lnlike_reduce_kernel, dim3(S), dim3(256), 0, st, sys,
                     (long)Kp, (long)Kp * Kp, K, M, info, lnlike, status);
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

int sp_launch_cho_solve(const double *L, int K, long ldl, long strideL, double *B,
                        int nrhs, int batch, hipStream_t st) {
  const int Kr = ((K + 63) / 64) * 64;
  const size_t lds = sizeof(double) * ((size_t)Kr + 64 * DLD);
  if (lds > 150 * 1024) return SP_ERR_INVALID;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(cho_solve_kernel),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(cho_solve_kernel, dim3(nrhs, batch), dim3(256), lds, st, L, K,
                     ldl, strideL, B, nrhs);
  SP_LAUNCH_CHECK();
  return SP_OK;
}
