// Shared by the assembly kernels of sp_assemble.hip and the planned step's assembly (sp_planasm.hip).
#ifndef SP_ASM_H
#define SP_ASM_H

#include "sp_internal.h"

// first tile of every chunk (strip-major order), by value in the kernel arguments
#define SP_ASM_MAX_CHUNKS 511
struct AsmChunks {
  unsigned short start[SP_ASM_MAX_CHUNKS + 1];
};

// alpha(z), beta(z) of the normalisation series (ops/norm/norm.py:26-44) and the star's coefficients
__device__ __forceinline__ SpCoef defer_coef(double m, double fmean, int order, double baseline_var) {
  const double mu = 1.0 + fmean;
  const double z = m / (mu * mu);
  double fac = 1.0, alpha = 0.0, beta = 0.0;
  for (int n = 0; n <= order; ++n) {
    alpha += fac;
    beta += 2 * n * fac;
    fac *= z * (2 * n + 3);
  }
  const double c1 = alpha / (mu * mu);
  SpCoef c;
  c.c1 = c1;
  c.zab = z * (alpha + beta) / c1;   // d_p
  c.za = -z * alpha / c1;            // d_q
  c.z = z;
  c.gpmean = 0.0;
  c.m = m;
  c.mu = mu;
  c.d1 = baseline_var / c1;          // d_1
  return c;
}

#endif
