// Issue rate of the vector instructions the VALU-heavy kernels are made of (assembly, tiles at first touch, the
// diagonal block's leaf), one wavefront on a SIMD and four: cycles per instruction from s_memtime around a loop of
// 8 independent chains x 64 repeats.          hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void k(long long *out, double seed, int iters) {
  double a[8], b = seed * 1.000001, c = seed * 0.999;
  int ia[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + i + threadIdx.x * 1e-3;
    ia[i] = (int)a[i] + i;
  }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#define ONE(i)                                                                                          \
  if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));                              \
  if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));                              \
  if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                  \
  if (OP == 3) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[i]) : "v"(a[i]));                          \
  if (OP == 4) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(ia[i]));                          \
  if (OP == 5) asm volatile("v_fract_f64 %0, %0" : "+v"(a[i]));                                         \
  if (OP == 6) asm volatile("v_min_i32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));              \
  if (OP == 7) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));      \
  if (OP == 8) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
  if (OP == 9) asm volatile("v_add_f64 %0, %0, -%1" : "+v"(a[i]) : "v"(b));                             \
  if (OP == 10) asm volatile("v_max_f64 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(b));                         \
  if (OP == 11) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                 \
  if (OP == 12) asm volatile("v_floor_f64 %0, %0" : "+v"(a[i]));                                        \
  if (OP == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]));              \
  if (OP == 14) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));                                          \
  if (OP == 15) asm volatile("v_mul_f64 %0, %1, |%0|" : "+v"(a[i]) : "v"(b));                           \
  if (OP == 16) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP8(ONE)
#undef ONE
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  int si = 0;
  for (int i = 0; i < 8; ++i) {
    s += a[i];
    si += ia[i];
  }
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (s == 12345.678 && si == 77) out[0] = 0;
}

static const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_fract_f64",
                              "v_min_i32", "v_lshl_add_u32", "v_mov_b32_dpp", "v_add_f64 neg", "v_max_f64 abs",
                              "v_cmp_gt_f64", "v_floor_f64", "v_add_u32", "v_rcp_f64", "v_mul_f64 abs", "v_fmac_f64"};

template <int OP>
void run(long long *d, int threads) {
  const int iters = 200;
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, 1.5, iters);
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, 1.5, iters);
  long long h[16], mx = 0;
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;     // (the oldest wavefront is served first: the slowest tells)
  const int per = threads / 256 ? threads / 256 : 1;
  printf("  %-16s %2d wavefront(s) per SIMD: %6.2f ticks per instruction of the slowest wavefront = %5.2f per instruction and SIMD\n",
         names[OP], per, (double)mx / (iters * 64.0), (double)mx / (iters * 64.0) / per);
}

#define RUNALL(T)                                                                                        \
  run<0>(d, T); run<1>(d, T); run<2>(d, T); run<16>(d, T); run<3>(d, T); run<4>(d, T); run<5>(d, T);     \
  run<12>(d, T); run<10>(d, T); run<11>(d, T); run<14>(d, T); run<6>(d, T); run<7>(d, T); run<13>(d, T); \
  run<8>(d, T);

int main() {
  long long *d;
  hipMalloc(&d, 64 * sizeof(long long));
  {
    // what is a tick?  one long launch between two events
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200000;
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, 1.5, 1000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, 1.5, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h;
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("tick calibration: %lld ticks in %.3f ms = %.1f ticks per us\n", h, ms, h / (ms * 1e3));
  }
  printf("one wavefront on its SIMD (64 threads):\n");
  RUNALL(64)
  printf("four wavefronts, one per SIMD (256 threads): same numbers if the SIMDs are independent\n");
  RUNALL(256)
  printf("eight wavefronts, two per SIMD (512 threads): cycles per instruction of ONE wavefront (x 1/2 = SIMD rate)\n");
  RUNALL(512)
  printf("sixteen wavefronts, four per SIMD (1024 threads): cycles per instruction of ONE wavefront (x 1/4 = SIMD rate)\n");
  RUNALL(1024)
  return 0;
}
