// Bandwidth of reading 64 x 64 fp64 tiles as the factorisation's kernels do (each row 512 B),
// with the rows of a tile `ld` doubles apart: ld = 64 is a tile-contiguous layout, ld = 1024 the
// row-major padded system of K = 1000.  hipcc --offload-arch=gfx950 -O3 tile_bw.hip -o tile_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

// one workgroup per tile visit: 256 threads, each 16 B x 8 (a 64 x 64 tile: 64 rows x 32 lanes x 16 B)
__global__ __launch_bounds__(256) void read_tiles(const double *base, long ld, long tile_stride_r,
                                                   long tile_stride_c, int ntr, int ntc, long mat_stride,
                                                   double *sink, int reps) {
  const int t = threadIdx.x, r = t >> 5, c = (t & 31) * 2;
  d2 acc = {0.0, 0.0};
  for (int rep = 0; rep < reps; ++rep) {
    const long item = (long)blockIdx.x + (long)rep * gridDim.x;
    const long per = (long)ntr * ntc;
    const long m = item / per, tt = item % per;
    const double *T = base + m * mat_stride + (tt / ntc) * tile_stride_r + (tt % ntc) * tile_stride_c;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const d2 v = *reinterpret_cast<const d2 *>(T + (long)(r + 8 * p) * ld + c);
      acc += v;
    }
  }
  if (acc.x == 123.456) sink[0] = acc.y;
}

int main() {
  const int S = 64, Kp = 1024, nt = Kp / 64;
  const size_t n = (size_t)S * Kp * Kp;
  double *buf, *sink;
  hipMalloc(&buf, n * 8); hipMalloc(&sink, 8);
  hipMemset(buf, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char *name; long ld, tsr, tsc; };
  Case cases[] = {{"row-major, ld 1024 (rows of a tile 8 KB apart)", 1024, 64L * 1024, 64},
                  {"tile-contiguous, ld 64 (a tile = 32 KB)", 64, 16L * 4096, 4096}};
  for (auto &cs : cases) {
    for (int wgs : {4096, 16384}) {
      const int reps = (int)((long)S * nt * nt / wgs);
      for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(read_tiles, dim3(wgs), dim3(256), 0, 0, buf, cs.ld, cs.tsr, cs.tsc, nt, nt,
                           (long)Kp * Kp, sink, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it == 2) printf("%-52s workgroups %6d: %.3f ms  %.2f TB/s\n", cs.name, wgs, ms,
                            (double)wgs * reps * 32768.0 / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
