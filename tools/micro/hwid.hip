// Which bits of HW_REG_HW_ID / HW_REG_XCC_ID identify a CU on this part?  One record per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void k(unsigned *out) {
  __shared__ double pad[4096];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
  }
  // stay a while so that the grid spreads over the CUs
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);
  if (pad[(threadIdx.x * 7) & 4095] < -1.0) out[0] = 0;
}
int main() {
  const int n = 1024;
  unsigned *d;
  hipMalloc(&d, 8 * n);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
  unsigned orv = 0, andv = ~0u, orx = 0;
  for (int i = 0; i < n; ++i) { orv |= h[2 * i]; andv &= h[2 * i]; orx |= h[2 * i + 1]; }
  printf("HW_ID bits that vary: %08x   XCC_ID bits set: %08x\n", orv & ~andv, orx);
  for (int lo = 0; lo < 32; lo += 4) {
    std::set<unsigned> v;
    for (int i = 0; i < n; ++i) v.insert((h[2 * i] >> lo) & 15u);
    printf("  HW_ID[%d:%d]: %zu distinct values\n", lo + 3, lo, v.size());
  }
  std::set<unsigned> k1, k2;
  std::map<unsigned, int> cnt;
  for (int i = 0; i < n; ++i) {
    k1.insert(((h[2 * i + 1] & 7u) << 8) | ((h[2 * i] >> 8) & 255u));
    k2.insert(((h[2 * i + 1] & 15u) << 16) | ((h[2 * i] >> 8) & 0xffffu));
    cnt[((h[2 * i + 1] & 7u) << 8) | ((h[2 * i] >> 8) & 255u)]++;
  }
  int mx = 0;
  for (auto &kv : cnt) mx = kv.second > mx ? kv.second : mx;
  printf("distinct keys xcc | HW_ID[15:8]: %zu (max %d workgroups on one key); xcc | HW_ID[23:8]: %zu\n", k1.size(), mx, k2.size());
  for (int i = 0; i < 12; ++i) printf("  wg %d: hw %08x xcc %08x\n", i, h[2 * i], h[2 * i + 1]);
  return 0;
}
