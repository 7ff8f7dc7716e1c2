// Microbenchmark behind DESIGN.md section 5 ("what k_pair_parts is bound by"): the rate at which an MI355X serves the galloping
// path's access pattern -- consecutive lanes read ONE 8-byte element each from an ascending sequence of positions that are r
// elements apart on average (r = |long| / |short|), four independent loads in flight per lane.
// build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 scripts/micro/gather_bw.hip -o /tmp/gather_bw && /tmp/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
__device__ __forceinline__ u32 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return (u32)x; }
template <int W>   // W = words of 8 bytes per access (1 or 2)
__global__ __launch_bounds__(64, 5) void k_gather(const u64* __restrict__ a, u64 n, u64 r, u64 elems, u64* __restrict__ out) {
  const u64 T = (u64)gridDim.x * blockDim.x, tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 acc = 0;
  for (u64 g0 = tid; g0 < elems; g0 += 4 * T) {
    u64 v[4][W];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 g = g0 + (u64)k * T;
      u64 idx = g * r + (r > 1 ? mix(g) % r : 0);
      if (idx + W > n) idx = 0;
#pragma unroll
      for (int w = 0; w < W; w++) v[k][w] = g < elems ? a[idx + w] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int w = 0; w < W; w++) acc += v[k][w];
  }
  if (acc == 0x1234567ull) out[0] = acc;
}
int main() {
  const u64 n = 1ull << 30;   // 8 GB
  u64 *a, *out;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4 * 5 * 4;
  for (int W = 1; W <= 2; W++)
    for (u64 r : {1ull, 2ull, 4ull, 8ull, 16ull, 32ull, 128ull, 1024ull}) {
      const u64 elems = (n - 2) / r;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (W == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(64), 0, 0, a, n, r, elems, out);
        else hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(64), 0, 0, a, n, r, elems, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double lines = r >= 8 ? (double)elems : (double)n / 8;   // distinct 64-byte lines touched (r < 8: all of them)
      printf("%2d B per access, one per %5llu elements: %8.3f ms  %7.1f G accesses/s  %7.2f TB/s of 64-byte lines  %6.2f TB/s of used bytes\n", 8 * W,
             (unsigned long long)r, best, elems / best / 1e6, lines * 64 / best / 1e9, (double)elems * 8 * W / best / 1e9);
    }
  return 0;
}
