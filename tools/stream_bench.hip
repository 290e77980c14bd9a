// tools/stream_bench.hip — achievable HBM streaming rate for the access patterns the pixel kernels use
// (development tool): in-place read-modify-write and out-of-place copy of uint4, 8 x 4K RGBA frames.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void rmw(uint4 *d, size_t n) {
  const size_t s = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) { uint4 v = d[i]; v.x ^= 1; v.y += 3; v.z ^= 7; v.w += 1; d[i] = v; }
}
__global__ __launch_bounds__(256) void cpy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n) {
  const size_t s = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) { uint4 v = a[i]; v.x ^= 1; b[i] = v; }
}
int main() {
  const size_t bytes = 3840ull * 2160 * 4 * 8, n = bytes / 16;
  uint4 *a[2], *b[2];
  for (int i = 0; i < 2; i++) { hipMalloc(&a[i], bytes); hipMalloc(&b[i], bytes); hipMemset(a[i], 1, bytes); }
  for (int grid : {2048, 4096, 8192, 32768}) {
    for (int mode = 0; mode < 2; mode++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int w = 0; w < 2; w++) { if (mode) hipLaunchKernelGGL(cpy, dim3(grid), dim3(256), 0, 0, a[w & 1], b[w & 1], n); else hipLaunchKernelGGL(rmw, dim3(grid), dim3(256), 0, 0, a[w & 1], n); }
      hipEventRecord(e0);
      const int it = 20;
      for (int w = 0; w < it; w++) { if (mode) hipLaunchKernelGGL(cpy, dim3(grid), dim3(256), 0, 0, a[w & 1], b[w & 1], n); else hipLaunchKernelGGL(rmw, dim3(grid), dim3(256), 0, 0, a[w & 1], n); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
      printf("%-22s grid=%6d  %.4f ms  %.0f GB/s\n", mode ? "copy a->b" : "in-place rmw", grid, ms, 2.0 * bytes / ms / 1e6);
    }
  }
  return 0;
}
