// anyorder_probe.hip — does hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) clear the barrier between two kernels of ONE stream on
// gfx950 (hip_ext.h says the flag is not supported on GFX9xx for the module-launch form)? Two independent copy kernels A (many small
// blocks, like the gather colorlut kernel) and B (grid-stride, like hsvfilter) over different 265 MB buffers, back to back, 50 pairs:
// all in order; B launched any-order (it may be dispatched while A's last blocks drain); both on two streams (the two-lane form).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kA(const u4_t *a, u4_t *b, size_t n) {   // one 4 KB tile per block
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { u4_t v = a[i]; v.x ^= 1; __builtin_nontemporal_store(v, b + i); }
}
__global__ __launch_bounds__(256) void kB(u4_t *a, size_t n) {   // in place, grid-stride
  const size_t s = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += s) { u4_t v = a[i]; v.y ^= 1; a[i] = v; }
}
int main() {
  const size_t n = (size_t)8 * 3840 * 2160 / 4, bytes = n * 16;
  u4_t *x[4], *y[4];
  for (int i = 0; i < 4; i++) { hipMalloc(&x[i], bytes); hipMalloc(&y[i], bytes); hipMemset(x[i], i, bytes); }
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned gA = (unsigned)((n + 255) / 256), gB = 256 * 64;
  for (int mode = 0; mode < 3; mode++)
    for (int rep = 0; rep < 3; rep++) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s0);
      for (int k = 0; k < 50; k++) {
        // B(k) in place on x[k & 3] (independent of A(k-1), which read x[(k - 1) & 3] and wrote y[(k - 1) & 3]); then A(k) reads what B(k) wrote
        if (mode == 1 && k > 0) hipExtLaunchKernelGGL(kB, dim3(gB), dim3(256), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, x[k & 3], n);
        else hipLaunchKernelGGL(kB, dim3(gB), dim3(256), 0, s0, x[k & 3], n);
        hipLaunchKernelGGL(kA, dim3(gA), dim3(256), 0, s0, (const u4_t *)x[k & 3], y[k & 3], n);
      }
      hipEventRecord(e1, s0);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.4f ms per (B, A) pair\n", mode == 0 ? "in order            " : (mode == 1 ? "B any-order         " : "in order (again)    "), ms / 50);
    }
  printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
