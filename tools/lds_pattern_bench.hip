// tools/lds_pattern_bench.hip — ds_read_b32 / ds_read2_b32 throughput vs address pattern (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define ITERS 2048
// pattern: per-lane byte offset table passed from host (64 entries per wave pattern, varied per wave)
__global__ __launch_bounds__(1024) void k_r32(float *out, const uint32_t *offs, int n_pat) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 38000; i += 1024) lds[i] = i * 0.5f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < ITERS; i++) {
    const uint32_t a = offs[((i + wave * 7) % n_pat) * 64 + lane];
    float v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:140\n ds_read_b32 %3, %8 offset:144\n"
                 "ds_read_b32 %4, %8 offset:4644\n ds_read_b32 %5, %8 offset:4648\n ds_read_b32 %6, %8 offset:4784\n ds_read_b32 %7, %8 offset:4788\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(a));
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(1024) void k_r2(float *out, const uint32_t *offs, int n_pat) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 38000; i += 1024) lds[i] = i * 0.5f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < ITERS; i++) {
    const uint32_t a = offs[((i + wave * 7) % n_pat) * 64 + lane];
    float2 v0, v1, v2, v3;
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %4 offset0:35 offset1:36\n ds_read2_b32 %2, %5 offset1:1\n ds_read2_b32 %3, %5 offset0:35 offset1:36\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a), "v"(a + 4644));
    acc += v0.x + v1.x + v2.x + v3.x + v0.y + v1.y + v2.y + v3.y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// ds_read_b64 at 4-byte (not 8-byte) aligned addresses: the x0 / x0+1 corner pair of the plane in ONE 8-byte read. Legal only
// with the LDS in unaligned-access mode (the ROCm default on gfx9); this measures what the hardware charges for it.
__global__ __launch_bounds__(1024) void k_r64(float *out, const uint32_t *offs, int n_pat) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 38000; i += 1024) lds[i] = i * 0.5f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int i = 0; i < ITERS; i++) {
    const uint32_t a = offs[((i + wave * 7) % n_pat) * 64 + lane];
    float2 v0, v1, v2, v3;
    asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:140\n ds_read_b64 %2, %4 offset:4644\n ds_read_b64 %3, %4 offset:4784\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a));
    acc += v0.x + v1.x + v2.x + v3.x + v0.y + v1.y + v2.y + v3.y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount;
  const int n_pat = 64;
  float *d; hipMalloc(&d, (size_t)cu * 1024 * 4);
  uint32_t *doffs; hipMalloc(&doffs, n_pat * 64 * 4);
  const size_t lds = 156000;
  hipFuncSetAttribute((const void *)k_r32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void *)k_r2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void *)k_r64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const char *names[] = {"consecutive words", "all lanes same address", "8 neighbour cells (x+3y+9z banks)", "8 neighbour cells natural 33/1089 strides",
                         "random cells (noise)", "2 addresses", "64 distinct rows same bank (worst)"};
  for (int pat = 0; pat < 7; pat++) {
    std::vector<uint32_t> offs(n_pat * 64);
    for (int k = 0; k < n_pat; k++) for (int l = 0; l < 64; l++) {
      uint32_t cx = rand() % 30 + 1, cy = rand() % 30 + 1, cz = rand() % 30 + 1;  // per-lane random cell
      static uint32_t bx, by, bz; if (l == 0) { bx = cx; by = cy; bz = cz; }
      uint32_t a = 0;
      switch (pat) {
        case 0: a = 4 * (k * 64 + l); break;
        case 1: a = 4 * (bx + 35 * by + 1161 * bz); break;
        case 2: a = 4 * ((bx + (rand() & 1)) + 35 * (by + (rand() & 1)) + 1161 * (bz + (rand() & 1))); break;
        case 3: a = 4 * ((bx + (rand() & 1)) + 33 * (by + (rand() & 1)) + 1089 * (bz + (rand() & 1))); break;
        case 4: a = 4 * (cx + 35 * cy + 1161 * cz); break;
        case 5: a = 4 * (bx + (l & 1) + 35 * by + 1161 * bz); break;
        case 6: a = 4 * (32 * l); break;
      }
      offs[k * 64 + l] = a;
    }
    hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
    for (int which = 0; which < 3; which++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto go = [&] {
        if (which == 0) hipLaunchKernelGGL(k_r32, dim3(cu), dim3(1024), lds, 0, d, doffs, n_pat);
        else if (which == 1) hipLaunchKernelGGL(k_r2, dim3(cu), dim3(1024), lds, 0, d, doffs, n_pat);
        else hipLaunchKernelGGL(k_r64, dim3(cu), dim3(1024), lds, 0, d, doffs, n_pat);
      };
      go();
      hipEventRecord(e0);
      go();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double dwords_per_cu = 16.0 * ITERS * 8;  // wave-level dword-reads per CU
      printf("%-44s %-9s %.3f ms  -> %.2f clk per wave dword-read (nominal 2.4 GHz)\n", names[pat], which == 0 ? "read_b32" : (which == 1 ? "read2_b32" : "read_b64*"), ms, ms * 1e-3 * 2.4e9 / dwords_per_cu);
    }
  }
  // correctness of the misaligned 8-byte read: one wave reads pairs at odd word addresses
  {
    std::vector<uint32_t> offs(n_pat * 64);
    for (size_t i = 0; i < offs.size(); i++) offs[i] = 4 * (2 * (i % 997) + 1);
    hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_r64, dim3(1), dim3(1024), lds, 0, d, doffs, n_pat);
    hipLaunchKernelGGL(k_r2, dim3(1), dim3(1024), lds, 0, d + 1024, doffs, n_pat);
    std::vector<float> h(2048);
    hipMemcpy(h.data(), d, 2048 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += h[i] != h[1024 + i];
    printf("misaligned ds_read_b64 vs ds_read2_b32 sums: %d of 1024 lanes differ\n", bad);
  }
  return 0;
}
