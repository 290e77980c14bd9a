// tools/lds_partial_exec_bench.hip — what does a ds_read_b128 / ds_read_b64 cost when only some lanes of the wave are active?
// (development tool: decides whether a second, predicated brick read for the lanes whose pixel pair straddles two cells is cheap)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define ITERS 4096
template <int WIDE>
__global__ __launch_bounds__(1024) void k(float *out, const uint32_t *offs, unsigned long long mask, int n_pat) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 30000; i += 1024) lds[i] = i * 0.5f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  const bool on = (mask >> lane) & 1;
  uint32_t a[4];
  for (int j = 0; j < 4; j++) a[j] = offs[((j + wave * 7) % n_pat) * 64 + lane];   // four address patterns per lane, kept in registers
  if (on) {
    for (int i = 0; i < ITERS; i++) {
      if (WIDE == 16) {
        float4 v[8];
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %9\n ds_read_b128 %2, %10\n ds_read_b128 %3, %11\n"
                     "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %9 offset:64\n ds_read_b128 %6, %10 offset:64\n ds_read_b128 %7, %11 offset:64\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
        for (int j = 0; j < 8; j++) acc += v[j].x;
      } else if (WIDE == 8) {
        float2 v[8];
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %9\n ds_read_b64 %2, %10\n ds_read_b64 %3, %11\n"
                     "ds_read_b64 %4, %8 offset:64\n ds_read_b64 %5, %9 offset:64\n ds_read_b64 %6, %10 offset:64\n ds_read_b64 %7, %11 offset:64\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
        for (int j = 0; j < 8; j++) acc += v[j].x;
      } else {
        float v[8];
        asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n"
                     "ds_read_b32 %4, %8 offset:64\n ds_read_b32 %5, %9 offset:64\n ds_read_b32 %6, %10 offset:64\n ds_read_b32 %7, %11 offset:64\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
        for (int j = 0; j < 8; j++) acc += v[j];
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cu = p.multiProcessorCount, n_pat = 64;
  float *d; hipMalloc(&d, (size_t)cu * 1024 * 4);
  uint32_t *doffs; hipMalloc(&doffs, n_pat * 64 * 4);
  const size_t lds = 120000;
  hipFuncSetAttribute((const void *)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<uint32_t> offs(n_pat * 64);
  for (int pat = 0; pat < 2; pat++) {
    for (auto &o : offs) o = pat == 0 ? 16 * (rand() % 7000) : 16 * (rand() % 4);   // random 16-byte granules / four hot granules (broadcast-like)
    hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
    struct { const char *name; unsigned long long m; } masks[] = {
      {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 0-7", 0xffull},
      {"every 2nd lane", 0x5555555555555555ull}, {"every 4th lane", 0x1111111111111111ull}, {"every 8th lane", 0x0101010101010101ull},
      {"every 16th lane", 0x0001000100010001ull}, {"one lane", 1ull}};
    for (auto &mk : masks) for (int wide : {16, 8, 4}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto go = [&] {
        if (wide == 16) hipLaunchKernelGGL(k<16>, dim3(cu), dim3(1024), lds, 0, d, doffs, mk.m, n_pat);
        else if (wide == 8) hipLaunchKernelGGL(k<8>, dim3(cu), dim3(1024), lds, 0, d, doffs, mk.m, n_pat);
        else hipLaunchKernelGGL(k<4>, dim3(cu), dim3(1024), lds, 0, d, doffs, mk.m, n_pat);
      };
      go(); hipEventRecord(e0); go(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_cu = 16.0 * ITERS * 8;
      printf("%-22s %-16s b%-3d %.3f ms -> %.2f clk per wave instruction (2.4 GHz nominal)\n", pat == 0 ? "random 16 B granules" : "four hot granules", mk.name, wide * 8, ms,
             ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
  }
  return 0;
}
