// tools/sem_probe.hip — exhaustive checks of gfx950 instruction semantics the FAST kernels want to
// rely on (development tool). Each probe sweeps every float in a range and counts mismatches
// against the exact definition computed with plain IEEE ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>

__device__ unsigned long long g_bad[8];
__device__ float g_first[8];

__device__ inline void report(int k, float x) {
  if (atomicAdd(&g_bad[k], 1ull) == 0) g_first[k] = x;
}

// reference round-half-away for x >= 0: exact via floor/fract
__device__ inline int ref_round_half_away(float x) {
  float f = floorf(x);
  return (int)f + ((x - f) >= 0.5f ? 1 : 0);
}

__global__ void probe(uint32_t lo_bits, uint32_t hi_bits) {
  const uint64_t n = (uint64_t)hi_bits - lo_bits + 1;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t b = lo_bits + (uint32_t)i;
    const float x = __uint_as_float(b);
    // 0: v_cvt_rpi_i32_f32 == round half away (x >= 0)
    int rpi;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(rpi) : "v"(x));
    if (rpi != ref_round_half_away(x)) report(0, x);
    // 1: v_cvt_pk_u8_f32 == trunc+saturate ?   2: == round half away ?  3: == rne ?
    uint32_t pk;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(pk) : "v"(x));
    const uint32_t tr = x >= 255.0f ? 255u : (uint32_t)x;
    if (pk != tr) report(1, x);
    const int rh = ref_round_half_away(x);
    if (pk != (uint32_t)(rh > 255 ? 255 : rh)) report(2, x);
    const float rne = rintf(x);
    if (pk != (uint32_t)(rne > 255.f ? 255.f : rne)) report(3, x);
    // 4: v_fract_f32 == x - floor(x)
    float fr;
    asm volatile("v_fract_f32 %0, %1" : "=v"(fr) : "v"(x));
    if (fr != x - floorf(x)) report(4, x);
    // 5: (trunc(x*2)+1)>>1 == round half away  (x = y, 2y exact)
    if ((int)(((uint32_t)(x * 2.0f) + 1u) >> 1) != rh) report(5, x);
    // 6: v_cvt_flr_i32_f32 == floor
    int fl;
    asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(fl) : "v"(x));
    if (fl != (int)floorf(x)) report(6, x);
    // 7: (uint)x == floor for x>=0
    if ((uint32_t)x != (uint32_t)floorf(x)) report(7, x);
  }
}

int main() {
  const char *names[8] = {"cvt_rpi_i32==round_half_away", "cvt_pk_u8==trunc_sat", "cvt_pk_u8==round_half_away", "cvt_pk_u8==rne",
                          "v_fract==x-floor(x)", "(trunc(2x)+1)>>1==round_half_away", "cvt_flr==floor", "cvt_u32==floor"};
  float lo = 0.0f, hi = 65536.0f;
  uint32_t lb, hb;
  memcpy(&lb, &lo, 4);
  memcpy(&hb, &hi, 4);
  unsigned long long zero[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_bad), zero, sizeof zero);
  hipLaunchKernelGGL(probe, dim3(256 * 8), dim3(256), 0, 0, lb, hb);
  hipDeviceSynchronize();
  unsigned long long bad[8];
  float first[8];
  hipMemcpyFromSymbol(bad, HIP_SYMBOL(g_bad), sizeof bad);
  hipMemcpyFromSymbol(first, HIP_SYMBOL(g_first), sizeof first);
  printf("swept %u floats in [0, 65536]\n", hb - lb + 1);
  for (int k = 0; k < 8; k++) printf("%-36s mismatches=%llu first=%a (%.9g)\n", names[k], bad[k], first[k], first[k]);
  return 0;
}
