// tools/lut_ablate.hip — ablation harness for the colorlut LDS kernel (development tool).
// Times variants of the three-pass kernel on synthetic smooth/noise 4K batches:
//   ABL bit0: skip plane refill after the first pass of the first tile   (cost of LDS staging)
//   ABL bit1: skip corner reads (use constants)                          (cost of LDS reads)
//   ABL bit2: skip lerp math (sum the corners)                           (cost of VALU lerps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kAxisTableBytes = 3 * 256 * 8;
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ uint32_t rha(float y) { int r; asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(y)); return (uint32_t)r; }

template <int NT, int P4, int ABL>
__global__ __launch_bounds__(NT) void k(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                        const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab, uint32_t plane_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = P4 * 4;
  constexpr int Sy = 35, Sz = 1161;
  const size_t tile_groups = (size_t)NT * P4;
  const size_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  __syncthreads();
  bool first = true;
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t px[P], base[P];
    float tx[P], ty[P], tz[P];
    const size_t g0 = tile * tile_groups + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < n_groups) v = src[g];
      px[4 * j + 0] = v.x; px[4 * j + 1] = v.y; px[4 * j + 2] = v.z; px[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      base[i] = ex.x + ey.x + ez.x;
      tx[i] = __uint_as_float(ex.y); ty[i] = __uint_as_float(ey.y); tz[i] = __uint_as_float(ez.y);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (!(ABL & 1) || first) {
        __syncthreads();
        if (ABL & 8) {
          const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
          const uint32_t chunks = plane_floats / 256;  // 1 KiB per wave-instruction
          const char *gsrc = (const char *)(planar + (size_t)c * plane_floats) + lane * 16;
          for (uint32_t kk = wave; kk < chunks; kk += NT / 64) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)kk * 1024),
                                             (__attribute__((address_space(3))) void *)(lds + kAxisTableBytes + kk * 1024), 16, 0, 0);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          const float4 *s4 = (const float4 *)(planar + (size_t)c * plane_floats);
          float4 *d4 = (float4 *)(lds + kAxisTableBytes);
          const uint32_t n4 = plane_floats / 4;
          for (uint32_t i = threadIdx.x; i < n4; i += NT) d4[i] = s4[i];
        }
        __syncthreads();
      }
      const uint32_t sel = c == 0 ? 0x07060500u : (c == 1 ? 0x07060004u : 0x07000504u);
#pragma unroll
      for (int i = 0; i < P; i++) {
        float a0, a1, b0, b1, c0, c1, d0, d1;
        if (ABL & 2) {
          a0 = tx[i]; a1 = ty[i]; b0 = tz[i]; b1 = tx[i] + 1.f; c0 = ty[i] + 1.f; c1 = tz[i] + 2.f; d0 = tx[i] * 2.f; d1 = ty[i] * 3.f;
        } else {
          const float *L1 = (const float *)(lds + base[i]);
          const float *L0 = L1 + Sz;
          a0 = L0[0]; a1 = L0[1]; b0 = L0[Sy]; b1 = L0[Sy + 1];
          c0 = L1[0]; c1 = L1[1]; d0 = L1[Sy]; d1 = L1[Sy + 1];
        }
        float o;
        if (ABL & 4) {
          o = ((a0 + a1) + (b0 + b1)) + ((c0 + c1) + (d0 + d1));
        } else {
          const float c00 = lerp1(a0, a1, tx[i]), c10 = lerp1(b0, b1, tx[i]);
          const float c01 = lerp1(c0, c1, tx[i]), c11 = lerp1(d0, d1, tx[i]);
          o = lerp1(lerp1(c00, c10, ty[i]), lerp1(c01, c11, ty[i]), tz[i]);
        }
        const uint32_t v8 = rha(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
        px[i] = __builtin_amdgcn_perm(px[i], v8, sel);
      }
    }
    first = false;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < n_groups) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
  }
}


template <int NT>
__device__ __forceinline__ void refill_dma(unsigned char *lds, const float *plane, uint32_t plane_floats) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t chunks = plane_floats / 256;
  const char *gsrc = (const char *)plane + lane * 16;
  for (uint32_t kk = wave; kk < chunks; kk += NT / 64)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)kk * 1024),
                                     (__attribute__((address_space(3))) void *)(lds + kAxisTableBytes + kk * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int P, int G, int C>
__device__ __forceinline__ void pass(const unsigned char *lds, uint32_t (&px)[P], const uint32_t (&base)[P], const float (&tx)[P],
                                     const float (&ty)[P], const float (&tz)[P]) {
  constexpr int Sy = 35, Sz = 1161;
  constexpr uint32_t sel = C == 0 ? 0x07060500u : (C == 1 ? 0x07060004u : 0x07000504u);
#pragma unroll
  for (int i = 0; i < P; i++) {
    const float *L1 = (const float *)(lds + base[i]);
    const float *L0 = L1 + Sz;
    const float a0 = L0[0], a1 = L0[1], b0 = L0[Sy], b1 = L0[Sy + 1];
    const float c0 = L1[0], c1 = L1[1], d0 = L1[Sy], d1 = L1[Sy + 1];
    const float c00 = lerp1(a0, a1, tx[i]), c10 = lerp1(b0, b1, tx[i]);
    const float c01 = lerp1(c0, c1, tx[i]), c11 = lerp1(d0, d1, tx[i]);
    const float o = lerp1(lerp1(c00, c10, ty[i]), lerp1(c01, c11, ty[i]), tz[i]);
    const uint32_t v8 = rha(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
    px[i] = __builtin_amdgcn_perm(px[i], v8, sel);
    if (G > 0 && (i % G) == G - 1) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT, int P4, int G, bool ALT>
__global__ __launch_bounds__(NT) void k2(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                         const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab, uint32_t plane_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = P4 * 4;
  const size_t tile_groups = (size_t)NT * P4;
  const size_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  __syncthreads();
  bool flip = false;
  int resident = -1;
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t px[P], base[P];
    float tx[P], ty[P], tz[P];
    const size_t g0 = tile * tile_groups + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < n_groups) v = src[g];
      px[4 * j + 0] = v.x; px[4 * j + 1] = v.y; px[4 * j + 2] = v.z; px[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      base[i] = ex.x + ey.x + ez.x;
      tx[i] = __uint_as_float(ex.y); ty[i] = __uint_as_float(ey.y); tz[i] = __uint_as_float(ez.y);
    }
#define STAGE(CH)                                                                   \
    if (resident != CH) {                                                           \
      __syncthreads();                                                              \
      refill_dma<NT>(lds, planar + (size_t)CH * plane_floats, plane_floats);        \
      __syncthreads();                                                              \
      resident = CH;                                                                \
    }
    if (!flip) {
      STAGE(0) pass<P, G, 0>(lds, px, base, tx, ty, tz);
      STAGE(1) pass<P, G, 1>(lds, px, base, tx, ty, tz);
      STAGE(2) pass<P, G, 2>(lds, px, base, tx, ty, tz);
    } else {
      STAGE(2) pass<P, G, 2>(lds, px, base, tx, ty, tz);
      STAGE(1) pass<P, G, 1>(lds, px, base, tx, ty, tz);
      STAGE(0) pass<P, G, 0>(lds, px, base, tx, ty, tz);
    }
    if (ALT) flip = !flip;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < n_groups) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
  }
}

template <int NT, int P4, int G, bool ALT>
static float run2(const char *name, const uint4 *src, uint4 *dst, size_t n_groups, const float *planar, const uint32_t *axis, uint32_t pf, int grid, size_t lds) {
  auto kern = k2<NT, P4, G, ALT>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e0));
  const int it = 10;
  for (int w = 0; w < it; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("  k2 %-31s NT=%d P4=%d G=%d ALT=%d  %.4f ms  (%.0f GB/s algorithmic)\n", name, NT, P4, G, (int)ALT, ms, n_groups * 32.0 / ms / 1e6);
  return ms;
}


// ---- k3: hand-pipelined pass: corner loads of pixel i+1 are in flight while pixel i is interpolated.
struct Corners { float a0, a1, b0, b1, c0, c1, d0, d1; };

__device__ __forceinline__ void issue_corners(Corners &q, uint32_t base) {
  // z0+1 layer at base (dword offsets 0,1 and Sy,Sy+1); z0 layer at base + 4*Sz (byte offsets)
  asm volatile(
      "ds_read2_b32 %0, %4 offset1:1\n\t"
      "ds_read2_b32 %1, %4 offset0:35 offset1:36\n\t"
      "ds_read2_b32 %2, %5 offset1:1\n\t"
      "ds_read2_b32 %3, %5 offset0:35 offset1:36"
      : "=&v"(*(float2 *)&q.c0), "=&v"(*(float2 *)&q.d0), "=&v"(*(float2 *)&q.a0), "=&v"(*(float2 *)&q.b0)
      : "v"(base), "v"(base + 4644u));
}
__device__ __forceinline__ void wait_corners_keep4(Corners &q) {
  asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(q.a0), "+v"(q.a1), "+v"(q.b0), "+v"(q.b1), "+v"(q.c0), "+v"(q.c1), "+v"(q.d0), "+v"(q.d1));
}
__device__ __forceinline__ void wait_corners_all(Corners &q) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q.a0), "+v"(q.a1), "+v"(q.b0), "+v"(q.b1), "+v"(q.c0), "+v"(q.c1), "+v"(q.d0), "+v"(q.d1));
}

template <int P, int C>
__device__ __forceinline__ void pass3(uint32_t (&px)[P], const uint32_t (&base)[P], const float (&tx)[P], const float (&ty)[P],
                                      const float (&tz)[P]) {
  constexpr uint32_t sel = C == 0 ? 0x07060500u : (C == 1 ? 0x07060004u : 0x07000504u);
  Corners q[2];
  issue_corners(q[0], base[0]);
#pragma unroll
  for (int i = 0; i < P; i++) {
    Corners &cur = q[i & 1];
    if (i + 1 < P) {
      issue_corners(q[(i + 1) & 1], base[i + 1]);
      wait_corners_keep4(cur);
    } else {
      wait_corners_all(cur);
    }
    const float c00 = lerp1(cur.a0, cur.a1, tx[i]), c10 = lerp1(cur.b0, cur.b1, tx[i]);
    const float c01 = lerp1(cur.c0, cur.c1, tx[i]), c11 = lerp1(cur.d0, cur.d1, tx[i]);
    const float o = lerp1(lerp1(c00, c10, ty[i]), lerp1(c01, c11, ty[i]), tz[i]);
    const uint32_t v8 = rha(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
    px[i] = __builtin_amdgcn_perm(px[i], v8, sel);
  }
}

template <int NT, int P4>
__global__ __launch_bounds__(NT) void k3(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                         const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab, uint32_t plane_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = P4 * 4;
  const size_t tile_groups = (size_t)NT * P4;
  const size_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  __syncthreads();
  bool flip = false;
  int resident = -1;
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t px[P], base[P];
    float tx[P], ty[P], tz[P];
    const size_t g0 = tile * tile_groups + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < n_groups) v = src[g];
      px[4 * j + 0] = v.x; px[4 * j + 1] = v.y; px[4 * j + 2] = v.z; px[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      base[i] = ex.x + ey.x + ez.x;
      tx[i] = __uint_as_float(ex.y); ty[i] = __uint_as_float(ey.y); tz[i] = __uint_as_float(ez.y);
    }
#define STAGE3(CH)                                                                  \
    if (resident != CH) {                                                           \
      __syncthreads();                                                              \
      refill_dma<NT>(lds, planar + (size_t)CH * plane_floats, plane_floats);        \
      __syncthreads();                                                              \
      resident = CH;                                                                \
    }
    if (!flip) {
      STAGE3(0) pass3<P, 0>(px, base, tx, ty, tz);
      STAGE3(1) pass3<P, 1>(px, base, tx, ty, tz);
      STAGE3(2) pass3<P, 2>(px, base, tx, ty, tz);
    } else {
      STAGE3(2) pass3<P, 2>(px, base, tx, ty, tz);
      STAGE3(1) pass3<P, 1>(px, base, tx, ty, tz);
      STAGE3(0) pass3<P, 0>(px, base, tx, ty, tz);
    }
    flip = !flip;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < n_groups) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
  }
}

template <int NT, int P4>
static float run3(const char *name, const uint4 *src, uint4 *dst, size_t n_groups, const float *planar, const uint32_t *axis, uint32_t pf, int grid, size_t lds) {
  auto kern = k3<NT, P4>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e0));
  const int it = 10;
  for (int w = 0; w < it; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("  k3 %-31s NT=%d P4=%d  %.4f ms  (%.0f GB/s algorithmic)\n", name, NT, P4, ms, n_groups * 32.0 / ms / 1e6);
  return ms;
}


// ---- Option Q: one launch per channel, plane resident for the whole launch, streaming waves.
template <int NT, int C, bool FINAL>
__global__ __launch_bounds__(NT) void kq(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t *__restrict__ planeC,
                                         const uint32_t *__restrict__ planeR, const uint32_t *__restrict__ planeG, size_t n_groups,
                                         const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab, uint32_t plane_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int Sy = 35, Sz = 1161;
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  refill_dma<NT>(lds, planar + (size_t)C * plane_floats, plane_floats);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * NT;
  size_t g = (size_t)blockIdx.x * NT + threadIdx.x;
  uint4 nxt = make_uint4(0, 0, 0, 0), nxt2 = make_uint4(0, 0, 0, 0);
  uint32_t nr = 0, ng = 0, nr2 = 0, ng2 = 0;
  if (g < n_groups) { nxt = src[g]; if (FINAL) { nr = planeR[g]; ng = planeG[g]; } }
  if (g + stride < n_groups) { nxt2 = src[g + stride]; if (FINAL) { nr2 = planeR[g + stride]; ng2 = planeG[g + stride]; } }
  for (; g < n_groups; g += stride) {
    const uint4 v = nxt;
    const uint32_t cr = nr, cg = ng;
    nxt = nxt2; nr = nr2; ng = ng2;
    if (g + 2 * stride < n_groups) { nxt2 = src[g + 2 * stride]; if (FINAL) { nr2 = planeR[g + 2 * stride]; ng2 = planeG[g + 2 * stride]; } }
    const uint32_t px[4] = {v.x, v.y, v.z, v.w};
    uint32_t out4 = 0;
    uint32_t res[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      const uint32_t base = ex.x + ey.x + ez.x;
      const float tx = __uint_as_float(ex.y), ty = __uint_as_float(ey.y), tz = __uint_as_float(ez.y);
      const float *L1 = (const float *)(lds + base);
      const float *L0 = L1 + Sz;
      const float a0 = L0[0], a1 = L0[1], b0 = L0[Sy], b1 = L0[Sy + 1];
      const float c0 = L1[0], c1 = L1[1], d0 = L1[Sy], d1 = L1[Sy + 1];
      const float c00 = lerp1(a0, a1, tx), c10 = lerp1(b0, b1, tx);
      const float c01 = lerp1(c0, c1, tx), c11 = lerp1(d0, d1, tx);
      const float o = lerp1(lerp1(c00, c10, ty), lerp1(c01, c11, ty), tz);
      res[i] = rha(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
    }
    if (!FINAL) {
      out4 = res[0] | (res[1] << 8) | (res[2] << 16) | (res[3] << 24);
      planeC[g] = out4;
    } else {
      const uint32_t r4 = cr, g4 = cg;
      uint4 o;
      o.x = (px[0] & 0xff000000u) | (r4 & 0xff) | ((g4 & 0xff) << 8) | (res[0] << 16);
      o.y = (px[1] & 0xff000000u) | ((r4 >> 8) & 0xff) | (((g4 >> 8) & 0xff) << 8) | (res[1] << 16);
      o.z = (px[2] & 0xff000000u) | ((r4 >> 16) & 0xff) | (((g4 >> 16) & 0xff) << 8) | (res[2] << 16);
      o.w = (px[3] & 0xff000000u) | (r4 >> 24) | ((g4 >> 24) << 8) | (res[3] << 16);
      dst[g] = o;
    }
  }
}

template <int NT>
static float runq(const char *name, const uint4 *src, uint4 *dst, uint32_t *pr, uint32_t *pg, size_t n_groups, int frames_per_group, int n_frames,
                  const float *planar, const uint32_t *axis, uint32_t pf, int grid, size_t lds) {
  auto k0 = kq<NT, 0, false>; auto k1 = kq<NT, 1, false>; auto k2 = kq<NT, 2, true>;
  CK(hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t gpf = n_groups / n_frames;  // groups per frame
  auto once = [&]() {
    for (int f0 = 0; f0 < n_frames; f0 += frames_per_group) {
      const size_t off = gpf * f0, n = gpf * frames_per_group;
      hipLaunchKernelGGL(k0, dim3(grid), dim3(NT), lds, 0, src + off, dst + off, pr + off, nullptr, nullptr, n, planar, axis, pf);
      hipLaunchKernelGGL(k1, dim3(grid), dim3(NT), lds, 0, src + off, dst + off, pg + off, nullptr, nullptr, n, planar, axis, pf);
      hipLaunchKernelGGL(k2, dim3(grid), dim3(NT), lds, 0, src + off, dst + off, nullptr, pr + off, pg + off, n, planar, axis, pf);
    }
  };
  once(); once();
  CK(hipEventRecord(e0));
  const int it = 10;
  for (int w = 0; w < it; w++) once();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("  kq %-24s NT=%d frames/group=%d  %.4f ms  (%.0f GB/s algorithmic)\n", name, NT, frames_per_group, ms, n_groups * 32.0 / ms / 1e6);
  return ms;
}


// ---- k4: k2 (alt order, DMA staging) + software prefetch of the next tile's pixels during the passes.
template <int NT, int P4>
__global__ __launch_bounds__(NT) void k4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                         const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab, uint32_t plane_floats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = P4 * 4;
  const size_t tile_groups = (size_t)NT * P4;
  const size_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  __syncthreads();
  bool flip = false;
  int resident = -1;
  uint4 nxt[P4];
  {
    const size_t g0 = (size_t)blockIdx.x * tile_groups + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) { const size_t g = g0 + (size_t)j * NT; nxt[j] = make_uint4(0, 0, 0, 0); if (blockIdx.x < n_tiles && g < n_groups) nxt[j] = src[g]; }
  }
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t px[P], base[P];
    float tx[P], ty[P], tz[P];
    const size_t g0 = tile * tile_groups + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) { px[4 * j + 0] = nxt[j].x; px[4 * j + 1] = nxt[j].y; px[4 * j + 2] = nxt[j].z; px[4 * j + 3] = nxt[j].w; }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      base[i] = ex.x + ey.x + ez.x;
      tx[i] = __uint_as_float(ex.y); ty[i] = __uint_as_float(ey.y); tz[i] = __uint_as_float(ez.y);
    }
#define STAGE4(CH)                                                                  \
    if (resident != CH) {                                                           \
      __syncthreads();                                                              \
      {                                                                             \
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; \
        const uint32_t chunks = plane_floats / 256;                                 \
        const char *gsrc = (const char *)(planar + (size_t)CH * plane_floats) + lane * 16; \
        for (uint32_t kk = wave; kk < chunks; kk += NT / 64)                        \
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)kk * 1024), \
                                           (__attribute__((address_space(3))) void *)(lds + kAxisTableBytes + kk * 1024), 16, 0, 0); \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
      }                                                                             \
      __syncthreads();                                                              \
      resident = CH;                                                                \
    }
#define PREFETCH4()                                                               \
    {                                                                               \
      const size_t nt = tile + gridDim.x;                                           \
      const size_t h0 = nt * tile_groups + threadIdx.x;                             \
      _Pragma("unroll") for (int j = 0; j < P4; j++) { const size_t g = h0 + (size_t)j * NT; if (nt < n_tiles && g < n_groups) nxt[j] = src[g]; } \
    }
    if (!flip) {
      STAGE4(0) pass<P, 0, 0>(lds, px, base, tx, ty, tz);
      STAGE4(1) pass<P, 0, 1>(lds, px, base, tx, ty, tz);
      STAGE4(2) PREFETCH4() pass<P, 0, 2>(lds, px, base, tx, ty, tz);
    } else {
      STAGE4(2) pass<P, 0, 2>(lds, px, base, tx, ty, tz);
      STAGE4(1) pass<P, 0, 1>(lds, px, base, tx, ty, tz);
      STAGE4(0) PREFETCH4() pass<P, 0, 0>(lds, px, base, tx, ty, tz);
    }
    flip = !flip;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < n_groups) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
  }
}
template <int NT, int P4>
static float run4(const char *name, const uint4 *src, uint4 *dst, size_t n_groups, const float *planar, const uint32_t *axis, uint32_t pf, int grid, size_t lds) {
  auto kern = k4<NT, P4>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e0));
  const int it = 10;
  for (int w = 0; w < it; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("  k4 %-31s NT=%d P4=%d  %.4f ms  (%.0f GB/s algorithmic)\n", name, NT, P4, ms, n_groups * 32.0 / ms / 1e6);
  return ms;
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint32_t xr() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }

template <int NT, int P4, int ABL>
static float run(const char *name, const uint4 *src, uint4 *dst, size_t n_groups, const float *planar, const uint32_t *axis, uint32_t pf, int grid, size_t lds) {
  auto kern = k<NT, P4, ABL>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e0));
  const int it = 10;
  for (int w = 0; w < it; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, src, dst, n_groups, planar, axis, pf);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  printf("  %-34s NT=%d P4=%d  %.4f ms  (%.0f GB/s algorithmic)\n", name, NT, P4, ms, n_groups * 32.0 / ms / 1e6);
  return ms;
}

int main(int argc, char **argv) {
  const int W = 3840, H = 2160, B = 8;
  const size_t npx = (size_t)W * H * B, n_groups = npx / 4;
  const int S = 33, Sy = 35, Sz = 1161;
  const size_t pf = ((size_t)S * Sz + Sy + 2 + 255) & ~(size_t)255;
  const size_t lds = kAxisTableBytes + 4 * pf;
  std::vector<float> planar(3 * pf, 0.f);
  for (int c = 0; c < 3; c++) for (int z = 0; z < S; z++) for (int y = 0; y < S; y++) for (int x = 0; x < S; x++) {
    float idv = (c == 0 ? x : (c == 1 ? y : z)) / 32.f;
    planar[c * pf + x + Sy * y + Sz * (S - 1 - z)] = idv + 0.04f * sinf(6.28f * (x + y * 0.5f + z * 0.25f) / 32.f);
  }
  std::vector<uint32_t> axis(3 * 256 * 2);
  for (int a = 0; a < 3; a++) for (int v = 0; v < 256; v++) {
    float xx = ((float)v / 255.0f) * 32.0f; int i0 = (int)floorf(xx); float t = xx - i0;
    int off = a == 0 ? 4 * i0 : (a == 1 ? 4 * Sy * i0 : kAxisTableBytes + 4 * Sz * (S - 2 - i0));
    axis[(a * 256 + v) * 2] = off; memcpy(&axis[(a * 256 + v) * 2 + 1], &t, 4);
  }
  std::vector<uint32_t> smooth(npx), noise(npx);
  for (size_t i = 0; i < npx; i++) {
    noise[i] = xr();
    size_t x = i % W, y = (i / W) % H;
    int g = (int)((x + y) * 255 / (W + H));
    int r = (g + (int)(xr() % 7) - 3), gg = (g / 2 + 60 + (int)(xr() % 7) - 3), b = (255 - g + (int)(xr() % 7) - 3);
    r = r < 0 ? 0 : r > 255 ? 255 : r; gg = gg < 0 ? 0 : gg > 255 ? 255 : gg; b = b < 0 ? 0 : b > 255 ? 255 : b;
    smooth[i] = r | (gg << 8) | (b << 16) | 0xff000000u;
  }
  uint4 *d_src, *d_dst; float *d_planar; uint32_t *d_axis;
  CK(hipMalloc(&d_src, npx * 4)); CK(hipMalloc(&d_dst, npx * 4));
  CK(hipMalloc(&d_planar, planar.size() * 4)); CK(hipMalloc(&d_axis, axis.size() * 4));
  CK(hipMemcpy(d_planar, planar.data(), planar.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_axis, axis.data(), axis.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount;
  for (int content = 0; content < 2; content++) {
    CK(hipMemcpy(d_src, content ? noise.data() : smooth.data(), npx * 4, hipMemcpyHostToDevice));
    printf("%s content, batch of %d 4K frames, lds=%zu B\n", content ? "noise" : "smooth", B, lds);
    run2<1024, 3, 0, true>("alt order (compiler sched)", d_src, d_dst, n_groups, d_planar, d_axis, (uint32_t)pf, grid, lds);
    {
      std::vector<uint32_t> ref(npx), got(npx);
      CK(hipMemcpy(ref.data(), d_dst, npx * 4, hipMemcpyDeviceToHost));
      run4<1024, 2>("prefetch next tile", d_src, d_dst, n_groups, d_planar, d_axis, (uint32_t)pf, grid, lds);
      CK(hipMemcpy(got.data(), d_dst, npx * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < npx; i++) bad += ref[i] != got[i];
      printf("  k4 vs k2 mismatching pixels: %zu\n", bad);
    }
    run4<1024, 3>("prefetch next tile", d_src, d_dst, n_groups, d_planar, d_axis, (uint32_t)pf, grid, lds);
    run2<1024, 2, 0, true>("alt order (compiler sched)", d_src, d_dst, n_groups, d_planar, d_axis, (uint32_t)pf, grid, lds);
    run4<768, 4>("prefetch next tile", d_src, d_dst, n_groups, d_planar, d_axis, (uint32_t)pf, grid, lds);
    run4<512, 6>("prefetch next tile", d_src, d_dst, n_groups, d_planar, d_axis, (uint32_t)pf, grid, lds);
  }
  return 0;
}
