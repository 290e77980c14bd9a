// tools/hsv_mem_probe.hip — development probe: how the hsvfilter arithmetic (csrc/hsv_device.hpp, FAST variant 5 =
// hue-shift 90, identity saturation / value) behaves under different memory policies and loop shapes, 8 x 4K RGBA in place.
// Not part of the product; numbers quoted in DESIGN.md §4.1 come from here.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I gst-plugins-rs_amd/csrc tools/hsv_mem_probe.hip -o tools/hsv_mem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "hsv_device.hpp"

using namespace mi355;

enum { NT_NONE = 0, NT_LOAD = 1, NT_STORE = 2, NT_BOTH = 3 };

template <int NT>
__device__ __forceinline__ uint4 ld(const uint4 *p) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  if constexpr (NT & NT_LOAD) { const u4 v = __builtin_nontemporal_load((const u4 *)p); return make_uint4(v.x, v.y, v.z, v.w); }
  else return *p;
}
template <int NT>
__device__ __forceinline__ void st(uint4 *p, uint4 v) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  if constexpr (NT & NT_STORE) { const u4 x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, (u4 *)p); }
  else *p = v;
}

// MATH: 0 = read-modify-write with one xor (the streaming floor), 1 = TAB 1 arithmetic, 2 = TAB 2 arithmetic
template <int MATH>
__device__ __forceinline__ void work(uint4 &p, const HsvK &k, const HsvLds *lds) {
  if constexpr (MATH == 0) {
    p.x ^= 1; p.y ^= 1; p.z ^= 1; p.w ^= 1;
  } else {
    hsvfilter_px2_lds<0, 1, 2, 3, HSV_SHIFT_POS, true, MATH>(p.x, p.y, k, lds);
    hsvfilter_px2_lds<0, 1, 2, 3, HSV_SHIFT_POS, true, MATH>(p.z, p.w, k, lds);
  }
}

// shape A: the product kernel's loop - grid-stride, two independent 16 B loads in flight per lane
template <int MATH, int NT, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void k_gridstride2(uint4 *__restrict__ data, size_t n_vec, HsvK k) {
  __shared__ HsvLds lds;
  hsv_lds_fill<0, 1, 2, 3, NTHREADS>(&lds);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * NTHREADS;
  size_t i = (size_t)blockIdx.x * NTHREADS + threadIdx.x;
  for (; i + stride < n_vec; i += 2 * stride) {
    uint4 p = ld<NT>(data + i), q = ld<NT>(data + i + stride);
    work<MATH>(p, k, &lds);
    st<NT>(data + i, p);
    work<MATH>(q, k, &lds);
    st<NT>(data + i + stride, q);
  }
  if (i < n_vec) { uint4 p = ld<NT>(data + i); work<MATH>(p, k, &lds); st<NT>(data + i, p); }
}

// shape B: grid-stride, one load in flight per lane (more waves instead)
template <int MATH, int NT, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void k_gridstride1(uint4 *__restrict__ data, size_t n_vec, HsvK k) {
  __shared__ HsvLds lds;
  hsv_lds_fill<0, 1, 2, 3, NTHREADS>(&lds);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * NTHREADS;
  for (size_t i = (size_t)blockIdx.x * NTHREADS + threadIdx.x; i < n_vec; i += stride) {
    uint4 p = ld<NT>(data + i);
    work<MATH>(p, k, &lds);
    st<NT>(data + i, p);
  }
}

// shape C: every block owns one contiguous chunk (U iterations of NTHREADS x 16 B back to back), software-pipelined:
// the next iteration's load is issued before the current one is worked on
template <int MATH, int NT, int NTHREADS>
__global__ __launch_bounds__(NTHREADS) void k_chunk(uint4 *__restrict__ data, size_t n_vec, HsvK k, int iters) {
  __shared__ HsvLds lds;
  hsv_lds_fill<0, 1, 2, 3, NTHREADS>(&lds);
  __syncthreads();
  size_t i = (size_t)blockIdx.x * NTHREADS * (size_t)iters + threadIdx.x;
  const size_t end = min(n_vec, (size_t)(blockIdx.x + 1) * NTHREADS * (size_t)iters);
  if (i >= end) return;
  uint4 p = ld<NT>(data + i);
  for (; i + NTHREADS < end; i += NTHREADS) {
    const uint4 q = ld<NT>(data + i + NTHREADS);
    work<MATH>(p, k, &lds);
    st<NT>(data + i, p);
    p = q;
  }
  work<MATH>(p, k, &lds);
  st<NT>(data + i, p);
}

static void fill_smooth(uint8_t *f, int w, int h, unsigned seed) {
  static const int bars[7][3] = {{192, 192, 192}, {192, 192, 0}, {0, 192, 192}, {0, 192, 0}, {192, 0, 192}, {192, 0, 0}, {0, 0, 192}};
  unsigned long long s = 0x9E3779B97F4A7C15ull ^ seed;
  auto rnd = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (unsigned)((s * 0x2545F4914F6CDD1Dull) >> 33); };
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const int *b = bars[(x * 7) / w];
      const int g = (x + y) * 255 / (w + h - 2);
      uint8_t *px = f + ((size_t)y * w + x) * 4;
      for (int c = 0; c < 3; c++) {
        int v = (b[c] * 2 + g) / 3 + (int)(rnd() % 7) - 3;
        px[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
      px[3] = (uint8_t)rnd();
    }
}

struct Case { const char *name; void (*launch)(uint4 *, size_t, HsvK, int grid, int arg); int threads; };

#define L_GS2(M, N, T) [](uint4 *d, size_t n, HsvK k, int grid, int) { hipLaunchKernelGGL((k_gridstride2<M, N, T>), dim3(grid), dim3(T), 0, 0, d, n, k); }
#define L_GS1(M, N, T) [](uint4 *d, size_t n, HsvK k, int grid, int) { hipLaunchKernelGGL((k_gridstride1<M, N, T>), dim3(grid), dim3(T), 0, 0, d, n, k); }
#define L_CH(M, N, T) [](uint4 *d, size_t n, HsvK k, int grid, int it) { hipLaunchKernelGGL((k_chunk<M, N, T>), dim3(grid), dim3(T), 0, 0, d, n, k, it); }

int main(int argc, char **argv) {
  const int W = 3840, H = 2160, B = 8, NBUF = 4;
  const size_t frame = (size_t)W * H * 4, bytes = frame * B, n_vec = bytes / 16;
  std::vector<uint8_t> host(frame);
  fill_smooth(host.data(), W, H, 1);
  uint4 *buf[NBUF];
  for (int i = 0; i < NBUF; i++) {
    hipMalloc(&buf[i], bytes);
    for (int f = 0; f < B; f++) hipMemcpy((uint8_t *)buf[i] + f * frame, host.data(), frame, hipMemcpyHostToDevice);
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  const HsvK k{90.0f, 1.0f, 0.0f, 1.0f, 0.0f};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char *name, void (*launch)(uint4 *, size_t, HsvK, int, int), int grid, int arg) {
    for (int w = 0; w < 6; w++) launch(buf[w % NBUF], n_vec, k, grid, arg);
    hipEventRecord(e0);
    const int it = 40;
    for (int w = 0; w < it; w++) launch(buf[w % NBUF], n_vec, k, grid, arg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    printf("%-44s grid=%6d arg=%3d  %.4f ms  %5.0f GB/s  %.3f of 8 TB/s\n", name, grid, arg, ms, 2.0 * bytes / ms / 1e6, 2.0 * bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
  };
  // warm the clocks
  for (int w = 0; w < 200; w++) hipLaunchKernelGGL((k_gridstride2<1, 0, 256>), dim3(n_cu * 64), dim3(256), 0, 0, buf[w % NBUF], n_vec, k);
  hipDeviceSynchronize();
  const int bpcs[] = {8, 16, 32, 64, 128};
  for (int bpc : bpcs) {
    run("rmw   gridstride2 nt0 256", L_GS2(0, 0, 256), n_cu * bpc, 0);
    run("rmw   gridstride2 nt3 256", L_GS2(0, 3, 256), n_cu * bpc, 0);
    run("tab1  gridstride2 nt0 256", L_GS2(1, 0, 256), n_cu * bpc, 0);
    run("tab1  gridstride2 nt1 256", L_GS2(1, 1, 256), n_cu * bpc, 0);
    run("tab1  gridstride2 nt2 256", L_GS2(1, 2, 256), n_cu * bpc, 0);
    run("tab1  gridstride2 nt3 256", L_GS2(1, 3, 256), n_cu * bpc, 0);
    run("tab2  gridstride2 nt0 256", L_GS2(2, 0, 256), n_cu * bpc, 0);
    run("tab2  gridstride2 nt3 256", L_GS2(2, 3, 256), n_cu * bpc, 0);
    run("tab1  gridstride1 nt0 256", L_GS1(1, 0, 256), n_cu * bpc, 0);
    run("tab1  gridstride1 nt3 256", L_GS1(1, 3, 256), n_cu * bpc, 0);
    run("tab2  gridstride1 nt0 256", L_GS1(2, 0, 256), n_cu * bpc, 0);
  }
  for (int bpc : {2, 4, 8}) {
    run("tab1  gridstride2 nt0 1024", L_GS2(1, 0, 1024), n_cu * bpc, 0);
    run("tab2  gridstride2 nt0 1024", L_GS2(2, 0, 1024), n_cu * bpc, 0);
    run("tab1  gridstride1 nt0 1024", L_GS1(1, 0, 1024), n_cu * bpc, 0);
  }
  // contiguous chunks: iterations per block
  for (int it : {4, 8, 16, 32, 64}) {
    const int grid = (int)((n_vec + (size_t)256 * it - 1) / ((size_t)256 * it));
    run("rmw   chunk nt0 256", L_CH(0, 0, 256), grid, it);
    run("tab1  chunk nt0 256", L_CH(1, 0, 256), grid, it);
    run("tab1  chunk nt3 256", L_CH(1, 3, 256), grid, it);
    run("tab2  chunk nt0 256", L_CH(2, 0, 256), grid, it);
  }
  return 0;
}
