// walk_dma_bench.hip — round 6, VERDICT r05 item 2: does the WALK of the LDS-cached table kernels (csrc/colorlut_window.hip) stream
// faster when the source tile goes global -> LDS by DMA (global_load_lds_dwordx4, no VGPR round trip) than through per-lane
// global_load_dwordx4 into VGPRs? Pure copies of 8 x 4K RGBA (265 MB in, 265 MB out), one 1024-lane block per CU, the aligned-fronts
// order of the real kernel (block b: strip b % 15, layer b / 15), LDS reserved as the real kernel's cache would be:
//   flat        grid-stride copy, 65,536 blocks of 256 lanes                                   (the floor every box gives: 0.082 ms)
//   walk_vgpr   the real kernel's walk: two steps of pixels in flight per lane in VGPRs        (0.087-0.104 ms by box, round 5)
//   walk_dma    every wave DMAs its own two rows of a step into its slot of an LDS ring DEPTH steps deep, reads them back with two
//               ds_read_b128 when they have landed (s_waitcnt vmcnt), stores. Bytes in flight per CU = DEPTH x 32 KB, no VGPRs held.
//   walk_loader one LOADER wave per block issues the DMA for everybody (32 x 1 KiB per step) and publishes "slot landed" through an
//               LDS word; 15 consumer waves read pixels from LDS and store (MI355X_MICROARCH.md: ldsdma-fill, one loader wave per CU)
// Build: hipcc --offload-arch=gfx950 -O3 tools/walk_dma_bench.hip -o tools/walk_dma_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

static constexpr unsigned W4 = 3840 / 4, ROWS = 2160 * 8;

__global__ __launch_bounds__(256) void flat(const u4_t *__restrict__ a, u4_t *__restrict__ b, size_t n) {
  const size_t s = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) { u4_t v = __builtin_nontemporal_load(a + i); v.x ^= 1; __builtin_nontemporal_store(v, b + i); }
}

// the real kernel's walk (aligned fronts, two rows per wave and step, DEPTH steps in VGPRs)
template <int DEPTH>
__global__ __launch_bounds__(1024) void walk_vgpr(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                                  unsigned steps_per_strip, unsigned share, unsigned layers) {
  extern __shared__ unsigned char dyn[];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];  // keeps the allocation
  const unsigned n_strips = (w4 + 63u) / 64u, strip = blockIdx.x % n_strips, layer = blockIdx.x / n_strips;
  unsigned first = layer * share, last = first + share;
  if (last > steps_per_strip) last = steps_per_strip;
  if (layer >= layers || first >= last) return;
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  struct Slot { u4_t p, q; uint32_t o0, o1; };
  Slot ring[DEPTH];
  auto fetch = [&](unsigned st_, Slot &S) {
    const unsigned k = st_ < last ? st_ : last - 1u;
    const unsigned col = strip * 64u + lane, r0 = k * 32u + 2u * wave, r1 = r0 + 1u;
    const unsigned cc = col < w4 ? col : w4 - 1u, c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
    S.o0 = col < w4 && r0 < rows ? (r0 * w4 + col) << 4 : 0x80000000u;
    S.o1 = col < w4 && r1 < rows ? (r1 * w4 + col) << 4 : 0x80000000u;
    S.p = __builtin_nontemporal_load(src + ((size_t)c0 * w4 + cc));
    S.q = __builtin_nontemporal_load(src + ((size_t)c1 * w4 + cc));
  };
  auto step = [&](unsigned st, Slot &S) {
    u4_t a = S.p, b = S.q;
    const uint32_t so0 = S.o0, so1 = S.o1;
    a.x ^= 1u; b.y ^= 1u;
    __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)so0, 0, 2);
    __builtin_amdgcn_raw_buffer_store_b128(b, dst_rsrc, (int)so1, 0, 2);
    fetch(st + DEPTH, S);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) fetch(first + d, ring[d]);
  unsigned st = first;
  for (; st + DEPTH <= last; st += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) step(st + d, ring[d]);
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++)
    if (st + d < last) step(st + d, ring[d]);
}

// Every wave streams its own two rows of each step through its own LDS ring: slot (wave, d) = 2 KiB = two 1 KiB DMA loads.
// A step's pixels are in flight from the issue of its DMA (DEPTH steps ahead) to the s_waitcnt that proves they have landed:
// DEPTH x 32 KB per CU without a single VGPR. The order of VM operations per wave is  S(i) S(i) L(i+D) L(i+D)  per step i, so when
// step i is consumed 4 (DEPTH - 1) younger operations may still be out: s_waitcnt vmcnt(4 (DEPTH - 1)) (VM operations retire in
// order on gfx9: the loads of step i are then done, and so are the older stores).
template <int DEPTH, int AUX>
__global__ __launch_bounds__(1024) void walk_dma(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                                 unsigned steps_per_strip, unsigned share, unsigned layers, unsigned ring_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];
  const unsigned n_strips = (w4 + 63u) / 64u, strip = blockIdx.x % n_strips, layer = blockIdx.x / n_strips;
  unsigned first = layer * share, last = first + share;
  if (last > steps_per_strip) last = steps_per_strip;
  if (layer >= layers || first >= last) return;
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  unsigned char *ring = dyn + ring_off + wave * (DEPTH * 2048u);
  const unsigned col = strip * 64u + lane, cc = col < w4 ? col : w4 - 1u;
  auto issue = [&](unsigned st_, unsigned d) {
    const unsigned k = st_ < last ? st_ : last - 1u;
    const unsigned r0 = k * 32u + 2u * wave, r1 = r0 + 1u;
    const unsigned c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + ((size_t)c0 * w4 + cc)), LDS_PTR(ring + d * 2048u), 16, 0, AUX);
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + ((size_t)c1 * w4 + cc)), LDS_PTR(ring + d * 2048u + 1024u), 16, 0, AUX);
  };
  auto consume = [&](unsigned st, unsigned d) {
    const unsigned r0 = st * 32u + 2u * wave, r1 = r0 + 1u;
    const uint32_t o0 = col < w4 && r0 < rows ? (r0 * w4 + col) << 4 : 0x80000000u;
    const uint32_t o1 = col < w4 && r1 < rows ? (r1 * w4 + col) << 4 : 0x80000000u;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
    u4_t a = *(const u4_t *)(ring + d * 2048u + lane * 16u), b = *(const u4_t *)(ring + d * 2048u + 1024u + lane * 16u);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");   // the slot has been read: the DMA of step st + DEPTH may overwrite it
    a.x ^= 1u; b.y ^= 1u;
    __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)o0, 0, 2);
    __builtin_amdgcn_raw_buffer_store_b128(b, dst_rsrc, (int)o1, 0, 2);
    asm volatile("" ::: "memory");
  };
  // prologue: DEPTH steps of loads, each followed by two dropped stores so that the operation order is the steady state's
#pragma unroll
  for (int d = 0; d < DEPTH; d++) {
    u4_t z = {0, 0, 0, 0};
    __builtin_amdgcn_raw_buffer_store_b128(z, dst_rsrc, (int)0x80000000u, 0, 2);
    __builtin_amdgcn_raw_buffer_store_b128(z, dst_rsrc, (int)0x80000000u, 0, 2);
    issue(first + d, d);
  }
  unsigned st = first;
  for (; st + DEPTH <= last; st += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      consume(st + d, d);
      issue(st + d + DEPTH, d);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++)
    if (st + d < last) {   // (wave-uniform)
      consume(st + d, d);
      issue(st + d + DEPTH, d);   // clamped: keeps the operation count of the wait
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One loader wave (wave 15) issues the DMA of a whole step (32 x 1 KiB) into ring slot d and, once it has landed, publishes the step
// number in s_ready[d]; the 15 consumer waves take the step's 32 rows (rows w, w + 15, w + 30 of the slot), store, and count the
// slot free again. LDS words are the only synchronisation (no barrier in the loop).
template <int DEPTH, int AUX>
__global__ __launch_bounds__(1024) void walk_loader(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                                    unsigned steps_per_strip, unsigned share, unsigned layers, unsigned ring_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __shared__ volatile unsigned s_ready[DEPTH], s_free[DEPTH];   // step + 1 that has landed in the slot / consumer waves done with it
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];
  if (threadIdx.x < DEPTH) { s_ready[threadIdx.x] = 0; s_free[threadIdx.x] = 0; }
  __syncthreads();
  const unsigned n_strips = (w4 + 63u) / 64u, strip = blockIdx.x % n_strips, layer = blockIdx.x / n_strips;
  unsigned first = layer * share, last = first + share;
  if (last > steps_per_strip) last = steps_per_strip;
  if (layer >= layers || first >= last) return;
  unsigned char *ring = dyn + ring_off;
  const unsigned col = strip * 64u + lane, cc = col < w4 ? col : w4 - 1u;
  constexpr unsigned kConsumers = 15;
  if (wave == 15) {
    // loader: keeps DEPTH slots in flight; slot d is reused for step s + DEPTH when the 15 consumers have finished step s
    for (unsigned s = first; s < last; s++) {
      const unsigned d = (s - first) % DEPTH, round = (s - first) / DEPTH;
      if (round > 0) {
        while (s_free[d] < round * kConsumers) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll 8
      for (unsigned r = 0; r < 32; r++) {
        const unsigned row = s * 32u + r, c = row < rows ? row : rows - 1u;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + ((size_t)c * w4 + cc)), LDS_PTR(ring + d * 32768u + r * 1024u), 16, 0, AUX);
      }
      // the PREVIOUS step's 32 loads have landed when at most these 32 are out (in-order retirement)
      if (s > first) {
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if (lane == 0) s_ready[(s - 1 - first) % DEPTH] = s;   // = (s - 1) + 1
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) s_ready[(last - 1 - first) % DEPTH] = last;
  } else {
    const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
    for (unsigned s = first; s < last; s++) {
      const unsigned d = (s - first) % DEPTH;
      while (s_ready[d] < s + 1u) __builtin_amdgcn_s_sleep(1);
      for (unsigned r = wave; r < 32; r += kConsumers) {
        const unsigned row = s * 32u + r;
        const uint32_t o = col < w4 && row < rows ? (row * w4 + col) << 4 : 0x80000000u;
        u4_t a = *(const u4_t *)(ring + d * 32768u + r * 1024u + lane * 16u);
        a.x ^= 1u;
        __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)o, 0, 2);
      }
      // the slot may be refilled once the LDS reads above have returned (they have: their data went into the stores' operands)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd((unsigned *)&s_free[d], 1u);
    }
  }
}

static u4_t *A[3], *B[3];
static hipEvent_t e0, e1;

template <class F>
static float timeit(F launch, int iters = 60) {
  for (int i = 0; i < 10; i++) launch(i % 3);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; i++) launch(i % 3);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

static bool check(int k, size_t n) {
  std::vector<uint32_t> a(4 * 4096), b(4 * 4096);
  bool ok = true;
  for (size_t off : {(size_t)0, n / 2, n - 4096}) {
    hipMemcpy(a.data(), A[k] + off, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), B[k] + off, b.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < a.size(); i++) ok &= (a[i] & ~1u) == (b[i] & ~1u);
  }
  return ok;
}

int main() {
  const size_t n = (size_t)ROWS * W4, bytes = n * 16;
  for (int i = 0; i < 3; i++) {
    hipMalloc(&A[i], bytes); hipMalloc(&B[i], bytes);
    std::vector<uint32_t> h(1 << 20);
    for (size_t j = 0; j < h.size(); j++) h[j] = (uint32_t)(j * 2654435761u + i) & ~1u;
    for (size_t off = 0; off < bytes; off += h.size() * 4) hipMemcpy((char *)A[i] + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice);
    hipMemset(B[i], 0, bytes);
  }
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned n_strips = (W4 + 63) / 64, steps_per_strip = (ROWS + 31) / 32, layers = 256 / n_strips, share = (steps_per_strip + layers - 1) / layers;
  const unsigned grid = n_strips * layers;
  for (int rep = 0; rep < 3; rep++) {
    {
      const float ms = timeit([&](int k) { hipLaunchKernelGGL(flat, dim3(65536), dim3(256), 0, 0, A[k], B[k], n); });
      printf("flat grid-stride copy, 65536 x 256            : %.4f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
    }
#define RUN_VGPR(D)                                                                                                                                              \
  {                                                                                                                                                              \
    hipFuncSetAttribute((const void *)walk_vgpr<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 139264);                                                         \
    const float ms = timeit([&](int k) { hipLaunchKernelGGL(walk_vgpr<D>, dim3(grid), dim3(1024), 139264, 0, A[k], B[k], W4, ROWS, (unsigned)bytes, steps_per_strip, share, layers); }); \
    printf("walk, VGPR ring depth %d, 139 KB LDS reserved   : %.4f ms  %.0f GB/s  %s\n", D, ms, 2.0 * bytes / ms / 1e6, check(0, n) ? "ok" : "WRONG");           \
  }
    RUN_VGPR(2) RUN_VGPR(3)
#define RUN_DMA(D, AUX, RES)                                                                                                                                     \
  {                                                                                                                                                              \
    const unsigned lds = (RES) + 16u * (D) * 2048u;                                                                                                              \
    hipFuncSetAttribute((const void *)walk_dma<D, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                                   \
    for (int k = 0; k < 3; k++) hipMemset(B[k], 0, bytes);                                                                                                       \
    const float ms = timeit([&](int k) { hipLaunchKernelGGL((walk_dma<D, AUX>), dim3(grid), dim3(1024), lds, 0, A[k], B[k], W4, ROWS, (unsigned)bytes, steps_per_strip, share, layers, (unsigned)(RES)); }); \
    printf("walk, own-wave LDS-DMA ring depth %d (%3u KB in flight per CU), aux %d, %3u KB reserved for the cache: %.4f ms  %.0f GB/s  %s\n", D, 32u * (D), AUX, (unsigned)(RES) / 1024u, ms, 2.0 * bytes / ms / 1e6, check(0, n) ? "ok" : "WRONG"); \
  }
    RUN_DMA(1, 0, 98304) RUN_DMA(2, 0, 98304) RUN_DMA(2, 2, 98304) RUN_DMA(3, 2, 65536) RUN_DMA(4, 0, 32768) RUN_DMA(4, 2, 32768) RUN_DMA(2, 2, 16384)
#define RUN_LOADER(D, AUX, RES)                                                                                                                                  \
  {                                                                                                                                                              \
    const unsigned lds = (RES) + (D) * 32768u;                                                                                                                   \
    hipFuncSetAttribute((const void *)walk_loader<D, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                                \
    for (int k = 0; k < 3; k++) hipMemset(B[k], 0, bytes);                                                                                                       \
    const float ms = timeit([&](int k) { hipLaunchKernelGGL((walk_loader<D, AUX>), dim3(grid), dim3(1024), lds, 0, A[k], B[k], W4, ROWS, (unsigned)bytes, steps_per_strip, share, layers, (unsigned)(RES)); }); \
    printf("walk, loader wave + 15 consumers, ring depth %d (%3u KB), aux %d, %3u KB reserved for the cache      : %.4f ms  %.0f GB/s  %s\n", D, 32u * (D), AUX, (unsigned)(RES) / 1024u, ms, 2.0 * bytes / ms / 1e6, check(0, n) ? "ok" : "WRONG"); \
  }
    RUN_LOADER(2, 2, 90112) RUN_LOADER(3, 2, 61440) RUN_LOADER(4, 2, 28672) RUN_LOADER(4, 0, 28672)
  }
  return 0;
}
