// hetero_probe.hip — round 6: do a TEXTURE-ADDRESS-bound body and a VALU-bound body ADD UP when their waves live on the same CUs?
// The memoised-table gather kernel (csrc/colorlut_kernels.hip: colorlut_table_tiled_kernel) retires a divergent dword gather at about
// one lane per clock and CU and leaves the VALU idle; the interpolating kernels (csrc/colorlut_brick.hip) issue ~52 VALU + ~10 LDS
// instructions per pixel and leave the texture path idle. Two separate launches do not overlap (tools/split_probe.py). Here: ONE
// persistent kernel whose waves take one of two roles; the 256 x 2 pixel tiles are split between the roles (the gather waves stride over
// the first tiles_g tiles, the arithmetic waves over the rest; the split is swept - a queue on one atomic costs 23 ns per claim):
//   role G  the gather kernel's body: LDS transpose, eight table gathers per lane (Morton table of 2^24 entries)
//   role A  a STAND-IN for the interpolating kernels: ~56 dependent-ish VALU operations and 10 LDS reads per pixel (not colorlut:
//           this probe asks about issue slots, not results)
// over 8 x 4K frames of a smooth picture + uniform noise of +-amp. Prints ms per batch for G alone, A alone, and mixes of G:A waves.
// Build: hipcc --offload-arch=gfx950 -O3 tools/hetero_probe.hip -o tools/hetero_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
static constexpr unsigned W = 3840, H = 2160, NF = 8, W4 = W / 4, ROWS = H * NF;
static constexpr unsigned TILE_COLS = W4 / 64, TILES = TILE_COLS * (ROWS / 2);   // a tile: 256 px x 2 rows = 64 lanes x 2 groups

__device__ __forceinline__ uint32_t spread3(uint32_t v) {
  v &= 0xffu;
  v = (v | (v << 8)) & 0x0000f00fu;
  v = (v | (v << 4)) & 0x000c30c3u;
  v = (v | (v << 2)) & 0x00249249u;
  return v;
}

__global__ void fill_table(uint32_t *t) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  t[i] = (i * 2654435761u) & 0x00ffffffu;
}

// roles: wave w of a block is an arithmetic wave if w < n_arith, else a gather wave. queue[0] = tiles claimed, [1] = taken from the
// front, [2] = taken from the back.
__global__ __launch_bounds__(256) void hetero(const u4_t *__restrict__ src, u4_t *__restrict__ dst, const uint32_t *__restrict__ table, unsigned tiles_g,
                                              unsigned n_arith, unsigned long long *__restrict__ taken, int other_role) {
  __shared__ uint32_t s_spread[256];
  __shared__ uint32_t s_strip[4][512];
  __shared__ float s_lut[1024];
  s_spread[threadIdx.x] = spread3(threadIdx.x);
  for (int i = threadIdx.x; i < 1024; i += 256) s_lut[i] = 0.25f + 0.0005f * (float)i;
  __syncthreads();
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool arith = wave < n_arith;
  uint32_t *x = s_strip[wave];
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  unsigned mine = 0;
  // this wave's place among the waves of its role, and how many of them there are
  const unsigned n_role = gridDim.x * (arith ? n_arith : 4u - n_arith), my = blockIdx.x * (arith ? n_arith : 4u - n_arith) + (arith ? wave : wave - n_arith);
  const unsigned t_first = arith ? tiles_g : 0u, t_end = arith ? TILES : tiles_g;
  for (unsigned t = t_first + my; t < t_end; t += n_role) {
    mine++;
    const unsigned tx = t % TILE_COLS, ty = t / TILE_COLS;
    const size_t i0 = (size_t)(2 * ty) * W4 + tx * 64 + lane, i1 = i0 + W4;
    const u4_t p = __builtin_nontemporal_load(src + i0), q = __builtin_nontemporal_load(src + i1);
    u4_t a, b;
    if (!arith) {
      *(u4_t *)(x + lane * 4) = p;
      *(u4_t *)(x + 256 + lane * 4) = q;
      wave_sync();
      uint32_t px[8], o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) px[j] = x[j * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = table[s_spread[px[j] & 0xffu] | (s_spread[(px[j] >> 8) & 0xffu] << 1) | (s_spread[(px[j] >> 16) & 0xffu] << 2)];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j * 64 + lane] = (o[j] & 0x00ffffffu) | (px[j] & 0xff000000u);
      wave_sync();
      a = *(u4_t *)(x + lane * 4);
      b = *(u4_t *)(x + 256 + lane * 4);
      wave_sync();
    } else if (other_role == 1) {
      // role S: the same table lookups through the SCALAR memory path: lane by lane, v_readlane -> s_buffer_load_dword -> v_writelane,
      // sixteen loads in flight per wave (what the texture-address path does for the gather waves, the scalar cache does here)
      const uint32_t px[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
      uint32_t o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t idx4 = (s_spread[px[j] & 0xffu] | (s_spread[(px[j] >> 8) & 0xffu] << 1) | (s_spread[(px[j] >> 16) & 0xffu] << 2)) << 2;
        uint32_t oj = 0;
#define HP_LOAD(K) { const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)idx4, (K)); asm volatile("s_load_dword %0, %1, %2" : "=s"(r[(K) & 15]) : "s"(table), "s"(off)); }
#define HP_PUT(K) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(oj) : "s"(r[(K) & 15]), "n"(K));
#define HP_GROUP(L0) { uint32_t r[16]; \
          HP_LOAD(L0 + 0) HP_LOAD(L0 + 1) HP_LOAD(L0 + 2) HP_LOAD(L0 + 3) HP_LOAD(L0 + 4) HP_LOAD(L0 + 5) HP_LOAD(L0 + 6) HP_LOAD(L0 + 7) \
          HP_LOAD(L0 + 8) HP_LOAD(L0 + 9) HP_LOAD(L0 + 10) HP_LOAD(L0 + 11) HP_LOAD(L0 + 12) HP_LOAD(L0 + 13) HP_LOAD(L0 + 14) HP_LOAD(L0 + 15) \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
          HP_PUT(L0 + 0) HP_PUT(L0 + 1) HP_PUT(L0 + 2) HP_PUT(L0 + 3) HP_PUT(L0 + 4) HP_PUT(L0 + 5) HP_PUT(L0 + 6) HP_PUT(L0 + 7) \
          HP_PUT(L0 + 8) HP_PUT(L0 + 9) HP_PUT(L0 + 10) HP_PUT(L0 + 11) HP_PUT(L0 + 12) HP_PUT(L0 + 13) HP_PUT(L0 + 14) HP_PUT(L0 + 15) }
        HP_GROUP(0) HP_GROUP(16) HP_GROUP(32) HP_GROUP(48)
        o[j] = (oj & 0x00ffffffu) | (px[j] & 0xff000000u);
      }
      a.x = o[0]; a.y = o[1]; a.z = o[2]; a.w = o[3];
      b.x = o[4]; b.y = o[5]; b.z = o[6]; b.w = o[7];
    } else {
      const uint32_t px[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
      uint32_t o[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        // ~56 VALU operations and 10 LDS reads per pixel (trilinear-shaped: three coordinates, eight corners x three channels of lerps)
        float r = (float)(px[j] & 0xffu) * (1.0f / 255.0f) * 32.0f, g = (float)((px[j] >> 8) & 0xffu) * (1.0f / 255.0f) * 32.0f,
              bl = (float)((px[j] >> 16) & 0xffu) * (1.0f / 255.0f) * 32.0f;
        const int ir = (int)r, ig = (int)g, ib = (int)bl;
        const float fr = r - (float)ir, fg = g - (float)ig, fb = bl - (float)ib;
        const int base = (ir + 7 * ig + 13 * ib) & 1023;
        float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const float v = s_lut[(base + 37 * c) & 1023];
          const float wgt = ((c & 1) ? fr : 1.0f - fr) * ((c & 2) ? fg : 1.0f - fg) * ((c & 4) ? fb : 1.0f - fb);
          acc[0] = acc[0] + v * wgt;
          acc[1] = acc[1] + (v + 0.1f) * wgt;
          acc[2] = acc[2] + (v + 0.2f) * wgt;
        }
        const float e0 = s_lut[(base + 5) & 1023], e1 = s_lut[(base + 11) & 1023];
        const uint32_t c0 = (uint32_t)(acc[0] * e0 * 255.0f) & 0xffu, c1 = (uint32_t)(acc[1] * e1 * 255.0f) & 0xffu, c2 = (uint32_t)(acc[2] * 255.0f) & 0xffu;
        o[j] = c0 | (c1 << 8) | (c2 << 16) | (px[j] & 0xff000000u);
      }
      a.x = o[0]; a.y = o[1]; a.z = o[2]; a.w = o[3];
      b.x = o[4]; b.y = o[5]; b.z = o[6]; b.w = o[7];
    }
    __builtin_nontemporal_store(a, dst + i0);
    __builtin_nontemporal_store(b, dst + i1);
  }
  if (lane == 0 && taken) atomicAdd(&taken[arith ? 1 : 0], (unsigned long long)mine);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const size_t n4 = (size_t)W4 * ROWS;
  u4_t *src, *dst;
  uint32_t *table;
  unsigned long long *taken;
  CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMalloc(&table, (size_t)(1u << 24) * 4)); CK(hipMalloc(&taken, 16));
  hipLaunchKernelGGL(fill_table, dim3((1u << 24) / 256), dim3(256), 0, 0, table);
  std::vector<uint32_t> h((size_t)W * ROWS), ref_g, ref_s;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks_per_cu = argc > 1 ? std::atoi(argv[1]) : 8;
  for (int amp : {0, 4, 8}) {
    uint64_t s = 88172645463325252ull;
    for (unsigned y = 0; y < ROWS; y++)
      for (unsigned xx = 0; xx < W; xx++) {
        const unsigned fy = y % H;
        int c[3] = {(int)(40 + 150.0 * xx / W + 30.0 * fy / H), (int)(200 - 120.0 * fy / H), (int)(60 + 80.0 * xx / W + 60.0 * fy / H)};
        for (int k = 0; k < 3; k++) {
          s ^= s << 13; s ^= s >> 7; s ^= s << 17;
          if (amp) c[k] += (int)(s % (2 * amp + 1)) - amp;
          c[k] = c[k] < 0 ? 0 : (c[k] > 255 ? 255 : c[k]);
        }
        h[(size_t)y * W + xx] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | 0xff000000u;
      }
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int other_role = 0; other_role < 2; other_role++) {
    std::printf("amp %d, %d blocks of 4 waves per CU, other role = %s:", amp, blocks_per_cu, other_role ? "S (scalar-path gather)" : "A (arithmetic stand-in)");
    struct Mix { unsigned n_arith; double share_a; };
    for (Mix m : {Mix{0u, 0.0}, Mix{4u, 1.0}, Mix{2u, 0.3}, Mix{2u, 0.4}, Mix{2u, 0.5}, Mix{2u, 0.6}, Mix{1u, 0.08}, Mix{1u, 0.12}, Mix{1u, 0.16}, Mix{1u, 0.2}, Mix{1u, 0.3}, Mix{3u, 0.6}, Mix{3u, 0.7}}) {
      const unsigned n_arith = m.n_arith;
      const unsigned tiles_g = (unsigned)((1.0 - m.share_a) * TILES);
      float best = 1e9f;
      unsigned long long tk[2] = {0, 0};
      for (int rep = 0; rep < 6; rep++) {
        CK(hipMemsetAsync(taken, 0, 16, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(hetero, dim3(256 * blocks_per_cu), dim3(256), 0, 0, src, dst, table, tiles_g, n_arith, taken, other_role);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 1 && ms < best) best = ms;
        CK(hipMemcpy(tk, taken, 16, hipMemcpyDeviceToHost));
      }
      std::printf("  G:%s %u:%u %.4f ms (%s took %.0f %%)", other_role ? "S" : "A", 4 - n_arith, n_arith, best, other_role ? "S" : "A", 100.0 * (double)tk[1] / (double)(tk[0] + tk[1] ? tk[0] + tk[1] : 1));
      if (other_role == 1 && (n_arith == 0u || n_arith == 4u)) {   // the scalar path must give what the gather path gives
        std::vector<uint32_t> &keep = n_arith == 0u ? ref_g : ref_s;
        keep.resize(h.size());
        CK(hipMemcpy(keep.data(), dst, h.size() * 4, hipMemcpyDeviceToHost));
        if (n_arith == 4u) std::printf(" [S == G: %s]", ref_g == ref_s ? "ok" : "WRONG");
      }
    }
    std::printf("\n");
    std::fflush(stdout);
    }
  }
  return 0;
}
