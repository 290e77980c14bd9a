// walk_bench.hip — what does the WALK of the LDS-cached table kernels (csrc/colorlut_window.hip) cost as a pure copy, and which
// shape of it streams best? One 1024-lane block per CU (134 KB of LDS reserved, as the real kernel has), a block takes a
// contiguous share of the column-major list of tiles, a tile = (256 * SW) pixels x (32 / SW) rows, two 16-byte groups per
// lane and step, DEPTH steps of pixels in flight. Against it: the flat grid-stride copy (tools/stream_bench.hip).
// Build: hipcc --offload-arch=gfx950 -O3 tools/walk_bench.hip -o tools/walk_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));

template <int SW, int DEPTH, int NTL, int NTS, int FETCH_FIRST, int ORDER>
__global__ __launch_bounds__(1024) void walk(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                             unsigned steps_per_strip, unsigned share, unsigned extra, unsigned long long *times) {
  extern __shared__ unsigned char dyn[];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];  // keeps the allocation
  const unsigned long long t_start = wall_clock64();
  constexpr unsigned RPS = 32 / SW;  // rows per step
  // ORDER 0: block b takes a contiguous share of the column-major tile list (round 4)
  // ORDER 1: aligned fronts - strip = b % n_strips, layer = b / n_strips; a layer is `share` steps tall; all blocks of a layer move down together
  // ORDER 2: row-major tile list, tile = s * grid + b (a compact window like the flat copy's; no locality for a block)
  // ORDER 3: column-major tile list, tile = s * grid + b
  const unsigned n_strips = (w4 + 64u * SW - 1u) / (64u * SW);
  unsigned first, last;
  if (ORDER == 0) {
    first = blockIdx.x * share + (blockIdx.x < extra ? blockIdx.x : extra);
    last = first + share + (blockIdx.x < extra ? 1u : 0u);
  } else if (ORDER == 1 || ORDER == 4 || ORDER == 6) {
    // ORDER 4: aligned fronts, odd layers walk bottom-up; ORDER 6: every strip starts its share at another phase (diagonal fronts)
    const unsigned strip = blockIdx.x % n_strips, layer = blockIdx.x / n_strips;   // share = steps per layer, extra = layers
    first = strip * steps_per_strip + layer * share;
    last = first + share;
    if (last > (strip + 1u) * steps_per_strip) last = (strip + 1u) * steps_per_strip;
    if (layer >= extra || first > last) first = last = 0;
  } else {
    first = 0;
    last = share + (blockIdx.x < extra ? 1u : 0u);
  }
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  struct Slot { u4_t p, q; uint32_t o0, o1; };
  Slot ring[DEPTH];
  auto fetch = [&](unsigned st_, Slot &S) {
    unsigned st = st_ < last ? st_ : last - 1u;
    if (ORDER == 4 && ((blockIdx.x / n_strips) & 1u)) st = first + (last - 1u - st);
    if (ORDER == 6) { const unsigned len = last - first, ph = ((blockIdx.x % n_strips) * len) / n_strips; st = first + ((st - first) + ph) % (len ? len : 1u); }
    unsigned strip, k;
    if (ORDER == 2) { st = st * gridDim.x + blockIdx.x; k = st / n_strips; strip = st - k * n_strips; }
    else {
      if (ORDER == 3) st = st * gridDim.x + blockIdx.x;
      strip = st / steps_per_strip; k = st - strip * steps_per_strip;
    }
    const unsigned col = strip * (64u * SW) + (wave % SW) * 64u + lane, r0 = k * RPS + 2u * (wave / SW), r1 = r0 + 1u;
    const unsigned cc = col < w4 ? col : w4 - 1u, c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
    S.o0 = col < w4 && r0 < rows ? (r0 * w4 + col) << 4 : 0x80000000u;
    S.o1 = col < w4 && r1 < rows ? (r1 * w4 + col) << 4 : 0x80000000u;
    if (NTL) {
      S.p = __builtin_nontemporal_load(src + ((size_t)c0 * w4 + cc));
      S.q = __builtin_nontemporal_load(src + ((size_t)c1 * w4 + cc));
    } else {
      S.p = src[(size_t)c0 * w4 + cc];
      S.q = src[(size_t)c1 * w4 + cc];
    }
  };
  auto step = [&](unsigned st, Slot &S) {
    u4_t a = S.p, b = S.q;
    const uint32_t so0 = S.o0, so1 = S.o1;
    a.x ^= 1u; b.y ^= 1u;
    if (FETCH_FIRST) fetch(st + DEPTH, S);
    __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)so0, 0, NTS ? 2 : 0);
    __builtin_amdgcn_raw_buffer_store_b128(b, dst_rsrc, (int)so1, 0, NTS ? 2 : 0);
    if (!FETCH_FIRST) fetch(st + DEPTH, S);
  };
  if (first < last) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(first + d, ring[d]);
  }
  unsigned st = first;
  for (; st + DEPTH <= last; st += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) step(st + d, ring[d]);
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++)
    if (st + d < last) step(st + d, ring[d]);
  __builtin_amdgcn_s_waitcnt(0);
  if (times && threadIdx.x == 0) { times[2 * blockIdx.x] = t_start; times[2 * blockIdx.x + 1] = wall_clock64(); }
}


// every WAVE claims its next piece (256 pixels x 2 rows) from one global counter, DEPTH pieces ahead: perfect balance, no
// locality of a block - the upper bound of what dynamic distribution can give this walk. MODE 0: pieces in row-major order
// (tile row by tile row); 1: column-major (down a strip).
template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void walk_dyn(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes, unsigned n_pieces,
                                                 unsigned *counter) {
  extern __shared__ unsigned char dyn[];
  const unsigned lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];
  const unsigned n_strips = (w4 + 63u) / 64u, pairs = (rows + 1u) / 2u;
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  struct Slot { u4_t p, q; uint32_t o0, o1; unsigned valid; };
  Slot ring[DEPTH];
  auto fetch = [&](Slot &S) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);
    t = __builtin_amdgcn_readfirstlane(t);
    S.valid = t < n_pieces;
    if (t >= n_pieces) t = n_pieces - 1u;
    unsigned strip, pr;
    if (MODE == 0) { pr = t / n_strips; strip = t - pr * n_strips; } else { strip = t / pairs; pr = t - strip * pairs; }
    const unsigned col = strip * 64u + lane, r0 = 2u * pr, r1 = r0 + 1u;
    const unsigned cc = col < w4 ? col : w4 - 1u, c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
    S.o0 = S.valid && col < w4 && r0 < rows ? (r0 * w4 + col) << 4 : 0x80000000u;
    S.o1 = S.valid && col < w4 && r1 < rows ? (r1 * w4 + col) << 4 : 0x80000000u;
    S.p = __builtin_nontemporal_load(src + ((size_t)c0 * w4 + cc));
    S.q = __builtin_nontemporal_load(src + ((size_t)c1 * w4 + cc));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; d++) fetch(ring[d]);
  for (;;) {
    bool any = false;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      Slot &S = ring[d];
      any |= S.valid != 0;
      u4_t a = S.p, b = S.q;
      a.x ^= 1u; b.y ^= 1u;
      __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)S.o0, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b128(b, dst_rsrc, (int)S.o1, 0, 2);
      fetch(S);
    }
    if (!any) break;
  }
}

// Teams: the blocks of a strip are grouped in teams of G (the last team of a strip may be smaller); a team owns a contiguous
// range of the strip's steps and its blocks take them one at a time from the team's counter (atomicAdd by wave 0, AHEAD steps
// before the pixels are fetched; the other waves find the step in LDS). Fast blocks take more steps: balance inside a team,
// and a block's consecutive steps are about G steps apart.
template <int DEPTH, int AHEAD>
__global__ __launch_bounds__(1024) void walk_team(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                                  unsigned steps_per_strip, unsigned blocks_per_strip, unsigned G, unsigned *counters, unsigned long long *times) {
  extern __shared__ unsigned char dyn[];
  __shared__ unsigned s_step[16], s_seq[16];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];
  if (threadIdx.x < 16) { s_seq[threadIdx.x] = 0xffffffffu; s_step[threadIdx.x] = 0; }
  __syncthreads();
  const unsigned long long t_start = wall_clock64();
  const unsigned n_strips = (w4 + 63u) / 64u;
  const unsigned strip = blockIdx.x % n_strips, in_strip = blockIdx.x / n_strips;
  if (in_strip >= blocks_per_strip) return;
  const unsigned teams = (blocks_per_strip + G - 1u) / G, team = in_strip / G;
  // the team's range of steps: proportional to its number of blocks
  const unsigned b0 = team * G, b1 = b0 + G < blocks_per_strip ? b0 + G : blocks_per_strip;
  const unsigned first = (unsigned)((unsigned long long)steps_per_strip * b0 / blocks_per_strip), last = (unsigned)((unsigned long long)steps_per_strip * b1 / blocks_per_strip);
  unsigned *counter = counters + strip * teams + team;
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  struct Slot { u4_t p, q; uint32_t o0, o1; unsigned valid; };
  Slot ring[DEPTH];
  unsigned claimed = 0;  // (wave 0) claims issued
  // wave 0, lane 0: claim number `claimed`, published for every wave as {seq = claim number, step}
  auto claim = [&]() {
    if (wave == 0) {
      if (lane == 0) {
        const unsigned t = first + atomicAdd(counter, 1u);
        *(volatile unsigned *)&s_step[claimed & 15u] = t;
        __threadfence_block();
        *(volatile unsigned *)&s_seq[claimed & 15u] = claimed;
      }
      claimed++;
    }
  };
  auto fetch = [&](unsigned k, Slot &S) {  // the pixels of this block's k-th step
    unsigned st;
    for (;;) {
      if (*(volatile unsigned *)&s_seq[k & 15u] == k) { st = *(volatile unsigned *)&s_step[k & 15u]; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    st = __builtin_amdgcn_readfirstlane(st);
    S.valid = st < last;
    if (st >= last) st = last - 1u;
    const unsigned col = strip * 64u + lane, r0 = st * 32u + 2u * wave, r1 = r0 + 1u;
    const unsigned cc = col < w4 ? col : w4 - 1u, c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
    S.o0 = S.valid && col < w4 && r0 < rows ? (r0 * w4 + col) << 4 : 0x80000000u;
    S.o1 = S.valid && col < w4 && r1 < rows ? (r1 * w4 + col) << 4 : 0x80000000u;
    S.p = __builtin_nontemporal_load(src + ((size_t)c0 * w4 + cc));
    S.q = __builtin_nontemporal_load(src + ((size_t)c1 * w4 + cc));
  };
  for (int a = 0; a < DEPTH + AHEAD; a++) claim();
  unsigned k = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; d++) fetch(k + d, ring[d]);
  for (;;) {
    bool any = false;
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      Slot &S = ring[d];
      any |= S.valid != 0;
      u4_t a = S.p, b = S.q;
      a.x ^= 1u; b.y ^= 1u;
      claim();
      __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)S.o0, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b128(b, dst_rsrc, (int)S.o1, 0, 2);
      fetch(k + DEPTH, S);
      k++;
    }
    if (!any) break;
  }
  __builtin_amdgcn_s_waitcnt(0);
  if (times && threadIdx.x == 0) { times[2 * blockIdx.x] = t_start; times[2 * blockIdx.x + 1] = wall_clock64(); }
}

// the same walk with SMALL blocks: NW waves per block (a step = 256 pixels x 2 NW rows), several blocks per CU
template <int NW, int DEPTH, int ORDER>
__global__ __launch_bounds__(NW * 64) void walk_small(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                                      unsigned steps_per_strip, unsigned share, unsigned extra) {
  extern __shared__ unsigned char dyn[];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];
  constexpr unsigned RPS = 2 * NW;
  const unsigned n_strips = (w4 + 63u) / 64u;
  unsigned first, last;
  if (ORDER == 0) {
    first = blockIdx.x * share + (blockIdx.x < extra ? blockIdx.x : extra);
    last = first + share + (blockIdx.x < extra ? 1u : 0u);
  } else {
    first = 0;
    last = share + (blockIdx.x < extra ? 1u : 0u);
  }
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  struct Slot { u4_t p, q; uint32_t o0, o1; };
  Slot ring[DEPTH];
  auto fetch = [&](unsigned st_, Slot &S) {
    unsigned st = st_ < last ? st_ : last - 1u;
    unsigned strip, k;
    if (ORDER == 2) { st = st * gridDim.x + blockIdx.x; k = st / n_strips; strip = st - k * n_strips; }
    else { strip = st / steps_per_strip; k = st - strip * steps_per_strip; }
    const unsigned col = strip * 64u + lane, r0 = k * RPS + 2u * wave, r1 = r0 + 1u;
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
  if (first < last) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fetch(first + d, ring[d]);
  }
  unsigned st = first;
  for (; st + DEPTH <= last; st += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) step(st + d, ring[d]);
  }
#pragma unroll
  for (int d = 0; d < DEPTH - 1; d++)
    if (st + d < last) step(st + d, ring[d]);
}

// the flat copy as a PERSISTENT, software-pipelined loop: DEPTH loads in flight per lane, a store never waited for before the next load's
// data is used (what the walk does, on the flat copy's linear addresses). Is it persistence or the 2D pattern that costs the walk 8-25 %?
template <int DEPTH>
__global__ __launch_bounds__(256) void flat_pipe(const u4_t *__restrict__ a, u4_t *__restrict__ b, size_t n) {
  const size_t s = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u4_t v[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; d++) v[d] = __builtin_nontemporal_load(a + (i + d * s < n ? i + d * s : n - 1));
  for (; i + (DEPTH - 1) * s < n; i += DEPTH * s) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      u4_t x = v[d];
      x.x ^= 1;
      __builtin_nontemporal_store(x, b + i + d * s);
      const size_t nx = i + (d + DEPTH) * s;
      v[d] = __builtin_nontemporal_load(a + (nx < n ? nx : n - 1));
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
    if (i + d * s < n) { u4_t x = v[d]; x.x ^= 1; __builtin_nontemporal_store(x, b + i + d * s); }
}

// aligned fronts with NARROW strips: a row segment of a wave is 64 / NR lanes (1024 / NR bytes), a wave's load covers NR rows, a step
// 32 * NR rows; strips = 15 * NR, layers = grid / strips: fewer, thicker bands of full rows (NR = 4: 60 strips x 4 layers of 128 rows)
template <int NR, int DEPTH>
__global__ __launch_bounds__(1024) void walk_narrow(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows, unsigned dst_bytes,
                                                    unsigned steps_per_strip, unsigned share, unsigned layers) {
  extern __shared__ unsigned char dyn[];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (dyn[threadIdx.x] == 77 && rows == 1234567u) dst[0] = src[1];
  constexpr unsigned LPR = 64 / NR, RPS = 32 * NR;  // lanes per row segment, rows per step
  const unsigned n_strips = (w4 + LPR - 1u) / LPR;
  const unsigned strip = blockIdx.x % n_strips, layer = blockIdx.x / n_strips;
  unsigned first = layer * share, last = first + share;
  if (last > steps_per_strip) last = steps_per_strip;
  if (layer >= layers || first >= last) return;
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  struct Slot { u4_t p, q; uint32_t o0, o1; };
  Slot ring[DEPTH];
  auto fetch = [&](unsigned st_, Slot &S) {
    const unsigned k = st_ < last ? st_ : last - 1u;
    const unsigned col = strip * LPR + (lane % LPR), r0 = k * RPS + wave * (2u * NR) + lane / LPR, r1 = r0 + NR;
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

__global__ __launch_bounds__(256) void flat(const u4_t *__restrict__ a, u4_t *__restrict__ b, size_t n) {
  const size_t s = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += s) { u4_t v = __builtin_nontemporal_load(a + i); v.x ^= 1; __builtin_nontemporal_store(v, b + i); }
}

static const unsigned W4 = 960, ROWS = 2160 * 8;
static u4_t *A[3], *B[3];
static hipEvent_t e0, e1;

template <class L>
static float timeit(L &&launch) {
  for (int w = 0; w < 6; w++) launch(w % 3);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    for (int w = 0; w < 30; w++) launch(w % 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms / 30 < best) best = ms / 30;
  }
  return best;
}

template <int SW, int DEPTH, int NTL, int NTS, int FF, int ORDER = 0>
static void run_walk(int blocks_per_cu_x1, int lds) {
  const unsigned n_strips = (W4 + 64 * SW - 1) / (64 * SW), sps = (ROWS + 32 / SW - 1) / (32 / SW), total = n_strips * sps;
  const unsigned grid = 256 * blocks_per_cu_x1;
  unsigned share = total / grid, extra = total % grid;
  if (ORDER == 1 || ORDER == 4 || ORDER == 6) { extra = grid / n_strips; share = (sps + extra - 1) / extra; }
  const float ms = timeit([&](int k) {
    hipLaunchKernelGGL((walk<SW, DEPTH, NTL, NTS, FF, ORDER>), dim3(grid), dim3(1024), lds, 0, A[k], B[k], W4, ROWS, (unsigned)(ROWS * W4 * 16), sps, share, extra, (unsigned long long *)nullptr);
  });
  {  // one more launch with the block clocks: when did each block start and end (100 MHz ticks)
    unsigned long long *d_t; std::vector<unsigned long long> h(2 * grid);
    hipMalloc(&d_t, 16 * grid); hipMemset(d_t, 0, 16 * grid);
    hipLaunchKernelGGL((walk<SW, DEPTH, NTL, NTS, FF, ORDER>), dim3(grid), dim3(1024), lds, 0, A[0], B[0], W4, ROWS, (unsigned)(ROWS * W4 * 16), sps, share, extra, d_t);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_t, 16 * grid, hipMemcpyDeviceToHost); hipFree(d_t);
    unsigned long long t0 = ~0ull; std::vector<double> ends, durs;
    for (unsigned b = 0; b < grid; b++) if (h[2 * b + 1]) t0 = h[2 * b] < t0 ? h[2 * b] : t0;
    for (unsigned b = 0; b < grid; b++) if (h[2 * b + 1]) { ends.push_back((h[2 * b + 1] - t0) * 0.01); durs.push_back((h[2 * b + 1] - h[2 * b]) * 0.01); }
    {  // by XCD (block b runs on XCD b % 8) and by the CU's place in the dispatch order (b / 8)
      double sx[8] = {0}, nx[8] = {0};
      for (unsigned b = 0; b < grid; b++) if (h[2 * b + 1]) { sx[b % 8] += (h[2 * b + 1] - h[2 * b]) * 0.01; nx[b % 8] += 1; }
      printf("   mean duration by XCD:");
      for (int x = 0; x < 8; x++) printf(" %.1f", sx[x] / (nx[x] > 0 ? nx[x] : 1));
      printf("\n   by b/8 (groups of 4):");
      for (unsigned g4 = 0; g4 < grid / 8; g4 += 4) { double a = 0; int n = 0; for (unsigned b = g4 * 8; b < (g4 + 4) * 8 && b < grid; b++) if (h[2 * b + 1]) { a += (h[2 * b + 1] - h[2 * b]) * 0.01; n++; } printf(" %.1f", a / (n ? n : 1)); }
      printf("\n");
    }
    std::sort(ends.begin(), ends.end()); std::sort(durs.begin(), durs.end());
    printf("   block ends (us after the first start): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f ; durations min %.1f median %.1f max %.1f\n", ends.front(), ends[ends.size() / 10],
           ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back(), durs.front(), durs[durs.size() / 2], durs.back());
  }
  printf("order %d  tile %4d x %2d  depth %d  nt load %d store %d  fetch-first %d  grid %4u lds %6d : %.4f ms  %.0f GB/s\n", ORDER, 256 * SW, 32 / SW, DEPTH, NTL, NTS, FF, grid, lds, ms,
         2.0 * ROWS * W4 * 16 / ms / 1e6);
  fflush(stdout);
}


template <int DEPTH, int MODE>
static void run_dyn(int lds) {
  const unsigned n_strips = (W4 + 63) / 64, n_pieces = n_strips * ((ROWS + 1) / 2);
  unsigned *d_c; hipMalloc(&d_c, 4);
  const float ms = timeit([&](int k) {
    hipMemsetAsync(d_c, 0, 4, 0);
    hipLaunchKernelGGL((walk_dyn<DEPTH, MODE>), dim3(256), dim3(1024), lds, 0, A[k], B[k], W4, ROWS, (unsigned)(ROWS * W4 * 16), n_pieces, d_c);
  });
  printf("dynamic per-wave claims  mode %d (%s)  depth %d  lds %6d : %.4f ms  %.0f GB/s (includes a 4-byte memset per launch)\n", MODE, MODE ? "column-major" : "row-major", DEPTH, lds, ms,
         2.0 * ROWS * W4 * 16 / ms / 1e6);
  hipFree(d_c);
}

template <int DEPTH, int AHEAD>
static void run_team(unsigned G, int lds) {
  const unsigned n_strips = (W4 + 63) / 64, sps = (ROWS + 31) / 32, bps = 256 / n_strips, teams = (bps + G - 1) / G;
  unsigned *d_c; hipMalloc(&d_c, 4 * n_strips * teams);
  const float ms = timeit([&](int k) {
    hipMemsetAsync(d_c, 0, 4 * n_strips * teams, 0);
    hipLaunchKernelGGL((walk_team<DEPTH, AHEAD>), dim3(n_strips * bps), dim3(1024), lds, 0, A[k], B[k], W4, ROWS, (unsigned)(ROWS * W4 * 16), sps, bps, G, d_c, (unsigned long long *)nullptr);
  });
  {
    const unsigned grid = n_strips * bps;
    unsigned long long *d_t; std::vector<unsigned long long> h(2 * grid);
    hipMalloc(&d_t, 16 * grid); hipMemset(d_t, 0, 16 * grid);
    hipMemsetAsync(d_c, 0, 4 * n_strips * teams, 0);
    hipLaunchKernelGGL((walk_team<DEPTH, AHEAD>), dim3(grid), dim3(1024), lds, 0, A[0], B[0], W4, ROWS, (unsigned)(ROWS * W4 * 16), sps, bps, G, d_c, d_t);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_t, 16 * grid, hipMemcpyDeviceToHost); hipFree(d_t);
    unsigned long long t0 = ~0ull; std::vector<double> ends;
    for (unsigned b = 0; b < grid; b++) if (h[2 * b + 1]) t0 = h[2 * b] < t0 ? h[2 * b] : t0;
    for (unsigned b = 0; b < grid; b++) if (h[2 * b + 1]) ends.push_back((h[2 * b + 1] - t0) * 0.01);
    std::sort(ends.begin(), ends.end());
    printf("   block ends: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f\n", ends.front(), ends[ends.size() / 10], ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back());
  }
  printf("teams of %2u  depth %d ahead %d  lds %6d : %.4f ms  %.0f GB/s (includes a small memset per launch)\n", G, DEPTH, AHEAD, lds, ms, 2.0 * ROWS * W4 * 16 / ms / 1e6);
  fflush(stdout);
  hipFree(d_c);
}

template <int NW, int DEPTH, int ORDER>
static void run_small(unsigned grid, int lds) {
  const unsigned n_strips = (W4 + 63) / 64, sps = (ROWS + 2 * NW - 1) / (2 * NW), total = n_strips * sps;
  const float ms = timeit([&](int k) {
    hipLaunchKernelGGL((walk_small<NW, DEPTH, ORDER>), dim3(grid), dim3(NW * 64), lds, 0, A[k], B[k], W4, ROWS, (unsigned)(ROWS * W4 * 16), sps, total / grid, total % grid);
  });
  printf("small blocks: %d waves (256 x %2d per step)  depth %d  order %d  grid %5u (%.1f steps each)  lds %6d : %.4f ms  %.0f GB/s\n", NW, 2 * NW, DEPTH, ORDER, grid, (double)total / grid, lds, ms,
         2.0 * ROWS * W4 * 16 / ms / 1e6);
  fflush(stdout);
}

template <int NR, int DEPTH>
static void run_narrow(int lds) {
  const unsigned n_strips = (W4 + 64 / NR - 1) / (64 / NR), sps = (ROWS + 32 * NR - 1) / (32 * NR), layers = 256 / n_strips, share = (sps + layers - 1) / layers;
  const float ms = timeit([&](int k) {
    hipLaunchKernelGGL((walk_narrow<NR, DEPTH>), dim3(n_strips * layers), dim3(1024), lds, 0, A[k], B[k], W4, ROWS, (unsigned)(ROWS * W4 * 16), sps, share, layers);
  });
  printf("narrow strips: %3d px wide, %3d rows per step, %2u strips x %2u layers (%3u blocks)  depth %d : %.4f ms  %.0f GB/s\n", 256 / NR, 32 * NR, n_strips, layers, n_strips * layers, DEPTH, ms,
         2.0 * ROWS * W4 * 16 / ms / 1e6);
  fflush(stdout);
}

int main() {
  const size_t bytes = (size_t)ROWS * W4 * 16;
  for (int i = 0; i < 3; i++) { hipMalloc(&A[i], bytes); hipMalloc(&B[i], bytes); hipMemset(A[i], 1 + i, bytes); hipMemset(B[i], 0, bytes); }
  hipEventCreate(&e0); hipEventCreate(&e1);

  for (int grid : {4096, 16384, 65536}) {
    const float ms = timeit([&](int k) { hipLaunchKernelGGL(flat, dim3(grid), dim3(256), 0, 0, A[k], B[k], bytes / 16); });
    printf("flat grid-stride copy (nt)  grid %6d : %.4f ms  %.0f GB/s\n", grid, ms, 2.0 * bytes / ms / 1e6);
  }
  for (int grid : {1024, 2048, 4096, 8192}) {
    const float m2 = timeit([&](int k) { hipLaunchKernelGGL(flat_pipe<2>, dim3(grid), dim3(256), 0, 0, A[k], B[k], bytes / 16); });
    const float m3 = timeit([&](int k) { hipLaunchKernelGGL(flat_pipe<3>, dim3(grid), dim3(256), 0, 0, A[k], B[k], bytes / 16); });
    const float m4 = timeit([&](int k) { hipLaunchKernelGGL(flat_pipe<4>, dim3(grid), dim3(256), 0, 0, A[k], B[k], bytes / 16); });
    printf("flat persistent pipelined copy  grid %5d : depth 2 %.4f  depth 3 %.4f  depth 4 %.4f ms\n", grid, m2, m3, m4);
  }
  const int L = 134144;
  for (int rep = 0; rep < 2; rep++) {
    run_walk<1, 3, 1, 1, 0, 1>(1, L);
    run_walk<1, 2, 1, 1, 0, 1>(1, L);
    run_narrow<1, 2>(L);
    run_narrow<2, 2>(L);
    run_narrow<4, 2>(L);
    run_narrow<8, 2>(L);
    run_narrow<16, 2>(L);
    run_narrow<4, 1>(L);
    run_narrow<4, 3>(L);
    run_walk<1, 3, 1, 1, 0, 2>(1, L);
  }
  return 0;
}
