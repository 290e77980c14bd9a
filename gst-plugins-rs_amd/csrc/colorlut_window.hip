// colorlut_window.hip — memoised-table colorlut (and fused hsvfilter -> colorlut) for packed RGBA8 with the table lookups
// served from LDS instead of the texture-address path.
//
// Replaces the per-pixel loop of video/colorlut/src/colorlut/imp.rs:267-294 (transform_rgba_3d over a frame) like the
// other table kernels (colorlut_kernels.hip §"full-domain table"): out = table[colour], alpha passed through; the table
// was produced by the exact interpolating kernel over all 2^24 colours, so the result is that kernel's, bit for bit.
//
// Why: a divergent dword gather retires at about ONE LANE per clock and CU even when every lane hits L1 (DESIGN 4.2b:
// 64-colour palette 0.119 ms per 8x4K against a 0.093 ms streaming floor), and an L1 miss moves a 128 B line for 4 bytes.
// A `ds_read_b32` serves 32 lanes per clock. So each block (ONE per CU, 16 waves) keeps a 2-way set-associative cache of
// table BRICKS (4x4x4 colours = 64 entries = 256 B, contiguous in a Morton-indexed table, colorlut_window.hpp) in 128 KiB of
// LDS: 256 sets = the brick's position in a box of 8x8x4 bricks (32x32x16 levels), so one compact cloud of colours (a
// smooth region + noise) never collides with itself and two clouds (an edge) get a way each.
//   per pixel: three axis-table reads (slot = X[r] + Y[g] + Z[b]), then ONE more LDS round trip: the set's {tags,
//   generation}, the entry in both ways (the ways of an entry are neighbours: one ds_read_b64), the generation again;
//   ~14 VALU instructions, no global access.
//   miss: the wave installs up to kWinFills of the bricks it missed: the first lane of each distinct brick takes the
//   set's lock (ds_cmpst), invalidates the victim way and bumps the generation in ONE 8-byte write - all leaders at once -,
//   the wave copies the 256 B bricks (all in flight together), the leaders publish the tags and drop the locks; the
//   missed pixels are looked up again, and a lane whose brick is still not there (set locked by another wave, too many
//   distinct bricks) reads table[slot] from global memory - always exact, the cache is only ever a copy.
//   No barrier after the prologue: readers validate instead - the LDS executes a wave's operations in order, so a reader
//   whose generation read AFTER its entry read still shows the value read BEFORE it cannot have overlapped an install
//   into that set (the install's first LDS operation is the generation bump).
// A block walks down a 256-pixel-wide strip of the batch (frames stacked), 32 rows per step, two rows per wave, the
// pixels of the next two steps in flight while this one is looked up; neighbouring rows share their colours, which is what
// keeps the cache warm (tools/window_cache_sim.py). Each block's first step is cold.
// Algorithmic traffic: 4 B read + 4 B written per pixel; the table traffic is 256 B per installed brick from L2.
#include "internal.hpp"
#include "colorlut_window.hpp"

namespace mi355 {

namespace {

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) u2_t lds_u2;
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
typedef volatile __attribute__((address_space(3))) u2_t lds_vu2;

// All LDS of the kernel is the dynamic allocation and there are no static __shared__ objects, so it starts at address 0
// and the map below is in absolute byte addresses (table bases fold into the ds offset field).
constexpr uint32_t kWinAxis = 0;              // 3 x 256 x u32: slot contributions per channel value
constexpr uint32_t kWinMeta = 3072;           // 256 sets x {tag way 0 | tag way 1 << 16, generation}; next victim = generation & 1
constexpr uint32_t kWinLock = 5120;           // 256 sets x lock word (0 = free)
constexpr uint32_t kWinProgress = 6144;       // 16 waves x steps done (WIN_PACE)
constexpr uint32_t kWinData = 8192;           // 16 K entries (slot & 0x3fff) x {way 0, way 1}: 128 KiB
constexpr uint32_t kWinLdsBytes = kWinData + 16384 * 8;  // 139,264 B: one block per CU
constexpr int kWinWaves = 16;
constexpr unsigned kWinRowsPerStep = 2 * kWinWaves;
#ifndef WIN_FILLS  // bricks a wave installs per step at most. Round 6: 4 (8 before): every election round and every brick in flight is paid by
#define WIN_FILLS 4  // all the waves that miss, and what a step misses is mostly the same few new bricks in all sixteen (tools/r06_miss_probe.sh)
#endif
#ifndef WIN_DEPTH
#define WIN_DEPTH 2
#endif
#ifndef WIN_PACE  // steps a wave may be ahead of the slowest wave of its block (0 = no pace keeping)
#define WIN_PACE 0
#endif
#ifndef WIN_WAIT  // 1: a lane whose brick another wave is installing waits for it before the second look (0: gathers from the table)
#define WIN_WAIT 0
#endif
#ifndef WIN_EXP
#define WIN_EXP 0
#endif
constexpr int kWinFills = WIN_FILLS;          // bricks a wave installs per step at most
constexpr int kWinCounterSlots = 1024;
static_assert(kWinLdsBytes <= 160 * 1024, "fits the CU");

__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(const lds_u32 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w32(uint32_t a, uint32_t v) { *(lds_u32 *)(lds_byte *)(uintptr_t)a = v; }
// the cache protocol's accesses: volatile, so that the compiler keeps them in program order (the LDS keeps a wave's
// operations in that order, which is what the validation relies on)
__device__ __forceinline__ uint32_t lds_r32v(uint32_t a) { return *(lds_vu32 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w32v(uint32_t a, uint32_t v) { *(lds_vu32 *)(lds_byte *)(uintptr_t)a = v; }
__device__ __forceinline__ u2_t lds_r64v(uint32_t a) { return *(lds_vu2 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w64v(uint32_t a, u2_t v) { *(lds_vu2 *)(lds_byte *)(uintptr_t)a = v; }

// byte BYTE of px times 4 (the axis-table entry's byte offset) in one SDWA shift
template <int BYTE>
__device__ __forceinline__ uint32_t byte_times4(uint32_t px, uint32_t two) {
  uint32_t o;
  if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "v"(two), "v"(px));
  else if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "v"(two), "v"(px));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "v"(two), "v"(px));
  return o;
}

// meta entry of the set of brick `brick` (brick = slot >> 6: set in the low 8 bits, tag above)
__device__ __forceinline__ uint32_t meta_addr(uint32_t brick) { return kWinMeta + ((brick & 255u) << 3); }
// entry `k` of brick `brick` in way `way`: the two ways of an entry are neighbours (one ds_read_b64 reads both)
__device__ __forceinline__ uint32_t data_addr(uint32_t brick, uint32_t k, uint32_t way) { return kWinData + ((((brick & 255u) << 6) + k) << 3) + 4u * way; }


}  // namespace

// grid = blocks of 1024 lanes, one per CU; block b takes a contiguous share of the column-major list of 256 x 32 pixel
// tiles (strip after strip, top to bottom).
// MULTI: the rows are those of up to kMultiFrames SEPARATE packed frames of frame_rows rows each (group.hip: the frames of many
// streams in one launch): row r belongs to frame r / frame_rows, whose base pointers travel in the kernel arguments; a wave's two
// rows of a step are wave-uniform, so the frame of a load or a store is an SGPR matter. The walk, the cache and the results are
// those of the one-buffer form.
template <bool MULTI>
__global__ __launch_bounds__(1024) void colorlut_window_kernel(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned sw4, unsigned dw4,
                                                                 unsigned rows, unsigned dst_bytes, unsigned steps_per_strip, unsigned share, unsigned extra,
                                                                 unsigned fronts, const uint32_t *__restrict__ table, unsigned long long *__restrict__ counters,
                                                                 MultiFramePtrs srcs, MultiFramePtrs dsts, unsigned frame_rows) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 768) lds_w32(kWinAxis + 4u * threadIdx.x, window_axis_entry((int)(threadIdx.x >> 8), threadIdx.x & 255u));
  if (threadIdx.x < 256) {
    const u2_t none = {0xffffffffu, 0u};  // tags are 10 bits: 0xffff matches nothing
    lds_w64v(kWinMeta + 8u * threadIdx.x, none);
    lds_w32v(kWinLock + 4u * threadIdx.x, 0u);
    if (threadIdx.x < kWinWaves) lds_w32v(kWinProgress + 4u * threadIdx.x, 0u);
  }
  __syncthreads();

  unsigned n_px = 0, n_miss = 0, n_fill = 0;
  {
    // fronts == 0: total_steps = share * gridDim.x + extra: the first `extra` blocks take one step more, block b a contiguous share
    // of the column-major step list. fronts != 0 (aligned fronts): strip = b % fronts, layer = b / fronts, a layer is `share` steps
    // tall, `extra` layers: the blocks of a layer walk down side by side, so that what the chip reads and writes at any time
    // is `extra` bands of full rows instead of 256 unrelated row segments (tools/walk_bench.hip: 3-5 % on the walk alone)
    unsigned first, last;
    if (fronts == 0u) {
      first = blockIdx.x * share + (blockIdx.x < extra ? blockIdx.x : extra);
      last = first + share + (blockIdx.x < extra ? 1u : 0u);
    } else {
      const unsigned strip = blockIdx.x % fronts, layer = blockIdx.x / fronts;
      first = strip * steps_per_strip + layer * share;
      last = first + share;
      if (last > (strip + 1u) * steps_per_strip) last = (strip + 1u) * steps_per_strip;
      if (layer >= extra || first > last) first = last = 0;
    }
    const uint32_t two = 2u;
    const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
    // pixel groups in flight: WIN_DEPTH steps, each in its own registers (the loop below is unrolled WIN_DEPTH times so that
    // no loaded value is ever copied before its step: a copy would wait for the load)
    struct Slot { u4_t p, q; uint32_t o0, o1; unsigned f0, f1; };  // o: byte offset of the results in dst (MULTI: in the row's frame, f0 / f1), 0x80000000 (out of range: dropped) for lanes outside the picture
    Slot ring[WIN_DEPTH];
    // this lane's two pixel groups of step st (clamped into the picture: loads are unconditional)
    auto fetch = [&](unsigned st_, Slot &S) {
      const unsigned st = st_ < last ? st_ : last - 1u;
      const unsigned strip = st / steps_per_strip, k = st - strip * steps_per_strip;
      const unsigned col = strip * 64u + lane, r0 = k * kWinRowsPerStep + 2u * wave, r1 = r0 + 1u;
      const unsigned cc = col < w4 ? col : w4 - 1u, c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
      if (MULTI) {
        // (wave is made uniform for the compiler: the frame numbers, and with them the base pointers, live in SGPRs)
        const unsigned uw = (unsigned)__builtin_amdgcn_readfirstlane((int)wave);
        const unsigned u0 = k * kWinRowsPerStep + 2u * uw, u1 = u0 + 1u, v0 = u0 < rows ? u0 : rows - 1u, v1 = u1 < rows ? u1 : rows - 1u;
        const unsigned f0 = v0 / frame_rows, f1 = v1 / frame_rows, y0 = v0 - f0 * frame_rows, y1 = v1 - f1 * frame_rows;
        S.f0 = f0; S.f1 = f1;
        S.o0 = col < w4 && r0 < rows ? (y0 * dw4 + col) << 4 : 0x80000000u;
        S.o1 = col < w4 && r1 < rows ? (y1 * dw4 + col) << 4 : 0x80000000u;
        S.p = __builtin_nontemporal_load((const u4_t *)srcs.p[f0] + ((size_t)y0 * sw4 + cc));
        S.q = __builtin_nontemporal_load((const u4_t *)srcs.p[f1] + ((size_t)y1 * sw4 + cc));
      } else {
        S.f0 = 0; S.f1 = 0;
        S.o0 = col < w4 && r0 < rows ? (r0 * dw4 + col) << 4 : 0x80000000u;
        S.o1 = col < w4 && r1 < rows ? (r1 * dw4 + col) << 4 : 0x80000000u;
        S.p = __builtin_nontemporal_load(src + ((size_t)c0 * sw4 + cc));
        S.q = __builtin_nontemporal_load(src + ((size_t)c1 * sw4 + cc));
      }
    };

#if WIN_PACE
    unsigned steps_done = 0;
#endif
    auto step = [&](unsigned st, Slot &S) {
#if WIN_PACE
      // the pace keeper of colorlut3d_shared_kernel (colorlut_brick.hip): a wave more than WIN_PACE steps ahead of the slowest sleeps
      if (lane == 0) lds_w32v(kWinProgress + 4u * wave, steps_done);
      for (int spin = 0; spin < 2048; spin++) {
        const uint32_t other = lds_r32v(kWinProgress + 4u * (lane & (kWinWaves - 1)));
        if (__builtin_amdgcn_ballot_w64(other + WIN_PACE < steps_done) == 0ull) break;
        __builtin_amdgcn_s_sleep(8);
      }
      steps_done++;
#endif
      const uint32_t px[8] = {S.p.x, S.p.y, S.p.z, S.p.w, S.q.x, S.q.y, S.q.z, S.q.w};
      const uint32_t so0 = S.o0, so1 = S.o1;
      const unsigned sf0 = S.f0, sf1 = S.f1;

      uint32_t val[8];
#if WIN_EXP == 1  // experiment: the walk alone (copy)
#pragma unroll
      for (int j = 0; j < 8; j++) val[j] = px[j];
#else
      uint32_t s[8];
      bool hit[8];
#pragma unroll
      for (int j = 0; j < 8; j++)
        s[j] = lds_r32(byte_times4<0>(px[j], two) + kWinAxis) + lds_r32(byte_times4<1>(px[j], two) + (kWinAxis + 1024u)) +
               lds_r32(byte_times4<2>(px[j], two) + (kWinAxis + 2048u));
      // ONE LDS round trip for the rest: {tags, generation}, the entry in both ways, the generation again - issued back to
      // back, executed by the LDS in this order
      u2_t m[8], d[8];
      uint32_t gen2[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t ma = kWinMeta + ((s[j] >> 3) & 0x7f8u);
        m[j] = lds_r64v(ma);
        d[j] = lds_r64v(kWinData + ((s[j] & 0x3fffu) << 3));
        gen2[j] = lds_r32v(ma + 4u);
      }
      bool miss_any = false;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t t = s[j] >> 14;
        const bool h0 = (m[j].x & 0xffffu) == t, h1 = (m[j].x >> 16) == t;
        val[j] = h1 ? d[j].y : d[j].x;
        hit[j] = (h0 | h1) & (gen2[j] == m[j].y);
        miss_any = miss_any | !hit[j];
      }
      n_px += 8;
#if WIN_EXP == 2  // experiment: the hit path alone (results wrong)
      miss_any = false;
#endif
      if (__builtin_amdgcn_ballot_w64(miss_any) != 0ull) {
        // Past the cache. First the installs (LDS only so far): leaders = the first lane of up to kWinFills distinct missed
        // bricks (a lane speaks for the brick of its first missed pixel).
        uint32_t mb = 0xffffffffu;
#pragma unroll
        for (int j = 7; j >= 0; j--)
          if (!hit[j]) mb = s[j] >> 6;
        // (each wave starts its search at another lane: the waves of a block miss the same new bricks at the same time,
        // and sixteen waves asking for the same eight leave the rest to the gather path)
        unsigned long long want = __builtin_amdgcn_ballot_w64(mb != 0xffffffffu), leaders = 0ull;
        const unsigned rot = (wave * 4u + 1u) & 63u;
#pragma unroll
        for (int f = 0; f < kWinFills; f++)
          if (want != 0ull) {
            const unsigned long long turned = (want >> rot) | (want << (64u - rot));
            const int l = (int)((__builtin_ctzll(turned) + rot) & 63u);
            leaders |= 1ull << l;
            want &= ~__builtin_amdgcn_ballot_w64(mb == (uint32_t)__builtin_amdgcn_readlane((int)mb, l));
          }
        // every leader claims its brick's set on its own (all of them at once): lock, look again, take the older way away
        bool claimed = false;
        uint32_t way = 0, tw_new = 0;
        if ((leaders >> lane) & 1ull) {
          const uint32_t meta = meta_addr(mb), tag = mb >> 8;
          uint32_t expect = 0;
          lds_u32 *lock = (lds_u32 *)(lds_byte *)(uintptr_t)(kWinLock + ((mb & 255u) << 2));
          if (__hip_atomic_compare_exchange_strong(lock, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            const u2_t mm = lds_r64v(meta);
            if ((mm.x & 0xffffu) == tag || (mm.x >> 16) == tag) {
              lds_w32v(kWinLock + ((mb & 255u) << 2), 0u);  // another wave has installed it since this wave's lookup
            } else {
              way = mm.y & 1u;
              tw_new = way ? ((mm.x & 0x0000ffffu) | (tag << 16)) : ((mm.x & 0xffff0000u) | tag);
              // victim tag gone and generation bumped in one LDS operation: the first thing any reader can see of this install
              const u2_t inv = {way ? (mm.x | 0xffff0000u) : (mm.x | 0x0000ffffu), mm.y + 1u};
              lds_w64v(meta, inv);
              claimed = true;
            }
          }
        }
        // the wave copies the claimed bricks: 256 B each, one entry per lane, all of them in flight together
        unsigned long long cm = __builtin_amdgcn_ballot_w64(claimed);
        uint32_t fa[kWinFills], fv[kWinFills];
        bool fo[kWinFills];
#pragma unroll
        for (int f = 0; f < kWinFills; f++) {
          fo[f] = cm != 0ull;
          if (fo[f]) {
            const int l = __builtin_ctzll(cm);
            cm &= cm - 1ull;
            const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)mb, l), w = (uint32_t)__builtin_amdgcn_readlane((int)way, l);
            fa[f] = data_addr(B, lane, w);
            fv[f] = table[((size_t)B << 6) + lane];
            n_fill++;
          }
        }
#pragma unroll
        for (int f = 0; f < kWinFills; f++)
          if (fo[f]) lds_w32v(fa[f], fv[f]);
        // ... the leaders publish (after the data in program order = in LDS order) and let go of the locks
        if (claimed) {
          lds_w32v(meta_addr(mb), tw_new);
          lds_w32v(kWinLock + ((mb & 255u) << 2), 0u);
        }
#if WIN_WAIT
        // (as in colorlut3d_shared_kernel: nobody waits with a lock in hand - this wave's were released above - so the wait ends)
        if (mb != 0xffffffffu) {
          const uint32_t meta = meta_addr(mb), tag = mb >> 8;
          for (int spin = 0; spin < 2048; spin++) {
            const uint32_t tg = lds_r32v(meta);
            if ((tg & 0xffffu) == tag || (tg >> 16) == tag || lds_r32v(kWinLock + ((mb & 255u) << 2)) == 0u) break;
            __builtin_amdgcn_s_sleep(2);
          }
        }
#endif
        // Second look for the pixels that missed: most of them belong to the bricks just installed (two coalesced lines
        // per brick from L2 instead of one 128 B line per pixel through the gather path).
        bool still = false;
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (!hit[j]) {
            const uint32_t ma = kWinMeta + ((s[j] >> 3) & 0x7f8u), t = s[j] >> 14;
            const u2_t mm = lds_r64v(ma), dd = lds_r64v(kWinData + ((s[j] & 0x3fffu) << 3));
            const uint32_t g2 = lds_r32v(ma + 4u);
            const bool h0 = (mm.x & 0xffffu) == t, h1 = (mm.x >> 16) == t;
            val[j] = h1 ? dd.y : dd.x;
            hit[j] = (h0 | h1) & (g2 == mm.y);
            still = still | !hit[j];
          }
        // what is left (sets locked by other waves, more distinct bricks than kWinFills) reads the table itself, per lane
        if (__builtin_amdgcn_ballot_w64(still) != 0ull) {
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (!hit[j]) {
              val[j] = table[s[j]];
              n_miss++;
            }
          // (a use inside the branch: the wait for these loads then sits here and not in front of the stores of every step,
          // where it would also wait for the pixels in flight)
#pragma unroll
          for (int j = 0; j < 8; j++) asm volatile("" : "+v"(val[j]));
        }
      }
#endif
      u4_t a, b;
      a.x = (val[0] & 0x00ffffffu) | (px[0] & 0xff000000u);
      a.y = (val[1] & 0x00ffffffu) | (px[1] & 0xff000000u);
      a.z = (val[2] & 0x00ffffffu) | (px[2] & 0xff000000u);
      a.w = (val[3] & 0x00ffffffu) | (px[3] & 0xff000000u);
      b.x = (val[4] & 0x00ffffffu) | (px[4] & 0xff000000u);
      b.y = (val[5] & 0x00ffffffu) | (px[5] & 0xff000000u);
      b.z = (val[6] & 0x00ffffffu) | (px[6] & 0xff000000u);
      b.w = (val[7] & 0x00ffffffu) | (px[7] & 0xff000000u);
      // unconditional buffer stores, range-checked by the hardware (a store inside a branch would make the number of
      // operations in flight unknown to the compiler, which then waits for ALL of them before the next step's pixels)
      if (MULTI) {
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(dsts.p[sf0], 0, (int)dst_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(dsts.p[sf1], 0, (int)dst_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(a, r0, (int)so0, 0, 2 /* nt */);
        __builtin_amdgcn_raw_buffer_store_b128(b, r1, (int)so1, 0, 2);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(a, dst_rsrc, (int)so0, 0, 2 /* nt */);
        __builtin_amdgcn_raw_buffer_store_b128(b, dst_rsrc, (int)so1, 0, 2);
      }
      // the slot's next pixels travel while the other slots' steps are looked up (issued last: a miss path's loads then do
      // not queue behind them)
      fetch(st + WIN_DEPTH, S);
    };

    if (first < last) {
#pragma unroll
      for (int d = 0; d < WIN_DEPTH; d++) fetch(first + d, ring[d]);
    }
    unsigned st = first;
    for (; st + WIN_DEPTH <= last; st += WIN_DEPTH) {
#pragma unroll
      for (int d = 0; d < WIN_DEPTH; d++) step(st + d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < WIN_DEPTH - 1; d++)
      if (st + d < last) step(st + d, ring[d]);
  }
#if WIN_PACE
  if (lane == 0) lds_w32v(kWinProgress + 4u * wave, 0x7fffffffu);
#endif
  // diagnostics: {pixels looked up, pixels served past the cache, bricks installed}, spread over slots so that the waves
  // of the chip do not queue up on one address
  unsigned long long miss_w = n_miss;
  for (int o = 32; o > 0; o >>= 1) miss_w += __shfl_xor(miss_w, o);
  if (lane == 0 && counters) {
    unsigned long long *c = counters + 3 * ((blockIdx.x * kWinWaves + wave) % kWinCounterSlots);
    atomicAdd(c, (unsigned long long)n_px * 64ull);
    atomicAdd(c + 1, miss_w);
    atomicAdd(c + 2, (unsigned long long)n_fill);
  }
}

bool window_applicable(const mi355_ctx *ctx, unsigned w4, unsigned dw4, size_t rows) {
  if (w4 < 1 || rows == 0 || rows >= (1u << 30)) return false;
  if (rows * dw4 * 16 > (1ull << 31)) return false;  // results are addressed by 32-bit byte offsets into one buffer descriptor
  const size_t steps = (size_t)((w4 + 63) / 64) * ((rows + kWinRowsPerStep - 1) / kWinRowsPerStep);
  return steps < (1u << 31) && steps >= (size_t)ctx->n_cu * (size_t)ctx->window_min_steps;
}

int launch_window_table(mi355_ctx *ctx, const uint32_t *table, const uint8_t *d_src, uint8_t *d_dst, unsigned w4, unsigned sw4, unsigned dw4, size_t rows) {
  int rc;
  if (!ctx->d_window_counters) {
    if ((rc = check_hip(ctx, hipMalloc((void **)&ctx->d_window_counters, 3 * kWinCounterSlots * sizeof(unsigned long long)), "hipMalloc(window counters)"))) return rc;
    if ((rc = check_hip(ctx, hipMemsetAsync(ctx->d_window_counters, 0, 3 * kWinCounterSlots * sizeof(unsigned long long), ctx->stream), "hipMemsetAsync(window counters)"))) return rc;
  }
  const unsigned n_strips = (w4 + 63) / 64, steps_per_strip = (unsigned)((rows + kWinRowsPerStep - 1) / kWinRowsPerStep);
  const unsigned total = n_strips * steps_per_strip;
  unsigned grid = (unsigned)ctx->n_cu;
  if (grid > total) grid = total;
  if (ctx->window_order == 1 && n_strips <= grid) {
    const unsigned layers = grid / n_strips, per_layer = (steps_per_strip + layers - 1) / layers;
    hipLaunchKernelGGL(colorlut_window_kernel<false>, dim3(n_strips * layers), dim3(64 * kWinWaves), kWinLdsBytes, ctx->stream, (const u4_t *)d_src, (u4_t *)d_dst, w4, sw4, dw4,
                       (unsigned)rows, (unsigned)(rows * dw4 * 16), steps_per_strip, per_layer, layers, n_strips, table, ctx->window_stats_on ? ctx->d_window_counters : nullptr,
                       MultiFramePtrs{}, MultiFramePtrs{}, 1u);
  } else
    hipLaunchKernelGGL(colorlut_window_kernel<false>, dim3(grid), dim3(64 * kWinWaves), kWinLdsBytes, ctx->stream, (const u4_t *)d_src, (u4_t *)d_dst, w4, sw4, dw4,
                       (unsigned)rows, (unsigned)(rows * dw4 * 16), steps_per_strip, total / grid, total % grid, 0u, table, ctx->window_stats_on ? ctx->d_window_counters : nullptr,
                       MultiFramePtrs{}, MultiFramePtrs{}, 1u);
  return check_hip(ctx, hipGetLastError(), "colorlut window kernel launch");
}

// n separate packed frames (w4 groups per row, frame_rows rows each, rows 16 B aligned and contiguous) in ONE launch on `stream`
bool window_multi_applicable(const mi355_ctx *ctx, unsigned w4, size_t frame_rows, int n_frames) {
  if (frame_rows == 0 || (size_t)frame_rows * w4 * 16 > (1ull << 31)) return false;
  return window_applicable(ctx, w4, w4, frame_rows * (size_t)n_frames) || ((size_t)frame_rows * (size_t)n_frames < (1u << 30) && w4 >= 1 &&
         (size_t)((w4 + 63) / 64) * ((frame_rows * (size_t)n_frames + kWinRowsPerStep - 1) / kWinRowsPerStep) >= (size_t)ctx->n_cu * (size_t)ctx->window_min_steps);
}

int launch_window_table_multi(mi355_ctx *ctx, hipStream_t stream, const uint32_t *table, uint8_t *const *srcs, uint8_t *const *dsts, int n_frames, unsigned w4,
                              size_t frame_rows) {
  int rc;
  if (!ctx->d_window_counters) {
    if ((rc = check_hip(ctx, hipMalloc((void **)&ctx->d_window_counters, 3 * kWinCounterSlots * sizeof(unsigned long long)), "hipMalloc(window counters)"))) return rc;
    if ((rc = check_hip(ctx, hipMemsetAsync(ctx->d_window_counters, 0, 3 * kWinCounterSlots * sizeof(unsigned long long), stream), "hipMemsetAsync(window counters)"))) return rc;
  }
  MultiFramePtrs s{}, d{};
  for (int f = 0; f < kMultiFrames; f++) { s.p[f] = srcs[f < n_frames ? f : 0]; d.p[f] = dsts[f < n_frames ? f : 0]; }
  const size_t rows = frame_rows * (size_t)n_frames;
  const unsigned n_strips = (w4 + 63) / 64, steps_per_strip = (unsigned)((rows + kWinRowsPerStep - 1) / kWinRowsPerStep);
  const unsigned total = n_strips * steps_per_strip;
  unsigned grid = (unsigned)ctx->n_cu;
  if (grid > total) grid = total;
  const unsigned frame_bytes = (unsigned)(frame_rows * w4 * 16);
  if (ctx->window_order == 1 && n_strips <= grid) {
    const unsigned layers = grid / n_strips, per_layer = (steps_per_strip + layers - 1) / layers;
    hipLaunchKernelGGL(colorlut_window_kernel<true>, dim3(n_strips * layers), dim3(64 * kWinWaves), kWinLdsBytes, stream, (const u4_t *)nullptr, (u4_t *)nullptr, w4, w4, w4,
                       (unsigned)rows, frame_bytes, steps_per_strip, per_layer, layers, n_strips, table, ctx->window_stats_on ? ctx->d_window_counters : nullptr, s, d,
                       (unsigned)frame_rows);
  } else
    hipLaunchKernelGGL(colorlut_window_kernel<true>, dim3(grid), dim3(64 * kWinWaves), kWinLdsBytes, stream, (const u4_t *)nullptr, (u4_t *)nullptr, w4, w4, w4,
                       (unsigned)rows, frame_bytes, steps_per_strip, total / grid, total % grid, 0u, table, ctx->window_stats_on ? ctx->d_window_counters : nullptr, s, d,
                       (unsigned)frame_rows);
  return check_hip(ctx, hipGetLastError(), "colorlut window kernel launch (multi-frame)");
}

int window_read_counters(mi355_ctx *ctx, unsigned long long out[3], bool reset) {
  out[0] = out[1] = out[2] = 0;
  if (!ctx->d_window_counters) return MI355_OK;
  unsigned long long h[3 * kWinCounterSlots];
  int rc;
  if ((rc = check_hip(ctx, hipMemcpyAsync(h, ctx->d_window_counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(window counters)"))) return rc;
  if (reset && (rc = check_hip(ctx, hipMemsetAsync(ctx->d_window_counters, 0, sizeof(h), ctx->stream), "hipMemsetAsync(window counters)"))) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize"))) return rc;
  for (int i = 0; i < kWinCounterSlots; i++)
    for (int k = 0; k < 3; k++) out[k] += h[3 * i + k];
  return MI355_OK;
}

void window_release(mi355_ctx *ctx) {
  if (ctx->d_window_counters) (void)hipFree(ctx->d_window_counters);
  ctx->d_window_counters = nullptr;
}

}  // namespace mi355
