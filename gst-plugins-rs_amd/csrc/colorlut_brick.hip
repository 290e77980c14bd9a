// colorlut_brick.hip — the interpolating ("arithmetic") colorlut kernel for packed RGBA8 frames and 3D LUTs:
// trilinear interpolation out of a per-wave LDS cache of LUT *bricks*.
//
// Reference loop replaced: transform_rgba_3d -> apply_3d -> sample_3d / lerp4 / float_to_u8
// (video/colorlut/src/colorlut/imp.rs:267-294, 431-449, 493-535, 537-539), Lut3D::at (video/colorlut/src/parser.rs:43-53).
//
// Why bricks. The reference fetches the 8 corners of the LUT cell (x0,y0,z0) a pixel falls into and makes 7 lerp4
// `a + (b - a) * t` (imp.rs:508-535). A *brick* is everything sample_3d reads for one cell, laid out for the lerp order:
// for each of the four (y,z) corner rows q = (y0|y1, z0|z1): c = cell(x0,yq,zq).rgb and d = cell(x1,yq,zq).rgb - c, the
// f32 difference the reference's x-lerp computes first (x1, y1, z1 clamped to size-1 exactly as imp.rs:499-501 does).
// 24 floats = 96 B, one 128 B line of the global brick table (size^3 lines, built on the host at mi355_colorlut_load with
// IEEE f32 subtraction, L2 / Infinity-Cache resident: 4.6 MB for 33^3). With the brick in registers a pixel costs
//   4 x (c + d*tx)  +  2 x (a + (b-a)*ty)  +  1 x (a + (b-a)*tz)   per channel = 17 unfused f32 ops (reference: 21),
// every one of them the reference's operation on the reference's operands: bit-identical for ANY table contents
// (non-finite entries included: the differences and products are the same IEEE operations), any size <= 64.
//
// Where the bricks live. Natural-like pictures are locally coherent in colour: the pixels of a 128 x 4 patch fall into a
// handful of neighbouring LUT cells. Each WAVE owns a direct-mapped cache of 64 bricks in LDS (slot = low 2 bits of x0,
// y0, z0: any 4 x 4 x 4 window of cells maps to distinct slots; 112 B per slot so that slot s starts on 16-byte column
// 7s mod 16 and the six 16-byte reads of neighbouring slots spread over the LDS banks; tag word = cell number) and walks
// down a 128-pixel-wide strip of the picture so the cache stays warm from tile to tile. Per pixel: three 8-byte axis-table
// reads ({t, packed slot/cell contribution} per input byte, computed on the host with the reference's coordinate
// arithmetic), one tag read + compare, six ds_read_b128, 51 + ~16 VALU ops. A wave whose 256-pixel step has a miss takes
// the careful path for that step: hit lanes read LDS, miss lanes read their brick's line from the global table (and one
// elected lane per slot refills the cache) — per lane, no loop, so hostile content (noise: every pixel its own cell) is
// slow but exact; the miss counters tell the host-side kernel choice (colorlut_kernels.hip) to use the three-pass
// whole-plane kernel for such streams.
// No block-level synchronisation anywhere after the prologue: waves run free, so HBM latency, L2 refills, LDS reads and
// VALU work of the 16 waves of a CU overlap by themselves.
#include "internal.hpp"
#include "hsv_device.hpp"
#include "exact_math.hpp"
#include "colorlut_brick.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace mi355 {

constexpr int kBrickSlotBytes = 112;   // 96 B brick + tag word at +96 + owner word at +100 (+8 spare)
constexpr int kBrickSlots = 64;        // per wave: 4 x 4 x 4 window
constexpr int kBrickTagShift = 15;     // packed = LDS byte address of the slot (< 32768) | window number << 15
constexpr int kBrickAxisBytes = 3 * 256 * 8;
constexpr int kBrickWaves = 4;         // waves per block
constexpr int kBrickWaveBytes = kBrickSlots * kBrickSlotBytes;            // 7,168 B of slots per wave
constexpr int kBrickAxisBase = kBrickWaves * kBrickWaveBytes;             // LDS: four slot regions, then the axis tables
constexpr int kBrickHsvSelBase = kBrickAxisBase + kBrickAxisBytes;        // 8 dwords: hsvfilter sextant selectors (fused form)
constexpr int kBrickLdsBytes = kBrickHsvSelBase + 32;                     // 34,848 B -> 4 blocks per CU
static_assert(kBrickAxisBase <= (1 << kBrickTagShift), "slot addresses must fit below the tag field");

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));

// input byte BYTE of px, times 8 (the axis-table entry's byte offset), in one SDWA shift
template <int BYTE>
__device__ __forceinline__ uint32_t byte_times8(uint32_t px, uint32_t three) {
  uint32_t o;
  if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "v"(three), "v"(px));
  else if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "v"(three), "v"(px));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "v"(three), "v"(px));
  return o;
}

// round-half-away(y) for 0 <= y <= 255 written straight into byte lane CH of `packed` (v_cvt_rpi_i32_f32 = floor(y + 0.5),
// exhaustively checked in tools/sem_probe.hip; the SDWA form drops the separate byte insert)
template <int CH>
__device__ __forceinline__ void brick_round_into(uint32_t &packed, float y) {
  if constexpr (CH == 0) asm("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(y));
  else if constexpr (CH == 1) asm("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(y));
  else asm("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(y));
}

template <int CH>
__device__ __forceinline__ void brick_channel(const float (&v)[24], float tx, float ty, float tz, uint32_t &out) {
  constexpr int ch = CH;
  {
    const float x00 = v[0 + ch] + v[3 + ch] * tx;    // (y0,z0)   imp.rs:518
    const float x10 = v[6 + ch] + v[9 + ch] * tx;    // (y1,z0)   imp.rs:519
    const float x01 = v[12 + ch] + v[15 + ch] * tx;  // (y0,z1)   imp.rs:520
    const float x11 = v[18 + ch] + v[21 + ch] * tx;  // (y1,z1)   imp.rs:521
    const float y0 = x00 + (x10 - x00) * ty;         // imp.rs:523
    const float y1 = x01 + (x11 - x01) * ty;         // imp.rs:524
    const float o = y0 + (y1 - y0) * tz;             // imp.rs:525
    // float_to_u8 (imp.rs:537-539): inherent clamp (NaN passes, then `as u8` gives 0) == max-then-min (NaN -> 0) on the byte
    brick_round_into<CH>(out, fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
  }
}

// lerps + float_to_u8 of one pixel from its brick v[24] (layout above); returns the three output bytes merged into `px`
__device__ __forceinline__ uint32_t brick_pixel(const float (&v)[24], float tx, float ty, float tz, uint32_t px) {
  uint32_t out = px;
  brick_channel<0>(v, tx, ty, tz, out);
  brick_channel<1>(v, tx, ty, tz, out);
  brick_channel<2>(v, tx, ty, tz, out);
  return out;
}

// HSV: kBrickNoHsv = plain colorlut; otherwise the fused hsvfilter -> colorlut chain (variant as in hsv_kernels.hip)
constexpr int kBrickNoHsv = -2;

// LDS by absolute byte address (see the kernel prologue)
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) u2_t lds_u2;
typedef __attribute__((address_space(3))) f4_t lds_f4;
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(const lds_u32 *)(lds_byte *)a; }
__device__ __forceinline__ void lds_w32(uint32_t a, uint32_t v) { *(lds_u32 *)(lds_byte *)a = v; }
__device__ __forceinline__ u2_t lds_r64(uint32_t a) { return *(const lds_u2 *)(lds_byte *)a; }
__device__ __forceinline__ f4_t lds_r128(uint32_t a) { return *(const lds_f4 *)(lds_byte *)a; }
__device__ __forceinline__ void lds_w128(uint32_t a, f4_t v) { *(lds_f4 *)(lds_byte *)a = v; }

template <int P, int HSV>  // P = 16-byte loads per lane and tile: a tile is 128 px x 2P rows
__global__ __launch_bounds__(256, 4) void colorlut3d_brick_kernel(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned rows,
                                                                   unsigned n_strips, unsigned tiles_per_run, unsigned n_runs,
                                                                   const f4_t *__restrict__ bricks, const u2_t *__restrict__ axis,
                                                                   const uint32_t *__restrict__ cellnum,
                                                                   unsigned long long *__restrict__ counters, HsvK hk) {
  // All LDS of this kernel is the dynamic allocation and there are no static __shared__ objects, so the allocation starts
  // at LDS address 0. LDS is addressed by absolute byte address (address-space-3 pointers made from integers): every table
  // base then folds into the ds_read offset field instead of costing an add of the link-time base symbol per access.
  const uint32_t *hsv_sel = (const uint32_t *)(lds_u32 *)(lds_byte *)(uint32_t)kBrickHsvSelBase;
  if constexpr (HSV != kBrickNoHsv) {
    if (threadIdx.x < 7)
      lds_w32(kBrickHsvSelBase + 4 * threadIdx.x, HSV >= 0 ? hsv_sel_entry_floor(threadIdx.x, 0, 1, 2, 3) : hsv_sel_entry(threadIdx.x, 0, 1, 2, 3));
  }
  for (int i = threadIdx.x; i < kBrickAxisBytes / 8; i += 256) *(lds_u2 *)(lds_byte *)(uint32_t)(kBrickAxisBase + 8 * i) = axis[i];
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t wave_base = __builtin_amdgcn_readfirstlane(wave * kBrickWaveBytes);
  lds_w32(wave_base + lane * kBrickSlotBytes + 96, 0xffffffffu);  // 64 lanes, 64 slots: every tag invalid
  __syncthreads();
  const uint32_t three = 3;

  const unsigned run = blockIdx.x * kBrickWaves + wave;
  if (run >= n_runs) return;
  // adjacent waves take adjacent strips of the same rows (a block covers 512 px x 2P rows: 2 KB row segments)
  const unsigned strip = run % n_strips, rr = run / n_strips;
  const unsigned sub = lane >> 5, g = lane & 31;
  const unsigned col = strip * 32 + g;
  const bool col_ok = col < w4;
  unsigned miss_steps = 0, miss_lanes = 0;

  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  const unsigned row_first = rr * tiles_per_run * (2 * P);
  u4_t cur[P], nxt[P];
  auto load_tile = [&](unsigned row0, u4_t(&t)[P]) {
#pragma unroll
    for (int j = 0; j < P; j++) {
      const unsigned r = row0 + 2 * j + sub;
      u4_t v = {0, 0, 0, 0};
      if (col_ok && r < rows) v = __builtin_nontemporal_load(src + (size_t)r * w4 + col);
      t[j] = v;
    }
  };
  load_tile(row_first, cur);
  for (unsigned t = 0; t < tiles_per_run; t++) {
    const unsigned row0 = row_first + t * (2 * P);
    if (row0 >= rows) break;  // wave-uniform
    if (t + 1 < tiles_per_run) load_tile(row0 + 2 * P, nxt);
#pragma unroll
    for (int j = 0; j < P; j++) {
      uint32_t px[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
      if constexpr (HSV >= 0) {
        hsvfilter_px2_fast<0, 1, 2, 3, HSV & 3, (HSV >> 2) != 0>(px[0], px[1], hk, hsv_sel);
        hsvfilter_px2_fast<0, 1, 2, 3, HSV & 3, (HSV >> 2) != 0>(px[2], px[3], hk, hsv_sel);
      } else if constexpr (HSV == -1) {
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = hsvfilter_px<false, 0, 1, 2, 3>(px[i], hk, hsv_sel);
      }
      // stage A: coordinates, slot, tag check
      float tx[4], ty[4], tz[4];
      uint32_t slot[4], tag[4];
      bool miss = false;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u2_t ex = lds_r64(byte_times8<0>(px[i], three) + (uint32_t)kBrickAxisBase);
        const u2_t ey = lds_r64(byte_times8<1>(px[i], three) + (uint32_t)(kBrickAxisBase + 2048));
        const u2_t ez = lds_r64(byte_times8<2>(px[i], three) + (uint32_t)(kBrickAxisBase + 4096));
        tx[i] = __uint_as_float(ex.x);
        ty[i] = __uint_as_float(ey.x);
        tz[i] = __uint_as_float(ez.x);
        const uint32_t packed = (ex.y + ey.y) + (ez.y + wave_base);  // v_add_u32 + v_add3_u32
        slot[i] = packed & ((1u << kBrickTagShift) - 1u);          // LDS byte address of the wave's slot for this cell
        tag[i] = packed >> kBrickTagShift;                         // which 4x4x4 window of cells the slot must hold
        miss |= lds_r32(slot[i] + 96u) != tag[i];
      }
      uint32_t out[4];
      if (__builtin_expect(!__any(miss), 1)) {
        // fast path: every lane's four bricks are resident
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float v[24];
#pragma unroll
          for (int k = 0; k < 6; k++) {
            const f4_t f = lds_r128(slot[i] + 16u * k);
            v[4 * k + 0] = f.x; v[4 * k + 1] = f.y; v[4 * k + 2] = f.z; v[4 * k + 3] = f.w;
          }
          out[i] = brick_pixel(v, tx[i], ty[i], tz[i], px[i]);
        }
      } else {
        // careful path, pixel by pixel: hit lanes read the cache, miss lanes read the global brick table; then one
        // elected miss lane per slot refills the cache (reads of this pixel come first in the wave's LDS order)
        miss_steps++;
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
          wave_sync();
          const bool m = lds_r32(slot[i] + 96u) != tag[i];
          miss_lanes += (unsigned)__builtin_popcountll(__ballot(m));
          f4_t f[6];
          if (m) {
            // cell number x0 + S*y0 + S*S*z0 from the per-axis contribution table (global, L1-resident; miss path only)
            const uint32_t cell = cellnum[px[i] & 0xffu] + cellnum[256 + ((px[i] >> 8) & 0xffu)] + cellnum[512 + ((px[i] >> 16) & 0xffu)];
            const f4_t *gb = bricks + (size_t)cell * 8;
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = gb[k];
          } else {
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = lds_r128(slot[i] + 16u * k);
          }
          wave_sync();
          if (m) lds_w32(slot[i] + 100u, lane);
          wave_sync();
          if (m && lds_r32(slot[i] + 100u) == lane) {
#pragma unroll
            for (int k = 0; k < 6; k++) lds_w128(slot[i] + 16u * k, f[k]);
            lds_w32(slot[i] + 96u, tag[i]);
          }
          float v[24];
#pragma unroll
          for (int k = 0; k < 6; k++) { v[4 * k + 0] = f[k].x; v[4 * k + 1] = f[k].y; v[4 * k + 2] = f[k].z; v[4 * k + 3] = f[k].w; }
          out[i] = brick_pixel(v, tx[i], ty[i], tz[i], px[i]);
        }
        wave_sync();
      }
      const unsigned r = row0 + 2 * j + sub;
      if (col_ok && r < rows) {
        const u4_t o = {out[0], out[1], out[2], out[3]};
        __builtin_nontemporal_store(o, dst + (size_t)r * w4 + col);
      }
    }
#pragma unroll
    for (int j = 0; j < P; j++) cur[j] = nxt[j];
  }
  if (counters && lane == 0 && miss_steps) {
    atomicAdd(counters + 0, (unsigned long long)miss_steps);
    atomicAdd(counters + 1, (unsigned long long)miss_lanes);
  }
}

// ---------------------------------------------------------------- host side

// One axis-table entry for input byte v: the reference's coordinate arithmetic, op for op (norm_comp imp.rs:471-474,
// apply_3d :438-440, sample_3d :496-506; this translation unit is compiled with -ffp-contract=off).
static void brick_axis_entry(int v, float scale, float offset, int S, float *t_out, int *i0_out) {
  volatile float n = (float)v / 255.0f;
  volatile float m = n * scale;
  volatile float a = m + offset;
  float cl = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);  // inherent clamp; the domain is finite (checked by the caller)
  volatile float x = cl * ((float)S - 1.0f);
  const float fl = std::floor(x);
  int i0 = (int)fl;
  if (i0 > S - 1) i0 = S - 1;
  if (i0 < 0) i0 = 0;
  volatile float t = x - (float)i0;
  *t_out = t;
  *i0_out = i0;
}

void brick_release(BrickLut &B) {
  if (B.h_counters) (void)hipHostFree(B.h_counters);
  if (B.ev) (void)hipEventDestroy(B.ev);
  if (B.d_bricks) (void)hipFree(B.d_bricks);
  if (B.d_axis) (void)hipFree(B.d_axis);
  if (B.d_cellnum) (void)hipFree(B.d_cellnum);
  if (B.d_counters) (void)hipFree(B.d_counters);
  B = BrickLut{};
}

int brick_upload(mi355_ctx *ctx, BrickLut &B, int S, const float *cells, const float scale[3], const float offset[3]) {
  brick_release(B);
  if (S < 2 || S > kBrickMaxSize) return MI355_OK;  // not applicable: B.ok stays false
  for (int c = 0; c < 3; c++)
    if (!std::isfinite(scale[c]) || !std::isfinite(offset[c])) return MI355_OK;
  const size_t n_cells = (size_t)S * S * S;
  std::vector<float> bricks(n_cells * 32, 0.0f);
  auto at = [&](int x, int y, int z) { return cells + 4 * ((size_t)x + (size_t)S * y + (size_t)S * S * z); };
  for (int z = 0; z < S; z++)
    for (int y = 0; y < S; y++)
      for (int x = 0; x < S; x++) {
        float *b = &bricks[((size_t)x + (size_t)S * y + (size_t)S * S * z) * 32];
        const int x1 = x + 1 < S ? x + 1 : S - 1, y1 = y + 1 < S ? y + 1 : S - 1, z1 = z + 1 < S ? z + 1 : S - 1;  // imp.rs:499-501
        const int ys[4] = {y, y1, y, y1}, zs[4] = {z, z, z1, z1};
        for (int q = 0; q < 4; q++) {
          const float *c0 = at(x, ys[q], zs[q]), *c1 = at(x1, ys[q], zs[q]);
          for (int ch = 0; ch < 3; ch++) {
            volatile float d = c1[ch] - c0[ch];  // the `(b - a)` of lerp4 (imp.rs:528-535), rounded to f32
            b[6 * q + ch] = c0[ch];
            b[6 * q + 3 + ch] = d;
          }
        }
      }
  std::vector<uint32_t> axis(3 * 256 * 2), cellnum(3 * 256);
  for (int a = 0; a < 3; a++)
    for (int v = 0; v < 256; v++) {
      float t;
      int i0;
      brick_axis_entry(v, scale[a], offset[a], S, &t, &i0);
      const uint32_t slot = (uint32_t)(i0 & 3) << (2 * a);                         // slot number: x bits 0-1, y 2-3, z 4-5
      const uint32_t window = (uint32_t)(i0 >> 2) << (5 * a);                      // tag: 5 bits per axis (size <= 128)
      std::memcpy(&axis[(size_t)(a * 256 + v) * 2 + 0], &t, 4);
      axis[(size_t)(a * 256 + v) * 2 + 1] = slot * (uint32_t)kBrickSlotBytes + (window << kBrickTagShift);
      cellnum[(size_t)a * 256 + v] = (uint32_t)i0 * (a == 0 ? 1u : (a == 1 ? (uint32_t)S : (uint32_t)S * S));
    }
  int rc = check_hip(ctx, hipMalloc((void **)&B.d_bricks, bricks.size() * sizeof(float)), "hipMalloc(lut bricks)");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(B.d_bricks, bricks.data(), bricks.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(lut bricks)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc((void **)&B.d_axis, axis.size() * sizeof(uint32_t)), "hipMalloc(brick axis tables)"))) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(B.d_axis, axis.data(), axis.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(brick axis tables)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc((void **)&B.d_cellnum, cellnum.size() * sizeof(uint32_t)), "hipMalloc(brick cell numbers)"))) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(B.d_cellnum, cellnum.data(), cellnum.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(brick cell numbers)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc((void **)&B.d_counters, 2 * sizeof(unsigned long long)), "hipMalloc(brick counters)"))) return rc;
  if ((rc = check_hip(ctx, hipMemset(B.d_counters, 0, 2 * sizeof(unsigned long long)), "hipMemset(brick counters)"))) return rc;
  if ((rc = check_hip(ctx, hipHostMalloc((void **)&B.h_counters, 2 * sizeof(unsigned long long), hipHostMallocDefault), "hipHostMalloc(brick counters)"))) return rc;
  B.h_counters[0] = B.h_counters[1] = 0;
  if ((rc = check_hip(ctx, hipEventCreateWithFlags(&B.ev, hipEventDisableTiming), "hipEventCreate(brick)"))) return rc;
  B.size = S;
  B.ok = true;
  return MI355_OK;
}

bool brick_applicable(const BrickLut &B, const uint8_t *d_src, size_t src_pitch, int src_stride, const uint8_t *d_dst, size_t dst_pitch,
                      int dst_stride, int n_frames, int width, int height) {
  if (!B.ok || width < 4 || width % 4 != 0) return false;
  const size_t row_bytes = (size_t)width * 4;
  const bool contiguous = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                          (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
  if (!contiguous || (uintptr_t)d_src % 16 != 0 || (uintptr_t)d_dst % 16 != 0) return false;
  return (size_t)n_frames * (size_t)height < (1u << 30);
}

template <int HSV>
static int brick_launch_t(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, uint8_t *d_dst, int n_frames, int width, int height, const HsvK &hk) {
  constexpr int P = 2;
  const unsigned w4 = (unsigned)width / 4, rows = (unsigned)((size_t)n_frames * height);
  const unsigned n_strips = (w4 + 31) / 32;
  const unsigned tile_rows = (rows + 2 * P - 1) / (2 * P);
  // run length: long enough to amortise the cold cache at the top of a run, short enough for >= 4 rounds of runs over the
  // 16 waves x n_cu the chip holds (tail balance)
  unsigned tpr = ctx->brick_tiles_per_run > 0 ? (unsigned)ctx->brick_tiles_per_run : 16;
  const size_t wave_slots = (size_t)ctx->n_cu * 16;
  while (tpr > 4 && (size_t)n_strips * ((tile_rows + tpr - 1) / tpr) < 4 * wave_slots) tpr /= 2;
  const unsigned runs_per_strip = (tile_rows + tpr - 1) / tpr;
  const size_t n_runs = (size_t)n_strips * runs_per_strip;
  if (n_runs >= (1u << 31)) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: frame batch too large");
  const unsigned grid = (unsigned)((n_runs + kBrickWaves - 1) / kBrickWaves);
  hipLaunchKernelGGL((colorlut3d_brick_kernel<P, HSV>), dim3(grid), dim3(256), kBrickLdsBytes, ctx->stream, (const u4_t *)d_src, (u4_t *)d_dst, w4, rows,
                     n_strips, tpr, (unsigned)n_runs, (const f4_t *)B.d_bricks, (const u2_t *)B.d_axis, (const uint32_t *)B.d_cellnum, B.d_counters, hk);
  return check_hip(ctx, hipGetLastError(), "colorlut3d_brick kernel launch");
}

int brick_launch(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, uint8_t *d_dst, int n_frames, int width, int height,
                 const mi355_hsv_settings *hs) {
  if (!hs) return brick_launch_t<kBrickNoHsv>(ctx, B, d_src, d_dst, n_frames, width, height, HsvK{});
  const HsvK hk{hs->hue_shift, hs->saturation_mul, hs->saturation_off, hs->value_mul, hs->value_off};
  switch (hsv_variant_for(*hs, false)) {
    case -1: return brick_launch_t<-1>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
    case 0: return brick_launch_t<0>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
    case 1: return brick_launch_t<1>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
    case 2: return brick_launch_t<2>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
    case 4: return brick_launch_t<4>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
    case 5: return brick_launch_t<5>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
    default: return brick_launch_t<6>(ctx, B, d_src, d_dst, n_frames, width, height, hk);
  }
}

// Content watch. The brick kernel is exact for any content but slow when most 256-pixel steps miss the cache (noise-like
// frames: every pixel in its own LUT cell); the three-pass whole-plane kernel does not care about content. Every
// kSnapEvery-th brick launch the miss counters are copied to pinned memory and reset, in stream order, behind an event
// that later launches poll (never wait for). A snapshot with more than kHostileFraction of its steps on the careful path
// hands the stream to the three-pass kernel for `retry_period` launches (64, doubling to 1024 while the verdict stays
// the same), after which the brick kernel gets kSnapEvery launches to prove itself again.
constexpr unsigned kSnapEvery = 4;
constexpr double kHostileFraction = 0.25;

static void brick_harvest(BrickLut &B) {
  if (!B.pending) return;
  if (hipEventQuery(B.ev) != hipSuccess) { (void)hipGetLastError(); return; }
  B.pending = false;
  const double steps = (double)B.px_snapshot / 256.0;
  B.last_miss_fraction = steps > 0.0 ? (double)B.h_counters[0] / steps : 0.0;
  if (B.last_miss_fraction > kHostileFraction) {
    B.hostile = true;
    B.retry_period = B.retry_period ? (B.retry_period < 1024 ? B.retry_period * 2 : 1024) : 64;
    B.retry_in = B.retry_period;
  } else {
    B.hostile = false;
    B.retry_period = 0;
  }
}

bool brick_choose(BrickLut &B) {
  brick_harvest(B);
  if (!B.hostile) return true;
  if (B.retry_in > 0) { B.retry_in--; return false; }
  // probation: brick kernel again until the next snapshot decides
  B.hostile = false;
  return true;
}

int brick_after_launch(mi355_ctx *ctx, BrickLut &B, unsigned long long pixels) {
  B.px_since += pixels;
  B.launches_since++;
  if (B.pending || B.launches_since < kSnapEvery) return MI355_OK;
  int rc = check_hip(ctx, hipMemcpyAsync(B.h_counters, B.d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream), "brick counters snapshot");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipMemsetAsync(B.d_counters, 0, 2 * sizeof(unsigned long long), ctx->stream), "brick counters reset"))) return rc;
  if ((rc = check_hip(ctx, hipEventRecord(B.ev, ctx->stream), "hipEventRecord(brick)"))) return rc;
  B.pending = true;
  B.px_snapshot = B.px_since;
  B.px_since = 0;
  B.launches_since = 0;
  return MI355_OK;
}

int brick_read_counters(mi355_ctx *ctx, const BrickLut &B, unsigned long long out[2], bool reset) {
  out[0] = out[1] = 0;
  if (!B.d_counters) return MI355_OK;
  int rc = check_hip(ctx, hipMemcpyAsync(out, B.d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream), "brick counters read");
  if (rc) return rc;
  if (reset && (rc = check_hip(ctx, hipMemsetAsync(B.d_counters, 0, 2 * sizeof(unsigned long long), ctx->stream), "brick counters reset"))) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "brick counters sync");
}

}  // namespace mi355
