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
// (non-finite entries included: the differences and products are the same IEEE operations), any size <= 65.
//
// Where the bricks live. Natural-like pictures are locally coherent in colour: the pixels of a 128 x 2 patch fall into a
// handful of neighbouring LUT cells, or into two such groups where the patch straddles an edge. Each WAVE owns a 2-way
// set-associative cache of bricks in LDS: set = low bits of (x0, y0, z0) (4 x 4 x 2 or 4 x 4 x 4 sets, so neighbouring
// cells never share a set), two ways per set (so two colour clusters do not evict each other), FIFO replacement;
// a set is 208 B = two 96 B bricks + {tag0, tag1, owner, fifo}: 13 sixteen-byte columns, so neighbouring sets start on
// different LDS bank groups. The wave walks down a 128-pixel-wide strip of the picture, so its cache stays warm from
// tile to tile. Per pixel: three 8-byte axis-table reads ({t, packed set address | tag contribution} per input byte,
// computed on the host with the reference's coordinate arithmetic), one 8-byte tag read, six ds_read_b128, 54 fast +
// 14 other VALU ops. A 256-pixel step with a miss first FILLS: per pixel slot, one elected lane per set reads its brick's
// line from the global table and writes it into the set (at most three rounds: a fill can evict a brick another lane
// of the same step still needs); then every lane takes the fast path. A step that still misses after that (more than
// two bricks per set among its pixels: noise) takes the slow path: hit lanes read the cache, miss lanes read the global
// table directly — per lane, no loop, always exact. The miss counters tell the host-side content watch to hand
// noise-like streams to the three-pass whole-plane kernel (colorlut_kernels.hip).
// No block-level synchronisation after the prologue: waves run free, so HBM latency, L2 refills, LDS reads and VALU work
// of the waves of a CU overlap by themselves. One block per CU holds every wave the LDS has a cache for; the waves of a
// block share their work through per-run tile deques in LDS (owners take from the front, finished waves steal from the
// back), so that a CU's waves finish together whatever their runs contain.
#include "internal.hpp"
#include "hsv_device.hpp"
#include "exact_math.hpp"
#include "colorlut_brick.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mi355 {

constexpr int kBrickSetBytes = 208;    // brick way 0 at +0, way 1 at +96, tags at +192 / +196, owner at +200, fifo at +204
// Block shape. ONE block per CU, as many waves as the LDS holds caches for: 16 (32 sets), 12 (48 sets), 10 (64 sets). The
// waves of a block share their work (see the deques in the kernel), so what matters is waves per CU, not blocks per CU.
#ifndef BRICK_W64
#define BRICK_W64 10
#endif
#ifndef BRICK_P64
#define BRICK_P64 4
#endif
#ifndef BRICK_G64
#define BRICK_G64 1
#endif
constexpr int brick_waves(int zn) { return zn == 4 ? BRICK_W64 : (zn == 3 ? 12 : 16); }
#ifndef BRICK_SB32
#define BRICK_SB32 2
#endif
#ifndef BRICK_SB64
#define BRICK_SB64 2
#endif
#ifndef BRICK_CHUNK32
#define BRICK_CHUNK32 2
#endif
#ifndef BRICK_CHUNK64
#define BRICK_CHUNK64 1
#endif
// 32-set geometry: set = (x0 + 3 y0 + 9 z0) mod 32 instead of the (4, 4, 2) box of residues. The box puts cells two apart
// in z into one set; the lattice of cells that collide under the linear hash has no vector shorter than 3 in any axis
// (searched over all (a, b, c) mod 32): ANY 3 x 3 x 3 block of cells maps to 27 different sets, whichever axis of the LUT
// the picture's gradients run along (hsvfilter's hue shift rotates them onto other axes). For 64 sets the (4, 4, 4) box
// is already optimal in that sense.
#ifndef BRICK_HASH32
#define BRICK_HASH32 1
#endif
constexpr bool brick_hashed(int zn) { return zn == 2 && BRICK_HASH32 != 0; }
constexpr int brick_chunk(int zn) { return zn == 4 ? BRICK_CHUNK64 : BRICK_CHUNK32; }  // tiles claimed (or stolen) at a time
constexpr int brick_group(int zn) { return zn == 4 ? BRICK_SB64 : (zn == 3 ? 4 : BRICK_SB32); }  // runs dealt out together (see the kernel)
// packed word of the axis tables: LDS byte address of the set / 16 in the low 16 bits (every set address is a multiple of
// 16 B; one SDWA shift turns WORD_0 into the byte address), tag in the high 16
constexpr int kBrickTagShift = 16;
constexpr int kBrickAxisBytes = 3 * 256 * 8;
constexpr int kBrickCounterSlots = 1024;
constexpr size_t kBrickCounterBytes = 2 * kBrickCounterSlots * sizeof(unsigned long long);
#ifdef BRICK_TIMING  // debug builds (tools/exp_brick_build.sh): per-run {start, end, miss | slow << 32, strip | rr << 32} after the counters
constexpr size_t kBrickTimingBytes = 16384 * 4 * sizeof(unsigned long long);
#else
constexpr size_t kBrickTimingBytes = 0;
#endif
// ZN = number of z0 residues in the set index: 2 -> 32 sets (6.9 KB per wave, 16 waves per CU), 3 -> 48 sets
// (10.4 KB, 12 waves per CU), 4 -> 64 sets (13.8 KB, 10 waves per CU)
// Set address = 208 x (x0 & 3) + 832 x (y0 & 3) + kBrickZStride x (z0 mod ZN). ds_read_b128 serves 16 lanes per LDS
// cycle over 16 sixteen-byte columns; with 13 columns per set the 16 (x, y) residues start on 16 different columns, way 1
// sits 6 columns after way 0, and the z stride adds 8 more: the two ways of a cell and of its x, y and z neighbours - the
// bricks the lanes of a wave actually read together on natural-like content - occupy 16 different columns.
constexpr int kBrickZStride = 16 * kBrickSetBytes + 128;
// (ZN == 2: 7,040 B so that the (2, 4, 4) and (4, 2, 4) residue layouts fit as well as (4, 4, 2))
constexpr int brick_wave_bytes(int zn) { return zn == 2 ? (brick_hashed(2) ? 32 * kBrickSetBytes : 7040) : zn * kBrickZStride; }
// LDS map: the small tables FIRST (their addresses then fit the 16-bit offset field of ds_read / ds_write: an axis-table
// read is one SDWA shift + the read; round 2 had the wave regions first and the axis tables at 104 KB, which cost one
// v_add_u32 per read - three per pixel), then the waves' brick caches.
constexpr int brick_axis_base(int) { return 0; }                                          // 3 x 256 x {t, packed}
constexpr int brick_cell_base(int zb) { return brick_axis_base(zb) + kBrickAxisBytes; }  // 3 x 256 bytes: lower cell index per axis and input byte (fill path)
constexpr int brick_sel_base(int zb) { return brick_cell_base(zb) + 768; }          // 8 dwords: hsvfilter sextant selectors
constexpr int kBrickQueueCap = 30;                                                       // fill queue entries per wave ({cell, destination})
constexpr int brick_queue_base(int zb) { return brick_sel_base(zb) + 32; }               // 256 B per wave
constexpr int kBrickScratch = 6;                                                         // per-step overflow bricks per wave
constexpr int brick_scratch_base(int zb) { return brick_queue_base(zb) + brick_waves(zb) * 256; }  // 6 x 96 B per wave
constexpr int brick_deque_base(int zb) { return brick_scratch_base(zb) + brick_waves(zb) * kBrickScratch * 96; }  // one word per wave: tiles {taken from the front, end}
constexpr int brick_runpos_base(int zb) { return brick_deque_base(zb) + 64; }  // per run of the block: {strip, first tile}
constexpr int brick_cache_base(int zb) { return (brick_runpos_base(zb) + 128 + 255) & ~255; }  // the wave regions (a multiple of 256 B: the column arithmetic above)
constexpr int brick_lds_bytes(int zb) { return brick_cache_base(zb) + brick_waves(zb) * brick_wave_bytes(zb); }
static_assert(brick_scratch_base(2) + brick_waves(2) * kBrickScratch * 96 < 65536 && brick_scratch_base(4) + brick_waves(4) * kBrickScratch * 96 < 65536, "table offsets fit ds offsets");
static_assert(brick_lds_bytes(2) <= 160 * 1024 && brick_lds_bytes(3) <= 160 * 1024 && brick_lds_bytes(4) <= 160 * 1024, "one block per CU");
static_assert(brick_waves(2) <= 16 && brick_waves(3) <= 16 && brick_waves(4) <= 16, "deque words / victim search");

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

// (packed & 0xffff) << 4: the set's LDS byte address out of the packed word, one SDWA shift
__device__ __forceinline__ uint32_t word0_times16(uint32_t packed, uint32_t four) {
  uint32_t o;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(o) : "v"(four), "v"(packed));
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

typedef float f2_t __attribute__((ext_vector_type(2)));

// Lerps + float_to_u8 of one pixel from its brick; returns the three output bytes merged into `px`.
// Brick register layout (six 16-byte rows, built by brick_upload): rows 0..3 = {c.r, c.g, d.r, d.g} of the corner rows
// q = (y0,z0), (y1,z0), (y0,z1), (y1,z1); row 4 = {c0.b, c2.b, d0.b, d2.b}; row 5 = {c1.b, c3.b, d1.b, d3.b}. Every operand
// pair of the 7 lerps then sits in an aligned register pair, so the arithmetic is packed-f32 (v_pk_mul_f32 / v_pk_add_f32,
// two IEEE results per instruction; never v_pk_fma: the reference's mul and add round separately): 24 packed + 3 scalar
// instructions instead of 51. On gfx950 a wave64 VALU instruction occupies its SIMD for 4 cycles whether packed or not.
#ifndef BRICK_DUMMY_VALU  // sensitivity probes (tools/exp_brick_build.sh): n extra VALU instructions / extra 16-byte LDS reads per pixel
#define BRICK_DUMMY_VALU 0
#endif
#ifndef BRICK_DUMMY_LDS
#define BRICK_DUMMY_LDS 0
#endif
// the seven lerps: O = clamped (r, g), ob = clamped b (float_to_u8 / float_to_u16's clamp rides on the last adds)
__device__ __forceinline__ void brick_lerps(const f4_t (&f)[6], float tx, float ty, float tz, uint32_t px, f2_t &O, float &ob) {
  if constexpr (BRICK_DUMMY_VALU > 0) {
    uint32_t sink = px;
#pragma unroll
    for (int d = 0; d < BRICK_DUMMY_VALU; d++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(sink) : "v"(px));
    asm volatile("" ::"v"(sink));
  }
  const f2_t TX = {tx, tx}, TY = {ty, ty}, TZ = {tz, tz};
  f2_t X[6];
#pragma unroll
  for (int q = 0; q < 6; q++) X[q] = f[q].xy + f[q].zw * TX;  // c + d * tx                 imp.rs:518-521
  const f2_t Y0 = X[0] + (X[1] - X[0]) * TY;                  // (r,g) at z0                imp.rs:523
  const f2_t Y1 = X[2] + (X[3] - X[2]) * TY;                  // (r,g) at z1                imp.rs:524
  const f2_t Yb = X[4] + (X[5] - X[4]) * TY;                  // b at (z0, z1)
  // imp.rs:525, and float_to_u8's clamp (imp.rs:537-539) folded into the last add: the VOP3P clamp bit clamps both halves to
  // [0, 1] and turns NaN into 0 (DX10_CLAMP is set for HSA kernels) - the inherent clamp lets NaN through and `as u8` then
  // makes it 0: the same byte
  const f2_t OZ = (Y1 - Y0) * TZ;
  asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(O) : "v"(Y0), "v"(OZ));
  // blue: the same clamp as an output modifier of the z-lerp's add (VOP3 clamp: [0, 1], NaN -> 0)
  const float obz = (Yb.y - Yb.x) * tz;
  asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(ob) : "v"(Yb.x), "v"(obz));
}

// RGBA8: lerps + float_to_u8 of one pixel from its brick; returns the three output bytes merged into `px`.
__device__ __forceinline__ uint32_t brick_pixel(const f4_t (&f)[6], float tx, float ty, float tz, uint32_t px) {
  f2_t O;
  float ob;
  brick_lerps(f, tx, ty, tz, px, O, ob);
  const f2_t O255 = O * (f2_t){255.0f, 255.0f};
  uint32_t out = px;
  brick_round_into<0>(out, O255.x);
  brick_round_into<1>(out, O255.y);
  brick_round_into<2>(out, ob * 255.0f);
  return out;
}

// RGBA64 (transform_rgba64_3d::<LE>, imp.rs:348-397): float_to_u16 = round-half-away(clamp(v) * 65535) (v_cvt_rpi: exact on
// [0, 65536], tools/sem_probe.hip) into the two colour words of `lo` and the low word of `hi`; the alpha word (high half of
// hi) is copied raw. SWAP: the samples are big-endian words in memory.
__device__ __forceinline__ uint32_t brick_swap_words(uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x02030001u); }  // bytes (1,0,3,2)
template <bool SWAP>
__device__ __forceinline__ void brick_pixel64(const f4_t (&f)[6], float tx, float ty, float tz, uint32_t &lo, uint32_t &hi) {
  f2_t O;
  float ob;
  brick_lerps(f, tx, ty, tz, lo, O, ob);
  const f2_t Os = O * (f2_t){65535.0f, 65535.0f};
  const float bs = ob * 65535.0f;
  uint32_t w0 = 0, w1 = SWAP ? brick_swap_words(hi) : hi;
  asm("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w0) : "v"(Os.x));
  asm("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w0) : "v"(Os.y));
  asm("v_cvt_rpi_i32_f32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(w1) : "v"(bs));
  lo = SWAP ? brick_swap_words(w0) : w0;
  hi = SWAP ? brick_swap_words(w1) : w1;
}

// coordinate arithmetic of the RGBA64 path (norm_comp_u16 imp.rs:476-480, apply_3d_u16 :451-469, sample_3d :496-506): 65,536
// input levels do not fit an LDS axis table, so x0 and t are computed - the expressions of colorlut3d_lds64_kernel
struct BrickK64 { float scale[3], offset[3], sm1; };

// HSV: kBrickNoHsv = plain colorlut; otherwise the fused hsvfilter -> colorlut chain (variant as in hsv_kernels.hip)
constexpr int kBrickNoHsv = -2;

// LDS by absolute byte address (see the kernel prologue)
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) u2_t lds_u2;
typedef __attribute__((address_space(3))) f4_t lds_f4;
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(const lds_u32 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w32(uint32_t a, uint32_t v) { *(lds_u32 *)(lds_byte *)(uintptr_t)a = v; }
__device__ __forceinline__ uint32_t lds_r8(uint32_t a) { return *(const lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w8(uint32_t a, uint32_t v) { *(lds_byte *)(uintptr_t)a = (unsigned char)v; }
__device__ __forceinline__ u2_t lds_r64(uint32_t a) { return *(const lds_u2 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w64(uint32_t a, u2_t v) { *(lds_u2 *)(lds_byte *)(uintptr_t)a = v; }
__device__ __forceinline__ f4_t lds_r128(uint32_t a) { return *(const lds_f4 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w128(uint32_t a, f4_t v) { *(lds_f4 *)(lds_byte *)(uintptr_t)a = v; }

// a wave-uniform global pointer the compiler will keep in SGPRs (global_load / global_store with an SGPR base and a 32-bit
// VGPR offset); address space 1 is kept through the integer round trip, or the accesses become flat_*
typedef __attribute__((address_space(1))) char gl_byte;
typedef __attribute__((address_space(1))) u4_t gl_u4;
__device__ __forceinline__ gl_byte *uniform_ptr(const void *p) {
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (gl_byte *)(((uint64_t)hi << 32) | lo);
}

// P = 16-byte loads per lane and tile (a tile is 32 lanes x 2P rows of 16-byte groups), G of them per step.
// FMT: 0 = RGBA8 (four pixels per group, coordinates from the LDS axis tables), 1 / 2 = RGBA64 little / big endian (two
// pixels per group, coordinates computed; everything else - cache, fills, work sharing, lerps - is the same code).
template <int P, int HSV, int ZN, int G, int FMT = 0>
__global__ __launch_bounds__(64 * brick_waves(ZN)) void colorlut3d_brick_kernel(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned sw4, unsigned dw4,
                                                                                 unsigned rows, unsigned n_strips, unsigned tiles_per_run, unsigned n_runs_flags,
                                                                                 const f4_t *__restrict__ bricks, const u2_t *__restrict__ axis,
                                                                                 const uint32_t *__restrict__ cellnum,
                                                                                 unsigned long long *__restrict__ counters, HsvK hk, unsigned fold_axis, unsigned lut_size,
                                                                                 BrickK64 k64) {
  static_assert(FMT == 0 || HSV == kBrickNoHsv, "the fused hsvfilter form is RGBA8 only");
  static_assert(FMT == 0 || ZN != 3, "RGBA64: 32 hashed sets or the 4 x 4 x 4 box");
  // All LDS of this kernel is the dynamic allocation and there are no static __shared__ objects, so the allocation starts
  // at LDS address 0. LDS is addressed by absolute byte address (address-space-3 pointers made from integers): every table
  // base then folds into the ds_read offset field instead of costing an add of the link-time base symbol per access.
  constexpr int NPX = FMT ? 2 : 4;  // pixels in a 16-byte group
  constexpr int NP = NPX * G, NT = 64 * brick_waves(ZN);
#ifndef BRICK_PIPE
#define BRICK_PIPE 1
#endif
  constexpr bool PIPE = BRICK_PIPE != 0;
  static_assert(P % G == 0, "a tile is a whole number of steps");
  constexpr uint32_t AX = brick_axis_base(ZN), CELL = brick_cell_base(ZN), SEL = brick_sel_base(ZN), SCR = brick_scratch_base(ZN), WB = brick_wave_bytes(ZN), SETS = 16u * ZN;
  const uint32_t *hsv_sel = (const uint32_t *)(lds_u32 *)(lds_byte *)(uintptr_t)SEL;
  if constexpr (HSV != kBrickNoHsv) {
    if (threadIdx.x < 7) lds_w32(SEL + 4 * threadIdx.x, HSV >= 0 ? hsv_sel_entry_floor(threadIdx.x, 0, 1, 2, 3) : hsv_sel_entry(threadIdx.x, 0, 1, 2, 3));
  }
  for (int i = threadIdx.x; i < kBrickAxisBytes / 8; i += NT) lds_w64(AX + 8 * i, axis[i]);
  for (int i = threadIdx.x; i < 768; i += NT) lds_w8(CELL + i, cellnum[i]);
  // cell number of a pixel (fill path only): x0 + S (y0 + S z0)
  auto cell_of = [&](uint32_t p) -> uint32_t {
    const uint32_t ix = lds_r8(CELL + (p & 0xffu)), iy = lds_r8(CELL + 256u + ((p >> 8) & 0xffu)), iz = lds_r8(CELL + 512u + ((p >> 16) & 0xffu));
    return ix + lut_size * (iy + lut_size * iz);
  };
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t queue = brick_queue_base(ZN) + wave * 256u;
  const uint32_t e_l = lane / 6u, k_l = lane % 6u;  // fill pass: lane -> (queue entry, 16-byte piece of its brick)
  uint32_t gen = 1;                                  // fill-round stamp (see the fifo word)
  const uint32_t wave_base = __builtin_amdgcn_readfirstlane((uint32_t)brick_cache_base(ZN) + wave * WB);
  if (lane < SETS) {
    const u2_t inval = {0xffffffffu, 0xffffffffu};
    // set residues per axis: (4, 4, ZN), or (2, 4, 4) / (4, 2, 4) when the host built the ZN == 2 tables with the 2 on x / y
    const uint32_t m0 = (ZN == 2 && fold_axis == 0) ? 2u : 4u, m1 = (ZN == 2 && fold_axis == 1) ? 2u : 4u;
    const uint32_t sa = brick_hashed(ZN) ? wave_base + lane * kBrickSetBytes
                                         : wave_base + (lane % m0) * kBrickSetBytes + ((lane / m0) % m1) * (m0 * kBrickSetBytes) +
                                               (lane / (m0 * m1)) * (m0 * m1 * kBrickSetBytes + 128u);
    lds_w64(sa + 192, inval);  // both tags invalid
    lds_w32(sa + 204, 0u);     // next victim: way 0
  }
  // Work. The picture is cut into 128-pixel-wide strips and every strip into RUNS of tiles_per_run tiles. Runs are dealt out
  // in groups of SB consecutive ones; a block takes the groups blockIdx, blockIdx + gridDim, ... (W / SB of them, from
  // different parts of the picture, so that the CUs get work of about the same difficulty) and wave w OWNS one run. A
  // wave walks down its run so that its cache stays warm from tile to tile - but runs differ in cost (one that straddles
  // colour regions refills far more often; the SIMDs of a CU do not hold the same number of waves; the SIMD arbitrates by
  // age), and a kernel that waits for its slowest wave loses 15-60 % to the tail. So every run is a DEQUE in LDS - one
  // word {tiles taken from the front, end} - the owner takes tiles from the front (ds_add), a wave whose own run is done
  // STEALS single tiles from the back of the run with the most tiles left (ds_cmpst), and keeps stealing from that run
  // while it lasts (its cache warms up on the neighbouring rows). The waves of a CU then finish within a tile of each other.
  constexpr uint32_t DQ = brick_deque_base(ZN), RP = brick_runpos_base(ZN), W = brick_waves(ZN), NONE = 0xffffffffu;
  const unsigned n_runs = n_runs_flags & 0x3fffffffu;  // bit 31: stealing enabled, bit 30: progress-based priorities
  const bool steal = (n_runs_flags >> 31) != 0, prio_quarters = ((n_runs_flags >> 30) & 1u) != 0;
  const unsigned runs_per_strip = (n_runs + n_strips - 1) / n_strips;
  const unsigned tile_rows = (rows + 2 * P - 1) / (2 * P);  // tiles per strip
  // run -> (strip, first tile). 32 sets: the runs of a block are the same rows of adjacent strips (2 KB row segments per
  // block measure faster on coherent content); otherwise consecutive runs of one strip
  auto run_pos = [&](unsigned run, unsigned &strip, unsigned &tile0) {
    const unsigned rr = ZN == 2 ? run / n_strips : run % runs_per_strip;
    strip = ZN == 2 ? run % n_strips : run / runs_per_strip;
    tile0 = rr * tiles_per_run;
  };
  constexpr unsigned SB = brick_group(ZN), C = brick_chunk(ZN);
  auto run_of = [&](unsigned w) { return ((w / SB) * gridDim.x + blockIdx.x) * SB + w % SB; };
  if (threadIdx.x < W) {
    const unsigned run = run_of(threadIdx.x);
    unsigned nt = 0, strip = 0, tile0 = 0;
    if (run < n_runs) {
      run_pos(run, strip, tile0);
      if (tile0 < tile_rows) nt = tile_rows - tile0 < tiles_per_run ? tile_rows - tile0 : tiles_per_run;
    }
    lds_w32(DQ + 4 * threadIdx.x, ((nt + C - 1) / C) << 16);  // the deque counts chunks of C tiles
    // (the divisions of run_pos are paid once per run here, not once per tile: there is no scalar integer divide)
    const u2_t rp = {strip, tile0};
    lds_w64(RP + 8 * threadIdx.x, rp);
  }
  __syncthreads();
  const uint32_t three = 3, four = 4;
#ifdef BRICK_TIMING
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  unsigned tiles_done = 0, tiles_stolen = 0;
#endif
  const unsigned sub = lane >> 5, g = lane & 31;
  unsigned miss_steps = 0, slow_steps = 0;

  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // claim a chunk of C tiles: (owning wave << 16) | index of its first tile within that wave's run, or NONE. Wave-uniform.
  bool own_done = false;
  uint32_t last_victim = NONE;
  unsigned prio_q = 0;
  if (prio_quarters) __builtin_amdgcn_s_setprio(3);
  auto claim = [&]() -> uint32_t {
    if (!own_done) {
      uint32_t old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add((lds_u32 *)(lds_byte *)(uintptr_t)(DQ + 4u * wave), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = __builtin_amdgcn_readfirstlane(old);
      if ((old & 0xffffu) < (old >> 16)) {
        // Fair share by hand. The SIMD arbitrates VALU issue by priority, then AGE: at equal priority the oldest wave of a
        // SIMD takes every slot it can use and is done with its run long before the youngest. A wave therefore lowers its
        // priority as it advances (3, 2, 1, 0 by quarter of its run; 0 when it lives on stolen tiles): whoever is behind
        // outranks whoever is ahead, so fewer tiles have to change hands (a stolen tile starts on a cold cache).
        if (prio_quarters) {
          const unsigned t4 = 4u * C * (old & 0xffffu);
          const unsigned q = (t4 >= tiles_per_run ? 1u : 0u) + (t4 >= 2u * tiles_per_run ? 1u : 0u) + (t4 >= 3u * tiles_per_run ? 1u : 0u);
          if (q != prio_q) {
            prio_q = q;
            if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
          }
        }
        return (wave << 16) | (C * (old & 0xffffu));
      }
      own_done = true;
      if (prio_quarters) __builtin_amdgcn_s_setprio(0);
    }
    if (!steal) return NONE;
    for (;;) {
      const uint32_t v = lane < W ? lds_r32(DQ + 4u * lane) : 0u;
      const uint32_t f = v & 0xffffu, e = v >> 16;
      uint32_t key = e > f ? (((e - f) << 6) | lane | (lane == last_victim ? 0x40000000u : 0u)) : 0u;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)key, o);
        key = other > key ? other : key;
      }
      key = __builtin_amdgcn_readfirstlane(key);
      if (key == 0u) return NONE;
      const uint32_t vic = key & 63u;
      const uint32_t vv = __builtin_amdgcn_readlane(v, vic);
      uint32_t ok = 0;
      if (lane == 0) {
        uint32_t expected = vv;
        ok = __hip_atomic_compare_exchange_strong((lds_u32 *)(lds_byte *)(uintptr_t)(DQ + 4u * vic), &expected, vv - 0x10000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
      }
      ok = __builtin_amdgcn_readfirstlane(ok);
      if (ok) {
        last_victim = vic;
#ifdef BRICK_TIMING
        tiles_stolen++;
#endif
        return (vic << 16) | (C * ((vv >> 16) - 1u));
      }
    }
  };
  unsigned tile0_of_cur = 0;  // first tile (within its strip) of the run the current chunk belongs to
  auto tile_pos = [&](uint32_t id, unsigned &strip, unsigned &row0) {
    const u2_t rp = lds_r64(RP + 8u * (id >> 16));  // wave-uniform address
    strip = __builtin_amdgcn_readfirstlane(rp.x);
    tile0_of_cur = __builtin_amdgcn_readfirstlane(rp.y);
    row0 = (tile0_of_cur + (id & 0xffffu)) * (2 * P);
  };

  u4_t cur[P], nxt[P];
  // Loads are unconditional (out-of-picture lanes re-read a clamped in-picture address; only the stores are predicated):
  // a load inside a branch makes the number of outstanding memory operations unknown to the compiler, which then waits
  // for ALL of them - next tile's prefetch included - with s_waitcnt vmcnt(0) before touching the current tile.
  // (addresses: wave-uniform 64-bit tile base in SGPRs + a 32-bit per-lane element offset - a tile may be anywhere now that
  // tiles change hands, and a full 64-bit address computation per load and lane is a dozen VALU instructions)
  auto load_tile = [&](unsigned row0, unsigned col_c, u4_t(&t)[P]) {
    gl_byte *base = uniform_ptr(src + (size_t)row0 * sw4);  // sw4 / dw4: row strides in 16-byte groups (== w4 for packed rows)
    const unsigned room = rows - 1u - row0;  // rows left below the tile's first (row0 < rows always)
#pragma unroll
    for (int j = 0; j < P; j++) {
      const unsigned ro = 2u * j + sub;
      const uint32_t byte_off = ((ro < room ? ro : room) * sw4 + col_c) * 16u;  // < 2^32: a tile spans 2P rows
      t[j] = __builtin_nontemporal_load((const gl_u4 *)(base + byte_off));
    }
  };
  uint32_t cur_id = claim();
  unsigned strip = 0, row0 = 0;
  if (cur_id != NONE) tile_pos(cur_id, strip, row0);
  unsigned col = strip * 32 + g;
  bool col_ok = col < w4;
  load_tile(row0, col_ok ? col : w4 - 1, cur);
  while (cur_id != NONE) {
    // the next tile is claimed - and its loads issued - before this one is worked on
    uint32_t nxt_id;
    unsigned nstrip = strip, nrow0 = row0;
    const uint32_t t1 = (cur_id & 0xffffu) + 1u;
    if (C > 1 && t1 % C != 0u && t1 < tiles_per_run && tile0_of_cur + t1 < tile_rows) {
      nxt_id = cur_id + 1u;  // the next tile of the chunk in hand
      nrow0 = row0 + 2 * P;
    } else {
      nxt_id = claim();
      if (nxt_id != NONE) tile_pos(nxt_id, nstrip, nrow0);
    }
    const unsigned ncol = nstrip * 32 + g;
    const bool ncol_ok = ncol < w4;
    load_tile(nrow0, ncol_ok ? ncol : w4 - 1, nxt);
    gl_byte *dst_tile = uniform_ptr(dst + (size_t)row0 * dw4);
#ifdef BRICK_TIMING
    tiles_done++;
#endif
#pragma unroll
    for (int j = 0; j < P; j += G) {
      // one STEP = G 16-byte groups per lane = NP pixels per lane (64 NP pixels per wave), checked and filled together
      uint32_t px[NP];                // RGBA8: the pixel; RGBA64: its r | g << 16 words
      uint32_t px_hi[FMT ? NP : 1];   // RGBA64: b | a << 16
#pragma unroll
      for (int g2 = 0; g2 < G; g2++) {
        if constexpr (FMT == 0) { px[4 * g2 + 0] = cur[j + g2].x; px[4 * g2 + 1] = cur[j + g2].y; px[4 * g2 + 2] = cur[j + g2].z; px[4 * g2 + 3] = cur[j + g2].w; }
        else { px[2 * g2] = cur[j + g2].x; px_hi[2 * g2] = cur[j + g2].y; px[2 * g2 + 1] = cur[j + g2].z; px_hi[2 * g2 + 1] = cur[j + g2].w; }
      }
      if constexpr (HSV >= 0) {
#pragma unroll
        for (int i = 0; i < NP; i += 2) hsvfilter_px2_fast<0, 1, 2, 3, HSV & 3, (HSV >> 2) != 0>(px[i], px[i + 1], hk, hsv_sel);
      } else if constexpr (HSV == -1) {
#pragma unroll
        for (int i = 0; i < NP; i++) px[i] = hsvfilter_px<false, 0, 1, 2, 3>(px[i], hk, hsv_sel);
      }
      // stage A: coordinates, set, tag. All twelve axis reads of the step are issued before the first one is used (one LDS
      // round trip for the step, not one per pixel: the compiler keeps the order it is given and waits with lgkmcnt(0))
      float tx[NP], ty[NP], tz[NP];
      uint32_t set[NP], tag[NP], baddr[NP];
      if constexpr (FMT != 0) {
        // RGBA64: x0 / t per axis by arithmetic; the tag is the cell number, the set its hash (32 sets) or residue box (64)
#pragma unroll
        for (int i = 0; i < NP; i++) {
          const uint32_t lo = FMT == 2 ? brick_swap_words(px[i]) : px[i], hi = FMT == 2 ? brick_swap_words(px_hi[i]) : px_hi[i];
          const uint32_t c16[3] = {lo & 0xffffu, lo >> 16, hi & 0xffffu};
          uint32_t idx[3];
          float t[3];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            float n = div65535_u16((float)c16[a]);
            n = fminf(fmaxf(n * k64.scale[a] + k64.offset[a], 0.0f), 1.0f);  // finite domain (brick_upload): == the inherent clamp
            const float x = n * k64.sm1;
            idx[a] = (uint32_t)x;                     // floor; x <= S - 1, so the min(.., size - 1) never bites
            t[a] = __builtin_amdgcn_fractf(x);        // x - x0
          }
          tx[i] = t[0]; ty[i] = t[1]; tz[i] = t[2];
          tag[i] = idx[0] + lut_size * (idx[1] + lut_size * idx[2]);
          if constexpr (brick_hashed(ZN)) set[i] = __umul24((idx[0] + 3u * idx[1] + 9u * idx[2]) & 31u, (uint32_t)kBrickSetBytes) + wave_base;
          else set[i] = wave_base + (idx[0] & 3u) * kBrickSetBytes + (idx[1] & 3u) * (4u * kBrickSetBytes) + (idx[2] & 3u) * (uint32_t)kBrickZStride;
        }
      } else {
        u2_t ex[NP], ey[NP], ez[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
          ex[i] = lds_r64(byte_times8<0>(px[i], three) + AX);
          ey[i] = lds_r64(byte_times8<1>(px[i], three) + (AX + 2048u));
          ez[i] = lds_r64(byte_times8<2>(px[i], three) + (AX + 4096u));
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
          tx[i] = __uint_as_float(ex[i].x);
          ty[i] = __uint_as_float(ey[i].x);
          tz[i] = __uint_as_float(ez[i].x);
          if constexpr (brick_hashed(ZN)) {
            // hashed 32-set geometry: the low byte sums the axes' set-number contributions (<= 93), the rest is the cell number
            const uint32_t packed = ex[i].y + ey[i].y + ez[i].y;                     // v_add3_u32
            set[i] = __umul24(packed & 31u, (uint32_t)kBrickSetBytes) + wave_base;  // v_and + v_mad_u32_u24
            tag[i] = packed;
          } else {
            const uint32_t packed = (ex[i].y + ey[i].y) + (ez[i].y + (wave_base >> 4));  // v_add_u32 + v_add3_u32
            set[i] = word0_times16(packed, four);                                        // LDS byte address of the wave's set for this cell
            tag[i] = packed;                                                             // the whole word identifies the brick (the set bits are redundant there)
          }
        }
      }
      // tag check: both ways' tags in one 8-byte read; baddr = the way that holds the brick
      auto check = [&]() -> bool {
        bool miss = false;
#pragma unroll
        for (int i = 0; i < NP; i++) {
          const u2_t tg = lds_r64(set[i] + 192u);
          const bool h1 = tg.y == tag[i];
          baddr[i] = h1 ? set[i] + 96u : set[i];
          miss |= !(h1 || tg.x == tag[i]);
        }
        return __any(miss);
      };
      bool slow = false;
      uint32_t lane_res = 0;  // bit i: this lane's pixel i is served from the global table (slow path only)
      if (__builtin_expect(check(), 0)) {
        miss_steps += G;
        // Fill rounds. Enqueue: per pixel slot, one elected lane per missing set claims the victim way (FIFO), installs
        // the tag at once (so that later pixel slots of this step see the brick as present and do not fetch it again) and
        // appends {cell number, destination} to the wave's queue. A set accepts two installs per round - one per way - so
        // no two queue entries of a round share a destination. Fetch: lanes 6e..6e+5 copy the six 16-byte pieces of entry
        // e from the global brick table into LDS, ten bricks per pass, ONE L2 round trip per pass. Then every tag is
        // checked again: a set asked for three different bricks by this step's pixels cannot hold them all, and after
        // three rounds the step takes the slow path below.
        int round = 0;
        do {
          uint32_t n = 0;  // wave-uniform
#pragma unroll
          for (int i = 0; i < NP; i++) {
            wave_sync();
            const u2_t tg = lds_r64(set[i] + 192u);
            const bool m = tg.x != tag[i] && tg.y != tag[i];
            if (__any(m)) {
              if (m) lds_w32(set[i] + 200u, lane);
              wave_sync();
              const bool own = m && lds_r32(set[i] + 200u) == lane;
              const unsigned long long ob = __ballot(own);
              const uint32_t pos = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(ob >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ob, 0u));
              if (own && pos < (uint32_t)kBrickQueueCap) {
                // fifo word: bit 0 = next victim way, bits 1-2 = installs in round `bits 3..`
                const uint32_t fw = lds_r32(set[i] + 204u);
                const uint32_t cnt = (fw >> 3) == gen ? ((fw >> 1) & 3u) : 0u;
                if (cnt < 2u) {
                  const uint32_t way = fw & 1u;
                  const uint32_t cell = FMT ? tag[i] : cell_of(px[i]);
                  const u2_t qe = {cell, set[i] + 96u * way};
                  lds_w64(queue + 8u * pos, qe);
                  lds_w32(set[i] + 192u + 4u * way, tag[i]);
                  lds_w32(set[i] + 204u, (way ^ 1u) | ((cnt + 1u) << 1) | (gen << 3));
                } else {
                  const u2_t qe = {0xffffffffu, 0u};  // claimed queue slot stays empty
                  lds_w64(queue + 8u * pos, qe);
                }
              }
              n += (uint32_t)__builtin_popcountll(ob);
            }
          }
          if (n > (uint32_t)kBrickQueueCap) n = kBrickQueueCap;
          n = __builtin_amdgcn_readfirstlane(n);
          wave_sync();
          for (uint32_t base = 0; base < n; base += 10u) {
            const uint32_t e = base + e_l;
            if (lane < 60u && e < n) {
              const u2_t qe = lds_r64(queue + 8u * e);
              if (qe.x != 0xffffffffu) lds_w128(qe.y + 16u * k_l, bricks[(size_t)qe.x * 8 + k_l]);
            }
          }
          gen++;
          wave_sync();
          slow = check();
        } while (slow && ++round < 2);
        if (slow) {
          // Overflow. Some set is asked for more bricks by this step's pixels than its two ways hold. The bricks that did not
          // fit go to the wave's SCRATCH slots, valid for this step only: per pixel slot, the first missing lane's brick is
          // given the next scratch slot and every lane waiting for that same brick is pointed at it (one wave-uniform
          // iteration per distinct brick); one more fetch pass brings them in. Lanes still without a brick when the scratch
          // slots run out (noise-like steps) are served one by one from the global table in the per-pixel path below.
          uint32_t n_ovf = 0;
#pragma unroll
          for (int i = 0; i < NP; i++) {
            const u2_t tg = lds_r64(set[i] + 192u);
            bool m = tg.x != tag[i] && tg.y != tag[i];
            unsigned long long mb = __ballot(m);
            if (mb) {
              const uint32_t cell = FMT ? tag[i] : cell_of(px[i]);
              while (mb && n_ovf < (uint32_t)kBrickScratch) {
                const int L = __builtin_ctzll(mb);
                const uint32_t cellL = __builtin_amdgcn_readlane(cell, L);
                const bool same = m && cell == cellL;
                const uint32_t dst = SCR + wave * (kBrickScratch * 96u) + 96u * n_ovf;
                if (same) baddr[i] = dst;
                if (lane == 0) { const u2_t qe = {cellL, dst}; lds_w64(queue + 8u * n_ovf, qe); }
                n_ovf++;
                m = m && !same;
                mb = __ballot(m);
              }
            }
            if (m) lane_res |= 1u << i;  // this lane's pixel i has no brick in LDS
          }
          n_ovf = __builtin_amdgcn_readfirstlane(n_ovf);
          wave_sync();
          {
            const uint32_t e = e_l;  // kBrickScratch <= 10: one pass
            if (lane < 60u && e < n_ovf) {
              const u2_t qe = lds_r64(queue + 8u * e);
              lds_w128(qe.y + 16u * k_l, bricks[(size_t)qe.x * 8 + k_l]);
            }
          }
          wave_sync();
          slow = __any(lane_res != 0u);
        }
      }
      uint32_t out[NP];
      // one pixel from its brick into out[i] (RGBA64: the pixel's two words are rewritten in place in px / px_hi)
      auto emit = [&](const f4_t (&f)[6], int i) {
        if constexpr (FMT == 0) out[i] = brick_pixel(f, tx[i], ty[i], tz[i], px[i]);
        else { brick_pixel64<FMT == 2>(f, tx[i], ty[i], tz[i], px[i], px_hi[i]); out[i] = px[i]; }
      };
      if (__builtin_expect(!slow, 1)) {
        // fast path: every lane's four bricks are resident. The next pixel's brick is read while this one's is worked on (two
        // register sets): a brick read left to its own devices waits out a full LDS round trip per pixel with nothing to issue
        if constexpr (PIPE) {
          f4_t f[2][6];
#pragma unroll
          for (int k = 0; k < 6; k++) f[0][k] = lds_r128(baddr[0] + 16u * k);
#pragma unroll
          for (int i = 0; i < NP; i++) {
            if (i + 1 < NP) {
              // (s_setprio 1 / 3 around these six reads, round 5: 0.1260 -> 0.1205 / 0.1267 ms at amp 0, 0.1492 -> 0.1577 / 0.1500 at
              //  +-4 - noise; not kept)
#pragma unroll
              for (int k = 0; k < 6; k++) f[(i + 1) & 1][k] = lds_r128(baddr[i + 1] + 16u * k);
            }
            if constexpr (BRICK_DUMMY_LDS > 0) {
#pragma unroll
              for (int d = 0; d < BRICK_DUMMY_LDS; d++) { const f4_t x = lds_r128(baddr[i] + 16u * (d % 6)); asm volatile("" ::"v"(x)); }
            }
            emit(f[i & 1], i);
          }
        } else {
#pragma unroll
          for (int i = 0; i < NP; i++) {
            f4_t f[6];
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = lds_r128(baddr[i] + 16u * k);
            emit(f, i);
          }
        }
      } else {
        // slow path: pixel slots in which some lane is still without a brick after fills and scratch (noise-like content):
        // those lanes read their brick's line from the global table, everyone else reads LDS (cache or scratch)
        slow_steps += G;
#pragma unroll 1
        for (int i = 0; i < NP; i++) {
          f4_t f[6];
          if ((lane_res >> i) & 1u) {
            const uint32_t cell = FMT ? tag[i] : cell_of(px[i]);
            const f4_t *gb = bricks + (size_t)cell * 8;
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = gb[k];
          } else {
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = lds_r128(baddr[i] + 16u * k);
          }
          emit(f, i);
        }
      }
#pragma unroll
      for (int g2 = 0; g2 < G; g2++) {
        const unsigned ro = 2u * (j + g2) + sub;
        if (col_ok && row0 + ro < rows) {
          u4_t o;
          if constexpr (FMT == 0) o = u4_t{out[4 * g2 + 0], out[4 * g2 + 1], out[4 * g2 + 2], out[4 * g2 + 3]};
          else o = u4_t{out[2 * g2], px_hi[2 * g2], out[2 * g2 + 1], px_hi[2 * g2 + 1]};
          const uint32_t byte_off = (ro * dw4 + col) * 16u;
          __builtin_nontemporal_store(o, (gl_u4 *)(dst_tile + byte_off));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < P; j++) cur[j] = nxt[j];
    cur_id = nxt_id;
    strip = nstrip;
    row0 = nrow0;
    col = ncol;
    col_ok = ncol_ok;
  }
  // counters are spread over kBrickCounterSlots slots (the host adds them up): thousands of waves finishing together and
  // adding to ONE address serialise in L2 at ~12 ns per atomic - a 0.1 ms tail on a 0.1 ms kernel
  const unsigned wave_id = blockIdx.x * W + wave;
#ifdef BRICK_TIMING
  if (lane == 0 && wave_id < 16384) {
    unsigned long long *tr = counters + 2 * kBrickCounterSlots + 4 * (size_t)wave_id;
    tr[0] = t_start;
    tr[1] = __builtin_amdgcn_s_memrealtime();
    tr[2] = miss_steps | ((unsigned long long)slow_steps << 32);
    tr[3] = tiles_done | ((unsigned long long)tiles_stolen << 32);
  }
#endif
  if (counters && lane == 0 && miss_steps) {
    unsigned long long *c = counters + 2 * (wave_id % kBrickCounterSlots);
    atomicAdd(c + 0, (unsigned long long)miss_steps);
    if (slow_steps) atomicAdd(c + 1, (unsigned long long)slow_steps);
  }
}

// ---------------------------------------------------------------- host side

// One axis-table entry for input byte v: the reference's coordinate arithmetic, op for op (norm_comp imp.rs:471-474,
// apply_3d :438-440, sample_3d :496-506; this translation unit is compiled with -ffp-contract=off).
// ---------------------------------------------------------------- the block-shared brick cache (round 4)
// colorlut3d_brick_kernel gives every WAVE 6-14 KB of bricks: enough for a smooth region, gone with +-8 of noise (73 % of the
// steps miss at 32 sets, the 64-set geometry has 8-10 waves per CU). The cells a noisy region needs are few - +-11 levels
// around a slowly moving base are 4-5 cells per axis of a 33^3 LUT, ~100 bricks, 10 KB - they are just not any one wave's.
// colorlut3d_shared_kernel gives the BLOCK (one per CU, 16 waves) one cache: 512 sets x 2 ways x 96 B = the cell's place in a
// box of 8 x 8 x 8 cells, so a cloud of colours up to 64 levels wide never collides with itself and two clouds (an edge) get
// a way each. It is the cache of colorlut_window.hip with LUT cells instead of table bricks - and the arithmetic of the brick
// kernel behind it (brick_pixel: the reference's lerps op for op, imp.rs:493-543):
//   per pixel: three axis-table reads {t, set | tag}, the set's {tag0, tag1, generation} (one 16-byte read), the six 16-byte
//   rows of the brick from the way that holds it, the generation again; 27 packed / scalar lerp instructions.
//   miss: the first lane of up to kShFills distinct missing bricks takes its set's lock (ds_cmpst), invalidates the older way
//   and bumps the generation; the wave copies the bricks from the global brick table (six lanes per brick, all in flight
//   together), the leaders publish and unlock; the pixels are looked up again and a lane whose brick is still not there reads
//   its six rows from the global table itself - always exact, the cache is only ever a copy.
//   No barrier after the prologue; a reader whose generation read AFTER its brick reads still shows the value read BEFORE them
//   cannot have overlapped an install into that set (the LDS executes a wave's operations in order; an install's first two
//   operations are "tag gone", "generation bumped", in that order).
// Walk, prefetch and stores are those of colorlut_window_kernel: a block walks down a 256-pixel-wide strip, 32 rows per step, two
// rows per wave, two steps of pixels in flight, range-checked buffer stores. 8 B/pixel algorithmic.
namespace {
constexpr uint32_t kShAxis = 0;               // 3 x 256 x {t, set | tag}
constexpr uint32_t kShCell = 6144;            // 3 x 256 bytes: cell index per axis (install / fallback path)
constexpr uint32_t kShQueue = 6912;           // 16 waves x 8 x {cell, destination}
constexpr uint32_t kShProgress = 8192;        // 16 waves x steps done (the pace keeper below)
constexpr uint32_t kShSets = 9216;            // 512 sets x 240 B: brick way 0 @0, way 1 @96, {tag0, tag1, generation0, generation1} @192, lock @208
constexpr uint32_t kShSetBytes = 240;         // (15 x 16 B: odd, see brick_upload)
constexpr uint32_t kShLdsBytes = kShSets + 512 * kShSetBytes;  // 132,096 B: one block per CU
#ifndef SH_SLACK  // (tools/exp_brick_build.sh)
#define SH_SLACK 2
#endif
constexpr int kShWaves = 16, kShFills = 8, kShDepth = 2, kShRounds = 2, kShSpins = 2048, kShSlack = SH_SLACK;
constexpr uint32_t kShNoTag = 0xffffffffu;    // a packed word never looks like this (the low half is a set offset / 16 < 8256)
static_assert(kShLdsBytes <= 160 * 1024 && kShSets % 256 == 0 && kShQueue + kShWaves * 64 <= kShSets, "LDS map");
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
typedef volatile __attribute__((address_space(3))) f4_t lds_vf4;
typedef volatile __attribute__((address_space(3))) u4_t lds_vu4;
__device__ __forceinline__ uint32_t lds_r32v(uint32_t a) { return *(lds_vu32 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w32v(uint32_t a, uint32_t v) { *(lds_vu32 *)(lds_byte *)(uintptr_t)a = v; }
__device__ __forceinline__ u4_t lds_r128uv(uint32_t a) { return *(lds_vu4 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ u2_t lds_r64v(uint32_t a) { return *(volatile __attribute__((address_space(3))) u2_t *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ f4_t lds_r128fv(uint32_t a) { return *(lds_vf4 *)(lds_byte *)(uintptr_t)a; }
__device__ __forceinline__ void lds_w128fv(uint32_t a, f4_t v) { *(lds_vf4 *)(lds_byte *)(uintptr_t)a = v; }
}  // namespace

__global__ __launch_bounds__(64 * kShWaves) void colorlut3d_shared_kernel(const u4_t *__restrict__ src, u4_t *__restrict__ dst, unsigned w4, unsigned sw4, unsigned dw4,
                                                                          unsigned rows, unsigned dst_bytes, unsigned steps_per_strip, unsigned share, unsigned extra,
                                                                          const f4_t *__restrict__ bricks, const u2_t *__restrict__ axis,
                                                                          const uint32_t *__restrict__ cellnum, unsigned lut_size, unsigned long long *__restrict__ counters) {
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned i = threadIdx.x; i < 768; i += 64 * kShWaves) {
    lds_w64(kShAxis + 8u * i, axis[3 * 768 + i]);
    lds_w8(kShCell + i, cellnum[i]);
  }
  if (threadIdx.x < kShWaves) lds_w32v(kShProgress + 4u * threadIdx.x, 0u);
  if (threadIdx.x < 512) {
    const uint32_t sa = kShSets + threadIdx.x * kShSetBytes;
    lds_w32v(sa + 192u, kShNoTag);
    lds_w32v(sa + 196u, kShNoTag);
    lds_w32v(sa + 200u, 0u);
    lds_w32v(sa + 204u, 0u);
    lds_w32v(sa + 208u, 0u);
  }
  __syncthreads();

  const unsigned first = blockIdx.x * share + (blockIdx.x < extra ? blockIdx.x : extra);
  const unsigned last = first + share + (blockIdx.x < extra ? 1u : 0u);
  const uint32_t three = 3u, four = 4u;
#ifdef BRICK_TIMING
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
  const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  unsigned miss_steps = 0, slow_steps = 0;

  struct Slot { u4_t p, q; unsigned strip, k; };  // (strip and k are the same in every lane: scalar registers)
  Slot ring[kShDepth];
  auto fetch = [&](unsigned st_, Slot &S) {
    const unsigned st = st_ < last ? st_ : last - 1u;
    const unsigned strip = st / steps_per_strip, k = st - strip * steps_per_strip;
    const unsigned col = strip * 64u + lane, r0 = k * (2u * kShWaves) + 2u * wave, r1 = r0 + 1u;
    const unsigned cc = col < w4 ? col : w4 - 1u, c0 = r0 < rows ? r0 : rows - 1u, c1 = r1 < rows ? r1 : rows - 1u;
    S.strip = st_ < last ? strip : 0xffffffffu;  // (a step past the end stores nothing)
    S.k = k;
    S.p = __builtin_nontemporal_load(src + ((size_t)c0 * sw4 + cc));
    S.q = __builtin_nontemporal_load(src + ((size_t)c1 * sw4 + cc));
  };

  // four pixels (one 16-byte row piece) through the cache
  auto half = [&](const u4_t pin) __attribute__((always_inline)) -> u4_t {
    const uint32_t px[4] = {pin.x, pin.y, pin.z, pin.w};
    float tx[4], ty[4], tz[4];
    uint32_t packed[4], set[4], out[4];
    {
      u2_t ex[4], ey[4], ez[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        ex[j] = lds_r64(byte_times8<0>(px[j], three) + kShAxis);
        ey[j] = lds_r64(byte_times8<1>(px[j], three) + (kShAxis + 2048u));
        ez[j] = lds_r64(byte_times8<2>(px[j], three) + (kShAxis + 4096u));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        tx[j] = __uint_as_float(ex[j].x);
        ty[j] = __uint_as_float(ey[j].x);
        tz[j] = __uint_as_float(ez[j].x);
        packed[j] = (ex[j].y + ey[j].y) + (ez[j].y + (kShSets >> 4));
        set[j] = word0_times16(packed[j], four);
      }
    }
    // the sets' {tag way 0, tag way 1, generation way 0, generation way 1}: which way holds the brick, and the generation its rows are read under
    u4_t meta[4];
    bool hit[4], miss_any = false;
#pragma unroll
    for (int j = 0; j < 4; j++) meta[j] = lds_r128uv(set[j] + 192u);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      hit[j] = (meta[j].x == packed[j]) | (meta[j].y == packed[j]);
      miss_any = miss_any | !hit[j];
    }
    if (__builtin_amdgcn_ballot_w64(miss_any) != 0ull) {
      miss_steps += 1;  // (the watch counts in 256-pixel steps)
#pragma unroll 1
      for (int round = 0; round < kShRounds; round++) {
        // leaders = the first lane of up to kShFills distinct missing bricks (a lane speaks for its first missed pixel); every
        // wave starts its search at another lane: the waves of a block miss the same new bricks at the same time
        uint32_t mb = kShNoTag, pxm = 0;
#pragma unroll
        for (int j = 3; j >= 0; j--)
          if (!hit[j]) { mb = packed[j]; pxm = px[j]; }
        unsigned long long want = __builtin_amdgcn_ballot_w64(mb != kShNoTag), leaders = 0ull;
        const unsigned rot = (wave * 4u + 1u) & 63u;
#pragma unroll
        for (int f = 0; f < kShFills; f++)
          if (want != 0ull) {
            const unsigned long long turned = (want >> rot) | (want << (64u - rot));
            const int l = (int)((__builtin_ctzll(turned) + rot) & 63u);
            leaders |= 1ull << l;
            want &= ~__builtin_amdgcn_ballot_w64(mb == (uint32_t)__builtin_amdgcn_readlane((int)mb, l));
          }
        bool claimed = false;
        uint32_t dest = 0;
        const uint32_t mset = word0_times16(mb, four);
        if ((leaders >> lane) & 1ull) {
          uint32_t expect = 0;
          lds_u32 *lock = (lds_u32 *)(lds_byte *)(uintptr_t)(mset + 208u);
          if (__hip_atomic_compare_exchange_strong(lock, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            const u4_t m = lds_r128uv(mset + 192u);
            if (m.x == mb || m.y == mb) {
              lds_w32v(mset + 208u, 0u);  // another wave has installed it since this wave's lookup
            } else {
              const uint32_t way = (m.z + m.w) & 1u;  // the ways take turns
              lds_w32v(mset + 192u + 4u * way, kShNoTag);               // the tag goes first,
              lds_w32v(mset + 200u + 4u * way, (way ? m.w : m.z) + 1u);  // then the way's generation: a reader that saw the tag sees this before any new byte
              dest = mset + 96u * way;
              claimed = true;
            }
          }
          // the queue entry: {cell number x0 + S (y0 + S z0), destination}, or "nothing" for a leader that did not claim
          const uint32_t cell = lds_r8(kShCell + (pxm & 0xffu)) + lut_size * (lds_r8(kShCell + 256u + ((pxm >> 8) & 0xffu)) + lut_size * lds_r8(kShCell + 512u + ((pxm >> 16) & 0xffu)));
          const u2_t qe = {claimed ? cell : kShNoTag, dest};
          lds_w64(kShQueue + 64u * wave + 8u * (uint32_t)__builtin_popcountll(leaders & ((1ull << lane) - 1ull)), qe);
        }
        if (__builtin_amdgcn_ballot_w64(claimed) != 0ull) {
          // six lanes per brick copy its six rows; all bricks of this round travel together
          const unsigned e = lane / 6u, r = lane - 6u * e;
          if (e < (unsigned)__builtin_popcountll(leaders)) {
            const u2_t qe = lds_r64(kShQueue + 64u * wave + 8u * e);
            if (qe.x != kShNoTag) lds_w128fv(qe.y + 16u * r, bricks[(size_t)qe.x * 8 + r]);
          }
          if (claimed) {  // publish after the rows (program order = LDS order), then let go
            lds_w32v(mset + 192u + 4u * ((dest - mset) / 96u), mb);
            lds_w32v(mset + 208u, 0u);
          }
        }
        // A brick this lane misses may be on its way in another wave's hands (its set locked): wait for it - the holder is
        // one memory round trip from publishing, and reading the brick from memory here would take as long and cost 6 loads
        // per lane. Nobody waits with a lock in hand (this wave's were released above), so the wait ends.
        if (mb != kShNoTag) {
          for (int spin = 0; spin < kShSpins; spin++) {
            const u2_t m = lds_r64v(mset + 192u);
            if ((m.x == mb) | (m.y == mb) | (lds_r32v(mset + 208u) == 0u)) break;
            __builtin_amdgcn_s_sleep(2);
          }
        }
        // the tags again (only: the pixels are computed once, below)
        bool still = false;
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (!hit[j]) {
            meta[j] = lds_r128uv(set[j] + 192u);
            hit[j] = (meta[j].x == packed[j]) | (meta[j].y == packed[j]);
            still = still | !hit[j];
          }
        if (__builtin_amdgcn_ballot_w64(still) == 0ull) break;
      }
    }
    // the pixels: six rows from the way that holds the brick, then the generation again - an install into the set in between
    // (it bumps the generation before it writes a byte) makes the pixel one for the careful path
    bool redo = false;
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool h1 = meta[j].y == packed[j];
      const uint32_t b = h1 ? set[j] + 96u : set[j];
      f4_t f[6];
#pragma unroll
      for (int r = 0; r < 6; r++) f[r] = lds_r128fv(b + 16u * r);
      const uint32_t gen2 = lds_r32v(h1 ? set[j] + 204u : set[j] + 200u);
      out[j] = brick_pixel(f, tx[j], ty[j], tz[j], px[j]);
      ok[j] = hit[j] & (gen2 == (h1 ? meta[j].w : meta[j].z));
      redo = redo | !ok[j];
    }
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
      // not in the cache (more bricks asked of one set than it has ways, or than the rounds install) or overtaken by an
      // install: the rows come from the global brick table
      slow_steps += 1;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (!ok[j]) {
          const uint32_t cell = lds_r8(kShCell + (px[j] & 0xffu)) + lut_size * (lds_r8(kShCell + 256u + ((px[j] >> 8) & 0xffu)) + lut_size * lds_r8(kShCell + 512u + ((px[j] >> 16) & 0xffu)));
          f4_t f[6];
#pragma unroll
          for (int r = 0; r < 6; r++) f[r] = bricks[(size_t)cell * 8 + r];
          out[j] = brick_pixel(f, tx[j], ty[j], tz[j], px[j]);
        }
      // (a use inside the branch: the wait for these loads then sits here and not in front of every step's stores)
#pragma unroll
      for (int j = 0; j < 4; j++) asm volatile("" : "+v"(out[j]));
    }
    const u4_t o = {out[0], out[1], out[2], out[3]};
    return o;
  };

  // The pace keeper. A SIMD issues for its oldest wave first: left alone, the first waves of a block finish a third of the
  // kernel before the last (measured: 50 of 135 us), the tail runs on a half-empty CU, and while they are ten steps apart the
  // block needs the bricks of both places - where two colour regions share their sets that is three bricks for two ways.
  // So a wave that is more than kShSlack steps ahead of the slowest sleeps until it is not (the slowest never waits; nobody
  // waits with a lock in hand). A little apart is good: the wave in front takes the misses, the others find the bricks there.
  unsigned steps_done = 0;
  auto keep_pace = [&]() __attribute__((always_inline)) {
    if (lane == 0) lds_w32v(kShProgress + 4u * wave, steps_done);
    for (int spin = 0; spin < kShSpins; spin++) {
      const uint32_t other = lds_r32v(kShProgress + 4u * (lane & (kShWaves - 1)));
      if (__builtin_amdgcn_ballot_w64(other + kShSlack < steps_done) == 0ull) break;
      __builtin_amdgcn_s_sleep(8);
    }
    steps_done++;
  };
  auto step = [&](unsigned st, Slot &S) __attribute__((always_inline)) {
    keep_pace();
    const unsigned col = S.strip * 64u + lane, r0 = S.k * (2u * kShWaves) + 2u * wave, r1 = r0 + 1u;
    const bool in = (S.strip != 0xffffffffu) & (col < w4);
    const u4_t oa = half(S.p), ob = half(S.q);
    const uint32_t so0 = (in & (r0 < rows)) ? (r0 * dw4 + col) << 4 : 0x80000000u, so1 = (in & (r1 < rows)) ? (r1 * dw4 + col) << 4 : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(oa, dst_rsrc, (int)so0, 0, 2 /* nt */);
    __builtin_amdgcn_raw_buffer_store_b128(ob, dst_rsrc, (int)so1, 0, 2);
    fetch(st + kShDepth, S);
  };

  if (first < last) {
#pragma unroll
    for (int d = 0; d < kShDepth; d++) fetch(first + d, ring[d]);
  }
  unsigned st = first;
  for (; st + kShDepth <= last; st += kShDepth) {
#pragma unroll
    for (int d = 0; d < kShDepth; d++) step(st + d, ring[d]);
  }
#pragma unroll
  for (int d = 0; d < kShDepth - 1; d++)
    if (st + d < last) step(st + d, ring[d]);
  if (lane == 0) lds_w32v(kShProgress + 4u * wave, 0x7fffffffu);  // done: nobody waits for this wave any more
#ifdef BRICK_TIMING
  if (lane == 0 && counters) {
    unsigned long long *tr = counters + 2 * kBrickCounterSlots + 4 * (size_t)(blockIdx.x * kShWaves + wave);
    tr[0] = t_start;
    tr[1] = __builtin_amdgcn_s_memrealtime();
    tr[2] = (unsigned long long)miss_steps | ((unsigned long long)slow_steps << 32);
    tr[3] = first | ((unsigned long long)last << 32);
  }
#endif
  if (lane == 0 && counters) {
    unsigned long long *c = counters + 2 * ((blockIdx.x * kShWaves + wave) % kBrickCounterSlots);
    atomicAdd(c, (unsigned long long)miss_steps);
    atomicAdd(c + 1, (unsigned long long)slow_steps);
  }
}

bool shared_applicable(const mi355_ctx *ctx, int width, int dst_stride, int n_frames, int height, bool *small) {
  const size_t rows = (size_t)n_frames * (size_t)height;
  if (small) *small = false;
  if (width % 4 != 0 || rows == 0 || rows * (size_t)dst_stride > (1ull << 31)) return false;
  const size_t steps = (size_t)(((unsigned)width / 4 + 63) / 64) * ((rows + 2 * kShWaves - 1) / (2 * kShWaves));
  // small: up to six steps per CU (one 4K frame is four) - there the shared cache is ahead on clean content as well
  if (small) *small = steps < (size_t)ctx->n_cu * 6;
  return steps < (1u << 31) && steps >= (size_t)ctx->n_cu * 3;  // a block's first step is cold: it needs a few behind it
}

int shared_launch(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int n_frames, int width, int height) {
  const unsigned w4 = (unsigned)width / 4, rows = (unsigned)((size_t)n_frames * height);
  const unsigned n_strips = (w4 + 63) / 64, steps_per_strip = (rows + 2 * kShWaves - 1) / (2 * kShWaves);
  const unsigned total = n_strips * steps_per_strip;
  unsigned grid = (unsigned)ctx->n_cu;
  if (grid > total) grid = total;
  hipLaunchKernelGGL(colorlut3d_shared_kernel, dim3(grid), dim3(64 * kShWaves), kShLdsBytes, ctx->stream, (const u4_t *)d_src, (u4_t *)d_dst, w4, (unsigned)src_stride / 16,
                     (unsigned)dst_stride / 16, rows, (unsigned)((size_t)rows * (size_t)dst_stride), steps_per_strip, total / grid, total % grid, (const f4_t *)B.d_bricks,
                     (const u2_t *)B.d_axis, B.d_cellnum, (unsigned)B.size, B.d_counters);
#ifdef BRICK_TIMING
  if (const char *path = getenv("BRICK_TIMING_FILE")) {
    std::vector<unsigned long long> tr(kBrickTimingBytes / 8);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(tr.data(), (const char *)B.d_counters + kBrickCounterBytes, kBrickTimingBytes, hipMemcpyDeviceToHost);
    if (FILE *f = fopen(path, "wb")) { unsigned long long hdr[4] = {grid * (unsigned long long)kShWaves, steps_per_strip, n_strips, 512ull}; fwrite(hdr, 8, 4, f); fwrite(tr.data(), 8, std::min<size_t>(tr.size(), (size_t)grid * kShWaves * 4), f); fclose(f); }
  }
#endif
  return check_hip(ctx, hipGetLastError(), "colorlut shared-cache kernel launch");
}

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
  const int fold_axis = ctx->brick_fold_axis;
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
          float cq[3], dq[3];
          for (int ch = 0; ch < 3; ch++) {
            volatile float d = c1[ch] - c0[ch];  // the `(b - a)` of lerp4 (imp.rs:528-535), rounded to f32
            cq[ch] = c0[ch];
            dq[ch] = d;
          }
          // register layout of brick_pixel: rows 0..3 = {c.r, c.g, d.r, d.g}; row 4 = b of q 0 and 2; row 5 = b of q 1 and 3
          b[4 * q + 0] = cq[0]; b[4 * q + 1] = cq[1]; b[4 * q + 2] = dq[0]; b[4 * q + 3] = dq[1];
          float *rb = b + 16 + 4 * (q & 1);
          rb[(q >> 1)] = cq[2];
          rb[2 + (q >> 1)] = dq[2];
        }
      }
  // axis tables for the three per-wave cache geometries (ZN = 2, 3, 4): [zn-2][axis][byte] = {t, set byte offset | tag contribution << 16},
  // and a fourth one ([3]) for the block-shared cache of colorlut3d_shared_kernel: 512 sets = the cell's place in a box of
  // 8 x 8 x 8 cells (set byte offset / 16 in the low half: 15 * set number; the set number takes its low three bits from the low
  // bits of x0, y0, z0 and a set is an odd number of 16-byte LDS bank groups long, so the eight bricks around any point - what the
  // lanes of a wave read together on smooth content - lie in eight different bank groups; no carry into the
  // high half), tag = which box (x0 >> 3 | (y0 >> 3) << 4 | (z0 >> 3) << 8, S <= 65)
  std::vector<uint32_t> axis(4 * 3 * 256 * 2), cellnum(3 * 256);
  for (int a = 0; a < 3; a++)
    for (int v = 0; v < 256; v++) {
      float t;
      int i0;
      brick_axis_entry(v, scale[a], offset[a], S, &t, &i0);
      uint32_t *e = &axis[((size_t)3 * 768 + (size_t)a * 256 + v) * 2];
      std::memcpy(&e[0], &t, 4);
      const uint32_t set_index = ((uint32_t)(i0 & 1) << a) + ((uint32_t)((i0 >> 1) & 3) << (3 + 2 * a));
      e[1] = 15u /* kShSetBytes / 16 */ * set_index + (((uint32_t)(i0 >> 3) << (4 * a)) << 16);
    }
  for (int zn = 2; zn <= 4; zn++)
    for (int a = 0; a < 3; a++)
      for (int v = 0; v < 256; v++) {
        float t;
        int i0;
        brick_axis_entry(v, scale[a], offset[a], S, &t, &i0);
        // residues per axis in the set index: (4, 4, zn), or - zn == 2 only - the 2 on the axis fold_axis instead of z
        int mods[3] = {4, 4, zn};
        if (zn == 2 && fold_axis != 2) { mods[2] = 4; mods[fold_axis] = 2; }
        const int mod = mods[a];
        const uint32_t r = (uint32_t)(i0 % mod);
        const uint32_t stride1 = (uint32_t)mods[0] * kBrickSetBytes;                   // y stride
        const uint32_t stride2 = (uint32_t)mods[1] * stride1 + 128u;                   // z stride (+128: see kBrickZStride)
        const uint32_t set_off = a == 0 ? r * kBrickSetBytes : (a == 1 ? r * stride1 : r * stride2);
        // tag fields packed axis after axis, each as wide as its largest value (S - 1) / mod needs: 16 bits in all for S <= 65
        auto bits_for = [&](int m) { int b = 0; while (((S - 1) / m) >> b) b++; return b; };
        const int tag_shift = a == 0 ? 0 : (a == 1 ? bits_for(mods[0]) : bits_for(mods[0]) + bits_for(mods[1]));
        if (brick_hashed(zn)) {
          if ((size_t)S * S * S >= (1u << 24)) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: brick cache geometry does not fit");
        } else if (bits_for(mods[0]) + bits_for(mods[1]) + bits_for(mods[2]) > 32 - kBrickTagShift || set_off % 16u != 0u ||
            (uint32_t)(mods[0] - 1) * kBrickSetBytes + (uint32_t)(mods[1] - 1) * stride1 + (uint32_t)(mods[2] - 1) * stride2 + kBrickSetBytes > (uint32_t)brick_wave_bytes(zn))
          return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: brick cache geometry does not fit");
        const uint32_t tag = (uint32_t)(i0 / mod) << tag_shift;
        uint32_t *e = &axis[((size_t)(zn - 2) * 768 + (size_t)a * 256 + v) * 2];
        std::memcpy(&e[0], &t, 4);
        e[1] = (set_off >> 4) + (tag << kBrickTagShift);
        if (brick_hashed(zn)) {
          const uint32_t mul = a == 0 ? 1u : (a == 1 ? 3u : 9u), cell = (uint32_t)i0 * (a == 0 ? 1u : (a == 1 ? (uint32_t)S : (uint32_t)S * S));
          e[1] = (((uint32_t)i0 * mul) & 31u) | (cell << 8);  // three of these add up without carries: 93 < 256, S^3 < 2^24
        }
        cellnum[(size_t)a * 256 + v] = (uint32_t)i0;
      }
  int rc = check_hip(ctx, hipMalloc((void **)&B.d_bricks, bricks.size() * sizeof(float)), "hipMalloc(lut bricks)");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(B.d_bricks, bricks.data(), bricks.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(lut bricks)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc((void **)&B.d_axis, axis.size() * sizeof(uint32_t)), "hipMalloc(brick axis tables)"))) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(B.d_axis, axis.data(), axis.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(brick axis tables)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc((void **)&B.d_cellnum, cellnum.size() * sizeof(uint32_t)), "hipMalloc(brick cell numbers)"))) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(B.d_cellnum, cellnum.data(), cellnum.size() * sizeof(uint32_t), hipMemcpyHostToDevice), "hipMemcpy(brick cell numbers)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc((void **)&B.d_counters, kBrickCounterBytes + kBrickTimingBytes), "hipMalloc(brick counters)"))) return rc;
  if ((rc = check_hip(ctx, hipMemset(B.d_counters, 0, kBrickCounterBytes), "hipMemset(brick counters)"))) return rc;
  if ((rc = check_hip(ctx, hipHostMalloc((void **)&B.h_counters, kBrickCounterBytes, hipHostMallocDefault), "hipHostMalloc(brick counters)"))) return rc;
  std::memset(B.h_counters, 0, kBrickCounterBytes);
  if ((rc = check_hip(ctx, hipEventCreateWithFlags(&B.ev, hipEventDisableTiming), "hipEventCreate(brick)"))) return rc;
  B.size = S;
  B.fold_axis = fold_axis;
  for (int c = 0; c < 3; c++) { B.scale[c] = scale[c]; B.offset[c] = offset[c]; }
  B.ok = true;
  return MI355_OK;
}

bool brick_applicable(const BrickLut &B, const uint8_t *d_src, size_t src_pitch, int src_stride, const uint8_t *d_dst, size_t dst_pitch,
                      int dst_stride, int n_frames, int width, int height, int bytes_per_pixel) {
  const int px_per_group = 16 / bytes_per_pixel;  // 4 (RGBA8) or 2 (RGBA64)
  if (!B.ok || width < px_per_group || width % px_per_group != 0) return false;
  // rows may be padded (strides that are multiples of 16 B - what aligned allocators negotiate); a batch must then be one
  // tall picture: frames `stride * height` apart
  const size_t row_bytes = (size_t)width * (size_t)bytes_per_pixel;
  if ((size_t)src_stride < row_bytes || (size_t)dst_stride < row_bytes || src_stride % 16 != 0 || dst_stride % 16 != 0) return false;
  if (n_frames != 1 && (src_pitch != (size_t)src_stride * (size_t)height || dst_pitch != (size_t)dst_stride * (size_t)height)) return false;
  if ((uintptr_t)d_src % 16 != 0 || (uintptr_t)d_dst % 16 != 0) return false;
  return (size_t)n_frames * (size_t)height < (1u << 30) && (size_t)src_stride * 4 < (1u << 28) && (size_t)dst_stride * 4 < (1u << 28);
}

template <int HSV, int ZN, int FMT = 0>
static int brick_launch_t(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int n_frames, int width, int height, const HsvK &hk) {
  // tile = 128 px x 2P rows, prefetched one tile ahead: with 8 waves per CU a 4-row tile is consumed faster than HBM answers
  constexpr int P = ZN == 4 ? BRICK_P64 : 2;
  const unsigned w4 = (unsigned)width / (FMT ? 2 : 4), rows = (unsigned)((size_t)n_frames * height);
  const unsigned n_strips = (w4 + 31) / 32;
  const unsigned tile_rows = (rows + 2 * P - 1) / (2 * P);
  // Run length: a wave's cache starts cold at the top of its run, so runs are as long as the launch allows while still
  // giving every wave of the chip (one block of W waves per CU) a run of its own: ONE round of runs, all of about the same
  // length; what unevenness remains (content, SIMD sharing) the waves of a block even out among themselves (the deques).
  constexpr unsigned W = brick_waves(ZN);
  const size_t wave_slots = (size_t)ctx->n_cu * W;
  unsigned tpr = (unsigned)(((size_t)n_strips * tile_rows + wave_slots - 1) / wave_slots);
  if (tpr < 8) tpr = 8;
  if (tpr > 512) tpr = 512;
  if (ctx->brick_tiles_per_run > 0) tpr = (unsigned)ctx->brick_tiles_per_run;
  const unsigned runs_per_strip = (tile_rows + tpr - 1) / tpr;
  const size_t n_runs = (size_t)n_strips * runs_per_strip;
  if (n_runs >= (1u << 30)) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: frame batch too large");
  // groups of SB runs, W / SB groups per block; small launches: as many blocks as there are groups, up to one per CU (the
  // waves of a block that own nothing steal)
  constexpr unsigned SB = brick_group(ZN), K = W / SB;
  static_assert(W % SB == 0, "whole groups per block");
  const size_t n_groups = (n_runs + SB - 1) / SB;
  size_t grid = (n_groups + K - 1) / K;
  const size_t spread = n_groups < (size_t)ctx->n_cu ? n_groups : (size_t)ctx->n_cu;
  if (grid < spread) grid = spread;
  // RGBA64: two pixels per 16-byte group, so two groups per step keep a step at four pixels per lane
  constexpr int G = FMT ? 2 : (ZN == 4 ? BRICK_G64 : 1);
  static_assert(P % G == 0, "a tile is a whole number of steps");
  BrickK64 k64;
  for (int c = 0; c < 3; c++) { k64.scale[c] = B.scale[c]; k64.offset[c] = B.offset[c]; }
  k64.sm1 = (float)B.size - 1.0f;
  hipLaunchKernelGGL((colorlut3d_brick_kernel<P, HSV, ZN, G, FMT>), dim3((unsigned)grid), dim3(64 * W), brick_lds_bytes(ZN), ctx->stream, (const u4_t *)d_src, (u4_t *)d_dst, w4,
                     (unsigned)src_stride / 16, (unsigned)dst_stride / 16, rows, n_strips, tpr, (unsigned)n_runs | ((unsigned)(ctx->brick_prio & 3) << 30), (const f4_t *)B.d_bricks,
                     (const u2_t *)B.d_axis + (size_t)(ZN - 2) * 768, (const uint32_t *)B.d_cellnum, B.d_counters, hk, (unsigned)B.fold_axis, (unsigned)B.size, k64);
#ifdef BRICK_TIMING
  if (const char *path = getenv("BRICK_TIMING_FILE")) {
    std::vector<unsigned long long> tr(kBrickTimingBytes / 8);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipMemcpy(tr.data(), (const char *)B.d_counters + kBrickCounterBytes, kBrickTimingBytes, hipMemcpyDeviceToHost);
    if (FILE *f = fopen(path, "wb")) { unsigned long long hdr[4] = {grid * W, tpr, n_strips, (unsigned long long)ZN}; fwrite(hdr, 8, 4, f); fwrite(tr.data(), 8, std::min<size_t>(tr.size(), grid * W * 4), f); fclose(f); }
  }
#endif
  return check_hip(ctx, hipGetLastError(), "colorlut3d_brick kernel launch");
}

template <int HSV>
static int brick_launch_z(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int n_frames, int width, int height, const HsvK &hk, int sets) {
  if (sets == 64) return brick_launch_t<HSV, 4>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk);
  if (sets == 48) return brick_launch_t<HSV, 3>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk);
  return brick_launch_t<HSV, 2>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk);
}

int brick_launch(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int n_frames, int width, int height,
                 const mi355_hsv_settings *hs, int sets, int fmt64) {
  if (fmt64) {  // RGBA64 (1 little endian, 2 big endian): plain colorlut only, 32 hashed sets or the 64-set box
    if (hs) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: the fused hsvfilter form is RGBA only");
    if (sets == 64) return fmt64 == 1 ? brick_launch_t<kBrickNoHsv, 4, 1>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, HsvK{})
                                      : brick_launch_t<kBrickNoHsv, 4, 2>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, HsvK{});
    return fmt64 == 1 ? brick_launch_t<kBrickNoHsv, 2, 1>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, HsvK{})
                      : brick_launch_t<kBrickNoHsv, 2, 2>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, HsvK{});
  }
  if (!hs) return brick_launch_z<kBrickNoHsv>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, HsvK{}, sets);
  const HsvK hk{hs->hue_shift, hs->saturation_mul, hs->saturation_off, hs->value_mul, hs->value_off};
  switch (hsv_variant_for(*hs, false)) {
    case -1: return brick_launch_z<-1>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
    case 0: return brick_launch_z<0>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
    case 1: return brick_launch_z<1>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
    case 2: return brick_launch_z<2>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
    case 4: return brick_launch_z<4>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
    case 5: return brick_launch_z<5>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
    default: return brick_launch_z<6>(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, hk, sets);
  }
}

// Content watch mechanism (policy: brickwatch.hpp). Every kWatchSnapEvery-th brick launch at one level the miss counters
// are copied to pinned memory and reset, in stream order, behind an event that later launches poll - never wait for.
static void brick_harvest(BrickLut &B) {
  if (!B.pending) return;
  if (hipEventQuery(B.ev) != hipSuccess) { (void)hipGetLastError(); return; }
  B.pending = false;
  const double steps = (double)B.px_snapshot / 256.0;
  unsigned long long tot[2] = {0, 0};
  for (int i = 0; i < kBrickCounterSlots; i++) { tot[0] += B.h_counters[2 * i]; tot[1] += B.h_counters[2 * i + 1]; }
  if (steps > 0.0) watch_snapshot(B.watch, B.level_snapshot, (double)tot[0] / steps, (double)tot[1] / steps, B.shared_snapshot);
}

int brick_choose(BrickLut &B, int min_level) {
  brick_harvest(B);
  return watch_level(B.watch, !B.pending, min_level);
}

// before a brick launch at `level`: a change of level starts a new count (what the counters hold belongs to the old one)
int brick_before_launch(mi355_ctx *ctx, BrickLut &B, int level) {
  if (level == B.level_since) return MI355_OK;
  int rc = MI355_OK;
  if (B.level_since >= 0) rc = check_hip(ctx, hipMemsetAsync(B.d_counters, 0, kBrickCounterBytes, ctx->stream), "brick counters reset");
  B.level_since = level;
  B.px_since = 0;
  B.launches_since = 0;
  return rc;
}

// a launch outside the watch's accounting - the three-pass kernel while the stream is at level 2, a pinned or table-build
// brick launch (which adds to d_counters without adding to px_since): whatever the counters hold no longer belongs to one
// level's launches, so the next watched launch starts a new count (counters reset in stream order, brick_before_launch).
// Without this a one-launch probe of level 1 after a stretch at level 2 was judged on counters and pixels left over from
// before the stream moved up (the selftest model always did this; the real path did not).
void brick_mark_unwatched(BrickLut &B) { B.level_since = 2; }

int brick_after_launch(mi355_ctx *ctx, BrickLut &B, unsigned long long pixels, int level, bool shared1) {
  int rc;
  B.px_since += pixels;
  // a probe of a lower level is judged on its first launch (it runs the kernel believed slower), the level in use on four
  const unsigned need = B.watch.probing == level ? 1u : kWatchSnapEvery;
  if (++B.launches_since < need || B.pending) return MI355_OK;
  if ((rc = check_hip(ctx, hipMemcpyAsync(B.h_counters, B.d_counters, kBrickCounterBytes, hipMemcpyDeviceToHost, ctx->stream), "brick counters snapshot"))) return rc;
  if ((rc = check_hip(ctx, hipMemsetAsync(B.d_counters, 0, kBrickCounterBytes, ctx->stream), "brick counters reset"))) return rc;
  if ((rc = check_hip(ctx, hipEventRecord(B.ev, ctx->stream), "hipEventRecord(brick)"))) return rc;
  B.pending = true;
  B.px_snapshot = B.px_since;
  B.level_snapshot = level;
  B.shared_snapshot = shared1;
  B.px_since = 0;
  B.launches_since = 0;
  return MI355_OK;
}

int brick_read_counters(mi355_ctx *ctx, const BrickLut &B, unsigned long long out[2], bool reset) {
  out[0] = out[1] = 0;
  if (!B.d_counters) return MI355_OK;
  std::vector<unsigned long long> tmp(2 * kBrickCounterSlots);
  int rc = check_hip(ctx, hipMemcpyAsync(tmp.data(), B.d_counters, kBrickCounterBytes, hipMemcpyDeviceToHost, ctx->stream), "brick counters read");
  if (rc) return rc;
  if (reset && (rc = check_hip(ctx, hipMemsetAsync(B.d_counters, 0, kBrickCounterBytes, ctx->stream), "brick counters reset"))) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "brick counters sync"))) return rc;
  for (int i = 0; i < kBrickCounterSlots; i++) { out[0] += tmp[2 * i]; out[1] += tmp[2 * i + 1]; }
  return MI355_OK;
}

}  // namespace mi355
