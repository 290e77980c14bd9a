// colorlut_kernels.hip — gfx950 kernels for the colorlut element.
//
// Reference loops replaced (gst-plugins-rs tree, video/colorlut/src/colorlut/imp.rs):
//   :237-265 transform_rgba_1d      :267-294 transform_rgba_3d
//   :308-346 transform_rgba64_1d    :348-397 transform_rgba64_3d
//   :399-469 apply_*                :471-479 norm_comp[_u16]
//   :482-526 sample_1d / sample_3d  :528-543 lerp4 / float_to_u8 / float_to_u16
// and Lut3D::at (video/colorlut/src/parser.rs:43-53).
//
// Kernels
//   colorlut_rows_kernel<...>  GENERIC: any LUT size (2..256 3D, 2..65536 1D), RGBA8 / RGBA64 LE/BE,
//                              any strides; literal arithmetic (IEEE `/`, floor, round, NaN paths);
//                              LUT cells gathered from global memory ([r,g,b,1] float4, L2-resident).
//   colorlut3d_lds_kernel<...> FAST path for RGBA8 + 3D LUTs whose single-channel plane fits LDS
//                              (size <= 34): the block keeps its pixels in VGPRs and makes three
//                              passes, one per output channel, with that channel's size^3 f32 plane
//                              staged in LDS (33^3*4 B = 143.7 KB of the CU's 160 KB). Bit-identical
//                              to the generic kernel for LUTs with finite, bounded entries and a finite
//                              domain (checked at load time; otherwise the generic kernel is used).
//   colorlut1d_lds_kernel / colorlut3d_lds64_kernel / hsv_colorlut3d_pipe_kernel / colorlut3d_lean_kernel
//                              1D LUTs from LDS tables; the RGBA64 form of the three-pass kernel; the fused
//                              hsvfilter -> colorlut forms (HSV template argument, software-pipelined variant);
//                              the lean-state experiment.
//   colorlut_table_tiled_kernel / colorlut_table_kernel + table_domain_kernel
//                              packed RGBA8 through a 2^24-entry memoised table that the kernels above build on
//                              the device (bit-identical by construction): one gather per pixel; 2D-tiled or flat
//                              pixel-to-wave mapping. Serves colorlut, the fused chain (table of the composed
//                              function) and, opt-in / for GENERIC settings, hsvfilter.
// Host side at the end of the file: LUT upload and layout (lut_upload), the launchers, and the run-time choice between
// the interpolating and the table kernels (auto_launch; policy in autopick.hpp).
// Algorithmic traffic of all of them: 4 B read + 4 B written per pixel (RGBA8; 8 + 8 for RGBA64); the LUT / table is a
// cache/LDS-resident constant.
#include "internal.hpp"
#include "hsv_device.hpp"
#include "exact_math.hpp"

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace mi355 {

struct LutK {
  float scale[3];
  float offset[3];
  int size;
};

// ---------------------------------------------------------------- generic arithmetic

// `x.floor() as usize` then `.min(max_idx)` (imp.rs:485,496-498)
__device__ __forceinline__ uint32_t floor_idx(float x, uint32_t max_idx) {
  const float f = floorf(x);
  // as usize: NaN -> 0, negative -> 0, huge -> saturate (then min)
  uint32_t i = (f > 0.0f) ? ((f >= 4294967040.0f) ? 0xffffffffu : (uint32_t)f) : 0u;
  return i < max_idx ? i : max_idx;
}

__device__ __forceinline__ float float_to_u8f(float v) { return roundf(rs_clamp(v, 0.0f, 1.0f) * 255.0f); }
__device__ __forceinline__ float float_to_u16f(float v) { return roundf(rs_clamp(v, 0.0f, 1.0f) * 65535.0f); }

template <bool BITS16>
__device__ __forceinline__ float norm_comp_generic(const LutK &k, int c, float value) {
  const float v = BITS16 ? value / 65535.0f : value / 255.0f;
  return rs_clamp(v * k.scale[c] + k.offset[c], 0.0f, 1.0f);
}

__device__ __forceinline__ float sample_1d_generic(const float *__restrict__ lut, uint32_t len, float x) {
  const uint32_t max_idx = len - 1;
  const uint32_t x0 = floor_idx(x, max_idx);
  const uint32_t x1 = (x0 + 1 < max_idx) ? x0 + 1 : max_idx;
  const float t = x - (float)x0;
  const float a = lut[x0], b = lut[x1];
  return a + (b - a) * t;
}

__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + (b - a) * t; }

// trilinear on the [r,g,b,1] cells; alpha lane is never consumed by the reference callers.
__device__ __forceinline__ void sample_3d_generic(const float4 *__restrict__ cells, uint32_t size, float x,
                                                  float y, float z, float out[3]) {
  const uint32_t max_idx = size - 1;
  const uint32_t x0 = floor_idx(x, max_idx), y0 = floor_idx(y, max_idx), z0 = floor_idx(z, max_idx);
  const uint32_t x1 = (x0 + 1 < max_idx) ? x0 + 1 : max_idx;
  const uint32_t y1 = (y0 + 1 < max_idx) ? y0 + 1 : max_idx;
  const uint32_t z1 = (z0 + 1 < max_idx) ? z0 + 1 : max_idx;
  const float tx = x - (float)x0, ty = y - (float)y0, tz = z - (float)z0;
  const uint32_t s2 = size * size;
  const float4 c000 = cells[x0 + y0 * size + z0 * s2], c100 = cells[x1 + y0 * size + z0 * s2];
  const float4 c010 = cells[x0 + y1 * size + z0 * s2], c110 = cells[x1 + y1 * size + z0 * s2];
  const float4 c001 = cells[x0 + y0 * size + z1 * s2], c101 = cells[x1 + y0 * size + z1 * s2];
  const float4 c011 = cells[x0 + y1 * size + z1 * s2], c111 = cells[x1 + y1 * size + z1 * s2];
#define MI355_TRI(ch)                                                              \
  lerp1(lerp1(lerp1(c000.ch, c100.ch, tx), lerp1(c010.ch, c110.ch, tx), ty),       \
        lerp1(lerp1(c001.ch, c101.ch, tx), lerp1(c011.ch, c111.ch, tx), ty), tz)
  out[0] = MI355_TRI(x);
  out[1] = MI355_TRI(y);
  out[2] = MI355_TRI(z);
#undef MI355_TRI
}

__device__ __forceinline__ uint16_t bswap16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }

// One pixel per lane, rows addressed with independent strides (imp.rs:281-286, :364-367).
// BITS16: RGBA64 (u16 samples, LE selects byte order); else RGBA8.
template <bool IS3D, bool BITS16, bool LE>
__global__ __launch_bounds__(256) void colorlut_rows_kernel(const uint8_t *__restrict__ src, size_t src_pitch,
                                                            int src_stride, uint8_t *__restrict__ dst,
                                                            size_t dst_pitch, int dst_stride, int n_frames, int width,
                                                            int height, const float *__restrict__ table, LutK k) {
  const size_t per_frame = (size_t)width * (size_t)height;
  const size_t total = per_frame * (size_t)n_frames;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  const float sm1 = (float)k.size - 1.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const size_t f = i / per_frame;
    const size_t r = i - f * per_frame;
    const size_t row = r / (size_t)width;
    const size_t col = r - row * (size_t)width;
    float in[3];
    if constexpr (BITS16) {
      // stride is taken in u16 units: plane_stride / 2 (imp.rs:358-359)
      const uint16_t *s = (const uint16_t *)(src + f * src_pitch) + row * (size_t)(src_stride / 2) + col * 4;
      uint16_t *d = (uint16_t *)(dst + f * dst_pitch) + row * (size_t)(dst_stride / 2) + col * 4;
      uint16_t raw[4] = {s[0], s[1], s[2], s[3]};
      for (int c = 0; c < 3; c++) in[c] = (float)(LE ? raw[c] : bswap16(raw[c]));
      float o[3];
      if constexpr (IS3D) {
        sample_3d_generic((const float4 *)table, (uint32_t)k.size, norm_comp_generic<true>(k, 0, in[0]) * sm1,
                          norm_comp_generic<true>(k, 1, in[1]) * sm1, norm_comp_generic<true>(k, 2, in[2]) * sm1, o);
      } else {
        for (int c = 0; c < 3; c++)
          o[c] = sample_1d_generic(table + (size_t)c * k.size, (uint32_t)k.size, norm_comp_generic<true>(k, c, in[c]) * sm1);
      }
      for (int c = 0; c < 3; c++) {
        const uint16_t v = (uint16_t)rs_as_u16(float_to_u16f(o[c]));
        d[c] = LE ? v : bswap16(v);
      }
      d[3] = raw[3];  // alpha word copied raw (imp.rs:344,394)
    } else {
      const uint8_t *s = src + f * src_pitch + row * (size_t)src_stride + col * 4;
      uint8_t *d = dst + f * dst_pitch + row * (size_t)dst_stride + col * 4;
      const uint8_t raw[4] = {s[0], s[1], s[2], s[3]};
      for (int c = 0; c < 3; c++) in[c] = (float)raw[c];
      float o[3];
      if constexpr (IS3D) {
        sample_3d_generic((const float4 *)table, (uint32_t)k.size, norm_comp_generic<false>(k, 0, in[0]) * sm1,
                          norm_comp_generic<false>(k, 1, in[1]) * sm1, norm_comp_generic<false>(k, 2, in[2]) * sm1, o);
      } else {
        for (int c = 0; c < 3; c++)
          o[c] = sample_1d_generic(table + (size_t)c * k.size, (uint32_t)k.size, norm_comp_generic<false>(k, c, in[c]) * sm1);
      }
      d[0] = (uint8_t)rs_as_u8(float_to_u8f(o[0]));
      d[1] = (uint8_t)rs_as_u8(float_to_u8f(o[1]));
      d[2] = (uint8_t)rs_as_u8(float_to_u8f(o[2]));
      d[3] = raw[3];
    }
  }
}

// ---------------------------------------------------------------- LDS three-pass kernel (RGBA8, 3D)
//
// Work decomposition: a block owns a tile of NT*P pixels held in VGPRs (packed pixel, LDS byte base
// of its cell, tx, ty, tz) and makes three passes, one per output channel, with that channel's f32
// plane staged in LDS. One 16 B load + one 16 B store per 4 pixels; the LUT never leaves LDS/L2.
//
// LDS image (bytes), identical for the three passes except for the plane contents:
//   [0, 6144)            three axis tables, 256 entries x {int32 byte offset, f32 t} per axis:
//                        for input byte v on axis a: x = clamp(v/255*scale+offset,0,1)*(S-1) computed on
//                        the host with the reference's IEEE operations (norm_comp, imp.rs:471-474, and
//                        sample_3d's floor/min/sub, imp.rs:496-506); entry = {offset of floor(x), x-floor(x)}.
//   [6144, 6144+4*S*Sz)  the plane, cell (x,y,z) at float index x + Sy*y + Sz*(S-1-z)   (z reversed).
//   + 4*(Sy+2)           zero tail.
// Strides Sy = 3 (mod 32) and Sz = 9 (mod 32): the LDS bank of a cell is (x + 3y + 9z) mod 32, so
// lanes whose cells differ by any (dx,dy,dz) in {-1,0,1}^3 — the normal case for neighbouring pixels
// of natural content — hit 27 distinct banks (a plain 33x33x33 layout has 33 = 33^2 = 1 mod 32, i.e.
// x+1, y+1 and z+1 neighbours all collide on one bank).
// A lane's base addresses the (x0,y0,z0+1) corner; its 8 corners sit at the fixed offsets
// {0,4,4Sy,4Sy+4} (z0+1) and 4Sz + {0,4,4Sy,4Sy+4} (z0). No x1/y1/z1 clamping is done: whenever the
// reference clamps (x0 == S-1 <=> x == S-1 exactly) the weight is exactly 0 and a + (b-a)*0 == a for
// every finite b with finite b-a. The out-of-cube reads land on the next row / plane, the zero tail,
// or — for z0 == S-1, thanks to the reversed z order — on the axis tables just below the plane
// (small integers and t values: finite). The load-time check guarantees all entries are finite with
// |v| <= 1e30. (A sign-of-zero difference cannot reach a non-zero result; both zeros give byte 0.)

constexpr int kAxisTableBytes = 3 * 256 * 8;

struct LdsLayout {
  int S, Sy, Sz;
  bool all_resident;    // all three planes fit LDS at once
  size_t plane_floats;  // S*Sz + Sy + 2 rounded up to 256 (1 KiB)
  size_t lds_bytes;     // kAxisTableBytes + 4*plane_floats
};

static LdsLayout lds_layout_for(int S) {
  LdsLayout L;
  L.S = S;
  L.Sy = S;
  while (L.Sy % 32 != 3) L.Sy++;
  L.Sz = L.Sy * S;
  while (L.Sz % 32 != 9) L.Sz++;
  size_t pf = (size_t)S * L.Sz + L.Sy + 2;
  L.plane_floats = (pf + 255) & ~(size_t)255;  // whole 1 KiB chunks for the async global->LDS staging
  L.lds_bytes = kAxisTableBytes + 4 * L.plane_floats;
  L.all_resident = kAxisTableBytes + 3 * 4 * L.plane_floats <= 160 * 1024;
  if (L.all_resident) L.lds_bytes = kAxisTableBytes + 3 * 4 * L.plane_floats;
  return L;
}

// round-half-away(y) for 0 <= y <= 65535: v_cvt_rpi_i32_f32 = (int)floor(y + 0.5), exact (verified
// exhaustively on gfx950 over every float in [0,65536], tools/sem_probe.hip).
__device__ __forceinline__ uint32_t round_half_away_nonneg(float y) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(y));
  return (uint32_t)r;
}

// Stage one channel plane into LDS with the async global->LDS path (global_load_lds_dwordx4: each
// wave-instruction moves 1 KiB, LDS destination = wave-uniform base + lane*16, no VGPR round trip).
template <int NT>
__device__ __forceinline__ void stage_plane(unsigned char *lds, const float *__restrict__ plane, uint32_t plane_floats) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t chunks = plane_floats / 256;  // plane_floats is a multiple of 256 (1 KiB)
  const char *gsrc = (const char *)plane + lane * 16;
  for (uint32_t k = wave; k < chunks; k += NT / 64)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)k * 1024),
                                     (__attribute__((address_space(3))) void *)(lds + kAxisTableBytes + k * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One channel pass over the P pixels a lane holds. C = output channel (byte index).
template <int P, int C, int S_CONST>
__device__ __forceinline__ void lut_pass(const unsigned char *lds, int Sy_rt, int Sz_rt, uint32_t plane_off, uint32_t (&px)[P],
                                         const uint32_t (&base)[P], const float (&tx)[P], const float (&ty)[P], const float (&tz)[P]) {
  // strides are compile-time for the common 33^3 case so every corner is base + immediate
  const int Sy = S_CONST == 33 ? 35 : Sy_rt;
  const int Sz = S_CONST == 33 ? 1161 : Sz_rt;
  // output byte C <- value byte 0, the other three bytes kept (v_perm_b32 selector)
  constexpr uint32_t sel = C == 0 ? 0x07060500u : (C == 1 ? 0x07060004u : 0x07000504u);
#pragma unroll
  for (int i = 0; i < P; i++) {
    const float *L1 = (const float *)(lds + base[i] + plane_off);  // z0+1 layer (plane_off != 0 only when all planes are resident)
    const float *L0 = L1 + Sz;                         // z0 layer
    const float a0 = L0[0], a1 = L0[1], b0 = L0[Sy], b1 = L0[Sy + 1];
    const float c0 = L1[0], c1 = L1[1], d0 = L1[Sy], d1 = L1[Sy + 1];
    const float c00 = lerp1(a0, a1, tx[i]), c10 = lerp1(b0, b1, tx[i]);
    const float c01 = lerp1(c0, c1, tx[i]), c11 = lerp1(d0, d1, tx[i]);
    const float o = lerp1(lerp1(c00, c10, ty[i]), lerp1(c01, c11, ty[i]), tz[i]);
    // float_to_u8 (imp.rs:537-539); o is never NaN here so max-then-min == the inherent clamp
    const uint32_t v8 = round_half_away_nonneg(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
    px[i] = __builtin_amdgcn_perm(px[i], v8, sel);
  }
}

// HSV: kNoHsv = plain colorlut. Otherwise the kernel is the fused hsvfilter -> colorlut chain on RGBA: the pixels
// are run through the hsvfilter arithmetic (variant as in hsv_kernels.hip: -1 GENERIC, 0..6 FAST) in registers
// right after the load, so the chain costs one read and one write per pixel instead of two of each.
constexpr int kNoHsv = -2;
template <int NT, int P4, int S_CONST, int HSV>
__device__ __forceinline__ void colorlut3d_lds_body(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                    size_t n_groups, const float *__restrict__ planar,
                                                    const uint32_t *__restrict__ axis_tab, int Sy_rt, int Sz_rt,
                                                    uint32_t plane_floats, int all_resident_and_stagger, HsvK hk) {
  const int all_resident = all_resident_and_stagger & 1;
  const unsigned stagger = (unsigned)all_resident_and_stagger >> 8;  // x256 clock ticks: spread of the per-block start delay (0 = off)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __shared__ uint32_t hsv_sel[HSV == kNoHsv ? 1 : 8];
  if constexpr (HSV != kNoHsv) {
    if (threadIdx.x < 7)
      hsv_sel[threadIdx.x] = HSV >= 0 ? hsv_sel_entry_floor(threadIdx.x, 0, 1, 2, 3) : hsv_sel_entry(threadIdx.x, 0, 1, 2, 3);
  }
  constexpr int P = P4 * 4;
  const size_t tile_groups = (size_t)NT * P4;
  // Work split: `full_rounds` rounds of whole tiles (tile t -> block t % grid), then the remaining
  // groups are cut into gridDim.x equal chunks so the last round keeps every CU busy instead of
  // leaving most of them idle behind a few whole tiles.
  const size_t full_rounds = (n_groups / tile_groups) / gridDim.x;
  const size_t full_tiles = full_rounds * gridDim.x;
  const size_t rem_start = full_tiles * tile_groups;
  const size_t rem_chunk = (n_groups - rem_start + gridDim.x - 1) / gridDim.x;  // <= tile_groups (+ rounding), see launcher
  const size_t my_rounds = full_rounds + (rem_chunk > 0 ? 1 : 0);

  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  // Small LUTs (size <= 18): all three planes fit LDS together and are staged once per block.
  const uint32_t plane_bytes = all_resident ? plane_floats * 4u : 0u;
  if (all_resident) {
    for (int c = 0; c < 3; c++) stage_plane<NT>(lds + (size_t)c * plane_bytes, planar + (size_t)c * plane_floats, plane_floats);
  }
  __syncthreads();

  // Channel order alternates R,G,B / B,G,R from tile to tile so the plane left in LDS by one tile's
  // last pass serves the next tile's first pass: two stagings per tile instead of three.
  if (stagger) {
    // De-synchronise the CUs: every block runs the same load / stage / pass / store cycle, so without this all 256 CUs
    // hit HBM in the same microsecond and then leave it idle together. A per-block start offset (hashed, up to
    // `stagger` ticks, of the order of one tile period) persists because all tiles take the same time.
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long d = ((unsigned long long)(((blockIdx.x * 2654435769u) >> 16) & 0xffffu) * ((unsigned long long)stagger << 8)) >> 16;
    while (__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
  bool flip = false;
  int resident = all_resident ? 3 : -1;  // 3 = "every plane"
  for (size_t round = 0; round < my_rounds; round++) {
    size_t t_begin, t_end;
    if (round < full_rounds) {
      t_begin = (round * gridDim.x + blockIdx.x) * tile_groups;
      t_end = t_begin + tile_groups;
    } else {
      t_begin = rem_start + (size_t)blockIdx.x * rem_chunk;
      t_end = t_begin + rem_chunk;
      if (t_end > n_groups) t_end = n_groups;
      if (t_begin >= t_end) break;  // block-uniform: nothing left for this block
    }
    uint32_t px[P];
    uint32_t base[P];
    float tx[P], ty[P], tz[P];
    const size_t g0 = t_begin + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < t_end) v = src[g];
      px[4 * j + 0] = v.x; px[4 * j + 1] = v.y; px[4 * j + 2] = v.z; px[4 * j + 3] = v.w;
    }
    if constexpr (HSV >= 0) {
#pragma unroll
      for (int i = 0; i < P; i += 2)
        hsvfilter_px2_fast<0, 1, 2, 3, HSV & 3, (HSV >> 2) != 0>(px[i], px[i + 1], hk, hsv_sel);
    } else if constexpr (HSV == -1) {
#pragma unroll
      for (int i = 0; i < P; i++) px[i] = hsvfilter_px<false, 0, 1, 2, 3>(px[i], hk, hsv_sel);
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      base[i] = ex.x + ey.x + ez.x;
      tx[i] = __uint_as_float(ex.y);
      ty[i] = __uint_as_float(ey.y);
      tz[i] = __uint_as_float(ez.y);
    }
#define MI355_STAGE(CH)                                                              \
  if (resident != CH && resident != 3) {                                             \
    __syncthreads(); /* everyone is done reading the previous plane */               \
    stage_plane<NT>(lds, planar + (size_t)CH * plane_floats, plane_floats);          \
    __syncthreads();                                                                 \
    resident = CH;                                                                   \
  }
    if (!flip) {
      MI355_STAGE(0) lut_pass<P, 0, S_CONST>(lds, Sy_rt, Sz_rt, 0u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE(1) lut_pass<P, 1, S_CONST>(lds, Sy_rt, Sz_rt, 1u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE(2) lut_pass<P, 2, S_CONST>(lds, Sy_rt, Sz_rt, 2u * plane_bytes, px, base, tx, ty, tz);
    } else {
      MI355_STAGE(2) lut_pass<P, 2, S_CONST>(lds, Sy_rt, Sz_rt, 2u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE(1) lut_pass<P, 1, S_CONST>(lds, Sy_rt, Sz_rt, 1u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE(0) lut_pass<P, 0, S_CONST>(lds, Sy_rt, Sz_rt, 0u * plane_bytes, px, base, tx, ty, tz);
    }
#undef MI355_STAGE
    flip = !flip;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < t_end) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
  }
}

template <int NT, int P4, int S_CONST, int HSV>
__global__ __launch_bounds__(NT) void colorlut3d_lds_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                            size_t n_groups, const float *__restrict__ planar,
                                                            const uint32_t *__restrict__ axis_tab, int Sy_rt, int Sz_rt,
                                                            uint32_t plane_floats, int all_resident, HsvK hk) {
  colorlut3d_lds_body<NT, P4, S_CONST, HSV>(src, dst, n_groups, planar, axis_tab, Sy_rt, Sz_rt, plane_floats, all_resident, hk);
}

// ---------------------------------------------------------------- LDS three-pass kernel, minimal per-pixel state
// Variant of colorlut3d_lds_kernel that keeps only TWO registers per pixel (input pixel, output pixel) instead of five
// (pixel, LDS base, tx, ty, tz): every pass re-reads the three {offset, t} axis entries (one ds_read_b64 each) and
// rebuilds base = ox+oy+oz. That costs 3 extra LDS reads per pixel and pass, but a lane can hold 2-2.7x more pixels,
// so each plane staging (and each barrier / load / store phase) is amortised over that many more pixels.
template <int P, int C, int S_CONST>
__device__ __forceinline__ void lut_pass_lean(const unsigned char *lds, int Sy_rt, int Sz_rt, uint32_t plane_off, const uint32_t (&pin)[P],
                                              uint32_t (&pout)[P]) {
  const int Sy = S_CONST == 33 ? 35 : Sy_rt;
  const int Sz = S_CONST == 33 ? 1161 : Sz_rt;
  constexpr uint32_t sel = C == 0 ? 0x07060500u : (C == 1 ? 0x07060004u : 0x07000504u);
#pragma unroll
  for (int i = 0; i < P; i++) {
    uint32_t pv = pin[i];
    asm volatile("" : "+v"(pv));  // opaque copy: keeps the byte-offset arithmetic from being computed once and kept live across the three passes
    const uint2 ex = *(const uint2 *)(lds + ((pv & 0xffu) << 3));
    const uint2 ey = *(const uint2 *)(lds + 2048 + (((pv >> 8) & 0xffu) << 3));
    const uint2 ez = *(const uint2 *)(lds + 4096 + (((pv >> 16) & 0xffu) << 3));
    const float tx = __uint_as_float(ex.y), ty = __uint_as_float(ey.y), tz = __uint_as_float(ez.y);
    const float *L1 = (const float *)(lds + (ex.x + ey.x + ez.x) + plane_off);
    const float *L0 = L1 + Sz;
    const float a0 = L0[0], a1 = L0[1], b0 = L0[Sy], b1 = L0[Sy + 1];
    const float c0 = L1[0], c1 = L1[1], d0 = L1[Sy], d1 = L1[Sy + 1];
    const float c00 = lerp1(a0, a1, tx), c10 = lerp1(b0, b1, tx);
    const float c01 = lerp1(c0, c1, tx), c11 = lerp1(d0, d1, tx);
    const float o = lerp1(lerp1(c00, c10, ty), lerp1(c01, c11, ty), tz);
    const uint32_t v8 = round_half_away_nonneg(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
    pout[i] = __builtin_amdgcn_perm(pout[i], v8, sel);
    // keep the scheduler from hoisting every pixel's axis reads to the front (that is what the five-register
    // variant stores): pixels are processed in groups of four
    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT, int P4, int S_CONST>
__global__ __launch_bounds__(NT) void colorlut3d_lean_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                                             const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab, int Sy_rt,
                                                             int Sz_rt, uint32_t plane_floats, int all_resident) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = P4 * 4;
  const size_t tile_groups = (size_t)NT * P4;
  const size_t full_rounds = (n_groups / tile_groups) / gridDim.x;
  const size_t full_tiles = full_rounds * gridDim.x;
  const size_t rem_start = full_tiles * tile_groups;
  const size_t rem_chunk = (n_groups - rem_start + gridDim.x - 1) / gridDim.x;
  const size_t my_rounds = full_rounds + (rem_chunk > 0 ? 1 : 0);
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  const uint32_t plane_bytes = all_resident ? plane_floats * 4u : 0u;
  if (all_resident) {
    for (int c = 0; c < 3; c++) stage_plane<NT>(lds + (size_t)c * plane_bytes, planar + (size_t)c * plane_floats, plane_floats);
  }
  __syncthreads();
  bool flip = false;
  int resident = all_resident ? 3 : -1;
  for (size_t round = 0; round < my_rounds; round++) {
    size_t t_begin, t_end;
    if (round < full_rounds) {
      t_begin = (round * gridDim.x + blockIdx.x) * tile_groups;
      t_end = t_begin + tile_groups;
    } else {
      t_begin = rem_start + (size_t)blockIdx.x * rem_chunk;
      t_end = t_begin + rem_chunk;
      if (t_end > n_groups) t_end = n_groups;
      if (t_begin >= t_end) break;
    }
    uint32_t pin[P], pout[P];
    const size_t g0 = t_begin + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < t_end) v = src[g];
      pin[4 * j + 0] = v.x; pin[4 * j + 1] = v.y; pin[4 * j + 2] = v.z; pin[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < P; i++) pout[i] = pin[i];
#define MI355_STAGE(CH)                                                              \
  if (resident != CH && resident != 3) {                                             \
    __syncthreads();                                                                 \
    stage_plane<NT>(lds, planar + (size_t)CH * plane_floats, plane_floats);          \
    __syncthreads();                                                                 \
    resident = CH;                                                                   \
  }
    if (!flip) {
      MI355_STAGE(0) lut_pass_lean<P, 0, S_CONST>(lds, Sy_rt, Sz_rt, 0u * plane_bytes, pin, pout);
      MI355_STAGE(1) lut_pass_lean<P, 1, S_CONST>(lds, Sy_rt, Sz_rt, 1u * plane_bytes, pin, pout);
      MI355_STAGE(2) lut_pass_lean<P, 2, S_CONST>(lds, Sy_rt, Sz_rt, 2u * plane_bytes, pin, pout);
    } else {
      MI355_STAGE(2) lut_pass_lean<P, 2, S_CONST>(lds, Sy_rt, Sz_rt, 2u * plane_bytes, pin, pout);
      MI355_STAGE(1) lut_pass_lean<P, 1, S_CONST>(lds, Sy_rt, Sz_rt, 1u * plane_bytes, pin, pout);
      MI355_STAGE(0) lut_pass_lean<P, 0, S_CONST>(lds, Sy_rt, Sz_rt, 0u * plane_bytes, pin, pout);
    }
#undef MI355_STAGE
    flip = !flip;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < t_end) dst[g] = make_uint4(pout[4 * j + 0], pout[4 * j + 1], pout[4 * j + 2], pout[4 * j + 3]);
    }
  }
}

// ---------------------------------------------------------------- fused hsvfilter -> colorlut, software pipelined
// Same three-pass structure, but the hsvfilter arithmetic of tile n+1 (pure VALU, ~100 instructions per pixel)
// runs inside the two plane-staging windows of tile n, where the CU otherwise only waits for the L2->LDS DMA:
//   setup(n) | issue loads(n+1) | pass A | [DMA B || hsv(first half of n+1)] | pass B | [DMA C || hsv(second half)] | pass C | store(n)
// Vector-memory results return in issue order, so the loads of tile n+1 (issued before DMA B) have landed once
// DMA B's own counter position is reached: for the 33^3 layout every wave issues exactly kDma33 DMA
// instructions (compile-time), which lets the compiler place `s_waitcnt vmcnt(kDma33)` in front of the hsv code
// instead of a full drain.
constexpr int kPlaneChunks33 = (33 * 1161 * 4 + 1023) / 1024;  // 150 KiB-chunks of the padded 33^3 plane

template <int NT, int S_CONST>
__device__ __forceinline__ void stage_plane_issue(unsigned char *lds, const float *__restrict__ plane, uint32_t plane_floats) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const char *gsrc = (const char *)plane + lane * 16;
  if constexpr (S_CONST == 33) {
    constexpr uint32_t per_wave = (kPlaneChunks33 + NT / 64 - 1) / (NT / 64);
#pragma unroll
    for (uint32_t i = 0; i < per_wave; i++) {
      uint32_t k = wave + i * (NT / 64);
      k = k < (uint32_t)kPlaneChunks33 ? k : (uint32_t)kPlaneChunks33 - 1;  // surplus slots re-copy the last chunk (same bytes)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)k * 1024),
                                       (__attribute__((address_space(3))) void *)(lds + kAxisTableBytes + k * 1024), 16, 0, 0);
    }
  } else {
    const uint32_t chunks = plane_floats / 256;
    for (uint32_t k = wave; k < chunks; k += NT / 64)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)k * 1024),
                                       (__attribute__((address_space(3))) void *)(lds + kAxisTableBytes + k * 1024), 16, 0, 0);
  }
}

template <int HSV, int I0, int I1, int P>
__device__ __forceinline__ void hsv_range(uint32_t (&nx)[P], const HsvK &hk, const uint32_t *hsv_sel) {
  if constexpr (HSV == kNoHsv) {
    // plain colorlut: nothing to do between load and setup
  } else if constexpr (HSV >= 0) {
#pragma unroll
    for (int i = I0; i < I1; i += 2) hsvfilter_px2_fast<0, 1, 2, 3, HSV & 3, (HSV >> 2) != 0>(nx[i], nx[i + 1], hk, hsv_sel);
  } else {
#pragma unroll
    for (int i = I0; i < I1; i++) nx[i] = hsvfilter_px<false, 0, 1, 2, 3>(nx[i], hk, hsv_sel);
  }
}

// LATE_LOAD: issue the next tile's loads before the LAST pass of the tile instead of before the first one. The CU's
// vector-memory pipe returns in order, so loads issued ahead of a plane DMA hold that DMA back; issued after the
// tile's last DMA they only have the stores behind them and fly during the last pass and the next tile's first one.
// Used for plain colorlut (HSV == kNoHsv), where nothing has to happen to the pixels between load and setup.
template <int NT, int P4, int S_CONST, int HSV, bool LATE_LOAD = false>
__global__ __launch_bounds__(NT) void hsv_colorlut3d_pipe_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                                 size_t n_groups, const float *__restrict__ planar,
                                                                 const uint32_t *__restrict__ axis_tab, int Sy_rt, int Sz_rt,
                                                                 uint32_t plane_floats, int all_resident, HsvK hk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __shared__ uint32_t hsv_sel[8];
  constexpr int P = P4 * 4;
  static_assert(P % 4 == 0, "pixel pairs are split in two halves");
  constexpr int PH = (P / 2 + 1) & ~1;  // first half, even
  const size_t tile_groups = (size_t)NT * P4;
  const size_t full_rounds = (n_groups / tile_groups) / gridDim.x;
  const size_t full_tiles = full_rounds * gridDim.x;
  const size_t rem_start = full_tiles * tile_groups;
  const size_t rem_chunk = (n_groups - rem_start + gridDim.x - 1) / gridDim.x;
  const size_t my_rounds = full_rounds + (rem_chunk > 0 ? 1 : 0);
  // [t_begin, t_end) of this block's tile in `round`; false when there is none (block-uniform)
  auto tile_range = [&](size_t round, size_t &t_begin, size_t &t_end) -> bool {
    if (round >= my_rounds) return false;
    if (round < full_rounds) {
      t_begin = (round * gridDim.x + blockIdx.x) * tile_groups;
      t_end = t_begin + tile_groups;
      return true;
    }
    t_begin = rem_start + (size_t)blockIdx.x * rem_chunk;
    t_end = t_begin + rem_chunk;
    if (t_end > n_groups) t_end = n_groups;
    return t_begin < t_end;
  };

  if (threadIdx.x < 7)
    hsv_sel[threadIdx.x] = HSV >= 0 ? hsv_sel_entry_floor(threadIdx.x, 0, 1, 2, 3) : hsv_sel_entry(threadIdx.x, 0, 1, 2, 3);
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  const uint32_t plane_bytes = all_resident ? plane_floats * 4u : 0u;
  if (all_resident) {
    for (int c = 0; c < 3; c++) stage_plane<NT>(lds + (size_t)c * plane_bytes, planar + (size_t)c * plane_floats, plane_floats);
  }
  __syncthreads();

  uint32_t nx[P];
  auto load_tile = [&](size_t t_begin, size_t t_end) {
    const size_t g0 = t_begin + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < t_end) v = src[g];
      nx[4 * j + 0] = v.x; nx[4 * j + 1] = v.y; nx[4 * j + 2] = v.z; nx[4 * j + 3] = v.w;
    }
  };

  size_t t_begin = 0, t_end = 0;
  bool have = tile_range(0, t_begin, t_end);
  if (have) {
    load_tile(t_begin, t_end);
    hsv_range<HSV, 0, P>(nx, hk, hsv_sel);
  }
  bool flip = false;
  int resident = all_resident ? 3 : -1;
  for (size_t round = 0; have; round++) {
    uint32_t px[P], base[P];
    float tx[P], ty[P], tz[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
      px[i] = nx[i];
      const uint2 ex = *(const uint2 *)(lds + ((px[i] & 0xffu) << 3));
      const uint2 ey = *(const uint2 *)(lds + 2048 + (((px[i] >> 8) & 0xffu) << 3));
      const uint2 ez = *(const uint2 *)(lds + 4096 + (((px[i] >> 16) & 0xffu) << 3));
      base[i] = ex.x + ey.x + ez.x;
      tx[i] = __uint_as_float(ex.y);
      ty[i] = __uint_as_float(ey.y);
      tz[i] = __uint_as_float(ez.y);
    }
    size_t n_begin = 0, n_end = 0;
    const bool have_next = tile_range(round + 1, n_begin, n_end);
    if (!LATE_LOAD && have_next) load_tile(n_begin, n_end);
    // stage plane CH if needed, running the hsv stage of pixels [I0, I1) of the next tile under the DMA
#define MI355_STAGE_HSV(CH, I0, I1)                                                             \
  {                                                                                             \
    const bool need = resident != CH && resident != 3;                                          \
    if (need) {                                                                                 \
      __syncthreads(); /* everyone is done reading the previous plane */                        \
      stage_plane_issue<NT, S_CONST>(lds, planar + (size_t)CH * plane_floats, plane_floats);    \
    }                                                                                           \
    if (!LATE_LOAD && have_next) hsv_range<HSV, I0, I1>(nx, hk, hsv_sel);                       \
    if (need) {                                                                                 \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
      __syncthreads();                                                                          \
      resident = CH;                                                                            \
    }                                                                                           \
    if (LATE_LOAD && (I1) == P && have_next) load_tile(n_begin, n_end); /* after the tile's last DMA */ \
  }
    if (!flip) {
      MI355_STAGE_HSV(0, 0, 0) lut_pass<P, 0, S_CONST>(lds, Sy_rt, Sz_rt, 0u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE_HSV(1, 0, PH) lut_pass<P, 1, S_CONST>(lds, Sy_rt, Sz_rt, 1u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE_HSV(2, PH, P) lut_pass<P, 2, S_CONST>(lds, Sy_rt, Sz_rt, 2u * plane_bytes, px, base, tx, ty, tz);
    } else {
      MI355_STAGE_HSV(2, 0, 0) lut_pass<P, 2, S_CONST>(lds, Sy_rt, Sz_rt, 2u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE_HSV(1, 0, PH) lut_pass<P, 1, S_CONST>(lds, Sy_rt, Sz_rt, 1u * plane_bytes, px, base, tx, ty, tz);
      MI355_STAGE_HSV(0, PH, P) lut_pass<P, 0, S_CONST>(lds, Sy_rt, Sz_rt, 0u * plane_bytes, px, base, tx, ty, tz);
    }
#undef MI355_STAGE_HSV
    flip = !flip;
    const size_t g0 = t_begin + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < t_end) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
    t_begin = n_begin; t_end = n_end; have = have_next;
  }
}

// ---------------------------------------------------------------- LDS three-pass kernel, RGBA64 (3D)
//
// transform_rgba64_3d::<LE> (imp.rs:348-397): same plane layout, staging and pass structure as the
// RGBA8 kernel; a pixel is two dwords (r16|g16, b16|a16, byte order per LE), the coordinates are
// computed with VALU (65536 input levels do not fit an LDS axis table): norm_comp_u16 = v/65535 via the
// exact two-operation quotient (div65535_u16), scale/offset, clamp, *(S-1), floor/fract.
// Output: float_to_u16 = round-half-away(clamp(v)*65535) (v_cvt_rpi, exact on [0,65536]).
struct Lut64K {
  float scale[3], offset[3];
  float sm1;
  uint32_t sy4, sz4;  // byte strides 4*Sy, 4*Sz
  uint32_t plane_base;  // kAxisTableBytes
  int S;
};

__device__ __forceinline__ uint32_t swap16(uint32_t v) { return ((v >> 8) & 0xffu) | ((v & 0xffu) << 8); }

template <int NT, int P2, bool LE>
__global__ __launch_bounds__(NT) void colorlut3d_lds64_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                                              const float *__restrict__ planar, const uint32_t *__restrict__ axis_tab,
                                                              uint32_t plane_floats, Lut64K k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int P = P2 * 2;  // pixels per lane (one uint4 = 2 pixels)
  const size_t tile_groups = (size_t)NT * P2;
  const size_t full_rounds = (n_groups / tile_groups) / gridDim.x;
  const size_t rem_start = full_rounds * gridDim.x * tile_groups;
  const size_t rem_chunk = (n_groups - rem_start + gridDim.x - 1) / gridDim.x;
  const size_t my_rounds = full_rounds + (rem_chunk > 0 ? 1 : 0);
  // the axis-table region is staged too: the z0 == S-1 pad reads land there (finite values)
  for (int i = threadIdx.x; i < kAxisTableBytes / 4; i += NT) ((uint32_t *)lds)[i] = axis_tab[i];
  __syncthreads();
  const int Sy = (int)(k.sy4 / 4), Sz = (int)(k.sz4 / 4);
  bool flip = false;
  int resident = -1;
  for (size_t round = 0; round < my_rounds; round++) {
    size_t t_begin, t_end;
    if (round < full_rounds) {
      t_begin = (round * gridDim.x + blockIdx.x) * tile_groups;
      t_end = t_begin + tile_groups;
    } else {
      t_begin = rem_start + (size_t)blockIdx.x * rem_chunk;
      t_end = t_begin + rem_chunk;
      if (t_end > n_groups) t_end = n_groups;
      if (t_begin >= t_end) break;
    }
    uint32_t lo[P], hi[P], base[P];
    float tx[P], ty[P], tz[P];
    const size_t g0 = t_begin + threadIdx.x;
#pragma unroll
    for (int j = 0; j < P2; j++) {
      const size_t g = g0 + (size_t)j * NT;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g < t_end) v = src[g];
      lo[2 * j] = v.x; hi[2 * j] = v.y; lo[2 * j + 1] = v.z; hi[2 * j + 1] = v.w;
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
      uint32_t c16[3] = {lo[i] & 0xffffu, lo[i] >> 16, hi[i] & 0xffffu};
      uint32_t idx[3];
      float t[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const uint32_t v = LE ? c16[a] : swap16(c16[a]);
        float n = div65535_u16((float)v);
        n = fminf(fmaxf(n * k.scale[a] + k.offset[a], 0.0f), 1.0f);  // finite domain: == inherent clamp
        const float x = n * k.sm1;
        idx[a] = (uint32_t)x;
        t[a] = __builtin_amdgcn_fractf(x);
      }
      base[i] = k.plane_base + 4u * idx[0] + k.sy4 * idx[1] + k.sz4 * (uint32_t)(k.S - 2 - (int)idx[2]);
      tx[i] = t[0]; ty[i] = t[1]; tz[i] = t[2];
    }
#define MI355_STAGE64(CH)                                                            \
  if (resident != CH) {                                                              \
    __syncthreads();                                                                 \
    stage_plane<NT>(lds, planar + (size_t)CH * plane_floats, plane_floats);          \
    __syncthreads();                                                                 \
    resident = CH;                                                                   \
  }
#define MI355_PASS64(CH)                                                                                     \
  _Pragma("unroll") for (int i = 0; i < P; i++) {                                                            \
    const float *L1 = (const float *)(lds + base[i]);                                                        \
    const float *L0 = L1 + Sz;                                                                               \
    const float a0 = L0[0], a1 = L0[1], b0 = L0[Sy], b1 = L0[Sy + 1];                                        \
    const float c0 = L1[0], c1 = L1[1], d0 = L1[Sy], d1 = L1[Sy + 1];                                        \
    const float c00 = lerp1(a0, a1, tx[i]), c10 = lerp1(b0, b1, tx[i]);                                      \
    const float c01 = lerp1(c0, c1, tx[i]), c11 = lerp1(d0, d1, tx[i]);                                      \
    const float o = lerp1(lerp1(c00, c10, ty[i]), lerp1(c01, c11, ty[i]), tz[i]);                            \
    uint32_t v16 = round_half_away_nonneg(fminf(fmaxf(o, 0.0f), 1.0f) * 65535.0f);                           \
    if (!LE) v16 = swap16(v16);                                                                              \
    if (CH == 0) lo[i] = (lo[i] & 0xffff0000u) | v16;                                                        \
    else if (CH == 1) lo[i] = (lo[i] & 0x0000ffffu) | (v16 << 16);                                           \
    else hi[i] = (hi[i] & 0xffff0000u) | v16;                                                                \
  }
    if (!flip) {
      MI355_STAGE64(0) MI355_PASS64(0)
      MI355_STAGE64(1) MI355_PASS64(1)
      MI355_STAGE64(2) MI355_PASS64(2)
    } else {
      MI355_STAGE64(2) MI355_PASS64(2)
      MI355_STAGE64(1) MI355_PASS64(1)
      MI355_STAGE64(0) MI355_PASS64(0)
    }
#undef MI355_STAGE64
#undef MI355_PASS64
    flip = !flip;
#pragma unroll
    for (int j = 0; j < P2; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < t_end) dst[g] = make_uint4(lo[2 * j], hi[2 * j], lo[2 * j + 1], hi[2 * j + 1]);
    }
  }
}

// ---------------------------------------------------------------- 1D LUT, RGBA8, tables in LDS
//
// transform_rgba_1d / apply_1d / sample_1d (imp.rs:237-265,399-414,482-490). LDS image: the three
// axis tables {byte offset of lut_c[x0], t} (same construction as the 3D kernel) followed by the three
// channel tables, each `size`+1 floats (one zero pad so x1 = x0+1 needs no clamp: when the reference
// clamps, t == 0). One lane = 4 pixels per iteration, 8 B/pixel of HBM traffic, everything else in LDS.
template <int NT>
__global__ __launch_bounds__(NT) void colorlut1d_lds_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_vec,
                                                            const uint32_t *__restrict__ lds_image, uint32_t image_dwords) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (uint32_t i = threadIdx.x; i < image_dwords; i += NT) ((uint32_t *)lds)[i] = lds_image[i];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * NT;
  for (size_t g = (size_t)blockIdx.x * NT + threadIdx.x; g < n_vec; g += stride) {
    const uint4 v = src[g];
    uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t out = px[i];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const uint2 e = *(const uint2 *)(lds + c * 2048 + (((px[i] >> (8 * c)) & 0xffu) << 3));
        const float *L = (const float *)(lds + e.x);
        const float o = lerp1(L[0], L[1], __uint_as_float(e.y));
        const uint32_t v8 = round_half_away_nonneg(fminf(fmaxf(o, 0.0f), 1.0f) * 255.0f);
        const uint32_t sel = c == 0 ? 0x07060500u : (c == 1 ? 0x07060004u : 0x07000504u);
        out = __builtin_amdgcn_perm(out, v8, sel);
      }
      px[i] = out;
    }
    dst[g] = make_uint4(px[0], px[1], px[2], px[3]);
  }
}

// ---------------------------------------------------------------- 1D LUT, RGBA64, tables in LDS
//
// transform_rgba64_1d::<LE> / apply_1d_u16 / sample_1d (imp.rs:296-346,416-430,482-490) from the LDS image of the RGBA8
// kernel (its three channel tables with the zero pad; the byte-indexed axis tables are not used: 65,536 input levels are
// computed with the expressions of colorlut3d_lds64_kernel - div65535_u16, scale / offset, clamp, * (size - 1), integer part
// and fraction). One lane = 2 pixels per 16-byte group; 16 B/pixel of HBM traffic, two 4-byte LDS reads per channel.
struct Lut1d64K { float scale[3], offset[3]; float sm1; uint32_t table_bytes; };   // table_bytes = (size + 1) * 4

template <int NT, bool LE>
__global__ __launch_bounds__(NT) void colorlut1d_lds64_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_groups,
                                                              const uint32_t *__restrict__ lds_image, uint32_t image_dwords, Lut1d64K k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (uint32_t i = threadIdx.x; i < image_dwords; i += NT) ((uint32_t *)lds)[i] = lds_image[i];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * NT;
  for (size_t g = (size_t)blockIdx.x * NT + threadIdx.x; g < n_groups; g += stride) {
    const uint4 v = src[g];
    uint32_t lo[2] = {v.x, v.z}, hi[2] = {v.y, v.w};
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint32_t c16[3] = {lo[i] & 0xffffu, lo[i] >> 16, hi[i] & 0xffffu};
      uint32_t o16[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const uint32_t in = LE ? c16[a] : swap16(c16[a]);
        float n = div65535_u16((float)in);
        n = fminf(fmaxf(n * k.scale[a] + k.offset[a], 0.0f), 1.0f);   // finite domain: == the inherent clamp
        const float x = n * k.sm1;
        const uint32_t idx = (uint32_t)x;
        const float t = __builtin_amdgcn_fractf(x);
        const float *L = (const float *)(lds + kAxisTableBytes + (uint32_t)a * k.table_bytes + 4u * idx);
        const float o = lerp1(L[0], L[1], t);   // idx == size - 1: t == 0 and the pad is never weighted
        const uint32_t v16 = round_half_away_nonneg(fminf(fmaxf(o, 0.0f), 1.0f) * 65535.0f);
        o16[a] = LE ? v16 : swap16(v16);
      }
      lo[i] = o16[0] | (o16[1] << 16);
      hi[i] = (hi[i] & 0xffff0000u) | o16[2];
    }
    dst[g] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
  }
}

// ---------------------------------------------------------------- host side: LUT upload + dispatch

static constexpr size_t kLdsBytes = 160 * 1024;

// ---------------------------------------------------------------- full-domain table ("memoised") kernel, RGBA8
// An 8-bit RGB -> 8-bit RGB per-pixel function has only 2^24 inputs. The exact LUT kernel is run ONCE over all of them
// when the table is requested (16.7 M pixels = two 4K frames' worth of work) and its outputs are kept as a 64 MiB table
// in HBM / Infinity Cache; a frame is then one 4-byte gather per pixel plus the alpha merge. Bit-identical to the
// kernel that built the table by construction, for every LUT kind and size. The table index is either the colour itself
// (linear: consecutive R share a cache line) or its 3-way bit interleave (Morton: a 128 B line holds a 4x4x2 colour
// cube, so pixels that differ a little in every channel still share lines).
constexpr uint32_t kTableEntries = 1u << 24;

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // bit k of an 8-bit value -> bit 3k
  v &= 0x000000ffu;
  v = (v | (v << 8)) & 0x0000f00fu;
  v = (v | (v << 4)) & 0x000c30c3u;
  v = (v | (v << 2)) & 0x00249249u;
  return v;
}
__device__ __forceinline__ uint32_t compact3(uint32_t v) {  // inverse of spread3: bit 3k -> bit k
  v &= 0x00249249u;
  v = (v | (v >> 2)) & 0x000c30c3u;
  v = (v | (v >> 4)) & 0x0000f00fu;
  v = (v | (v >> 8)) & 0x000000ffu;
  return v;
}

// in[i] = the colour whose table slot is i (alpha 0xff)
__global__ __launch_bounds__(256) void table_domain_kernel(uint32_t *__restrict__ in, int morton) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  uint32_t c = i;
  if (morton) c = compact3(i) | (compact3(i >> 1) << 8) | (compact3(i >> 2) << 16);
  in[i] = c | 0xff000000u;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// A wave handles 512 consecutive pixels per iteration: two 16-byte loads per lane (1 KiB contiguous per instruction,
// non-temporal so that the pixel stream does not evict table lines from L2), a transpose through a wave-private 2 KiB LDS
// strip so that gather j of lane l serves pixel 64 j + l - neighbouring lanes then look up neighbouring pixels, whose
// colours tend to share table lines (0.187 -> 0.160 ms per 8x4K against each lane gathering its own four pixels) - eight
// gathers in flight per lane, and the way back through the strip to two 16-byte stores. The time does not react to the
// grid (4..32 blocks per CU), to the block size (256..1024), or to loading the next chunk early: 8x4K natural-like
// frames take 0.155-0.165 ms = 0.41 Tpixel/s = ~0.7 gathered lanes per clock and CU whatever the latency hiding, i.e. the
// texture-address path works through a divergent dword gather at about one lane per clock (uniform noise, where every
// lane also misses L1 and L2: 1.19 ms).
template <bool MORTON>
__global__ __launch_bounds__(256) void colorlut_table_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_vec,
                                                             const uint32_t *__restrict__ table) {
  __shared__ uint32_t s_spread[256];
  __shared__ uint32_t s_strip[4][512];
  if (MORTON) {
    s_spread[threadIdx.x] = spread3(threadIdx.x);
    __syncthreads();
  }
  auto index = [&](uint32_t p) -> uint32_t {
    if (MORTON) return s_spread[p & 0xffu] | (s_spread[(p >> 8) & 0xffu] << 1) | (s_spread[(p >> 16) & 0xffu] << 2);
    return p & 0x00ffffffu;
  };
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t *x = s_strip[wave];
  const u32x4_t *s4 = (const u32x4_t *)src;
  u32x4_t *d4 = (u32x4_t *)dst;
  const size_t n_chunks = n_vec / 128;
  const size_t wstride = (size_t)gridDim.x * 4;
  auto wave_sync = [] {  // LDS writes of this wave visible to its other lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  size_t c = (size_t)blockIdx.x * 4 + wave;
  u32x4_t p = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (c < n_chunks) {
    p = __builtin_nontemporal_load(s4 + c * 128 + lane);
    q = __builtin_nontemporal_load(s4 + c * 128 + 64 + lane);
  }
  for (; c < n_chunks; c += wstride) {
    *(u32x4_t *)(x + lane * 4) = p;
    *(u32x4_t *)(x + 256 + lane * 4) = q;
    // the next chunk's pixels travel while this chunk's gathers do
    const size_t cn = c + wstride < n_chunks ? c + wstride : c;
    p = __builtin_nontemporal_load(s4 + cn * 128 + lane);
    q = __builtin_nontemporal_load(s4 + cn * 128 + 64 + lane);
    wave_sync();
    uint32_t px[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) px[j] = x[j * 64 + lane];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = table[index(px[j])];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j * 64 + lane] = (o[j] & 0x00ffffffu) | (px[j] & 0xff000000u);
    wave_sync();
    const u32x4_t r0 = *(u32x4_t *)(x + lane * 4), r1 = *(u32x4_t *)(x + 256 + lane * 4);
    wave_sync();
    __builtin_nontemporal_store(r0, d4 + c * 128 + lane);
    __builtin_nontemporal_store(r1, d4 + c * 128 + 64 + lane);
  }
  // fewer than 128 pixel groups left over: the first block's lanes take one group each, twice
  if (blockIdx.x == 0 && threadIdx.x < 128) {
    const size_t i = n_chunks * 128 + threadIdx.x;
    if (i < n_vec) {
      const u32x4_t p = s4[i];
      const uint32_t pp[4] = {p.x, p.y, p.z, p.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; j++) o[j] = (table[index(pp[j])] & 0x00ffffffu) | (pp[j] & 0xff000000u);
      const u32x4_t r = {o[0], o[1], o[2], o[3]};
      d4[i] = r;
    }
  }
}

// Tiled form for frames whose rows are whole pixel groups (width % 4 == 0): a wave's 512 pixels are a 256 x 2 (or 128 x 4)
// patch of the picture instead of 512 consecutive pixels of one row, and the four waves of a block take four patches
// stacked vertically (256 x 8). Vertically adjacent pixels have (almost) the same colours, so the table lines a block
// touches are reused more often before they fall out of the 32 KB L1: 0.163 -> 0.132-0.137 ms per 8x4K with a persistent
// grid (128 x 4: 0.140, 64 x 8: 0.143, 32 x 16: 0.157 - narrow patches cut the loads into short segments; blocks of
// 512 / 1024 lanes: 0.143 / 0.149), 0.117 ms with one tile per block. What costs time in this kernel is not the divergent gather
// itself (64 random colours, all L1 hits: 0.119 ms per 8x4K, next to 0.093 ms for a single colour) but L1 misses: each
// moves a 128 B line from L2 for 4 useful bytes (1024..4096 random colours, every gather an L2 hit: 0.28-0.31 ms =
// 30 TB/s of line traffic = the L2's bandwidth). tools/table_gather_probe.py.
// The frames are contiguous, so the batch is one picture of `rows` = n_frames * height rows of w4 pixel groups.
template <bool MORTON, int TW4>  // TW4 = patch width in pixel groups (64 = 256 px); a wave's patch is TW4*4 px x 128/TW4 rows
__global__ __launch_bounds__(256) void colorlut_table_tiled_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, unsigned w4, unsigned sw4, unsigned dw4,
                                                                   unsigned rows, unsigned n_cols, const uint32_t *__restrict__ table) {
  __shared__ uint32_t s_spread[256];
  __shared__ uint32_t s_strip[4][512];
  if (MORTON) {
    s_spread[threadIdx.x] = spread3(threadIdx.x);
    __syncthreads();
  }
  auto index = [&](uint32_t p) -> uint32_t {
    if (MORTON) return s_spread[p & 0xffu] | (s_spread[(p >> 8) & 0xffu] << 1) | (s_spread[(p >> 16) & 0xffu] << 2);
    return p & 0x00ffffffu;
  };
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t *x = s_strip[wave];
  const u32x4_t *s4 = (const u32x4_t *)src;
  u32x4_t *d4 = (u32x4_t *)dst;
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  constexpr unsigned RPL = 64 / TW4;            // rows covered by one 64-lane load
  const unsigned sub = lane / TW4, g = lane % TW4;
  // ONE tile per block, tiles in row-major order: the blocks resident at any time then work on neighbouring tiles, so
  // the table lines in flight across the chip belong to one band of the picture (a persistent grid whose blocks stride
  // over the tile list spreads them over the whole batch: 0.133 ms per 8x4K with 16 blocks per CU, 0.128 / 0.123 with
  // 32 / 64, 0.117 with one tile per block; column-major or banded tile orders: 0.120-0.124)
  const unsigned t = blockIdx.x;
  const unsigned tx = t % n_cols, ty = t / n_cols;
  const unsigned col = tx * TW4 + g;
  const unsigned r0 = ty * (8 * RPL) + wave * (2 * RPL) + sub, r1 = r0 + RPL;
  const bool ok0 = col < w4 && r0 < rows, ok1 = col < w4 && r1 < rows;
  // sw4 / dw4: row strides in 16-byte groups (== w4 for packed rows; larger when the rows are padded)
  const size_t i0 = (size_t)r0 * sw4 + col, i1 = (size_t)r1 * sw4 + col, o0 = (size_t)r0 * dw4 + col, o1 = (size_t)r1 * dw4 + col;
  u32x4_t p = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (ok0) p = __builtin_nontemporal_load(s4 + i0);
  if (ok1) q = __builtin_nontemporal_load(s4 + i1);
  *(u32x4_t *)(x + lane * 4) = p;
  *(u32x4_t *)(x + 256 + lane * 4) = q;
  wave_sync();
  uint32_t px[8], o[8];
#pragma unroll
  for (int j = 0; j < 8; j++) px[j] = x[j * 64 + lane];
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = table[index(px[j])];
#pragma unroll
  for (int j = 0; j < 8; j++) x[j * 64 + lane] = (o[j] & 0x00ffffffu) | (px[j] & 0xff000000u);
  wave_sync();
  const u32x4_t r0v = *(u32x4_t *)(x + lane * 4), r1v = *(u32x4_t *)(x + 256 + lane * 4);
  if (ok0) __builtin_nontemporal_store(r0v, d4 + o0);
  if (ok1) __builtin_nontemporal_store(r1v, d4 + o1);
}

// The tiled kernel over up to kMultiFrames SEPARATE packed frames of one size: tile t belongs to frame t / tiles_per_frame. The
// frames of different streams batched into one launch by the group dispatcher (group.hip). Morton table only.
template <int TW4>
__global__ __launch_bounds__(256) void colorlut_table_tiled_multi_kernel(MultiFramePtrs srcs, MultiFramePtrs dsts, unsigned w4, unsigned rows, unsigned n_cols,
                                                                         unsigned tiles_per_frame, const uint32_t *__restrict__ table) {
  __shared__ uint32_t s_spread[256];
  __shared__ uint32_t s_strip[4][512];
  s_spread[threadIdx.x] = spread3(threadIdx.x);
  __syncthreads();
  auto index = [&](uint32_t p) -> uint32_t { return s_spread[p & 0xffu] | (s_spread[(p >> 8) & 0xffu] << 1) | (s_spread[(p >> 16) & 0xffu] << 2); };
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t *x = s_strip[wave];
  const unsigned frame = blockIdx.x / tiles_per_frame, t = blockIdx.x - frame * tiles_per_frame;
  const u32x4_t *s4 = (const u32x4_t *)srcs.p[frame];
  u32x4_t *d4 = (u32x4_t *)dsts.p[frame];
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  constexpr unsigned RPL = 64 / TW4;
  const unsigned sub = lane / TW4, g = lane % TW4;
  const unsigned tx = t % n_cols, ty = t / n_cols;
  const unsigned col = tx * TW4 + g;
  const unsigned r0 = ty * (8 * RPL) + wave * (2 * RPL) + sub, r1 = r0 + RPL;
  const bool ok0 = col < w4 && r0 < rows, ok1 = col < w4 && r1 < rows;
  const size_t i0 = (size_t)r0 * w4 + col, i1 = (size_t)r1 * w4 + col;
  u32x4_t p = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  if (ok0) p = __builtin_nontemporal_load(s4 + i0);
  if (ok1) q = __builtin_nontemporal_load(s4 + i1);
  *(u32x4_t *)(x + lane * 4) = p;
  *(u32x4_t *)(x + 256 + lane * 4) = q;
  wave_sync();
  uint32_t px[8], o[8];
#pragma unroll
  for (int j = 0; j < 8; j++) px[j] = x[j * 64 + lane];
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = table[index(px[j])];
#pragma unroll
  for (int j = 0; j < 8; j++) x[j * 64 + lane] = (o[j] & 0x00ffffffu) | (px[j] & 0xff000000u);
  wave_sync();
  const u32x4_t r0v = *(u32x4_t *)(x + lane * 4), r1v = *(u32x4_t *)(x + 256 + lane * 4);
  if (ok0) __builtin_nontemporal_store(r0v, d4 + i0);
  if (ok1) __builtin_nontemporal_store(r1v, d4 + i1);
}

void shared_table_release(mi355_ctx *ctx, std::shared_ptr<void> *ref);
void lut_release(mi355_ctx *ctx) {
  for (int i = 0; i < 2; i++) {
    shared_table_release(ctx, &ctx->lut.table_ref[i]);  // the shared table goes when its last user does
    if (ctx->lut.pick[i].ev0) (void)hipEventDestroy(ctx->lut.pick[i].ev0);
    if (ctx->lut.pick[i].ev1) (void)hipEventDestroy(ctx->lut.pick[i].ev1);
  }
  if (ctx->lut.d_cells) (void)hipFree(ctx->lut.d_cells);
  if (ctx->lut.d_planar) (void)hipFree(ctx->lut.d_planar);
  if (ctx->lut.d_axis) (void)hipFree(ctx->lut.d_axis);
  brick_release(ctx->lut.brick);
  ctx->lut = LutDevice{};
}

// One axis-table entry for input byte v: the reference's coordinate arithmetic, op for op
// (this translation unit is compiled with -ffp-contract=off; host float ops are IEEE).
static void axis_entry(int v, float scale, float offset, int S, float *t_out, int *i0_out) {
  volatile float n = (float)v / 255.0f;            // norm_comp (imp.rs:472)
  volatile float m = n * scale;
  volatile float a = m + offset;
  float cl = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);  // inherent clamp (finite domain: no NaN)
  volatile float x = cl * ((float)S - 1.0f);       // imp.rs:438-440
  const float fl = std::floor(x);
  int i0 = (int)fl;
  if (i0 > S - 1) i0 = S - 1;                       // .min(max_idx), never taken since x <= S-1
  if (i0 < 0) i0 = 0;
  volatile float t = x - (float)i0;                 // imp.rs:504-506
  *t_out = t;
  *i0_out = i0;
}

int lut_upload(mi355_ctx *ctx, int is3d, size_t size, const float *table, const float scale[3],
               const float offset[3]) {
  // What the choice between the interpolating and the table kernels has learnt survives a reload as the PRIOR: which kind is
  // faster depends on the content (colour locality) and hardly on the LUT (the table kernel does not see it at all). Without it a
  // used context spends its first dozen launches after every reload re-learning through the slower kind
  // (profiles/r05_configs_elements.txt: 17^3 at 0.134 ms next to 33^3 at 0.096). The other kind is probed again after kProbeMin launches.
  AutoPolicy prior[2];
  bool have_prior = ctx->lut.loaded && ctx->lut.is3d && is3d;
  for (int i = 0; i < 2 && have_prior; i++) prior[i] = ctx->lut.pick[i];
  lut_release(ctx);
  LutDevice &L = ctx->lut;
  for (int i = 0; i < 2 && have_prior; i++) {
    if (prior[i].learn < 4 || prior[i].table_unavailable || !(prior[i].t_compute > 0.0 && prior[i].t_table > 0.0)) continue;
    AutoPolicy &P = L.pick[i];
    P.learn = 4; P.table = prior[i].table; P.t_compute = prior[i].t_compute; P.t_table = prior[i].t_table; P.vec = prior[i].vec;
    P.calls = 1;   // (not a multiple of the sampling period: the first launch builds the table, the sampling starts behind it)
    P.probe_period = kProbeMin; P.since_probe = 0; P.pending_kind = -1;
  }
  L.is3d = is3d;
  L.size = (int)size;
  for (int c = 0; c < 3; c++) { L.scale[c] = scale[c]; L.offset[c] = offset[c]; }
  const size_t n_floats = is3d ? 4 * size * size * size : 3 * size;
  int rc = check_hip(ctx, hipMalloc((void **)&L.d_cells, n_floats * sizeof(float)), "hipMalloc(lut cells)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpy(L.d_cells, table, n_floats * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(lut cells)");
  if (rc) return rc;

  bool domain_finite = true;
  for (int c = 0; c < 3; c++)
    if (!std::isfinite(scale[c]) || !std::isfinite(offset[c])) domain_finite = false;
  L.lds_ok = false;
  if (is3d && domain_finite && size <= 64) {
    const int S = (int)size;
    const LdsLayout lay = lds_layout_for(S);
    if (lay.lds_bytes <= kLdsBytes && (size_t)lay.Sz * 4 <= (size_t)kAxisTableBytes) {
      bool bounded = true;
      const size_t cells = size * size * size;
      for (size_t i = 0; i < cells && bounded; i++)
        for (int c = 0; c < 3; c++) {
          const float v = table[i * 4 + c];
          if (!(std::fabs(v) <= 1e30f)) { bounded = false; break; }
        }
      if (bounded) {
        std::vector<float> planar(3 * lay.plane_floats, 0.0f);
        for (int c = 0; c < 3; c++)
          for (int z = 0; z < S; z++)
            for (int y = 0; y < S; y++)
              for (int x = 0; x < S; x++)
                planar[(size_t)c * lay.plane_floats + (size_t)x + (size_t)lay.Sy * y + (size_t)lay.Sz * (S - 1 - z)] =
                    table[4 * ((size_t)x + (size_t)S * y + (size_t)S * S * z) + c];
        std::vector<uint32_t> axis(3 * 256 * 2);
        for (int a = 0; a < 3; a++)
          for (int v = 0; v < 256; v++) {
            float t;
            int i0;
            axis_entry(v, scale[a], offset[a], S, &t, &i0);
            int off;
            // Every offset field must stay a small non-negative integer: the z0 == S-1 pad reads
            // interpret these dwords as floats (denormals: finite), so no negative/huge bit patterns.
            if (a == 0) off = 4 * i0;                                       // x
            else if (a == 1) off = 4 * lay.Sy * i0;                         // y
            else off = kAxisTableBytes + 4 * lay.Sz * (S - 2 - i0);         // z: plane base + plane of z0+1 (reversed order)
            axis[(size_t)(a * 256 + v) * 2 + 0] = (uint32_t)off;
            std::memcpy(&axis[(size_t)(a * 256 + v) * 2 + 1], &t, 4);
          }
        rc = check_hip(ctx, hipMalloc((void **)&L.d_planar, planar.size() * sizeof(float)), "hipMalloc(lut planar)");
        if (rc) return rc;
        rc = check_hip(ctx, hipMemcpy(L.d_planar, planar.data(), planar.size() * sizeof(float), hipMemcpyHostToDevice),
                       "hipMemcpy(lut planar)");
        if (rc) return rc;
        rc = check_hip(ctx, hipMalloc((void **)&L.d_axis, axis.size() * sizeof(uint32_t)), "hipMalloc(lut axis tables)");
        if (rc) return rc;
        rc = check_hip(ctx, hipMemcpy(L.d_axis, axis.data(), axis.size() * sizeof(uint32_t), hipMemcpyHostToDevice),
                       "hipMemcpy(lut axis tables)");
        if (rc) return rc;
        L.planar_plane_floats = lay.plane_floats;
        L.lds_Sy = lay.Sy;
        L.lds_Sz = lay.Sz;
        L.lds_bytes = lay.lds_bytes;
        L.lds_all_resident = lay.all_resident;
        L.lds_ok = true;
      }
    }
  }
  if (!is3d && domain_finite) {
    const size_t plane = size + 1;
    const size_t bytes = kAxisTableBytes + 3 * plane * sizeof(float);
    bool bounded = true;
    for (size_t i = 0; i < 3 * size && bounded; i++)
      if (!(std::fabs(table[i]) <= 1e30f)) bounded = false;
    if (bytes <= kLdsBytes && bounded) {
      std::vector<uint32_t> image((bytes + 3) / 4, 0u);
      for (int a = 0; a < 3; a++) {
        for (int v = 0; v < 256; v++) {
          float t;
          int i0;
          axis_entry(v, scale[a], offset[a], (int)size, &t, &i0);
          image[(size_t)(a * 256 + v) * 2 + 0] = (uint32_t)(kAxisTableBytes + (a * plane + (size_t)i0) * sizeof(float));
          std::memcpy(&image[(size_t)(a * 256 + v) * 2 + 1], &t, 4);
        }
        std::memcpy(&image[kAxisTableBytes / 4 + a * plane], table + (size_t)a * size, size * sizeof(float));
      }
      rc = check_hip(ctx, hipMalloc((void **)&L.d_axis, image.size() * sizeof(uint32_t)), "hipMalloc(1D LUT LDS image)");
      if (rc) return rc;
      rc = check_hip(ctx, hipMemcpy(L.d_axis, image.data(), image.size() * sizeof(uint32_t), hipMemcpyHostToDevice),
                     "hipMemcpy(1D LUT LDS image)");
      if (rc) return rc;
      L.lds_bytes = image.size() * sizeof(uint32_t);
      L.lds_ok = true;
    }
  }
  if (is3d && (rc = brick_upload(ctx, L.brick, (int)size, table, scale, offset))) return rc;
  {
    // identity of this LUT for the shared-table registry: kind, size, domain and two independent 64-bit hashes of the table
    uint64_t h1 = 1469598103934665603ull, h2 = 0x9e3779b97f4a7c15ull;
    const unsigned char *b = (const unsigned char *)table;
    for (size_t i = 0; i < n_floats * sizeof(float); i++) {
      h1 = (h1 ^ b[i]) * 1099511628211ull;
      h2 = (h2 + b[i] + 1) * 0xff51afd7ed558ccdull;
      h2 ^= h2 >> 29;
    }
    L.digest.assign((const char *)&h1, 8);
    L.digest.append((const char *)&h2, 8);
    L.digest.append((const char *)&is3d, sizeof(is3d));
    L.digest.append((const char *)&size, sizeof(size));
    L.digest.append((const char *)scale, 12);
    L.digest.append((const char *)offset, 12);
  }
  L.loaded = true;
  return MI355_OK;
}

template <int NT, int P4, int S_CONST, int HSV = kNoHsv>
static int launch_lds_variant(mi355_ctx *ctx, const uint4 *src, uint4 *dst, size_t n_groups, const HsvK &hk = HsvK{}) {
  auto kern = colorlut3d_lds_kernel<NT, P4, S_CONST, HSV>;
  const LutDevice &L = ctx->lut;
  const size_t lds = L.lds_bytes;
  int rc = check_hip(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                     "hipFuncSetAttribute(max dynamic LDS)");
  if (rc) return rc;
  // one block per CU (the plane takes the CU's whole LDS); small inputs use fewer blocks so that a
  // block always has at least 1024 pixels to amortise its plane stagings
  size_t grid = (size_t)ctx->n_cu;
  const size_t min_blocks = (n_groups + 255) / 256;
  if (grid > min_blocks) grid = min_blocks;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, ctx->stream, src, dst, n_groups, (const float *)L.d_planar,
                     (const uint32_t *)L.d_axis, L.lds_Sy, L.lds_Sz, (uint32_t)L.planar_plane_floats,
                     (L.lds_all_resident ? 1 : 0) | (ctx->lut_stagger << 8), hk);
  return check_hip(ctx, hipGetLastError(), "colorlut3d_lds kernel launch");
}

template <int NT, int P4>
static int launch_lean_variant(mi355_ctx *ctx, const uint4 *src, uint4 *dst, size_t n_groups) {
  const LutDevice &L = ctx->lut;
  const size_t lds = L.lds_bytes;
  const void *k33 = (const void *)colorlut3d_lean_kernel<NT, P4, 33>, *k0 = (const void *)colorlut3d_lean_kernel<NT, P4, 0>;
  int rc = check_hip(ctx, hipFuncSetAttribute(L.size == 33 ? k33 : k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute(max dynamic LDS)");
  if (rc) return rc;
  size_t grid = (size_t)ctx->n_cu;
  const size_t min_blocks = (n_groups + 255) / 256;
  if (grid > min_blocks) grid = min_blocks;
  if (grid < 1) grid = 1;
  if (L.size == 33)
    hipLaunchKernelGGL((colorlut3d_lean_kernel<NT, P4, 33>), dim3((unsigned)grid), dim3(NT), lds, ctx->stream, src, dst, n_groups, (const float *)L.d_planar,
                       (const uint32_t *)L.d_axis, L.lds_Sy, L.lds_Sz, (uint32_t)L.planar_plane_floats, L.lds_all_resident ? 1 : 0);
  else
    hipLaunchKernelGGL((colorlut3d_lean_kernel<NT, P4, 0>), dim3((unsigned)grid), dim3(NT), lds, ctx->stream, src, dst, n_groups, (const float *)L.d_planar,
                       (const uint32_t *)L.d_axis, L.lds_Sy, L.lds_Sz, (uint32_t)L.planar_plane_floats, L.lds_all_resident ? 1 : 0);
  return check_hip(ctx, hipGetLastError(), "colorlut3d_lean kernel launch");
}

// Is the RGBA8 3D LDS kernel applicable to this geometry? (contiguous rows and frames, 16 B aligned)
static bool lds3d_rgba_applicable(const mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, const uint8_t *d_dst,
                                  size_t dst_pitch, int dst_stride, int n_frames, int width, int height, size_t *n_groups) {
  const LutDevice &L = ctx->lut;
  if (!(L.is3d && L.lds_ok) || ctx->force_generic) return false;
  const size_t row_bytes = (size_t)width * 4;
  const bool contiguous = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                          (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
  const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
  if (!(contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && (total_bytes % 16 == 0))) return false;
  *n_groups = total_bytes / 16;
  return true;
}

template <int NT, int P4, int S_CONST, int HSV, bool LATE_LOAD = false>
static int launch_pipe_variant(mi355_ctx *ctx, const uint4 *src, uint4 *dst, size_t n_groups, const HsvK &hk) {
  auto kern = hsv_colorlut3d_pipe_kernel<NT, P4, S_CONST, HSV, LATE_LOAD>;
  const LutDevice &L = ctx->lut;
  const size_t lds = L.lds_bytes;
  int rc = check_hip(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                     "hipFuncSetAttribute(max dynamic LDS)");
  if (rc) return rc;
  size_t grid = (size_t)ctx->n_cu;
  const size_t min_blocks = (n_groups + 255) / 256;
  if (grid > min_blocks) grid = min_blocks;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, ctx->stream, src, dst, n_groups, (const float *)L.d_planar,
                     (const uint32_t *)L.d_axis, L.lds_Sy, L.lds_Sz, (uint32_t)L.planar_plane_floats, L.lds_all_resident ? 1 : 0, hk);
  return check_hip(ctx, hipGetLastError(), "hsv_colorlut3d_pipe kernel launch");
}

template <int HSV>
static int launch_fused_variant(mi355_ctx *ctx, const uint4 *s, uint4 *d, size_t n_groups, const HsvK &hk) {
  const bool s33 = ctx->lut.size == 33;
  // MI355_FLAG_FUSED_VARIANT 1: software-pipelined kernel (hsv of the next tile under the plane DMA). Measured on
  // 8x4K: 0.316 ms smooth / 0.476 ms noise at 1024x2 vs 0.330 / 0.435 ms for the inline kernel at 1024x3; the
  // 1024x3 pipelined tiling spills at the 128-VGPR cap (0.49 ms), so the inline kernel stays the default.
  if (ctx->fused_variant == 1)
    return s33 ? launch_pipe_variant<1024, 2, 33, HSV>(ctx, s, d, n_groups, hk) : launch_pipe_variant<1024, 2, 0, HSV>(ctx, s, d, n_groups, hk);
  return s33 ? launch_lds_variant<1024, 3, 33, HSV>(ctx, s, d, n_groups, hk) : launch_lds_variant<1024, 3, 0, HSV>(ctx, s, d, n_groups, hk);
}

// hsvfilter followed by colorlut on RGBA frames. One fused launch when the 3D LDS kernel applies; otherwise the
// two element kernels back to back (copy src -> dst, hsvfilter in place on dst, colorlut dst -> dst; every colorlut
// kernel reads a pixel before it writes the same pixel, so in-place is safe).
static int launch_colorlut_compute(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst, size_t dst_pitch,
                                  int dst_stride, int n_frames, int width, int height, int format);

// `plain`: the colorlut half of the unfused fallback uses the compute kernel directly (table builds)
static int launch_hsv_colorlut_compute(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                                       size_t dst_pitch, int dst_stride, int n_frames, int width, int height,
                                       const mi355_hsv_settings &hs, bool plain) {
  const LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  size_t n_groups = 0;
  const bool three_pass_ok = lds3d_rgba_applicable(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, &n_groups);
  const int lv = ctx->lut_variant;
  if (!ctx->force_generic && lv != 3 && lv != 1 && lv != 2 && ctx->fused_variant == 0 &&
      brick_applicable(ctx->lut.brick, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height)) {
    BrickLut &B = ctx->lut.brick;
    const bool build = ctx->lut.building_table;
    const bool pinned = lv == 7 || !three_pass_ok || ctx->brick_sets != 0;
    int level = pinned ? (ctx->brick_sets == 64 ? 1 : 0) : (build ? 2 : brick_choose(B));
    if (level == 2 && !three_pass_ok) level = 1;
    if (level == 2 || build || pinned) brick_mark_unwatched(B);
    if (level < 2) {
      const int sets = pinned && ctx->brick_sets ? ctx->brick_sets : (level ? 64 : 32);
      ctx->lut.last_kernel = sets == 64 ? "colorlut3d_brick_kernel<HSV> (64 sets)" : (sets == 48 ? "colorlut3d_brick_kernel<HSV> (48 sets)" : "colorlut3d_brick_kernel<HSV>");
      int rc = (build || pinned) ? MI355_OK : brick_before_launch(ctx, B, level);
      if (rc) return rc;
      rc = brick_launch(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, &hs, sets);
      if (rc || build || pinned) return rc;
      return brick_after_launch(ctx, B, (unsigned long long)width * height * n_frames, level);
    }
  }
  if (three_pass_ok) {
    ctx->lut.last_kernel = "colorlut3d_lds_kernel<HSV>";
    const HsvK hk{hs.hue_shift, hs.saturation_mul, hs.saturation_off, hs.value_mul, hs.value_off};
    const uint4 *s = (const uint4 *)d_src;
    uint4 *d = (uint4 *)d_dst;
    switch (hsv_variant_for(hs, false)) {
      case -1: return launch_fused_variant<-1>(ctx, s, d, n_groups, hk);
      case 0: return launch_fused_variant<0>(ctx, s, d, n_groups, hk);
      case 1: return launch_fused_variant<1>(ctx, s, d, n_groups, hk);
      case 2: return launch_fused_variant<2>(ctx, s, d, n_groups, hk);
      case 4: return launch_fused_variant<4>(ctx, s, d, n_groups, hk);
      case 5: return launch_fused_variant<5>(ctx, s, d, n_groups, hk);
      default: return launch_fused_variant<6>(ctx, s, d, n_groups, hk);
    }
  }
  PixFmt fmt;
  pixfmt_of(MI355_FMT_RGBA, &fmt);
  if (d_src != d_dst) {
    for (int f = 0; f < n_frames; f++) {
      int rc = check_hip(ctx, hipMemcpy2DAsync(d_dst + (size_t)f * dst_pitch, (size_t)dst_stride, d_src + (size_t)f * src_pitch, (size_t)src_stride,
                                               (size_t)width * 4, (size_t)height, hipMemcpyDeviceToDevice, ctx->stream),
                         "hsv+colorlut: device copy");
      if (rc) return rc;
    }
  }
  int rc = launch_hsvfilter_compute(ctx, d_dst, n_frames, dst_pitch, width, height, dst_stride, fmt, hs);
  if (rc) return rc;
  if (plain) return launch_colorlut_compute(ctx, d_dst, dst_pitch, dst_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, MI355_FMT_RGBA);
  return launch_colorlut(ctx, d_dst, dst_pitch, dst_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, MI355_FMT_RGBA);
}

static int launch_colorlut_compute(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                    size_t dst_pitch, int dst_stride, int n_frames, int width, int height, int format) {
  const LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  LutK k;
  for (int c = 0; c < 3; c++) { k.scale[c] = L.scale[c]; k.offset[c] = L.offset[c]; }
  k.size = L.size;

  const bool rgba8 = (format == MI355_FMT_RGBA);
  if (rgba8 && !L.is3d && L.lds_ok && !ctx->force_generic) {
    const size_t row_bytes = (size_t)width * 4;
    const bool contiguous = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                            (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
    const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
    if (contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && (total_bytes % 16 == 0)) {
      constexpr int NT = 512;
      ctx->lut.last_kernel = "colorlut1d_lds_kernel";
      auto kern = colorlut1d_lds_kernel<NT>;
      int rc = check_hip(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bytes),
                         "hipFuncSetAttribute(max dynamic LDS)");
      if (rc) return rc;
      const size_t n_vec = total_bytes / 16;
      // blocks per CU limited by the LDS image; a few hundred pixels per lane amortise the table staging
      size_t per_cu = kLdsBytes / (L.lds_bytes ? L.lds_bytes : 1);
      if (per_cu < 1) per_cu = 1;
      if (per_cu > 4) per_cu = 4;
      size_t grid = (size_t)ctx->n_cu * per_cu;
      const size_t max_blocks = (n_vec + NT - 1) / NT;
      if (grid > max_blocks) grid = max_blocks;
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), L.lds_bytes, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_vec,
                         (const uint32_t *)L.d_axis, (uint32_t)(L.lds_bytes / 4));
      return check_hip(ctx, hipGetLastError(), "colorlut1d_lds kernel launch");
    }
  }
  {
    size_t n_groups = 0;
    // MI355_FLAG_LUT_VARIANT among the interpolating kernels: 0 / 6 = brick-cache kernel with the content watch handing
    // noise-like streams to the three-pass whole-plane kernel (brick_choose); 7 = brick kernel only; 3 = three-pass only;
    // 1 / 2 = the three-pass kernel's late-prefetch / lean-state experiments.
    const int v = ctx->lut_variant;
    const bool three_pass_ok = rgba8 && lds3d_rgba_applicable(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, &n_groups);
    if (rgba8 && !ctx->force_generic && (v == 0 || v == 6 || v == 7 || v == 4 || v == 5) &&
        brick_applicable(ctx->lut.brick, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height)) {
      BrickLut &B = ctx->lut.brick;
      // a table build runs over the all-colours frame, which is as hostile to the brick cache as noise: three-pass kernel
      // when it applies, and no entry in the stream's content watch either way
      const bool build = ctx->lut.building_table;
      // MI355_FLAG_BRICK_SETS = 512: the block-shared cache (round 4), pinned
      if (ctx->brick_sets == 512 && shared_applicable(ctx, width, dst_stride, n_frames, height)) {
        brick_mark_unwatched(B);
        ctx->lut.last_kernel = "colorlut3d_shared_kernel";
        return shared_launch(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height);
      }
      // MI355_FLAG_BRICK_SETS pins the cache geometry; 0 lets the content watch pick (and leave for the three-pass kernel)
      const bool pinned = v == 7 || !three_pass_ok || ctx->brick_sets != 0;
      // level 1 of the watch is the block-shared cache where the launch is large enough for one
      bool small = false;
      const bool shared1 = !pinned && !build && shared_applicable(ctx, width, dst_stride, n_frames, height, &small);
      int level = pinned ? (ctx->brick_sets == 64 ? 1 : 0) : (build ? 2 : brick_choose(B, shared1 && small ? 1 : 0));
      if (level == 2 && !three_pass_ok) level = 1;
      if (level == 2 || build || pinned) brick_mark_unwatched(B);
      if (level < 2) {
        const int sets = pinned && ctx->brick_sets ? ctx->brick_sets : (level ? 64 : 32);
        const bool shared = shared1 && level == 1;
        ctx->lut.last_kernel = shared ? "colorlut3d_shared_kernel" : (sets == 64 ? "colorlut3d_brick_kernel (64 sets)" : (sets == 48 ? "colorlut3d_brick_kernel (48 sets)" : "colorlut3d_brick_kernel"));
        int rc = (build || pinned) ? MI355_OK : brick_before_launch(ctx, B, level);
        if (rc) return rc;
        rc = shared ? shared_launch(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height)
                    : brick_launch(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, nullptr, sets);
        if (rc || build || pinned) return rc;
        return brick_after_launch(ctx, B, (unsigned long long)width * height * n_frames, level, shared1);
      }
    }
    if (three_pass_ok) {
      ctx->lut.last_kernel = "colorlut3d_lds_kernel";
      const uint4 *s = (const uint4 *)d_src;
      uint4 *d = (uint4 *)d_dst;
      constexpr int NT = 1024, P4 = 3;
      // lean-state kernel (2 registers per pixel, 32 pixels per lane). Measured on 8x4K: 0.243-0.258 ms smooth /
      // 0.440 ms noise vs 0.249-0.253 / 0.402 ms for the default: the larger tile amortises staging and barriers, the
      // three extra axis reads per pixel and pass give it back. Not the default.
      if (ctx->lut_variant == 2) return launch_lean_variant<1024, 8>(ctx, s, d, n_groups);
      if (ctx->lut_variant == 1)  // MI355_FLAG_LUT_VARIANT: next tile's pixels prefetched before the last pass
        return L.size == 33 ? launch_pipe_variant<NT, P4, 33, kNoHsv, true>(ctx, s, d, n_groups, HsvK{})
                            : launch_pipe_variant<NT, P4, 0, kNoHsv, true>(ctx, s, d, n_groups, HsvK{});
      if (L.size == 33) return launch_lds_variant<NT, P4, 33>(ctx, s, d, n_groups);
      return launch_lds_variant<NT, P4, 0>(ctx, s, d, n_groups);
    }
  }

  // RGBA64 with a 3D LUT: the brick-cache kernel (round 3: coordinates computed instead of tabled) under the same content
  // watch as RGBA8 - noise-like streams go on to the three-pass 16-bit kernel below
  if ((format == MI355_FMT_RGBA64_LE || format == MI355_FMT_RGBA64_BE) && L.is3d && !ctx->force_generic && ctx->lut_variant != 3 &&
      brick_applicable(ctx->lut.brick, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, 8)) {
    BrickLut &B = ctx->lut.brick;
    const size_t row_bytes64 = (size_t)width * 8;
    const bool three_pass64_ok = L.lds_ok && (size_t)src_stride == row_bytes64 && (size_t)dst_stride == row_bytes64 &&
                                 (n_frames == 1 || (src_pitch == row_bytes64 * (size_t)height && dst_pitch == row_bytes64 * (size_t)height));
    const bool pinned = ctx->lut_variant == 7 || !three_pass64_ok || ctx->brick_sets != 0;
    int level = pinned ? (ctx->brick_sets == 64 ? 1 : 0) : brick_choose(B);
    if (level == 2 && !three_pass64_ok) level = 1;
    if (level == 2 || pinned) brick_mark_unwatched(B);
    if (level < 2) {
      const int sets = pinned && ctx->brick_sets == 64 ? 64 : (level ? 64 : 32);
      ctx->lut.last_kernel = sets == 64 ? "colorlut3d_brick_kernel<RGBA64> (64 sets)" : "colorlut3d_brick_kernel<RGBA64>";
      int rc = pinned ? MI355_OK : brick_before_launch(ctx, B, level);
      if (rc) return rc;
      rc = brick_launch(ctx, B, d_src, src_stride, d_dst, dst_stride, n_frames, width, height, nullptr, sets, format == MI355_FMT_RGBA64_LE ? 1 : 2);
      if (rc || pinned) return rc;
      return brick_after_launch(ctx, B, (unsigned long long)width * height * n_frames, level);
    }
  }

  if ((format == MI355_FMT_RGBA64_LE || format == MI355_FMT_RGBA64_BE) && L.is3d && L.lds_ok && !ctx->force_generic) {
    const size_t row_bytes = (size_t)width * 8;
    const bool contiguous = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                            (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
    const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
    if (contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && (total_bytes % 16 == 0)) {
      constexpr int NT = 1024, P2 = 4;
      ctx->lut.last_kernel = "colorlut3d_lds64_kernel";
      Lut64K k64;
      for (int c = 0; c < 3; c++) { k64.scale[c] = L.scale[c]; k64.offset[c] = L.offset[c]; }
      k64.sm1 = (float)L.size - 1.0f;
      k64.sy4 = 4u * (uint32_t)L.lds_Sy; k64.sz4 = 4u * (uint32_t)L.lds_Sz; k64.plane_base = kAxisTableBytes; k64.S = L.size;
      const size_t n_groups = total_bytes / 16;
      size_t grid = (size_t)ctx->n_cu;
      const size_t min_blocks = (n_groups + 255) / 256;
      if (grid > min_blocks) grid = min_blocks;
      if (grid < 1) grid = 1;
      const size_t lds = kAxisTableBytes + 4 * L.planar_plane_floats;
      const bool le = format == MI355_FMT_RGBA64_LE;
      auto kle = colorlut3d_lds64_kernel<NT, P2, true>;
      auto kbe = colorlut3d_lds64_kernel<NT, P2, false>;
      int rc = check_hip(ctx, hipFuncSetAttribute(le ? (const void *)kle : (const void *)kbe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                         "hipFuncSetAttribute(max dynamic LDS)");
      if (rc) return rc;
      if (le) hipLaunchKernelGGL(kle, dim3((unsigned)grid), dim3(NT), lds, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_groups,
                                 (const float *)L.d_planar, (const uint32_t *)L.d_axis, (uint32_t)L.planar_plane_floats, k64);
      else hipLaunchKernelGGL(kbe, dim3((unsigned)grid), dim3(NT), lds, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_groups,
                              (const float *)L.d_planar, (const uint32_t *)L.d_axis, (uint32_t)L.planar_plane_floats, k64);
      return check_hip(ctx, hipGetLastError(), "colorlut3d_lds64 kernel launch");
    }
  }

  if ((format == MI355_FMT_RGBA64_LE || format == MI355_FMT_RGBA64_BE) && !L.is3d && L.lds_ok && !ctx->force_generic) {
    const size_t row_bytes = (size_t)width * 8;
    const bool contiguous = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                            (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
    const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
    if (contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && (total_bytes % 16 == 0)) {
      constexpr int NT = 512;
      ctx->lut.last_kernel = "colorlut1d_lds64_kernel";
      Lut1d64K k1;
      for (int c = 0; c < 3; c++) { k1.scale[c] = L.scale[c]; k1.offset[c] = L.offset[c]; }
      k1.sm1 = (float)L.size - 1.0f;
      k1.table_bytes = ((uint32_t)L.size + 1u) * 4u;
      const bool le = format == MI355_FMT_RGBA64_LE;
      auto kle = colorlut1d_lds64_kernel<NT, true>;
      auto kbe = colorlut1d_lds64_kernel<NT, false>;
      int rc = check_hip(ctx, hipFuncSetAttribute(le ? (const void *)kle : (const void *)kbe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bytes),
                         "hipFuncSetAttribute(max dynamic LDS)");
      if (rc) return rc;
      const size_t n_groups = total_bytes / 16;
      size_t per_cu = kLdsBytes / (L.lds_bytes ? L.lds_bytes : 1);
      if (per_cu < 1) per_cu = 1;
      if (per_cu > 4) per_cu = 4;
      size_t grid = (size_t)ctx->n_cu * per_cu;
      const size_t max_blocks = (n_groups + NT - 1) / NT;
      if (grid > max_blocks) grid = max_blocks;
      if (le) hipLaunchKernelGGL(kle, dim3((unsigned)grid), dim3(NT), L.lds_bytes, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_groups,
                                 (const uint32_t *)L.d_axis, (uint32_t)(L.lds_bytes / 4), k1);
      else hipLaunchKernelGGL(kbe, dim3((unsigned)grid), dim3(NT), L.lds_bytes, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_groups,
                              (const uint32_t *)L.d_axis, (uint32_t)(L.lds_bytes / 4), k1);
      return check_hip(ctx, hipGetLastError(), "colorlut1d_lds64 kernel launch");
    }
  }

  ctx->lut.last_kernel = "colorlut_rows_kernel";
  const size_t total = (size_t)width * (size_t)height * (size_t)n_frames;
  size_t blocks = (total + 255) / 256;
  const size_t cap = (size_t)ctx->n_cu * 8;
  if (blocks > cap) blocks = cap;
  dim3 g((unsigned)blocks), b(256);
#define MI355_ROWS(IS3D, B16, LE)                                                                               \
  hipLaunchKernelGGL((colorlut_rows_kernel<IS3D, B16, LE>), g, b, 0, ctx->stream, d_src, src_pitch, src_stride, \
                     d_dst, dst_pitch, dst_stride, n_frames, width, height, (const float *)L.d_cells, k)
  if (format == MI355_FMT_RGBA) {
    if (L.is3d) MI355_ROWS(true, false, true); else MI355_ROWS(false, false, true);
  } else if (format == MI355_FMT_RGBA64_LE) {
    if (L.is3d) MI355_ROWS(true, true, true); else MI355_ROWS(false, true, true);
  } else if (format == MI355_FMT_RGBA64_BE) {
    if (L.is3d) MI355_ROWS(true, true, false); else MI355_ROWS(false, true, false);
  } else {
    return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: format must be RGBA, RGBA64_LE or RGBA64_BE");
  }
#undef MI355_ROWS
  return check_hip(ctx, hipGetLastError(), "colorlut rows kernel launch");
}


// geometry test shared by the table path and the auto-selector: RGBA8 frames on 16-byte-aligned storage, either packed
// (rows and frames back to back: one flat array of pixel groups) or with padded rows (strides that are multiples of 16 B,
// frames `stride * height` apart so that a batch is one tall picture: the tiled kernel takes row strides)
struct Rgba8Geom {
  size_t n_vec = 0;      // pixel groups (4 px) of the batch
  unsigned sw4 = 0, dw4 = 0;  // row strides in 16-byte groups
  bool packed = false;
};
static bool rgba8_geom(const uint8_t *d_src, size_t src_pitch, int src_stride, const uint8_t *d_dst, size_t dst_pitch, int dst_stride, int n_frames,
                       int width, int height, Rgba8Geom *g) {
  const size_t row_bytes = (size_t)width * 4;
  if (((uintptr_t)d_src % 16 != 0) || ((uintptr_t)d_dst % 16 != 0)) return false;
  const bool packed = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                      (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
  const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
  if (packed) {
    if (total_bytes % 16 != 0) return false;
    g->n_vec = total_bytes / 16;
    g->sw4 = g->dw4 = (unsigned)(row_bytes / 16);
    g->packed = true;
    return true;
  }
  // padded rows: only what the tiled kernel takes (whole 16-byte groups per row, width >= 128)
  if (width % 4 != 0 || width < 128 || (size_t)src_stride < row_bytes || (size_t)dst_stride < row_bytes || src_stride % 16 != 0 || dst_stride % 16 != 0) return false;
  if (n_frames != 1 && (src_pitch != (size_t)src_stride * (size_t)height || dst_pitch != (size_t)dst_stride * (size_t)height)) return false;
  g->n_vec = total_bytes / 16;
  g->sw4 = (unsigned)src_stride / 16;
  g->dw4 = (unsigned)dst_stride / 16;
  g->packed = false;
  return true;
}

static bool same_hs(const mi355_hsv_settings &a, const mi355_hsv_settings &b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

// ---- memoised tables are shared between contexts
// A table is a pure function of what it was built from (LUT contents and domain, index layout, hsv settings), so every
// context of a process that asks for the same one on the same device gets the same 64 MiB: 32 streams with one LUT hold one
// table, not 32 that evict each other from L2 and the Infinity Cache. The registry holds weak references; the context
// that creates an entry enqueues the build on its own stream and records `ready` there, all under the registry lock, and
// every other context makes its stream wait for that event once (hipStreamWaitEvent: no host wait). The table is freed
// when its last user lets go.
struct SharedTable {
  uint32_t *d = nullptr;
  hipEvent_t ready = nullptr;
  // "the last launch that reads this table on my stream has been enqueued": one event per context that let go of the table
  // while others kept it (shared_table_release). Whoever later rebuilds the buffer in place makes its stream wait for them.
  std::vector<hipEvent_t> released;
  int device = 0;
  ~SharedTable() {
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); }
    for (hipEvent_t e : released) (void)hipEventDestroy(e);
    if (d) (void)hipFree(d);  // (a device-wide synchronisation: nothing reads the buffer any more when it returns)
    (void)hipSetDevice(cur);
  }
};
static std::mutex g_tables_mu;
static std::map<std::string, std::weak_ptr<SharedTable>> g_tables;

// Lets go of *ref. If other contexts keep the table, this context's reads of it - all enqueued on ctx->stream before this
// call - are marked with an event the table remembers: a later sole owner may rebuild the buffer in place, on ITS stream,
// and must not overtake them. Call with g_tables_mu held.
static void shared_table_release_locked(mi355_ctx *ctx, std::shared_ptr<void> *ref) {
  if (!*ref) return;
  if (ref->use_count() > 1) {
    SharedTable *sp = static_cast<SharedTable *>(ref->get());
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
      if (hipEventRecord(e, ctx->stream) == hipSuccess) sp->released.push_back(e);
      else { (void)hipEventDestroy(e); (void)hipStreamSynchronize(ctx->stream); }
    } else {
      (void)hipStreamSynchronize(ctx->stream);  // no event to be had: make the statement true the slow way
    }
    (void)hipGetLastError();
  }
  ref->reset();
}
void shared_table_release(mi355_ctx *ctx, std::shared_ptr<void> *ref) {
  std::shared_ptr<void> last;  // if this was the last reference, the table is destroyed (device-wide wait) outside the lock
  {
    std::lock_guard<std::mutex> g(g_tables_mu);
    if (*ref && ref->use_count() == 1) last = std::move(*ref);
    else shared_table_release_locked(ctx, ref);
  }
}

// `build(table)` enqueues the kernels that fill a table on ctx->stream. On return *out is usable on ctx->stream.
// *ref may hold the table this context used under ANOTHER key (new hsv settings, a reloaded LUT). If nobody else has that
// table it is rebuilt in place under the new key, in stream order - everything that read the old contents was enqueued on
// this stream earlier - instead of being destroyed (hipEventSynchronize + hipFree + a fresh 64 MiB hipMalloc on the
// streaming thread: device-wide waits in a path that otherwise never blocks its caller).
template <class Build>
static int shared_table_acquire(mi355_ctx *ctx, const std::string &key, std::shared_ptr<void> *ref, uint32_t **out, Build &&build) {
  std::lock_guard<std::mutex> g(g_tables_mu);
  for (auto it = g_tables.begin(); it != g_tables.end();) it = it->second.expired() ? g_tables.erase(it) : std::next(it);
  const std::string full = std::string((const char *)&ctx->device, sizeof(ctx->device)) + key;
  if (auto sp = g_tables[full].lock()) {
    int rc = check_hip(ctx, hipStreamWaitEvent(ctx->stream, sp->ready, 0), "hipStreamWaitEvent(shared table)");
    if (rc) return rc;
    if (ref->get() != sp.get()) shared_table_release_locked(ctx, ref);  // the table used under the old key may live on in other contexts
    *out = sp->d;
    *ref = sp;
    return MI355_OK;
  }
  std::shared_ptr<SharedTable> sp;
  if (*ref && ref->use_count() == 1) {
    // sole user (strong references are only ever taken under this lock): keep the buffer, give it the new key. Contexts
    // that shared it earlier may still have launches in flight on THEIR streams that read the old contents: wait for the
    // events they left behind (in stream order, no host wait) before the build overwrites it.
    sp = std::static_pointer_cast<SharedTable>(*ref);
    for (hipEvent_t e : sp->released) {
      int rc = check_hip(ctx, hipStreamWaitEvent(ctx->stream, e, 0), "hipStreamWaitEvent(released table)");
      if (rc) return rc;
    }
    for (hipEvent_t e : sp->released) (void)hipEventDestroy(e);  // (the waits are enqueued; the runtime keeps what they need)
    sp->released.clear();
    for (auto it = g_tables.begin(); it != g_tables.end();) it = it->second.lock() == sp ? g_tables.erase(it) : std::next(it);
  } else {
    shared_table_release_locked(ctx, ref);
    sp = std::make_shared<SharedTable>();
    sp->device = ctx->device;
    int rc = check_hip(ctx, hipMalloc((void **)&sp->d, (size_t)kTableEntries * 4), "hipMalloc(memoised table)");
    if (rc) return rc;
    if ((rc = check_hip(ctx, hipEventCreateWithFlags(&sp->ready, hipEventDisableTiming), "hipEventCreate(shared table)"))) return rc;
  }
  int rc = build(sp->d);
  if (rc) { ref->reset(); return rc; }
  if ((rc = check_hip(ctx, hipEventRecord(sp->ready, ctx->stream), "hipEventRecord(shared table)"))) { ref->reset(); return rc; }
  g_tables[full] = sp;
  *out = sp->d;
  *ref = sp;
  return MI355_OK;
}

extern "C" int mi355_shared_table_count(void) {
  std::lock_guard<std::mutex> g(g_tables_mu);
  int n = 0;
  for (auto &kv : g_tables) n += kv.second.expired() ? 0 : 1;
  return n;
}

// Makes table `which` (0: colorlut, 1: hsvfilter -> colorlut under *hs) in the requested layout current for this context:
// shared with every context that loaded the same LUT (and, for 1, uses the same settings); built - table filled with the
// colour of every slot, exact compute path run over it in place, on the stream, no host wait - only if nobody has it.
static int table_ensure(mi355_ctx *ctx, int which, int morton, const mi355_hsv_settings *hs) {
  LutDevice &L = ctx->lut;
  std::string key(which ? "F" : "L");
  key.push_back((char)('0' + morton));
  key += L.digest;
  if (which) key.append((const char *)hs, sizeof(*hs));
  if (L.table_ref[which] && L.table_key[which] == key) return MI355_OK;
  L.d_table[which] = nullptr;  // (the reference stays: shared_table_acquire reuses the buffer if nobody else holds it)
  L.table_morton[which] = -1;
  L.table_key[which].clear();
  int rc = shared_table_acquire(ctx, key, &L.table_ref[which], &L.d_table[which], [&](uint32_t *tab) {
    uint8_t *t = (uint8_t *)tab;
    hipLaunchKernelGGL(table_domain_kernel, dim3(kTableEntries / 256), dim3(256), 0, ctx->stream, tab, morton);
    L.building_table = true;
    const char *serving = L.last_kernel;
    const int r = which == 0 ? launch_colorlut_compute(ctx, t, 0, 4096 * 4, t, 0, 4096 * 4, 1, 4096, 4096, MI355_FMT_RGBA)
                             : launch_hsv_colorlut_compute(ctx, t, 0, 4096 * 4, t, 0, 4096 * 4, 1, 4096, 4096, *hs, true);
    L.building_table = false;
    L.last_kernel = serving;
    return r;
  });
  if (rc) { L.table_ref[which].reset(); L.d_table[which] = nullptr; return rc; }
  L.table_key[which] = key;
  L.table_morton[which] = morton;
  if (which == 1) L.table_hs = *hs;
  return MI355_OK;
}

// `sub`: which kernel reads a Morton table - the gather kernels or the LDS-cached one (colorlut_window.hip)
enum TableKernel { kTableGather = 0, kTableWindow = 1, kTableEither = 2 };
template <class Compute, class Ensure, class Table>
static int auto_launch(mi355_ctx *ctx, AutoPick &A, size_t n_vec, Compute &&compute, Ensure &&ensure, Table &&table);
static int launch_table_gather(mi355_ctx *ctx, const uint32_t *t, const uint8_t *d_src, uint8_t *d_dst, const Rgba8Geom &geo, int width, size_t rows, int morton);
static int launch_table_raw(mi355_ctx *ctx, const uint32_t *t, const uint8_t *d_src, uint8_t *d_dst, const Rgba8Geom &geo, int width, size_t rows, int morton,
                            TableKernel sub = kTableGather, int *last_sub = nullptr);
static int launch_table(mi355_ctx *ctx, int which, const uint8_t *d_src, uint8_t *d_dst, const Rgba8Geom &geo, int width, size_t rows, int morton,
                        const mi355_hsv_settings *hs, TableKernel sub = kTableGather) {
  int rc = table_ensure(ctx, which, morton, hs);
  if (rc) return rc;
  return launch_table_raw(ctx, ctx->lut.d_table[which], d_src, d_dst, geo, width, rows, morton, sub, &ctx->lut.last_sub[which]);
}

// Both kernels read the same table and give the same bytes; which is faster depends on WHERE THE PIXELS COME FROM: the gather
// kernel wins when its input was just written by the previous launch and is still on-die (Infinity Cache; behind hsvfilter
// 0.086-0.089 against 0.093-0.099 ms at amp 0, 0.114 against 0.142 at +-4), the LDS-cached kernel when the frames come from HBM
// (0.096 against 0.117; 0.143 against 0.176; 0.171 against 0.231 at +-8 - every box of rounds 5 and 6). Rounds 4-5
// decided this by a second, nested measured choice (autopick.hpp inside autopick.hpp): its samples of the kernel not in use
// were up to 1024 launches old, so a choice made on input from HBM survived on on-die input and the other way round
// (BENCH_r05 content_sweep.amp4: 35.3 k instead of 37.6 k). The provenance is known exactly (note_written / recently_written):
// kTableEither is that rule now, with no state, no probes and nothing to settle.
static int launch_table_raw(mi355_ctx *ctx, const uint32_t *t, const uint8_t *d_src, uint8_t *d_dst, const Rgba8Geom &geo, int width, size_t rows, int morton,
                            TableKernel sub, int *last_sub) {
  const bool window_ok = morton && sub != kTableGather && width % 4 == 0 && window_applicable(ctx, (unsigned)width / 4, geo.dw4, rows);
  const size_t src_bytes = (size_t)geo.sw4 * 16 * rows;
  const bool window = window_ok && (sub == kTableWindow || !recently_written(ctx->device, d_src, src_bytes));
  if (last_sub) *last_sub = window ? 1 : 0;
  int rc;
  if (window) {
    ctx->lut.last_kernel = "colorlut_window_kernel";
    rc = launch_window_table(ctx, t, d_src, d_dst, (unsigned)width / 4, geo.sw4, geo.dw4, rows);
  } else {
    rc = launch_table_gather(ctx, t, d_src, d_dst, geo, width, rows, morton);
  }
  return rc;
}

static int launch_table_gather(mi355_ctx *ctx, const uint32_t *t, const uint8_t *d_src, uint8_t *d_dst, const Rgba8Geom &geo, int width, size_t rows, int morton) {
  const size_t n_vec = geo.n_vec;
  ctx->lut.last_kernel = "colorlut_table_tiled_kernel";
  if (width % 4 == 0 && width >= 128 && rows < (1u << 30)) {
    const unsigned w4 = (unsigned)width / 4;
    // 256-pixel patches unless 128-pixel ones waste fewer masked lanes in the last column
    const unsigned pad64 = (w4 + 63) / 64 * 64 - w4, pad32 = (w4 + 31) / 32 * 32 - w4;
    const unsigned tw4 = pad64 <= pad32 ? 64 : 32;
    const unsigned n_cols = (w4 + tw4 - 1) / tw4, rpb = 8 * (64 / tw4);
    const size_t n_tiles = (size_t)n_cols * ((rows + rpb - 1) / rpb);
    if (n_tiles < (1u << 31)) {
      const size_t grid = n_tiles;
#define MI355_LT(M, T) hipLaunchKernelGGL((colorlut_table_tiled_kernel<M, T>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint4 *)d_src, \
                                          (uint4 *)d_dst, w4, geo.sw4, geo.dw4, (unsigned)rows, n_cols, t)
      if (morton && tw4 == 64) MI355_LT(true, 64);
      else if (morton) MI355_LT(true, 32);
      else if (tw4 == 64) MI355_LT(false, 64);
      else MI355_LT(false, 32);
#undef MI355_LT
      return check_hip(ctx, hipGetLastError(), "colorlut table kernel launch");
    }
  }
  if (!geo.packed) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: table kernel geometry");  // rgba8_geom only admits padded rows the tiled kernel takes
  ctx->lut.last_kernel = "colorlut_table_kernel";
  size_t grid = (size_t)ctx->n_cu * 16;
  const size_t max_blocks = n_vec / 512 + 1;  // a block's four waves take one 128-group chunk each per iteration
  if (grid > max_blocks) grid = max_blocks;
  if (morton) hipLaunchKernelGGL((colorlut_table_kernel<true>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_vec, t);
  else hipLaunchKernelGGL((colorlut_table_kernel<false>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, (const uint4 *)d_src, (uint4 *)d_dst, n_vec, t);
  return check_hip(ctx, hipGetLastError(), "colorlut table kernel launch");
}

// MI355_FLAG_LUT_VARIANT: 0 = auto (default). The interpolating ("compute") kernel and the table kernel are both exact;
// which one is faster depends on the content (the table kernel's gathers need colour locality: 0.18 vs 0.25 ms per
// 8x4K on natural-like frames, 1.2 vs 0.40 ms on uniform noise), on the LUT size (the table kernel does not care) and on
// the launch size. Auto keeps a per-pixel-group time for each kind, measured with event pairs that are recorded around
// every n-th launch (n = 8..32, about one sample per 128 Mpixel) and read back at the start of a later launch, by polling
// only (hipEventQuery): nothing here ever blocks the calling thread on the device:
//   * after a LUT load four measured launches run compute, compute, table build + table, table (unmeasured launches of
//     the same kind in between while a measurement is still in flight); the first measurement of each kind is not used
//     (its interval holds one-off costs: code upload, cold caches); from then on the kind with the smaller time serves
//     the launches;
//   * the kind in use keeps being sampled, so content that turns hostile to the table (its time rises above the compute
//     kernel's last time) flips the choice at the next sample;
//   * the kind NOT in use is tried again after `probe_period` launches (64, doubling up to 1024 while the answer stays
//     the same, back to 64 when it changes);
//   * measurements are dropped when the launch size changes by more than 2x.
// 1, 2, 6 = compute kernels only (6 = the default compute kernel), 4 / 5 = table kernel only (linear / Morton index).
// The same policy, with its own state and table, serves the fused hsvfilter -> colorlut entry point.
// The policy itself is autopick.hpp (HIP-free, unit-tested on the CPU); this is the mechanism around it.
static void auto_harvest(AutoPick &A) {
  if (A.pending_kind < 0) return;
  // poll, never wait: the streaming thread must not block on the device for a bookkeeping measurement
  if (hipEventQuery(A.ev1) != hipSuccess) {
    (void)hipGetLastError();
    return;  // not finished yet: look again at a later call
  }
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, A.ev0, A.ev1) != hipSuccess) { (void)hipGetLastError(); ms = 0.0f; }
  auto_complete(A, (double)ms);
}

template <class Compute, class Ensure, class Table>
static int auto_launch(mi355_ctx *ctx, AutoPick &A, size_t n_vec, Compute &&compute, Ensure &&ensure, Table &&table) {
  int rc;
  if (A.table_unavailable) return compute();
  if (!A.ev0) {
    if ((rc = check_hip(ctx, hipEventCreate(&A.ev0), "hipEventCreate"))) return rc;
    if ((rc = check_hip(ctx, hipEventCreate(&A.ev1), "hipEventCreate"))) return rc;
  }
  auto_harvest(A);
  const AutoDecision D = auto_decide(A, n_vec);
  if (D.kind == 1 && (rc = ensure())) {  // a table build stays outside the measurement
    // no memory for the 64 MiB table (or the build failed): this entry point stays on the compute kernel for good
    auto_give_up_table(A);
    (void)hipGetLastError();
    return compute();
  }
  if (D.measure && (rc = check_hip(ctx, hipEventRecord(A.ev0, ctx->stream), "hipEventRecord"))) return rc;
  if ((rc = D.kind ? table() : compute())) return rc;
  if (D.measure) {
    if ((rc = check_hip(ctx, hipEventRecord(A.ev1, ctx->stream), "hipEventRecord"))) return rc;
    auto_sampled(A, D, n_vec);
  }
  return MI355_OK;
}

// below 64 K pixels a launch is all fixed cost and a 64 MiB table is not worth building
constexpr size_t kAutoMinVec = 16384;
// MI355_FLAG_LUT_VARIANT values that pin a table kernel: 4 linear index, 5 Morton index (gather kernels), 8 Morton index
// through the LDS-cached kernel (colorlut_window.hip) where the launch is large enough for it
// 9 = Morton table, the kernel that reads it chosen by measurement as in auto
static bool table_variant(int v) { return v == 4 || v == 5 || v == 8 || v == 9; }
static TableKernel table_variant_kernel(int v) { return v == 8 ? kTableWindow : (v == 9 ? kTableEither : kTableGather); }

int launch_colorlut(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                    size_t dst_pitch, int dst_stride, int n_frames, int width, int height, int format) {
  LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  Rgba8Geom geo;
  const bool table_ok = format == MI355_FMT_RGBA && !ctx->force_generic &&
                        rgba8_geom(d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, &geo);
  const size_t n_vec = geo.n_vec;
  const int v = ctx->lut_variant;
  auto compute = [&]() { return launch_colorlut_compute(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, format); };
  if (table_ok && table_variant(v))
    return launch_table(ctx, 0, d_src, d_dst, geo, width, (size_t)n_frames * height, v == 4 ? 0 : 1, nullptr, table_variant_kernel(v));
  if (!table_ok || v != 0 || n_vec < kAutoMinVec) return compute();
  return auto_launch(ctx, L.pick[0], n_vec, compute, [&]() { return table_ensure(ctx, 0, 1, nullptr); },
                     [&]() { return launch_table(ctx, 0, d_src, d_dst, geo, width, (size_t)n_frames * height, 1, nullptr, kTableEither); });
}

// The fused entry point: hsvfilter -> colorlut is also a function of the colour alone, so the same memoisation applies
// with a table built by the fused compute path under the call's hsv settings. A table is only built for settings that
// have been the same for kStableCalls consecutive calls (an animated property would otherwise rebuild it every buffer).
constexpr unsigned kStableCalls = 8;
int launch_hsv_colorlut(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                        size_t dst_pitch, int dst_stride, int n_frames, int width, int height,
                        const mi355_hsv_settings &hs) {
  LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  Rgba8Geom geo;
  const bool table_ok = !ctx->force_generic && rgba8_geom(d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, &geo);
  const size_t n_vec = geo.n_vec;
  const int v = ctx->lut_variant;
  auto compute = [&]() { return launch_hsv_colorlut_compute(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, hs, false); };
  if (table_ok && table_variant(v))
    return launch_table(ctx, 1, d_src, d_dst, geo, width, (size_t)n_frames * height, v == 4 ? 0 : 1, &hs, table_variant_kernel(v));
  if (!table_ok || v != 0 || n_vec < kAutoMinVec) return compute();
  if (same_hs(hs, L.seen_hs)) { if (L.seen_stable < kStableCalls) L.seen_stable++; }
  else { L.seen_hs = hs; L.seen_stable = 0; }
  const bool have = L.table_ref[1] && L.table_morton[1] == 1 && same_hs(L.table_hs, hs);
  // (Settings that change with every call - a hue shift animated by a controller - never settle and keep the arithmetic kernel. Round 6
  // measured the alternative for large launches, a table built per call: 0.219 ms per 8 x 4K against 0.19 - the build is the fused
  // arithmetic over ALL 2^24 colours, the one frame on which every cache of the arithmetic kernels misses: profiles/r06_fused_animated_probe.txt)
  if (!have && L.seen_stable < kStableCalls) return compute();
  if (!have && L.pick[1].learn > 2) {  // the table kernel's time belongs to the old table only loosely: measure it again
    L.pick[1].t_table = 0.0;
    L.pick[1].learn = 2;
  }
  return auto_launch(ctx, L.pick[1], n_vec, compute, [&]() { return table_ensure(ctx, 1, 1, &hs); },
                     [&]() { return launch_table(ctx, 1, d_src, d_dst, geo, width, (size_t)n_frames * height, 1, &hs, kTableEither); });
}

// ---- frames of several streams in one launch (group.hip)
int colorlut_multi_table(mi355_ctx *ctx, const uint32_t **table_out) {
  LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  int rc = table_ensure(ctx, 0, 1, nullptr);
  if (rc) return rc;
  *table_out = L.d_table[0];
  return MI355_OK;
}

// the composed hsvfilter -> colorlut table of these settings (built, or found in the registry, if the context does not hold it)
int colorlut_multi_fused_table(mi355_ctx *ctx, const mi355_hsv_settings *hs, const uint32_t **table_out) {
  LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (!hs) return set_error(ctx, MI355_ERR_INVALID_ARG, "null settings");
  int rc = table_ensure(ctx, 1, 1, hs);
  if (rc) return rc;
  *table_out = L.d_table[1];
  return MI355_OK;
}
int launch_colorlut_multi(mi355_ctx *ctx, hipStream_t stream, const uint32_t *table, uint8_t *const *srcs, uint8_t *const *dsts, int n_frames, int width,
                          int height, bool from_hbm) {
  if (n_frames < 1 || n_frames > kMultiFrames || width % 4 != 0 || width < 128 || height <= 0) return MI355_ERR_UNSUPPORTED;
  MultiFramePtrs s{}, d{};
  for (int f = 0; f < n_frames; f++) {
    if (!srcs[f] || !dsts[f] || (uintptr_t)srcs[f] % 16 != 0 || (uintptr_t)dsts[f] % 16 != 0) return MI355_ERR_UNSUPPORTED;
    s.p[f] = srcs[f];
    d.p[f] = dsts[f];
  }
  const unsigned w4 = (unsigned)width / 4;
  // the provenance rule of launch_table_raw: frames that come from HBM (the fused form: each stream's own source, uploaded or
  // decoded, not written by a launch just before) go through the LDS-cached kernel where the launch is large enough for it
  if (from_hbm && ctx->lut_variant == 0 && window_multi_applicable(ctx, w4, (size_t)height, n_frames)) {
    ctx->lut.last_kernel = "colorlut_window_kernel";
    return launch_window_table_multi(ctx, stream, table, srcs, dsts, n_frames, w4, (size_t)height);
  }
  ctx->lut.last_kernel = "colorlut_table_tiled_multi_kernel";
  const unsigned pad64 = (w4 + 63) / 64 * 64 - w4, pad32 = (w4 + 31) / 32 * 32 - w4;
  const unsigned tw4 = pad64 <= pad32 ? 64 : 32;
  const unsigned n_cols = (w4 + tw4 - 1) / tw4, rpb = 8 * (64 / tw4);
  const size_t tiles_per_frame = (size_t)n_cols * (((size_t)height + rpb - 1) / rpb);
  const size_t grid = tiles_per_frame * (size_t)n_frames;
  if (grid >= (1u << 31)) return MI355_ERR_UNSUPPORTED;
  if (tw4 == 64) hipLaunchKernelGGL((colorlut_table_tiled_multi_kernel<64>), dim3((unsigned)grid), dim3(256), 0, stream, s, d, w4, (unsigned)height, n_cols, (unsigned)tiles_per_frame, table);
  else hipLaunchKernelGGL((colorlut_table_tiled_multi_kernel<32>), dim3((unsigned)grid), dim3(256), 0, stream, s, d, w4, (unsigned)height, n_cols, (unsigned)tiles_per_frame, table);
  return check_hip(ctx, hipGetLastError(), "colorlut multi-frame table kernel launch");
}

// hsvfilter alone is a function of the colour too, and MI355_FLAG_HSV_TABLE = 1 / 2 runs it through the same machinery
// (auto choice for all settings / table only; the table is built by the arithmetic kernel, in place over the colours of all slots, once the
// settings and the byte order have been the same for kStableCalls calls; colour-first packed 4-byte formats only: RGBx,
// RGBA, BGRx, BGRA). By default (0) only settings outside the FAST envelope (huge / non-finite hue-shift: GENERIC
// arithmetic, 0.336 ms per 32 x 1080p against 0.117 ms) get the auto choice; for FAST settings it is off (3 = off for
// all settings): the arithmetic kernel is close to the streaming floor for large launches from
// HBM (0.107-0.12 ms per 8x4K against 0.132-0.137 ms for the table kernel), and for per-buffer launches, where the table
// kernel measures faster on its own (0.0025 against 0.0029 ms per Mpixel at one 4K frame per launch), a second 64 MiB
// table competes with colorlut's for L2 and the Infinity Cache: hsvfilter -> colorlut at one frame per launch drops from
// 26.3-27.4 k to 24.6 k frames/s with both tables live.
void hsv_table_release(mi355_ctx *ctx) {
  HsvTable &T = ctx->hsv_table;
  shared_table_release(ctx, &T.table_ref);
  if (T.pick.ev0) (void)hipEventDestroy(T.pick.ev0);
  if (T.pick.ev1) (void)hipEventDestroy(T.pick.ev1);
  T = HsvTable{};
}

static int hsv_table_ensure(mi355_ctx *ctx, const PixFmt &fmt, const mi355_hsv_settings &hs) {
  HsvTable &T = ctx->hsv_table;
  if (T.valid && T.bgr == fmt.bgr && same_hs(T.hs, hs)) return MI355_OK;
  T.valid = false;
  T.d_table = nullptr;  // (the reference stays: shared_table_acquire reuses the buffer if nobody else holds it)
  std::string key("H");
  key.push_back((char)('0' + fmt.bgr));
  key.append((const char *)&hs, sizeof(hs));
  int rc = shared_table_acquire(ctx, key, &T.table_ref, &T.d_table, [&](uint32_t *tab) {
    hipLaunchKernelGGL(table_domain_kernel, dim3(kTableEntries / 256), dim3(256), 0, ctx->stream, tab, 1);
    return launch_hsvfilter_compute(ctx, (uint8_t *)tab, 1, 0, 4096, 4096, 4096 * 4, fmt, hs);
  });
  if (rc) { T.table_ref.reset(); T.d_table = nullptr; return rc; }
  T.valid = true;
  T.bgr = fmt.bgr;
  T.hs = hs;
  return MI355_OK;
}

int launch_hsvfilter(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width, int height, int stride, const PixFmt &fmt,
                     const mi355_hsv_settings &hs) {
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  HsvTable &T = ctx->hsv_table;
  auto compute = [&]() { return launch_hsvfilter_compute(ctx, d_data, n_frames, frame_pitch, width, height, stride, fmt, hs); };
  Rgba8Geom geo;
  // mode 0 (default): only settings that need the GENERIC arithmetic (3x slower than the FAST kernels) are candidates
  const bool candidate = ctx->hsv_table_mode == 1 || ctx->hsv_table_mode == 2 || (ctx->hsv_table_mode == 0 && hsv_variant_for(hs, false, true) < 0);
  const bool table_ok = candidate && !ctx->force_generic && fmt.pixel_stride == 4 && fmt.first == 0 &&
                        rgba8_geom(d_data, frame_pitch, stride, d_data, frame_pitch, stride, n_frames, width, height, &geo);
  const size_t n_vec = geo.n_vec;
  T.last_table = false;
  if (!table_ok) return compute();
  const size_t rows = (size_t)n_frames * height;
  if (ctx->hsv_table_mode == 2) {
    int rc = hsv_table_ensure(ctx, fmt, hs);
    T.last_table = true;
    return rc ? rc : launch_table_raw(ctx, T.d_table, d_data, d_data, geo, width, rows, 1, kTableEither, nullptr);
  }
  if (n_vec < kAutoMinVec) return compute();
  if (same_hs(hs, T.seen_hs) && fmt.bgr == T.seen_bgr) { if (T.seen_stable < kStableCalls) T.seen_stable++; }
  else { T.seen_hs = hs; T.seen_bgr = fmt.bgr; T.seen_stable = 0; }
  const bool have = T.valid && T.bgr == fmt.bgr && same_hs(T.hs, hs);
  if (!have && T.seen_stable < kStableCalls) return compute();
  if (!have && T.pick.learn > 2) {
    T.pick.t_table = 0.0;
    T.pick.learn = 2;
  }
  return auto_launch(ctx, T.pick, n_vec, compute, [&]() { return hsv_table_ensure(ctx, fmt, hs); },
                     [&]() { T.last_table = true; return launch_table_raw(ctx, T.d_table, d_data, d_data, geo, width, rows, 1, kTableEither, nullptr); });
}

}  // namespace mi355

// Runs the kernel-choice policy (autopick.hpp) against a scripted device: call i has n_vec[i] pixel groups; if it is
// measured, its interval is ms_compute[i] or ms_table[i] depending on the kind it ran, and the measurement becomes readable
// `lag` calls later (the policy never waits). kind_out[i] = 0 / 1, measured_out[i] = 0 / 1.
// No device, no context: host logic only (tests/test_autopick.py).
// Runs the content-watch policy (brickwatch.hpp) against a scripted stream: call i would produce the miss / slow fractions
// miss0[i], slow0[i] on the 32-set brick kernel and miss1[i], slow1[i] on the 64-set one; a snapshot covers kWatchSnapEvery
// consecutive launches at one level and becomes readable `lag` calls after its last launch. level_out[i] = 0 / 1 / 2.
extern "C" int mi355_selftest_brickwatch(int n_calls, const double *miss0, const double *slow0, const double *miss1, const double *slow1, int lag,
                                         int *level_out) {
  if (n_calls < 0 || !miss0 || !slow0 || !miss1 || !slow1 || !level_out || lag < 0) return MI355_ERR_INVALID_ARG;
  // lag: bits 0-15 the snapshot lag in launches; bit 16: level 1 is the block-shared cache (its thresholds); bits 17-18: the
  // lowest level worth running (small launches)
  const bool shared1 = (lag >> 16) & 1;
  const int min_level = (lag >> 17) & 3;
  lag &= 0xffff;
  mi355::BrickWatch W;
  int level_since = -1, launches_since = 0, pend_level = -1, pend_ready = 0;
  double acc_m = 0, acc_s = 0, pend_m = 0, pend_s = 0;
  for (int i = 0; i < n_calls; i++) {
    if (pend_level >= 0 && i >= pend_ready) { const int l = pend_level; pend_level = -1; mi355::watch_snapshot(W, l, pend_m, pend_s, shared1); }
    const int level = mi355::watch_level(W, pend_level < 0, min_level);
    level_out[i] = level;
    if (level == 2) { level_since = 2; continue; }
    if (level != level_since) { level_since = level; launches_since = 0; acc_m = acc_s = 0; }
    acc_m += level ? miss1[i] : miss0[i];
    acc_s += level ? slow1[i] : slow0[i];
    launches_since++;
    if (pend_level < 0 && launches_since >= (W.probing == level ? 1 : (int)mi355::kWatchSnapEvery)) {
      pend_level = level; pend_m = acc_m / launches_since; pend_s = acc_s / launches_since; pend_ready = i + 1 + lag;
      launches_since = 0; acc_m = acc_s = 0;
    }
  }
  return MI355_OK;
}

extern "C" int mi355_selftest_autopick(int n_calls, const uint64_t *n_vec, const double *ms_compute, const double *ms_table, int lag,
                                       int *kind_out, int *measured_out) {
  if (n_calls < 0 || !n_vec || !ms_compute || !ms_table || !kind_out || lag < 0) return MI355_ERR_INVALID_ARG;
  mi355::AutoPolicy A;
  double pending_ms = 0.0;
  for (int i = 0; i < n_calls; i++) {
    const size_t nv = (size_t)n_vec[i];
    if (A.pending_kind >= 0 && A.calls - A.pending_call >= (unsigned)lag) mi355::auto_complete(A, pending_ms);
    const mi355::AutoDecision D = mi355::auto_decide(A, nv);
    kind_out[i] = D.kind;
    if (measured_out) measured_out[i] = D.measure ? 1 : 0;
    if (D.measure) {
      mi355::auto_sampled(A, D, nv);
      pending_ms = D.kind ? ms_table[i] : ms_compute[i];
    }
  }
  return MI355_OK;
}
