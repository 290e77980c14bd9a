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
// Algorithmic traffic of both: 4 B read + 4 B written per pixel (RGBA8); the LUT is a
// cache/LDS-resident constant.
#include "internal.hpp"
#include "exact_math.hpp"

#include <cmath>
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
// LDS plane layout for channel c (floats): plane[x + y*S + z*S*S] = cell(x,y,z).c for the S^3 real
// cells, followed by S*S + S + 4 zero floats. The padding lets every lane read its 8 corners at the
// fixed offsets {0,1,S,S+1,S^2,S^2+1,S^2+S,S^2+S+1} from its base without clamping x1/y1/z1:
// whenever the reference clamps (x0 == S-1, i.e. x == S-1 exactly) the interpolation weight is 0,
// and a + (b - a)*0 == a for every finite b with |b - a| < inf — guaranteed by the load-time check
// that all entries are finite with |v| <= 1e30 (a sign-of-zero difference cannot reach a non-zero
// result and both zeros convert to byte 0).
//
// Block = NT lanes, each owning P = 4*P4 pixels held in VGPRs: packed pixel, LDS byte base, tx, ty, tz.
// Per tile: load pixels, compute coordinates once, then for c in R,G,B: stage plane c into LDS,
// interpolate channel c for all P pixels, insert the byte. One 16 B load + one 16 B store per 4 pixels.

template <bool UNIT>
__device__ __forceinline__ void lut_axis(float c8, float scale, float offset, float sm1, uint32_t &i0, float &t) {
  float n = div255_u8(c8);
  if constexpr (!UNIT) {
    // finite scale/offset: inherent clamp == max-then-min (no NaN can occur)
    n = fminf(fmaxf(n * scale + offset, 0.0f), 1.0f);
  }
  const float x = n * sm1;      // in [0, S-1]
  i0 = (uint32_t)x;             // floor; never exceeds S-1 so `.min(max_idx)` is the identity
  t = __builtin_amdgcn_fractf(x);  // x - floor(x), exact
}

template <int S_CONST>
__device__ __forceinline__ float lut_tri_lds(const float *__restrict__ lut, uint32_t base_bytes, int S_rt, float tx,
                                             float ty, float tz) {
  const int S = S_CONST > 0 ? S_CONST : S_rt;
  const float *L = (const float *)((const char *)lut + base_bytes);
  const float a0 = L[0], a1 = L[1];
  const float b0 = L[S], b1 = L[S + 1];
  const float c0 = L[S * S], c1 = L[S * S + 1];
  const float d0 = L[S * S + S], d1 = L[S * S + S + 1];
  const float c00 = lerp1(a0, a1, tx), c10 = lerp1(b0, b1, tx);
  const float c01 = lerp1(c0, c1, tx), c11 = lerp1(d0, d1, tx);
  return lerp1(lerp1(c00, c10, ty), lerp1(c01, c11, ty), tz);
}

// round-half-away(clamp(v,0,1)*255) as u8 == (trunc(clamp(v,0,1)*510) + 1) >> 1 (2*RN(c*255) == RN(c*510)).
__device__ __forceinline__ uint32_t float_to_u8_fast(float v) {
  const float c = fminf(fmaxf(v, 0.0f), 1.0f);
  return ((uint32_t)(c * 510.0f) + 1u) >> 1;
}

template <int NT, int P4, bool UNIT, int S_CONST>
__global__ __launch_bounds__(NT) void colorlut3d_lds_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                            size_t n_groups, const float *__restrict__ planar,
                                                            uint32_t plane_floats, LutK k) {
  extern __shared__ float lut[];
  constexpr int P = P4 * 4;
  const int S = S_CONST > 0 ? S_CONST : k.size;
  const float sm1 = (float)S - 1.0f;
  const size_t tile_groups = (size_t)NT * P4;
  const size_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;

  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t px[P];
    uint32_t base[P];
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
      uint32_t x0, y0, z0;
      lut_axis<UNIT>((float)(px[i] & 0xffu), k.scale[0], k.offset[0], sm1, x0, tx[i]);
      lut_axis<UNIT>((float)((px[i] >> 8) & 0xffu), k.scale[1], k.offset[1], sm1, y0, ty[i]);
      lut_axis<UNIT>((float)((px[i] >> 16) & 0xffu), k.scale[2], k.offset[2], sm1, z0, tz[i]);
      base[i] = (x0 + (uint32_t)S * (y0 + (uint32_t)S * z0)) * 4u;
    }
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
      __syncthreads();  // everyone is done reading the previous plane
      {
        const float4 *s4 = (const float4 *)(planar + (size_t)c * plane_floats);
        float4 *d4 = (float4 *)lut;
        const uint32_t n4 = plane_floats / 4;
        for (uint32_t i = threadIdx.x; i < n4; i += NT) d4[i] = s4[i];
      }
      __syncthreads();
      const uint32_t shift = 8u * (uint32_t)c;
      const uint32_t keep = ~(0xffu << shift);
#pragma unroll
      for (int i = 0; i < P; i++) {
        const float o = lut_tri_lds<S_CONST>(lut, base[i], S, tx[i], ty[i], tz[i]);
        px[i] = (px[i] & keep) | (float_to_u8_fast(o) << shift);
      }
    }
#pragma unroll
    for (int j = 0; j < P4; j++) {
      const size_t g = g0 + (size_t)j * NT;
      if (g < n_groups) dst[g] = make_uint4(px[4 * j + 0], px[4 * j + 1], px[4 * j + 2], px[4 * j + 3]);
    }
  }
}

// ---------------------------------------------------------------- host side: LUT upload + dispatch

static constexpr size_t kLdsBytes = 160 * 1024;

static size_t planar_plane_floats(size_t S) {
  size_t n = S * S * S + S * S + S + 4;
  return (n + 3) & ~(size_t)3;
}

void lut_release(mi355_ctx *ctx) {
  if (ctx->lut.d_cells) (void)hipFree(ctx->lut.d_cells);
  if (ctx->lut.d_planar) (void)hipFree(ctx->lut.d_planar);
  ctx->lut = LutDevice{};
}

int lut_upload(mi355_ctx *ctx, int is3d, size_t size, const float *table, const float scale[3],
               const float offset[3]) {
  lut_release(ctx);
  LutDevice &L = ctx->lut;
  L.is3d = is3d;
  L.size = (int)size;
  for (int c = 0; c < 3; c++) { L.scale[c] = scale[c]; L.offset[c] = offset[c]; }
  const size_t n_floats = is3d ? 4 * size * size * size : 3 * size;
  int rc = check_hip(ctx, hipMalloc((void **)&L.d_cells, n_floats * sizeof(float)), "hipMalloc(lut cells)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpy(L.d_cells, table, n_floats * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(lut cells)");
  if (rc) return rc;

  L.unit_domain = true;
  bool domain_finite = true;
  for (int c = 0; c < 3; c++) {
    if (!(scale[c] == 1.0f && offset[c] == 0.0f)) L.unit_domain = false;
    if (!std::isfinite(scale[c]) || !std::isfinite(offset[c])) domain_finite = false;
  }
  L.lds_ok = false;
  if (is3d && domain_finite) {
    const size_t pf = planar_plane_floats(size);
    if (pf * sizeof(float) <= kLdsBytes) {
      bool bounded = true;
      const size_t cells = size * size * size;
      for (size_t i = 0; i < cells && bounded; i++)
        for (int c = 0; c < 3; c++) {
          const float v = table[i * 4 + c];
          if (!(std::fabs(v) <= 1e30f)) { bounded = false; break; }
        }
      if (bounded) {
        std::vector<float> planar(3 * pf, 0.0f);
        for (int c = 0; c < 3; c++)
          for (size_t i = 0; i < cells; i++) planar[(size_t)c * pf + i] = table[i * 4 + c];
        rc = check_hip(ctx, hipMalloc((void **)&L.d_planar, planar.size() * sizeof(float)), "hipMalloc(lut planar)");
        if (rc) return rc;
        rc = check_hip(ctx, hipMemcpy(L.d_planar, planar.data(), planar.size() * sizeof(float), hipMemcpyHostToDevice),
                       "hipMemcpy(lut planar)");
        if (rc) return rc;
        L.planar_plane_floats = pf;
        L.lds_ok = true;
      }
    }
  }
  L.loaded = true;
  return MI355_OK;
}

template <int NT, int P4, bool UNIT, int S_CONST>
static int launch_lds_variant(mi355_ctx *ctx, const uint4 *src, uint4 *dst, size_t n_groups, const LutK &k) {
  auto kern = colorlut3d_lds_kernel<NT, P4, UNIT, S_CONST>;
  const LutDevice &L = ctx->lut;
  const size_t lds = L.planar_plane_floats * sizeof(float);
  int rc = check_hip(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                     "hipFuncSetAttribute(max dynamic LDS)");
  if (rc) return rc;
  const size_t tile_groups = (size_t)NT * P4;
  size_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;
  size_t grid = n_tiles < (size_t)ctx->n_cu ? n_tiles : (size_t)ctx->n_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, ctx->stream, src, dst, n_groups, L.d_planar,
                     (uint32_t)L.planar_plane_floats, k);
  return check_hip(ctx, hipGetLastError(), "colorlut3d_lds kernel launch");
}

int launch_colorlut(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                    size_t dst_pitch, int dst_stride, int n_frames, int width, int height, int format) {
  const LutDevice &L = ctx->lut;
  if (!L.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  LutK k;
  for (int c = 0; c < 3; c++) { k.scale[c] = L.scale[c]; k.offset[c] = L.offset[c]; }
  k.size = L.size;

  const bool rgba8 = (format == MI355_FMT_RGBA);
  if (rgba8 && L.is3d && L.lds_ok && !ctx->force_generic) {
    const size_t row_bytes = (size_t)width * 4;
    const bool contiguous = (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                            (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
    const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
    if (contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && (total_bytes % 16 == 0)) {
      const size_t n_groups = total_bytes / 16;
      const uint4 *s = (const uint4 *)d_src;
      uint4 *d = (uint4 *)d_dst;
      constexpr int NT = 512, P4 = 6;
      if (L.size == 33) {
        return L.unit_domain ? launch_lds_variant<NT, P4, true, 33>(ctx, s, d, n_groups, k)
                             : launch_lds_variant<NT, P4, false, 33>(ctx, s, d, n_groups, k);
      }
      return L.unit_domain ? launch_lds_variant<NT, P4, true, 0>(ctx, s, d, n_groups, k)
                           : launch_lds_variant<NT, P4, false, 0>(ctx, s, d, n_groups, k);
    }
  }

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

}  // namespace mi355
