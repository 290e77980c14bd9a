// dssim_kernels.hip — gfx950 kernels for videocompare's optional Dssim engine (BASELINE config 5: "videocompare SSIM").
//
// Reference path replaced: HasherEngine::hash_image / compare with HashAlgorithm::Dssim (cargo feature `dssim`,
// video/videofx/src/videocompare/hashed_image.rs:41-53,66-70,92-94) -> crate dssim-core 3.4.0
// `Dssim::new / create_image_rgb / create_image_rgba / compare` (sources not in the reference tree; the published
// multi-scale SSIM-in-LAB algorithm is restated in oracle/dssim_restate.py, PARITY UNPINNED; the reference test pins
// identical frames -> 0.0).
//
// A DssimImage lives on the device: 5 scales x 3 LAB planes x {img, mu, img_sq_blur} f32. All kernels are pointwise or
// 3x3 stencils (HBM/L2-bound), the scores are block-reduced f64 sums finished on the host in a fixed order:
//   dssim_downsample_u8_kernel / dssim_downsample_kernel   u8 sRGB(A) -> premultiplied linear float4 at half size; 2x2
//                               box average chain (scale 0's linear values are converted on the fly by the scale kernel)
//   dssim_scale_fused_kernel    per scale: LAB conversion (polynomial + 2x Halley cube root), chroma pre-blur and the
//                               mu / img_sq_blur planes, all four blur passes through LDS tiles (halo 4)
//   dssim_compare_fused_kernel  per scale: blur(img1*img2) through LDS (halo 2) + SSIM map + f64 block partials
//   dssim_avg / absdev2 / sum   deterministic reductions; one D2H of 15 doubles per comparison
// Every f32 expression is written in the operation order of the restatement (-ffp-contract=off), so the per-pixel maps
// are bit-identical to it; only the f64 reductions differ in summation order.
#include "internal.hpp"

#include <cmath>
#include <vector>

namespace mi355 {

constexpr int kDssimScales = 5;
static const double kDssimWeights[kDssimScales] = {0.028, 0.197, 0.322, 0.298, 0.155};

struct DssimScale {
  int w = 0, h = 0;
  float *img[3] = {nullptr, nullptr, nullptr}, *mu[3] = {nullptr, nullptr, nullptr}, *sq[3] = {nullptr, nullptr, nullptr};
};

}  // namespace mi355

struct mi355_dssim_image {
  int n_scales = 0;
  mi355::DssimScale s[mi355::kDssimScales];
  float *pool = nullptr;
  size_t pool_bytes = 0;
};

namespace mi355 {

__global__ __launch_bounds__(256) void dssim_downsample_kernel(const float4 *__restrict__ src, int w, int h, float4 *__restrict__ dst) {
  const int w2 = w / 2, h2 = h / 2;
  const size_t n = (size_t)w2 * h2, gs = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
    const int y = (int)(i / w2), x = (int)(i - (size_t)y * w2);
    const float4 a = src[(size_t)(2 * y) * w + 2 * x], b = src[(size_t)(2 * y) * w + 2 * x + 1];
    const float4 c = src[(size_t)(2 * y + 1) * w + 2 * x], d = src[(size_t)(2 * y + 1) * w + 2 * x + 1];
    dst[i] = make_float4((((a.x + b.x) + c.x) + d.x) * 0.25f, (((a.y + b.y) + c.y) + d.y) * 0.25f, (((a.z + b.z) + c.z) + d.z) * 0.25f,
                         (((a.w + b.w) + c.w) + d.w) * 0.25f);
  }
}

// scale 1's linear image straight from the packed bytes: each of the four source pixels is converted (gamma table,
// premultiplied by alpha/255) and averaged in dssim_downsample_kernel's order, so the full-size float4 image (133 MB for 4K)
// is never written or read.
__global__ __launch_bounds__(256) void dssim_downsample_u8_kernel(const uint8_t *__restrict__ src, int stride, int w, int h, int channels,
                                                                  const float *__restrict__ lut, float4 *__restrict__ dst) {
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const int w2 = w / 2, h2 = h / 2;
  const size_t n = (size_t)w2 * h2, gs = (size_t)gridDim.x * 256;
  const bool wide = channels == 4 && ((uintptr_t)src % 8 == 0) && (stride % 8 == 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
    const int y = (int)(i / w2), x = (int)(i - (size_t)y * w2);
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const uint8_t *p = src + (size_t)(2 * y + r) * stride + (size_t)(2 * x) * channels;
      uint8_t b[8];
      if (wide) {
        const uint2 q = *(const uint2 *)p;
        b[0] = q.x & 255; b[1] = (q.x >> 8) & 255; b[2] = (q.x >> 16) & 255; b[3] = q.x >> 24;
        b[4] = q.y & 255; b[5] = (q.y >> 8) & 255; b[6] = (q.y >> 16) & 255; b[7] = q.y >> 24;
      } else if (channels == 4) {
#pragma unroll
        for (int j = 0; j < 8; j++) b[j] = p[j];
      } else {
        b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; b[3] = 255; b[4] = p[3]; b[5] = p[4]; b[6] = p[5]; b[7] = 255;
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (channels == 4) {
          const float a = (float)b[4 * j + 3] / 255.0f;
          v[2 * r + j] = make_float4(s_lut[b[4 * j]] * a, s_lut[b[4 * j + 1]] * a, s_lut[b[4 * j + 2]] * a, a);
        } else {
          v[2 * r + j] = make_float4(s_lut[b[4 * j]], s_lut[b[4 * j + 1]], s_lut[b[4 * j + 2]], 1.0f);
        }
      }
    }
    dst[i] = make_float4((((v[0].x + v[1].x) + v[2].x) + v[3].x) * 0.25f, (((v[0].y + v[1].y) + v[2].y) + v[3].y) * 0.25f,
                         (((v[0].z + v[1].z) + v[2].z) + v[3].z) * 0.25f, (((v[0].w + v[1].w) + v[2].w) + v[3].w) * 0.25f);
  }
}

typedef float dssim_f2 __attribute__((ext_vector_type(2)));   // a pair of f32 in an even-aligned register pair: v_pk_* = two IEEE operations

// n / d, correctly rounded, for operands that need no range scaling. The compiler's IEEE division is this very sequence
// (reciprocal, one Newton step on it, quotient, two residual corrections) bracketed by v_div_scale (which returns its
// operand unchanged unless an exponent is near the ends of the range) and v_div_fixup (which passes the quotient through
// unless an operand is zero, infinite, NaN or the result denormal): for the Halley steps below - numerator and denominator
// in [2^-8, 2^3] - those three instructions do nothing, and leaving them out saves 3 of 11 VALU slots per division, six
// divisions per LAB cell. mi355_selftest_dssim_cbrt replays every f32 in (216/24389, 2] through both forms.
__device__ __forceinline__ float dssim_div_unscaled(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  const float e2 = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e2, r, q);
  const float e3 = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e3, r, q);
}

template <bool LITERAL = false>
__device__ __forceinline__ float dssim_cbrt_poly(float x) {
  float y = (-0.5f * x + 1.51f) * x + 0.2f;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const float y3 = y * y * y;
    const float num = y * (y3 + 2.0f * x), den = 2.0f * y3 + x;
    y = LITERAL ? num / den : dssim_div_unscaled(num, den);
  }
  return y;
}

__device__ __forceinline__ float dssim_f(float t) {
  const float eps = 216.0f / 24389.0f, kk = 24389.0f / (27.0f * 116.0f);
  return t > eps ? dssim_cbrt_poly<>(t) - 16.0f / 116.0f : kk * t;
}

// The same for two values at once (the same channel of two neighbouring cells): every operation of dssim_cbrt_poly /
// dssim_div_unscaled as its packed-f32 twin (v_pk_mul / v_pk_add / v_pk_fma: one IEEE operation per element, nothing fused
// that the scalar form does not fuse), the two reciprocals scalar. Half the VALU slots per cube root.
__device__ __forceinline__ dssim_f2 dssim_pk_fma(dssim_f2 a, dssim_f2 b, dssim_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ dssim_f2 dssim_div_unscaled2(dssim_f2 n, dssim_f2 d) {
  dssim_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const dssim_f2 one = {1.0f, 1.0f};
  const dssim_f2 e = dssim_pk_fma(-d, r, one);
  r = dssim_pk_fma(e, r, r);
  dssim_f2 q = n * r;
  const dssim_f2 e2 = dssim_pk_fma(-d, q, n);
  q = dssim_pk_fma(e2, r, q);
  const dssim_f2 e3 = dssim_pk_fma(-d, q, n);
  return dssim_pk_fma(e3, r, q);
}
__device__ __forceinline__ dssim_f2 dssim_cbrt_poly2(dssim_f2 x) {
  dssim_f2 y = (-0.5f * x + 1.51f) * x + 0.2f;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const dssim_f2 y3 = y * y * y;
    const dssim_f2 num = y * (y3 + 2.0f * x), den = 2.0f * y3 + x;
    y = dssim_div_unscaled2(num, den);
  }
  return y;
}
// both branches for both elements (the cube-root branch is finite for every t >= 0), then a per-element select
__device__ __forceinline__ dssim_f2 dssim_f_pair(dssim_f2 t) {
  const float eps = 216.0f / 24389.0f, kk = 24389.0f / (27.0f * 116.0f);
  const dssim_f2 cb = dssim_cbrt_poly2(t) - 16.0f / 116.0f, lin = kk * t;
  dssim_f2 o;
  o.x = t.x > eps ? cb.x : lin.x;
  o.y = t.y > eps ? cb.y : lin.y;
  return o;
}

// ---- fused per-scale kernel: LAB conversion, chroma pre-blur and the mu / sq blurs of all three planes in one pass
// over the scale. A block owns a 32x16 output tile; the linear-RGB source is read for the tile plus a halo of 4 (two
// blur passes for the chroma pre-blur + two for mu/sq), everything in between lives in LDS. Replicated edges are
// applied PER PASS as in the unfused form: an out-of-image neighbour of pass k reads pass k-1's value at the clamped
// coordinate (always inside the tile's region because the region contains the border pixel), never a value computed
// at a virtual position. Arithmetic and tap order are those of dssim_blur_kernel, so the planes are bit-identical.
#ifndef DSSIM_TW
#define DSSIM_TW 32
#define DSSIM_TH 16
#define DSSIM_NT 256
#endif
constexpr int kTw = DSSIM_TW, kTh = DSSIM_TH, kHalo = 4;
constexpr int kNt = DSSIM_NT;                      // lanes of a tile block
constexpr int kOwners = (kTw / 2) * (kTh / 2);     // lanes that own a 2 x 2 block of tile outputs in the register-tiled passes
constexpr int kPpl = (kTw * kTh) / kNt;            // tile pixels per lane in the per-pixel passes
static_assert((kTw * kTh) % kNt == 0 && kOwners <= kNt && kNt % 64 == 0 && kNt >= 256, "tile geometry");
constexpr int kRw = kTw + 2 * kHalo, kRh = kTh + 2 * kHalo;  // 40 x 24 region

struct DssimSrc {          // source of the scale's linear RGB
  const uint8_t *u8; int stride, channels; const float *lut;   // scale 0: packed sRGB(A) bytes + gamma table
  int pattern;                                                  // translucent pixels: 1 = over the crate's coloured pattern, 0 = over black
  const float4 *lin;                                           // other scales: premultiplied linear float4
};

// Translucent pixels (premultiplied linear r, g, b with alpha a < 1): dssim composes them on a coloured, position-dependent
// background "to better judge dissimilarity with various backgrounds" - each channel gets the missing coverage 1 - a where
// a bit of n = (x + 11) ^ (y + 11) is set (r: 16, g: 8, b: 32), in the coordinates of the scale being converted. Written
// from memory of dssim-core's to_lab for RGBA (sources not in the reference tree): PARITY UNPINNED like the rest of this
// engine. For a == 1 the term is + 0.0f: opaque frames - what videocompare's tests and config 5 feed - are unaffected.
__device__ __forceinline__ void dssim_pattern(float &r, float &g, float &b, float a, int gx, int gy) {
  const int n = (gx + 11) ^ (gy + 11);
  const float t = 1.0f - a;
  if (n & 16) r = r + t;
  if (n & 8) g = g + t;
  if (n & 32) b = b + t;
}

__device__ __forceinline__ void dssim_lab_px(float r, float g, float b, float &L, float &A, float &B) {
  const float dx = 0.9505f, dy = 1.0f, dz = 1.089f;
  const float fx = (r * (0.4124f / dx) + g * (0.3576f / dx)) + b * (0.1805f / dx);
  const float fy = (r * (0.2126f / dy) + g * (0.7152f / dy)) + b * (0.0722f / dy);
  const float fz = (r * (0.0193f / dz) + g * (0.1192f / dz)) + b * (0.9505f / dz);
  const float X = dssim_f(fx), Y = dssim_f(fy), Z = dssim_f(fz);
  L = Y * 1.05f;
  A = (500.0f / 220.0f) * (X - Y) + 86.2f / 220.0f;
  B = (200.0f / 220.0f) * (Y - Z) + 107.9f / 220.0f;
}

// two cells at once: element k of every pair belongs to cell k; the operations per element are those of dssim_lab_px
__device__ __forceinline__ void dssim_lab_px2(dssim_f2 r, dssim_f2 g, dssim_f2 b, dssim_f2 &L, dssim_f2 &A, dssim_f2 &B) {
  const float dx = 0.9505f, dy = 1.0f, dz = 1.089f;
  const dssim_f2 fx = (r * (0.4124f / dx) + g * (0.3576f / dx)) + b * (0.1805f / dx);
  const dssim_f2 fy = (r * (0.2126f / dy) + g * (0.7152f / dy)) + b * (0.0722f / dy);
  const dssim_f2 fz = (r * (0.0193f / dz) + g * (0.1192f / dz)) + b * (0.9505f / dz);
  const dssim_f2 X = dssim_f_pair(fx), Y = dssim_f_pair(fy), Z = dssim_f_pair(fz);
  L = Y * 1.05f;
  A = (500.0f / 220.0f) * (X - Y) + 86.2f / 220.0f;
  B = (200.0f / 220.0f) * (Y - Z) + 107.9f / 220.0f;
}

// one 3x3 pass inside the LDS region: dst(lx,ly) for region coords in [M, kRw-M) x [M, kRh-M), reading src with
// per-pass edge replication in IMAGE coordinates. SQ: square the input on the fly. INTERIOR (block-uniform: the whole
// region lies inside the image) drops the clamps and the in-image test - the per-tap coordinate arithmetic was ~80 % of
// the kernel's instructions - and leaves nine LDS reads at constant offsets; the taps and their order are the same.
template <bool SQ, int M, bool INTERIOR>
__device__ __forceinline__ void dssim_region_pass(const float *src, float *dst, int x0, int y0, int w, int h) {
  const float K[9] = {0.095332f, 0.118095f, 0.095332f, 0.118095f, 0.146293f, 0.118095f, 0.095332f, 0.118095f, 0.095332f};
  constexpr int rw = kRw - 2 * M, rh = kRh - 2 * M;
  for (int e = threadIdx.x; e < rw * rh; e += kNt) {
    const int ly = M + e / rw, lx = M + e - (e / rw) * rw;
    float acc = 0.0f;
    if (INTERIOR) {
      const float *p = src + ly * kRw + lx;
#pragma unroll
      for (int dyy = 0; dyy < 3; dyy++) {
#pragma unroll
        for (int dxx = 0; dxx < 3; dxx++) {
          float v = p[(dyy - 1) * kRw + (dxx - 1)];
          if (SQ) v = v * v;
          acc = acc + v * K[dyy * 3 + dxx];
        }
      }
    } else {
      const int gx = x0 + lx, gy = y0 + ly;            // image coordinates of this region cell (may be outside the image)
      if (gx >= 0 && gx < w && gy >= 0 && gy < h) {    // cells outside the image are never read (clamping maps inside)
#pragma unroll
        for (int dyy = 0; dyy < 3; dyy++) {
          int yy = gy + dyy - 1; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
          for (int dxx = 0; dxx < 3; dxx++) {
            int xx = gx + dxx - 1; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
            float v = src[(yy - y0) * kRw + (xx - x0)];
            if (SQ) v = v * v;
            acc = acc + v * K[dyy * 3 + dxx];
          }
        }
      }
    }
    dst[ly * kRw + lx] = acc;
  }
}

// The 4 x 4 window of a 2 x 2 output block, first cell at plane index j (row stride `stride`, even), as 8-byte LDS reads: lanes
// that are neighbours in x are 8 bytes apart, so 4-byte reads use every second bank (two passes per instruction) where 8-byte
// reads are conflict-free. ALIGNED: j is even - two reads per row; otherwise the row's four cells straddle three aligned
// pairs and the outer halves are dropped (the cells j - 1 and j + 4 exist in every caller: a window never starts in column 0
// or ends in the last column of a misaligned pass).
template <bool ALIGNED>
__device__ __forceinline__ void dssim_window_load(const float *plane, int j, int stride, float (&v)[4][4]) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (ALIGNED) {
      const dssim_f2 a = *(const dssim_f2 *)&plane[j + r * stride], b = *(const dssim_f2 *)&plane[j + r * stride + 2];
      v[r][0] = a.x; v[r][1] = a.y; v[r][2] = b.x; v[r][3] = b.y;
    } else {
      const dssim_f2 a = *(const dssim_f2 *)&plane[j + r * stride - 1], b = *(const dssim_f2 *)&plane[j + r * stride + 1],
                     c = *(const dssim_f2 *)&plane[j + r * stride + 3];
      v[r][0] = a.y; v[r][1] = b.x; v[r][2] = b.y; v[r][3] = c.x;
    }
  }
}

// The 3 x 3 blur of a 2 x 2 block of outputs from its 4 x 4 window, the two outputs of a row as one packed-f32 pair:
// v_pk_mul_f32 / v_pk_add_f32 are two IEEE operations per instruction (never fused: -ffp-contract=off), so every output is
// accumulated over its nine taps in the same order, with the same roundings, as the scalar loop - in half the instructions.
__device__ __forceinline__ void dssim_blur_2x2(const float (&v)[4][4], float (&o)[2][2]) {
  const float K[9] = {0.095332f, 0.118095f, 0.095332f, 0.118095f, 0.146293f, 0.118095f, 0.095332f, 0.118095f, 0.095332f};
#pragma unroll
  for (int oy = 0; oy < 2; oy++) {
    dssim_f2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int dyy = 0; dyy < 3; dyy++)
#pragma unroll
      for (int dxx = 0; dxx < 3; dxx++) {
        const dssim_f2 p = {v[oy + dyy][dxx], v[oy + dyy][dxx + 1]};
        const dssim_f2 k = {K[dyy * 3 + dxx], K[dyy * 3 + dxx]};
        acc = acc + p * k;
      }
    o[oy][0] = acc.x;
    o[oy][1] = acc.y;
  }
}

// Interior regions, register-tiled: a lane produces a 2 x 2 block of outputs from one 4 x 4 window (16 LDS reads instead
// of 36), each output accumulated over its nine taps in the usual order. DUAL also produces the blur of the squares from
// the same window (the squares are formed once per cell: the same values as squaring per tap).
template <int M, bool DUAL>
__device__ __forceinline__ void dssim_pass_2x2(const float *src, float *dst, float *dst_sq, int first_cell) {
  constexpr int cw = (kRw - 2 * M) / 2, ch = (kRh - 2 * M) / 2;
  static_assert((kRw - 2 * M) % 2 == 0 && (kRh - 2 * M) % 2 == 0, "even pass extents");
  for (int e = first_cell; e < cw * ch; e += kNt) {
    const int cy = e / cw, cx = e - cy * cw;
    const int ly = M + 2 * cy, lx = M + 2 * cx;
    float v[4][4];
    dssim_window_load<(M - 1) % 2 == 0>(src, (ly - 1) * kRw + (lx - 1), kRw, v);
    float o[2][2];
    dssim_blur_2x2(v, o);
#pragma unroll
    for (int oy = 0; oy < 2; oy++)
#pragma unroll
      for (int ox = 0; ox < 2; ox++) dst[(ly + oy) * kRw + lx + ox] = o[oy][ox];
    if (DUAL) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) v[r][c] = v[r][c] * v[r][c];
      dssim_blur_2x2(v, o);
#pragma unroll
      for (int oy = 0; oy < 2; oy++)
#pragma unroll
        for (int ox = 0; ox < 2; ox++) dst_sq[(ly + oy) * kRw + lx + ox] = o[oy][ox];
    }
  }
}

struct DssimPlanes { float *img[3], *mu[3], *sq[3]; };

// step 1 of a scale: LAB of every in-image cell of the region
// linear r, g, b of one in-image cell (gamma table, premultiplication, background pattern)
__device__ __forceinline__ void dssim_linear_px(const DssimSrc &S, int w, const float *s_lut, int gx, int gy, float &r, float &g, float &b) {
  if (S.u8) {
    const uint8_t *p = S.u8 + (size_t)gy * S.stride + (size_t)gx * S.channels;
    if (S.channels == 4) {
      const float a = s_lut[256 + p[3]]; r = s_lut[p[0]] * a; g = s_lut[p[1]] * a; b = s_lut[p[2]] * a;
      if (S.pattern && a != 1.0f) dssim_pattern(r, g, b, a, gx, gy);   // a == 1 adds + 0.0f to values >= + 0: skipped, same bits
    }
    else { r = s_lut[p[0]]; g = s_lut[p[1]]; b = s_lut[p[2]]; }
  } else {
    const float4 v = S.lin[(size_t)gy * w + gx];
    r = v.x; g = v.y; b = v.z;
    if (S.pattern && v.w != 1.0f) dssim_pattern(r, g, b, v.w, gx, gy);
  }
}

template <bool INTERIOR>
__device__ __forceinline__ void dssim_lab_region(const DssimSrc &S, int w, int h, float (*s_lab)[kRw * kRh], const float *s_lut, int x0, int y0) {
  if (INTERIOR) {
    // two horizontally adjacent cells per step, their cube roots as packed pairs; the region's rows are even, so a pair never
    // straddles two rows and the three results go to LDS as 8-byte words
    static_assert(kRw % 2 == 0, "pairs of cells per row");
    for (int e = threadIdx.x; e < kRw * kRh / 2; e += kNt) {
      const int ly = e / (kRw / 2), lx = 2 * (e - ly * (kRw / 2));
      float r0, g0, b0, r1, g1, b1;
      dssim_linear_px(S, w, s_lut, x0 + lx, y0 + ly, r0, g0, b0);
      dssim_linear_px(S, w, s_lut, x0 + lx + 1, y0 + ly, r1, g1, b1);
      dssim_f2 L, A, B;
      dssim_lab_px2(dssim_f2{r0, r1}, dssim_f2{g0, g1}, dssim_f2{b0, b1}, L, A, B);
      *(dssim_f2 *)&s_lab[0][2 * e] = L; *(dssim_f2 *)&s_lab[1][2 * e] = A; *(dssim_f2 *)&s_lab[2][2 * e] = B;
    }
    return;
  }
  for (int e = threadIdx.x; e < kRw * kRh; e += kNt) {
    const int ly = e / kRw, lx = e - ly * kRw;
    const int gx = x0 + lx, gy = y0 + ly;
    float L = 0, A = 0, B = 0;
    if (INTERIOR || (gx >= 0 && gx < w && gy >= 0 && gy < h)) {
      float r, g, b;
      dssim_linear_px(S, w, s_lut, gx, gy, r, g, b);
      dssim_lab_px(r, g, b, L, A, B);
    }
    s_lab[0][e] = L; s_lab[1][e] = A; s_lab[2][e] = B;
  }
}

// step 2: chroma pre-blur (two passes through s_a): the a / b planes are valid on the region minus a margin of 2 afterwards
template <bool INTERIOR>
__device__ __forceinline__ void dssim_chroma_preblur(float (*s_lab)[kRw * kRh], float *s_a, int x0, int y0, int w, int h) {
  for (int c = 1; c < 3; c++) {
    if (INTERIOR) dssim_pass_2x2<1, false>(s_lab[c], s_a, nullptr, threadIdx.x);
    else dssim_region_pass<false, 1, INTERIOR>(s_lab[c], s_a, x0, y0, w, h);
    __syncthreads();
    if (INTERIOR) dssim_pass_2x2<2, false>(s_a, s_lab[c], nullptr, threadIdx.x);
    else dssim_region_pass<false, 2, INTERIOR>(s_a, s_lab[c], x0, y0, w, h);
    __syncthreads();
  }
}

// the second blur pass for a lane's 2 x 2 block of TILE outputs (cell cx, cy of the 16 x 8 grid), from the first pass in LDS
__device__ __forceinline__ void dssim_tile_pass2_2x2(const float *src, int cx, int cy, float (&o)[2][2]) {
  const int j0 = (kHalo + 2 * cy - 1) * kRw + (kHalo + 2 * cx - 1);
  float v[4][4];
  dssim_window_load<(kHalo - 1) % 2 == 0>(src, j0, kRw, v);
  dssim_blur_2x2(v, o);
}

// the same for ONE tile pixel with per-pass edge replication in image coordinates (tiles that touch the image border)
__device__ __forceinline__ float dssim_tile_pass2_px(const float *src, int gx, int gy, int x0, int y0, int w, int h) {
  const float K[9] = {0.095332f, 0.118095f, 0.095332f, 0.118095f, 0.146293f, 0.118095f, 0.095332f, 0.118095f, 0.095332f};
  float acc = 0.0f;
#pragma unroll
  for (int dyy = 0; dyy < 3; dyy++) {
    int yy = gy + dyy - 1; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
    for (int dxx = 0; dxx < 3; dxx++) {
      int xx = gx + dxx - 1; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
      acc = acc + src[(yy - y0) * kRw + (xx - x0)] * K[dyy * 3 + dxx];
    }
  }
  return acc;
}

template <bool INTERIOR>
__device__ __forceinline__ void dssim_scale_body(const DssimSrc &S, int w, int h, const DssimPlanes &O, float (*s_lab)[kRw * kRh], float *s_a,
                                                 float *s_b, const float *s_lut, int x0, int y0) {
  dssim_lab_region<INTERIOR>(S, w, h, s_lab, s_lut, x0, y0);
  __syncthreads();
  dssim_chroma_preblur<INTERIOR>(s_lab, s_a, x0, y0, w, h);
  // 3. per plane: img = plane (tile), mu = blur(plane), sq = blur(plane^2); margins 3 and 4
  for (int c = 0; c < 3; c++) {
    if (INTERIOR) {
      dssim_pass_2x2<3, true>(s_lab[c], s_a, s_b, threadIdx.x);
    } else {
      dssim_region_pass<false, 3, INTERIOR>(s_lab[c], s_a, x0, y0, w, h);
      dssim_region_pass<true, 3, INTERIOR>(s_lab[c], s_b, x0, y0, w, h);
    }
    __syncthreads();
    // second passes straight to global for the tile cells
    const float K[9] = {0.095332f, 0.118095f, 0.095332f, 0.118095f, 0.146293f, 0.118095f, 0.095332f, 0.118095f, 0.095332f};
    if (INTERIOR) {
      for (int e = threadIdx.x; e < (kTw / 2) * (kTh / 2); e += kNt) {
        const int cy = e / (kTw / 2), cx = e - cy * (kTw / 2);
        const int ly = kHalo + 2 * cy, lx = kHalo + 2 * cx;
        const int j0 = (ly - 1) * kRw + (lx - 1);
        float va[4][4], vb[4][4];
        dssim_window_load<(kHalo - 1) % 2 == 0>(s_a, j0, kRw, va);
        dssim_window_load<(kHalo - 1) % 2 == 0>(s_b, j0, kRw, vb);
        float om[2][2], os[2][2];
        dssim_blur_2x2(va, om);
        dssim_blur_2x2(vb, os);
#pragma unroll
        for (int oy = 0; oy < 2; oy++)
#pragma unroll
          for (int ox = 0; ox < 2; ox++) {
            const size_t o = (size_t)(y0 + ly + oy) * w + (x0 + lx + ox);
            O.img[c][o] = s_lab[c][(ly + oy) * kRw + lx + ox];
            O.mu[c][o] = om[oy][ox];
            O.sq[c][o] = os[oy][ox];
          }
      }
    } else
    for (int e = threadIdx.x; e < kTw * kTh; e += kNt) {
      const int ly = kHalo + e / kTw, lx = kHalo + e - (e / kTw) * kTw;
      const int gx = x0 + lx, gy = y0 + ly;
      if (INTERIOR || (gx < w && gy < h)) {
        float am = 0.0f, as = 0.0f;
#pragma unroll
        for (int dyy = 0; dyy < 3; dyy++) {
          int yy = gy + dyy - 1;
          if (!INTERIOR) yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
          for (int dxx = 0; dxx < 3; dxx++) {
            int xx = gx + dxx - 1;
            if (!INTERIOR) xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
            const int j = INTERIOR ? (ly + dyy - 1) * kRw + (lx + dxx - 1) : (yy - y0) * kRw + (xx - x0);
            am = am + s_a[j] * K[dyy * 3 + dxx];
            as = as + s_b[j] * K[dyy * 3 + dxx];
          }
        }
        const size_t o = (size_t)gy * w + gx;
        O.img[c][o] = s_lab[c][ly * kRw + lx];
        O.mu[c][o] = am;
        O.sq[c][o] = as;
      }
    }
    __syncthreads();
  }
}

// up to three scales per launch (the small ones are launch-bound on their own): blocks [first[j], first[j+1]) belong
// to entry j
struct DssimScaleJob { DssimSrc S; int w, h; DssimPlanes O; };
struct DssimScaleJobs { DssimScaleJob job[3]; unsigned first[4]; };

__global__ __launch_bounds__(kNt) void dssim_scale_fused_kernel(DssimScaleJobs J) {
  __shared__ __attribute__((aligned(16))) float s_lab[3][kRw * kRh];   // LAB planes of the region
  __shared__ __attribute__((aligned(16))) float s_a[kRw * kRh], s_b[kRw * kRh];
  __shared__ float s_lut[512];   // [0, 256): sRGB -> linear; [256, 512): alpha byte / 255 (the IEEE quotient, once per block instead of per cell)
  const int j = blockIdx.x >= J.first[2] ? 2 : (blockIdx.x >= J.first[1] ? 1 : 0);
  const DssimSrc &S = J.job[j].S;
  const DssimPlanes &O = J.job[j].O;
  const int w = J.job[j].w, h = J.job[j].h;
  const unsigned tile = blockIdx.x - J.first[j];
  if (S.u8 && threadIdx.x < 256) { s_lut[threadIdx.x] = S.lut[threadIdx.x]; s_lut[256 + threadIdx.x] = (float)threadIdx.x / 255.0f; }
  const int tiles_x = (w + kTw - 1) / kTw;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * kTw - kHalo, y0 = ty * kTh - kHalo;  // image coords of region cell (0,0)
  __syncthreads();
  if (x0 >= 0 && y0 >= 0 && x0 + kRw <= w && y0 + kRh <= h) dssim_scale_body<true>(S, w, h, O, s_lab, s_a, s_b, s_lut, x0, y0);
  else dssim_scale_body<false>(S, w, h, O, s_lab, s_a, s_b, s_lut, x0, y0);
}

template <int NW = 4>
__device__ __forceinline__ double dssim_block_sum(double v, double *s_w) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    t = s_w[0];
#pragma unroll
    for (int k = 1; k < NW; k++) t = t + s_w[k];   // wave order, left to right
  }
  __syncthreads();
  return t;
}

// ---- fused comparison kernel: i12 = blur(img1*img2) (two passes through LDS, halo 2) and the SSIM map + its f64 block
// partial in one pass. Tile 32x16, region 36x20. Same per-pass edge replication and arithmetic as the unfused form.
constexpr int kCh = 2, kCw = kTw + 2 * kCh, kChh = kTh + 2 * kCh;  // 36 x 20
struct DssimCmp { const float *img1[3], *img2[3], *mu1[3], *mu2[3], *sq1[3], *sq2[3]; };

// SSIM of one pixel from the channel-averaged moments (1 = original, 2 = modified)
__device__ __forceinline__ float dssim_ssim_px(const float (&m1)[3], const float (&m2)[3], const float (&q1)[3], const float (&q2)[3], const float (&x12)[3]) {
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f, third = 1.0f / 3.0f;
  const float mu1mu1 = ((m1[0] * m1[0] + m1[1] * m1[1]) + m1[2] * m1[2]) * third;
  const float mu2mu2 = ((m2[0] * m2[0] + m2[1] * m2[1]) + m2[2] * m2[2]) * third;
  const float mu1mu2 = ((m1[0] * m2[0] + m1[1] * m2[1]) + m1[2] * m2[2]) * third;
  const float sig1 = (((q1[0] - m1[0] * m1[0]) + (q1[1] - m1[1] * m1[1])) + (q1[2] - m1[2] * m1[2])) * third;
  const float sig2 = (((q2[0] - m2[0] * m2[0]) + (q2[1] - m2[1] * m2[1])) + (q2[2] - m2[2] * m2[2])) * third;
  const float sig12 = (((x12[0] - m1[0] * m2[0]) + (x12[1] - m1[1] * m2[1])) + (x12[2] - m1[2] * m2[2])) * third;
  return ((2.0f * mu1mu2 + c1) * (2.0f * sig12 + c2)) / (((mu1mu1 + mu2mu2) + c1) * ((sig1 + sig2) + c2));
}

template <bool INTERIOR>
__device__ __forceinline__ double dssim_compare_body(const DssimCmp &P, int w, int h, float *__restrict__ ssim_map, float (*s_p)[kCw * kChh],
                                                     float (*s_q)[kCw * kChh], int x0, int y0) {
  const float K[9] = {0.095332f, 0.118095f, 0.095332f, 0.118095f, 0.146293f, 0.118095f, 0.095332f, 0.118095f, 0.095332f};
  for (int e = threadIdx.x; e < kCw * kChh; e += kNt) {
    const int ly = e / kCw, lx = e - ly * kCw;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool in = INTERIOR || (gx >= 0 && gx < w && gy >= 0 && gy < h);
    const size_t o = in ? (size_t)gy * w + gx : 0;
#pragma unroll
    for (int c = 0; c < 3; c++) s_p[c][e] = in ? P.img1[c][o] * P.img2[c][o] : 0.0f;
  }
  __syncthreads();
  // pass 1 on the region minus a margin of 1; interior regions: 2 x 2 outputs per lane from one 4 x 4 window per plane
  if (INTERIOR) {
    constexpr int cw = (kCw - 2) / 2, ch = (kChh - 2) / 2;
    static_assert((kCw - 2) % 2 == 0 && (kChh - 2) % 2 == 0, "even pass extents");
    for (int e = threadIdx.x; e < cw * ch; e += kNt) {
      const int cy = e / cw, cx = e - cy * cw;
      const int ly = 1 + 2 * cy, lx = 1 + 2 * cx;
      const int j0 = (ly - 1) * kCw + (lx - 1);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float v[4][4];
        dssim_window_load<true>(s_p[c], j0, kCw, v);   // lx - 1 = 2 cx: even
        float o[2][2];
        dssim_blur_2x2(v, o);
#pragma unroll
        for (int oy = 0; oy < 2; oy++)
#pragma unroll
          for (int ox = 0; ox < 2; ox++) s_q[c][(ly + oy) * kCw + lx + ox] = o[oy][ox];
      }
    }
  } else
  for (int e = threadIdx.x; e < (kCw - 2) * (kChh - 2); e += kNt) {
    const int ly = 1 + e / (kCw - 2), lx = 1 + e - (e / (kCw - 2)) * (kCw - 2);
    const int gx = x0 + lx, gy = y0 + ly;
    if (INTERIOR || (gx >= 0 && gx < w && gy >= 0 && gy < h)) {
      float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int dyy = 0; dyy < 3; dyy++) {
        int yy = gy + dyy - 1;
        if (!INTERIOR) yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
        for (int dxx = 0; dxx < 3; dxx++) {
          int xx = gx + dxx - 1;
          if (!INTERIOR) xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
          const int j = INTERIOR ? (ly + dyy - 1) * kCw + (lx + dxx - 1) : (yy - y0) * kCw + (xx - x0);
#pragma unroll
          for (int c = 0; c < 3; c++) acc[c] = acc[c] + s_p[c][j] * K[dyy * 3 + dxx];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; c++) s_q[c][ly * kCw + lx] = acc[c];
    }
  }
  __syncthreads();
  double dsum = 0.0;
  for (int e = threadIdx.x; e < kTw * kTh; e += kNt) {
    const int ly = kCh + e / kTw, lx = kCh + e - (e / kTw) * kTw;
    const int gx = x0 + lx, gy = y0 + ly;
    if (INTERIOR || (gx < w && gy < h)) {
      float x12[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int dyy = 0; dyy < 3; dyy++) {
        int yy = gy + dyy - 1;
        if (!INTERIOR) yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
#pragma unroll
        for (int dxx = 0; dxx < 3; dxx++) {
          int xx = gx + dxx - 1;
          if (!INTERIOR) xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
          const int j = INTERIOR ? (ly + dyy - 1) * kCw + (lx + dxx - 1) : (yy - y0) * kCw + (xx - x0);
#pragma unroll
          for (int c = 0; c < 3; c++) x12[c] = x12[c] + s_q[c][j] * K[dyy * 3 + dxx];
        }
      }
      const size_t i = (size_t)gy * w + gx;
      float m1[3], m2[3], q1[3], q2[3];
#pragma unroll
      for (int c = 0; c < 3; c++) { m1[c] = P.mu1[c][i]; m2[c] = P.mu2[c][i]; q1[c] = P.sq1[c][i]; q2[c] = P.sq2[c][i]; }
      const float ssim = dssim_ssim_px(m1, m2, q1, q2, x12);
      ssim_map[i] = ssim;
      dsum += (double)ssim;
    }
  }
  return dsum;
}

__global__ __launch_bounds__(kNt) void dssim_compare_fused_kernel(DssimCmp P, int w, int h, float *__restrict__ ssim_map, double *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float s_p[3][kCw * kChh];   // products img1*img2 of the region
  __shared__ __attribute__((aligned(16))) float s_q[3][kCw * kChh];   // first blur pass
  __shared__ double s_w[kNt / 64];
  const int tiles_x = (w + kTw - 1) / kTw;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * kTw - kCh, y0 = ty * kTh - kCh;
  const double dsum = (x0 >= 0 && y0 >= 0 && x0 + kCw <= w && y0 + kChh <= h) ? dssim_compare_body<true>(P, w, h, ssim_map, s_p, s_q, x0, y0)
                                                                             : dssim_compare_body<false>(P, w, h, ssim_map, s_p, s_q, x0, y0);
  const double t = dssim_block_sum<kNt / 64>(dsum, s_w);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---- hash-and-compare in one pass (videocompare's per-aggregate loop, imp.rs:316-345: every non-reference pad's frame is
// hashed, compared against the reference pad's image, and its hash is dropped). The MODIFIED frame's scale is converted and
// blurred exactly as dssim_scale_fused_kernel does, but its img / mu / sq planes never leave the CU: the products with the
// original's img plane, their two blur passes and the SSIM map of dssim_compare_fused_kernel follow in the same block. Per
// scale that is 4 + 9 * 4 B/pixel read (bytes or float4, the original's nine planes, its img plane with a halo of 2) and
// 4 B/pixel written (the SSIM map) instead of 36 written + 72 + 4 read and written again by the two-kernel form, under the
// arithmetic of the scale kernel, which bounds it. Every value is produced by the operations of the two-kernel form in the
// same order; the block partials are summed in the comparison kernel's pixel order, so the f64 scores are the same bits.
struct DssimFusedJob { DssimSrc S; int w, h; const float *img1[3], *mu1[3], *sq1[3]; float *map; double *partial; };
struct DssimFusedJobs { DssimFusedJob job[3]; unsigned first[4]; };

#ifdef DSSIM_EXPERIMENT_NO_BARRIERS   /* timing experiment only (wrong results): what do the phase barriers cost? */
#define DSSIM_PHASE_SYNC() ((void)0)
#else
#define DSSIM_PHASE_SYNC() __syncthreads()
#endif
template <bool INTERIOR>
__device__ __forceinline__ double dssim_fused_body(const DssimFusedJob &J, float (*s_lab)[kRw * kRh], float *s_a, float *s_b, float *s_p,
                                                   const float *s_lut, int x0, int y0) {
  const int w = J.w, h = J.h;
  constexpr int pw = kRw - 4, ph = kRh - 4;   // the products' region: the tile plus a halo of 2 (36 x 20)
  float pimg[(pw * ph + kNt - 1) / kNt];      // interior tiles: the original's img values of this lane's product cells, one channel ahead
  if (INTERIOR) {
#pragma unroll
    for (int k = 0; k < (pw * ph + kNt - 1) / kNt; k++) {
      const int e = (int)threadIdx.x + k * kNt;
      pimg[k] = e < pw * ph ? J.img1[0][(size_t)(y0 + 2 + e / pw) * w + (x0 + 2 + e - (e / pw) * pw)] : 0.0f;
    }
  }
  dssim_lab_region<INTERIOR>(J.S, w, h, s_lab, s_lut, x0, y0);
  DSSIM_PHASE_SYNC();
  dssim_chroma_preblur<INTERIOR>(s_lab, s_a, x0, y0, w, h);
  double dsum = 0.0;
  if (INTERIOR) {
    // lanes 0..127 own a 2 x 2 block of tile outputs each (their moments stay in registers over the channel loop); while they
    // run the second passes of mu / sq, the other lanes (if the block has any) start the first pass of the products. The
    // original's img values for the products are requested one channel ahead (`pimg`: before the LAB conversion for channel 0),
    // so the product phase does not wait on memory
    const bool owner = threadIdx.x < kOwners;
    constexpr int kPimg = (pw * ph + kNt - 1) / kNt;
    const int cy = (int)threadIdx.x / (kTw / 2), cx = (int)threadIdx.x - cy * (kTw / 2);
    float mu2[3][2][2], sq2[3][2][2], x12[3][2][2];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      // first pass of plane and squares; the products of the region (modified: s_lab, post pre-blur; original: from memory)
      dssim_pass_2x2<3, true>(s_lab[c], s_a, s_b, threadIdx.x);
#pragma unroll
      for (int k = 0; k < kPimg; k++) {
        const int e = (int)threadIdx.x + k * kNt;
        if (e < pw * ph) {
          const int ly = 2 + e / pw, lx = 2 + e - (e / pw) * pw;
          s_p[ly * kRw + lx] = pimg[k] * s_lab[c][ly * kRw + lx];
          // the next channel's values of the same cells: requested now, multiplied a channel later
          if (c < 2) pimg[k] = J.img1[c + 1][(size_t)(y0 + ly) * w + (x0 + lx)];
        }
      }
      if (c > 0 && owner) dssim_tile_pass2_2x2(s_lab[c - 1], cx, cy, x12[c - 1]);   // the previous channel's products, second pass
      DSSIM_PHASE_SYNC();
      if (owner) {
        dssim_tile_pass2_2x2(s_a, cx, cy, mu2[c]);
        dssim_tile_pass2_2x2(s_b, cx, cy, sq2[c]);
      }
      // first pass of the products into s_lab[c] (the plane itself is not needed any more)
      dssim_pass_2x2<3, false>(s_p, s_lab[c], nullptr, kOwners < kNt ? (int)(threadIdx.x + kNt - kOwners) % kNt : (int)threadIdx.x);
      DSSIM_PHASE_SYNC();
    }
    if (owner) {
      dssim_tile_pass2_2x2(s_lab[2], cx, cy, x12[2]);
#pragma unroll
      for (int oy = 0; oy < 2; oy++)
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
          const int ty = 2 * cy + oy, tx = 2 * cx + ox;
          const size_t i = (size_t)(y0 + kHalo + ty) * w + (x0 + kHalo + tx);
          float m1[3], q1[3], m2[3], q2[3], xx[3];
#pragma unroll
          for (int c = 0; c < 3; c++) { m1[c] = J.mu1[c][i]; q1[c] = J.sq1[c][i]; m2[c] = mu2[c][oy][ox]; q2[c] = sq2[c][oy][ox]; xx[c] = x12[c][oy][ox]; }
          const float ssim = dssim_ssim_px(m1, m2, q1, q2, xx);
          J.map[i] = ssim;
          s_p[ty * kTw + tx] = ssim;
        }
    }
    DSSIM_PHASE_SYNC();
    // the block partial in the comparison kernel's order: lane t adds pixels t, t + kNt, ... of the tile
    for (int e = threadIdx.x; e < kTw * kTh; e += kNt) dsum += (double)s_p[e];
  } else {
    // tiles that touch the image border: lane t owns tile pixels t, t + kNt, ..., every pass replicates edges in image coordinates
    float mu2[3][kPpl], sq2[3][kPpl], x12[3][kPpl];
    int gx[kPpl], gy[kPpl];
    bool in[kPpl];
#pragma unroll
    for (int k = 0; k < kPpl; k++) {
      const int e = (int)threadIdx.x + kNt * k;
      gx[k] = x0 + kHalo + e % kTw; gy[k] = y0 + kHalo + e / kTw;
      in[k] = gx[k] < w && gy[k] < h;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      dssim_region_pass<false, 3, false>(s_lab[c], s_a, x0, y0, w, h);
      dssim_region_pass<true, 3, false>(s_lab[c], s_b, x0, y0, w, h);
      for (int e = threadIdx.x; e < pw * ph; e += kNt) {
        const int ly = 2 + e / pw, lx = 2 + e - (e / pw) * pw;
        const int px = x0 + lx, py = y0 + ly;
        const bool inside = px >= 0 && px < w && py >= 0 && py < h;
        s_p[ly * kRw + lx] = inside ? J.img1[c][(size_t)py * w + px] * s_lab[c][ly * kRw + lx] : 0.0f;
      }
      if (c > 0) {
#pragma unroll
        for (int k = 0; k < kPpl; k++) x12[c - 1][k] = in[k] ? dssim_tile_pass2_px(s_lab[c - 1], gx[k], gy[k], x0, y0, w, h) : 0.0f;
      }
      DSSIM_PHASE_SYNC();
#pragma unroll
      for (int k = 0; k < kPpl; k++) {
        mu2[c][k] = in[k] ? dssim_tile_pass2_px(s_a, gx[k], gy[k], x0, y0, w, h) : 0.0f;
        sq2[c][k] = in[k] ? dssim_tile_pass2_px(s_b, gx[k], gy[k], x0, y0, w, h) : 0.0f;
      }
      dssim_region_pass<false, 3, false>(s_p, s_lab[c], x0, y0, w, h);
      DSSIM_PHASE_SYNC();
    }
#pragma unroll
    for (int k = 0; k < kPpl; k++) {
      if (!in[k]) continue;
      x12[2][k] = dssim_tile_pass2_px(s_lab[2], gx[k], gy[k], x0, y0, w, h);
      const size_t i = (size_t)gy[k] * w + gx[k];
      float m1[3], q1[3], m2[3], q2[3], xx[3];
#pragma unroll
      for (int c = 0; c < 3; c++) { m1[c] = J.mu1[c][i]; q1[c] = J.sq1[c][i]; m2[c] = mu2[c][k]; q2[c] = sq2[c][k]; xx[c] = x12[c][k]; }
      const float ssim = dssim_ssim_px(m1, m2, q1, q2, xx);
      J.map[i] = ssim;
      dsum += (double)ssim;
    }
  }
  return dsum;
}

__global__ __launch_bounds__(kNt, kNt == 256 ? 7 : 1) void dssim_hash_compare_kernel(DssimFusedJobs JJ) {
  // 6 planes + the block-sum words = 23,072 B and <= 72 VGPRs: seven blocks per CU (the gamma / alpha tables are only read by
  // the LAB conversion and live in the products' plane, which is first written after it)
  __shared__ __attribute__((aligned(16))) float s_lab[3][kRw * kRh];
  __shared__ __attribute__((aligned(16))) float s_a[kRw * kRh], s_b[kRw * kRh], s_p[kRw * kRh];
  float *const s_lut = s_p;
  static_assert(kRw * kRh >= 512, "the tables fit the products' plane");
  __shared__ double s_w[kNt / 64];
  const int j = blockIdx.x >= JJ.first[2] ? 2 : (blockIdx.x >= JJ.first[1] ? 1 : 0);
  const DssimFusedJob &J = JJ.job[j];
  const unsigned tile = blockIdx.x - JJ.first[j];
  if (J.S.u8 && threadIdx.x < 256) { s_lut[threadIdx.x] = J.S.lut[threadIdx.x]; s_lut[256 + threadIdx.x] = (float)threadIdx.x / 255.0f; }
  const int tiles_x = (J.w + kTw - 1) / kTw;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int x0 = tx * kTw - kHalo, y0 = ty * kTh - kHalo;
  __syncthreads();
  const double dsum = (x0 >= 0 && y0 >= 0 && x0 + kRw <= J.w && y0 + kRh <= J.h) ? dssim_fused_body<true>(J, s_lab, s_a, s_b, s_p, s_lut, x0, y0)
                                                                               : dssim_fused_body<false>(J, s_lab, s_a, s_b, s_p, s_lut, x0, y0);
  const double t = dssim_block_sum<kNt / 64>(dsum, s_w);
  if (threadIdx.x == 0) J.partial[tile] = t;
}

// The reductions of all scales of a comparison run in three launches (one block per scale, or a block range per scale):
// per scale the work, the grid share and the summation order are what one launch per scale did, so the values are the
// same; only ~12 launch overheads per comparison are gone.
struct DssimRed {
  const float *map[kDssimScales];        // SSIM map of the scale
  const double *part_a[kDssimScales];    // block partials of the SSIM sum (n_a of them)
  double *part_b[kDssimScales];          // block partials of the absolute deviations (first_b[k+1] - first_b[k] of them)
  unsigned n_a[kDssimScales];
  unsigned first_b[kDssimScales + 1];
  double len[kDssimScales], exponent[kDssimScales];
  size_t n[kDssimScales];
  double *slots;                         // per scale: [sum, avg, dev]
  int n_scales;
  // blockIdx.y = frame of a call that reduces several comparisons at once: its maps / partials / slots lie these strides
  // behind frame 0's (all 0 for one comparison per launch)
  size_t frame_map = 0, frame_pa = 0, frame_pb = 0, frame_slots = 0;
};

// block k: sum the block partials of scale k (256 lanes, strided, then the fixed-order block reduction: deterministic);
// slot[0] = sum, slot[1] = avg = max(sum/len, 0)^(0.5^scale)
__global__ __launch_bounds__(256) void dssim_avg_kernel(DssimRed R) {
  __shared__ double s_w[4];
  const int k = blockIdx.x;
  double acc = 0.0;
  // eight loads in flight; the additions stay in index order
  const unsigned n_a = R.n_a[k];
  const double *pa = R.part_a[k] + blockIdx.y * R.frame_pa;
  unsigned i = threadIdx.x;
  for (; i + 7 * 256 < n_a; i += 8 * 256) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = pa[i + j * 256];
#pragma unroll
    for (int j = 0; j < 8; j++) acc += v[j];
  }
  for (; i < n_a; i += 256) acc += pa[i];
  const double sum = dssim_block_sum(acc, s_w);
  if (threadIdx.x == 0) {
    double *slots = R.slots + blockIdx.y * R.frame_slots;
    slots[3 * k] = sum;
    slots[3 * k + 1] = pow(fmax(sum / R.len[k], 0.0), R.exponent[k]);
  }
}

__global__ __launch_bounds__(256) void dssim_absdev2_kernel(DssimRed R) {
  __shared__ double s_w[4];
  int k = 0;
  while (k + 1 < R.n_scales && blockIdx.x >= R.first_b[k + 1]) k++;
  const unsigned j = blockIdx.x - R.first_b[k], g = R.first_b[k + 1] - R.first_b[k];
  const double avg = R.slots[blockIdx.y * R.frame_slots + 3 * k + 1];
  const size_t gs = (size_t)g * 256, n = R.n[k];
  const float *map = R.map[k] + blockIdx.y * R.frame_map;
  double acc = 0.0;
  for (size_t i = (size_t)j * 256 + threadIdx.x; i < n; i += gs) acc += fabs(avg - (double)map[i]);
  const double t = dssim_block_sum(acc, s_w);
  if (threadIdx.x == 0) R.part_b[k][blockIdx.y * R.frame_pb + j] = t;
}

__global__ __launch_bounds__(256) void dssim_sum_kernel(DssimRed R) {
  __shared__ double s_w[4];
  const int k = blockIdx.x;
  const unsigned g = R.first_b[k + 1] - R.first_b[k];
  double acc = 0.0;
  const double *pb = R.part_b[k] + blockIdx.y * R.frame_pb;
  for (unsigned i = threadIdx.x; i < g; i += 256) acc += pb[i];
  const double sum = dssim_block_sum(acc, s_w);
  if (threadIdx.x == 0) R.slots[blockIdx.y * R.frame_slots + 3 * k + 2] = sum;
}

// ------------------------------------------------------------------ host side

// per-context cache: the gamma table (uploaded once) and released image pools kept for reuse (a stream of frames of one
// size allocates once instead of once per frame; hipMalloc/hipFree of ~400 MB cost more than the kernels)
struct DssimCache {
  float *d_lut = nullptr;
  struct Pool { float *p; size_t bytes; };
  std::vector<Pool> free_pools;
};
static DssimCache *dssim_cache(mi355_ctx *ctx) {
  if (!ctx->dssim_cache) ctx->dssim_cache = new DssimCache();
  return (DssimCache *)ctx->dssim_cache;
}
void dssim_release(mi355_ctx *ctx) {
  DssimCache *c = (DssimCache *)ctx->dssim_cache;
  if (!c) return;
  if (c->d_lut) (void)hipFree(c->d_lut);
  for (auto &p : c->free_pools) (void)hipFree(p.p);
  delete c;
  ctx->dssim_cache = nullptr;
}

static unsigned dssim_grid(mi355_ctx *ctx, size_t n) {
  size_t b = (n + 255) / 256;
  const size_t cap = (size_t)ctx->n_cu * 8;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

static int dssim_scratch(mi355_ctx *ctx, int slot, size_t bytes, void **out) {
  if (ctx->d_stage_bytes[slot] < bytes) {
    if (ctx->d_stage[slot]) (void)hipFree(ctx->d_stage[slot]);
    ctx->d_stage[slot] = nullptr;
    ctx->d_stage_bytes[slot] = 0;
    int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[slot], bytes), "hipMalloc(dssim scratch)");
    if (rc) return rc;
    ctx->d_stage_bytes[slot] = bytes;
  }
  *out = ctx->d_stage[slot];
  return MI355_OK;
}

// the sRGB -> linear table (uploaded once per context)
static int dssim_gamma_table(mi355_ctx *ctx, DssimCache *cache) {
  if (cache->d_lut) return MI355_OK;
  float lut[256];
  for (int i = 0; i < 256; i++) {
    const double s = (double)i / 255.0;
    lut[i] = (float)(s <= 0.04045 ? s / 12.92 : std::pow((s + 0.055) / 1.055, 2.4));
  }
  int rc = check_hip(ctx, hipMalloc((void **)&cache->d_lut, sizeof lut), "hipMalloc(dssim gamma table)");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipMemcpy(cache->d_lut, lut, sizeof lut, hipMemcpyHostToDevice), "dssim: gamma table"))) {
    (void)hipFree(cache->d_lut);
    cache->d_lut = nullptr;
  }
  return rc;
}

// the dssim value from the per-scale [sum, avg, dev] slots
static double dssim_value_of(const mi355_dssim_image *a, const double *slots) {
  double ssim_sum = 0.0, weight_sum = 0.0;
  for (int k = 0; k < a->n_scales; k++) {
    const double len = (double)a->s[k].w * (double)a->s[k].h;
    const double score = 1.0 - slots[3 * k + 2] / len;
    ssim_sum += score * kDssimWeights[k];
    weight_sum += kDssimWeights[k];
  }
  const double total = ssim_sum / weight_sum;
  return 1.0 / std::fmax(total, 2.220446049250313e-16) - 1.0;
}

void dssim_free_image(mi355_ctx *ctx, mi355_dssim_image *img) {
  if (!img) return;
  // no host wait: the pool goes back to THIS context's free list and is handed out again by a later create_image on the same
  // stream, i.e. behind every kernel that still reads it (an image is created, compared and freed through one context)
  if (img->pool) {
    DssimCache *c = dssim_cache(ctx);
    if (c->free_pools.size() < 8) c->free_pools.push_back({img->pool, img->pool_bytes});
    else { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(img->pool); }
  }
  delete img;
}

// Dssim::create_image_rgb / create_image_rgba on a device-resident packed frame
int dssim_create_image(mi355_ctx *ctx, const uint8_t *d_frame, int stride, int width, int height, int channels, mi355_dssim_image **out) {
  *out = nullptr;
  // scale geometry
  int ws[kDssimScales], hs[kDssimScales], ns = 0;
  for (int w = width, h = height; ns < kDssimScales; ) {
    ws[ns] = w; hs[ns] = h; ns++;
    if (w < 8 || h < 8) break;  // downsample() -> None
    w /= 2; h /= 2;
  }
  size_t planes_px = 0;
  for (int k = 0; k < ns; k++) planes_px += (size_t)ws[k] * hs[k];
  mi355_dssim_image *img = new mi355_dssim_image();
  int rc = MI355_OK;
  img->pool_bytes = planes_px * 9 * sizeof(float);
  {
    DssimCache *c = dssim_cache(ctx);
    for (size_t i = 0; i < c->free_pools.size(); i++)
      if (c->free_pools[i].bytes == img->pool_bytes) { img->pool = c->free_pools[i].p; c->free_pools.erase(c->free_pools.begin() + (std::ptrdiff_t)i); break; }
  }
  if (!img->pool) rc = check_hip(ctx, hipMalloc((void **)&img->pool, img->pool_bytes), "hipMalloc(dssim image)");
  if (rc) { delete img; return rc; }
  float *p = img->pool;
  img->n_scales = ns;
  for (int k = 0; k < ns; k++) {
    DssimScale &s = img->s[k];
    s.w = ws[k]; s.h = hs[k];
    const size_t n = (size_t)s.w * s.h;
    for (int c = 0; c < 3; c++) { s.img[c] = p; p += n; s.mu[c] = p; p += n; s.sq[c] = p; p += n; }
  }
  // scratch: the linear float4 images of scales 1.. (scale 0 is converted from the bytes on the fly)
  size_t lin_px = 0;
  for (int k = 1; k < ns; k++) lin_px += (size_t)ws[k] * hs[k];
  void *scr = nullptr;
  if ((rc = dssim_scratch(ctx, 1, (lin_px + 16) * 16 + 1024, &scr))) { dssim_free_image(ctx, img); return rc; }
  float4 *lin[kDssimScales] = {nullptr};
  {
    float4 *q = (float4 *)scr;
    for (int k = 1; k < ns; k++) { lin[k] = q; q += (size_t)ws[k] * hs[k]; }
  }
  DssimCache *cache = dssim_cache(ctx);
  if ((rc = dssim_gamma_table(ctx, cache))) { dssim_free_image(ctx, img); return rc; }
  float *d_lut = cache->d_lut;
  // the downsampling chain first (scale 1 straight from the bytes), then the per-scale kernels: scales 0 and 1 on their
  // own, the small ones in one launch
  for (int k = 1; k < ns; k++) {
    const unsigned g = dssim_grid(ctx, (size_t)ws[k] * hs[k]);
    if (k == 1) hipLaunchKernelGGL(dssim_downsample_u8_kernel, dim3(g), dim3(256), 0, ctx->stream, d_frame, stride, width, height, channels, (const float *)d_lut, lin[1]);
    else hipLaunchKernelGGL(dssim_downsample_kernel, dim3(g), dim3(256), 0, ctx->stream, (const float4 *)lin[k - 1], ws[k - 1], hs[k - 1], lin[k]);
  }
  auto job_of = [&](int k) {
    DssimScaleJob j;
    const DssimScale &s = img->s[k];
    if (k == 0) { j.S.u8 = d_frame; j.S.stride = stride; j.S.channels = channels; j.S.lut = d_lut; j.S.lin = nullptr; }
    else { j.S.u8 = nullptr; j.S.stride = 0; j.S.channels = 0; j.S.lut = nullptr; j.S.lin = lin[k]; }
    j.S.pattern = (channels == 4 && !ctx->dssim_translucent) ? 1 : 0;
    j.w = s.w; j.h = s.h;
    for (int c = 0; c < 3; c++) { j.O.img[c] = s.img[c]; j.O.mu[c] = s.mu[c]; j.O.sq[c] = s.sq[c]; }
    return j;
  };
  auto tiles_of = [&](int k) { return (unsigned)(((img->s[k].w + kTw - 1) / kTw) * ((img->s[k].h + kTh - 1) / kTh)); };
  for (int k = 0; k < ns; ) {
    DssimScaleJobs J;
    const int count = k < 2 ? 1 : (ns - k < 3 ? ns - k : 3);
    unsigned total = 0;
    for (int j = 0; j < 3; j++) {
      J.first[j] = total;
      if (j < count) { J.job[j] = job_of(k + j); total += tiles_of(k + j); }
      else J.job[j] = J.job[0];
    }
    J.first[3] = total;
    for (int j = count; j < 3; j++) J.first[j] = total;  // empty ranges
    hipLaunchKernelGGL(dssim_scale_fused_kernel, dim3(total), dim3(kNt), 0, ctx->stream, J);
    k += count;
  }
  rc = check_hip(ctx, hipGetLastError(), "dssim kernel launch");
  if (rc) { dssim_free_image(ctx, img); return rc; }
  // No host wait here: the scratch (linear images of scales 1..) is reused by the next call on this context, which is behind
  // these kernels in stream order; compare() ends with the only synchronisation of a comparison.
  *out = img;
  return MI355_OK;
}

// diagnostics: copy one plane (kind 0 img, 1 mu, 2 img_sq_blur) of one scale/channel to the host; dims via w/h
int dssim_image_plane(mi355_ctx *ctx, const mi355_dssim_image *img, int scale, int channel, int kind, float *out, int *w, int *h) {
  if (scale < 0 || scale >= img->n_scales || channel < 0 || channel > 2 || kind < 0 || kind > 2)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: bad plane selector");
  const DssimScale &s = img->s[scale];
  if (w) *w = s.w;
  if (h) *h = s.h;
  if (!out) return MI355_OK;
  const float *p = kind == 0 ? s.img[channel] : (kind == 1 ? s.mu[channel] : s.sq[channel]);
  int rc = check_hip(ctx, hipMemcpyAsync(out, p, (size_t)s.w * s.h * sizeof(float), hipMemcpyDeviceToHost, ctx->stream), "dssim: plane D2H");
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "dssim: sync");
}

// Dssim::compare -> the f64 value. One fused kernel per scale + two tiny reductions; a single D2H/sync at the end.
int dssim_compare(mi355_ctx *ctx, const mi355_dssim_image *a, const mi355_dssim_image *b, double *out) {
  if (a->n_scales != b->n_scales || a->s[0].w != b->s[0].w || a->s[0].h != b->s[0].h)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: images differ in size");
  // scratch: the SSIM maps of all scales, the two families of block partials, the result slots
  DssimRed R;
  R.n_scales = a->n_scales;
  size_t map_px = 0, n_pa = 0;
  unsigned n_pb = 0;
  unsigned tiles[kDssimScales];
  for (int k = 0; k < a->n_scales; k++) {
    const size_t n = (size_t)a->s[k].w * a->s[k].h;
    tiles[k] = (unsigned)(((a->s[k].w + kTw - 1) / kTw) * ((a->s[k].h + kTh - 1) / kTh));
    R.n[k] = n;
    R.n_a[k] = tiles[k];
    R.first_b[k] = n_pb;
    R.len[k] = (double)n;
    R.exponent[k] = std::pow(0.5, (double)k);
    map_px += n + (n & 1);
    n_pa += tiles[k];
    n_pb += dssim_grid(ctx, n);
  }
  R.first_b[a->n_scales] = n_pb;
  for (int k = a->n_scales; k < kDssimScales; k++) { R.n[k] = 0; R.n_a[k] = 0; R.first_b[k + 1] = n_pb; R.len[k] = 1.0; R.exponent[k] = 1.0; R.map[k] = nullptr; R.part_a[k] = nullptr; R.part_b[k] = nullptr; }
  void *scr = nullptr;
  int rc = dssim_scratch(ctx, 1, map_px * 4 + (n_pa + n_pb) * 8 + 64 * 8 + 64, &scr);
  if (rc) return rc;
  float *map = (float *)scr;
  double *d_pa = (double *)(map + map_px), *d_pb = d_pa + n_pa;
  double *d_slots = d_pb + n_pb;  // per scale: [sum, avg, dev]
  R.slots = d_slots;
  {
    float *m = map;
    double *pa = d_pa;
    for (int k = 0; k < a->n_scales; k++) {
      R.map[k] = m; m += R.n[k] + (R.n[k] & 1);
      R.part_a[k] = pa; pa += tiles[k];
      R.part_b[k] = d_pb + R.first_b[k];
    }
  }
  for (int k = 0; k < a->n_scales; k++) {
    const DssimScale &s1 = a->s[k], &s2 = b->s[k];
    DssimCmp P;
    for (int c = 0; c < 3; c++) {
      P.img1[c] = s1.img[c]; P.img2[c] = s2.img[c]; P.mu1[c] = s1.mu[c]; P.mu2[c] = s2.mu[c]; P.sq1[c] = s1.sq[c]; P.sq2[c] = s2.sq[c];
    }
    hipLaunchKernelGGL(dssim_compare_fused_kernel, dim3(tiles[k]), dim3(kNt), 0, ctx->stream, P, s1.w, s1.h, (float *)R.map[k], (double *)R.part_a[k]);
  }
  hipLaunchKernelGGL(dssim_avg_kernel, dim3(a->n_scales), dim3(256), 0, ctx->stream, R);
  hipLaunchKernelGGL(dssim_absdev2_kernel, dim3(n_pb), dim3(256), 0, ctx->stream, R);
  hipLaunchKernelGGL(dssim_sum_kernel, dim3(a->n_scales), dim3(256), 0, ctx->stream, R);
  if ((rc = check_hip(ctx, hipGetLastError(), "dssim kernel launch"))) return rc;
  double slots[3 * kDssimScales];
  if ((rc = check_hip(ctx, hipMemcpyAsync(slots, d_slots, sizeof(double) * 3 * a->n_scales, hipMemcpyDeviceToHost, ctx->stream), "dssim: scores D2H"))) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "dssim: sync"))) return rc;
  *out = dssim_value_of(a, slots);
  return MI355_OK;
}

// videocompare's loop over the non-reference pads of one aggregate (imp.rs:316-345): frames[i] is hashed and compared against
// `a` (the reference pad's image) in one pass per scale, nothing of frames[i] is kept. All frames are queued before the single
// synchronisation that returns the values; the scratch (linear images, SSIM maps, partials) is shared in stream order.
// `d_refs` (optional): frame f is compared against the image of d_refs[f] instead of `a` - a PAIR per frame, what n independent
// two-pad videocompare elements hand over in one interval (group.hip: mi355_group_submit_compare). The reference's image is
// created right before the kernels that read it and released right behind them: one pool serves the whole call in stream order.
// `h_slots` (optional, pinned, n_frames x 3 x kDssimScales doubles): the per-scale [sum, avg, dev] slots are copied there and the
// call returns WITHOUT waiting (the caller owns an event behind it and finishes with dssim_scores_from_slots); otherwise the call
// waits and writes the values to `out`.
static int dssim_compare_frames_impl(mi355_ctx *ctx, const mi355_dssim_image *a, const uint8_t *const *d_refs, const uint8_t *const *d_frames, int n_frames,
                                     int stride, int width, int height, int channels, double *out, double *h_slots) {
  mi355_dssim_image geometry;   // scale sizes only (pairs: there is no image yet)
  if (!a) {
    int ns0 = 0;
    for (int w = width, h = height; ns0 < kDssimScales; ) {
      geometry.s[ns0].w = w; geometry.s[ns0].h = h; ns0++;
      if (w < 8 || h < 8) break;
      w /= 2; h /= 2;
    }
    geometry.n_scales = ns0;
    a = &geometry;
  }
  if (a->s[0].w != width || a->s[0].h != height) return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: frame and image differ in size");
  const int ns = a->n_scales;
  if (n_frames > 64) return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: at most 64 frames per call");
  DssimRed R;
  R.n_scales = ns;
  size_t map_px = 0, n_pa = 0, lin_px = 0;
  unsigned n_pb = 0;
  unsigned tiles[kDssimScales];
  for (int k = 0; k < ns; k++) {
    const size_t n = (size_t)a->s[k].w * a->s[k].h;
    tiles[k] = (unsigned)(((a->s[k].w + kTw - 1) / kTw) * ((a->s[k].h + kTh - 1) / kTh));
    R.n[k] = n;
    R.n_a[k] = tiles[k];
    R.first_b[k] = n_pb;
    R.len[k] = (double)n;
    R.exponent[k] = std::pow(0.5, (double)k);
    map_px += n + (n & 1);
    n_pa += tiles[k];
    n_pb += dssim_grid(ctx, n);
    if (k > 0) lin_px += n;
  }
  R.first_b[ns] = n_pb;
  for (int k = ns; k < kDssimScales; k++) { R.n[k] = 0; R.n_a[k] = 0; R.first_b[k + 1] = n_pb; R.len[k] = 1.0; R.exponent[k] = 1.0; R.map[k] = nullptr; R.part_a[k] = nullptr; R.part_b[k] = nullptr; }
  void *scr = nullptr;
  const size_t lin_bytes = (lin_px + 16) * 16;
  // per frame: SSIM maps and both families of partials (the reductions of all frames run after the last frame's kernels)
  const size_t F = (size_t)n_frames;
  int rc = dssim_scratch(ctx, 1, lin_bytes + F * (map_px * 4 + (n_pa + n_pb) * 8 + 3 * kDssimScales * 8) + 1024, &scr);
  if (rc) return rc;
  float4 *lin[kDssimScales] = {nullptr};
  {
    float4 *q = (float4 *)scr;
    for (int k = 1; k < ns; k++) { lin[k] = q; q += (size_t)a->s[k].w * a->s[k].h; }
  }
  float *map = (float *)((char *)scr + lin_bytes);
  double *d_pa = (double *)(map + F * map_px), *d_pb = d_pa + F * n_pa;
  double *d_slots = d_pb + F * n_pb;  // per frame, per scale: [sum, avg, dev]
  R.frame_map = map_px; R.frame_pa = n_pa; R.frame_pb = n_pb; R.frame_slots = 3 * kDssimScales;
  R.slots = d_slots;
  {
    float *m = map;
    double *pa = d_pa;
    for (int k = 0; k < ns; k++) {
      R.map[k] = m; m += R.n[k] + (R.n[k] & 1);
      R.part_a[k] = pa; pa += tiles[k];
      R.part_b[k] = d_pb + R.first_b[k];
    }
  }
  DssimCache *cache = dssim_cache(ctx);
  if ((rc = dssim_gamma_table(ctx, cache))) return rc;
  float *d_lut = cache->d_lut;
  for (int f = 0; f < n_frames; f++) {
    const uint8_t *d_frame = d_frames[f];
    mi355_dssim_image *own = nullptr;   // pairs: this frame's reference image (its scratch use - the linear images at the start of
                                        // slot 1 - is what the loop below reuses right behind it, in stream order; never larger)
    if (d_refs) {
      if ((rc = dssim_create_image(ctx, d_refs[f], stride, width, height, channels, &own))) return rc;
      if (ctx->d_stage[1] != scr) { dssim_free_image(ctx, own); return set_error(ctx, MI355_ERR_HIP, "dssim: scratch moved under a pair"); }
    }
    const mi355_dssim_image *orig = own ? own : a;
    for (int k = 1; k < ns; k++) {
      const unsigned g = dssim_grid(ctx, R.n[k]);
      if (k == 1) hipLaunchKernelGGL(dssim_downsample_u8_kernel, dim3(g), dim3(256), 0, ctx->stream, d_frame, stride, width, height, channels, (const float *)d_lut, lin[1]);
      else hipLaunchKernelGGL(dssim_downsample_kernel, dim3(g), dim3(256), 0, ctx->stream, (const float4 *)lin[k - 1], a->s[k - 1].w, a->s[k - 1].h, lin[k]);
    }
    auto job_of = [&](int k) {
      DssimFusedJob j;
      const DssimScale &s = orig->s[k];
      if (k == 0) { j.S.u8 = d_frame; j.S.stride = stride; j.S.channels = channels; j.S.lut = d_lut; j.S.lin = nullptr; }
      else { j.S.u8 = nullptr; j.S.stride = 0; j.S.channels = 0; j.S.lut = nullptr; j.S.lin = lin[k]; }
      j.S.pattern = (channels == 4 && !ctx->dssim_translucent) ? 1 : 0;
      j.w = s.w; j.h = s.h;
      for (int c = 0; c < 3; c++) { j.img1[c] = s.img[c]; j.mu1[c] = s.mu[c]; j.sq1[c] = s.sq[c]; }
      j.map = (float *)R.map[k] + (size_t)f * map_px;
      j.partial = (double *)R.part_a[k] + (size_t)f * n_pa;
      return j;
    };
    for (int k = 0; k < ns; ) {
      DssimFusedJobs J;
      const int count = k < 2 ? 1 : (ns - k < 3 ? ns - k : 3);
      unsigned total = 0;
      for (int j = 0; j < 3; j++) {
        J.first[j] = total;
        if (j < count) { J.job[j] = job_of(k + j); total += tiles[k + j]; }
        else J.job[j] = J.job[0];
      }
      J.first[3] = total;
      for (int j = count; j < 3; j++) J.first[j] = total;
      hipLaunchKernelGGL(dssim_hash_compare_kernel, dim3(total), dim3(kNt), 0, ctx->stream, J);
      k += count;
    }
    if (own) dssim_free_image(ctx, own);   // back to the context's free list: the next pair's create_image takes it, behind these kernels
  }
  hipLaunchKernelGGL(dssim_avg_kernel, dim3(ns, n_frames), dim3(256), 0, ctx->stream, R);
  hipLaunchKernelGGL(dssim_absdev2_kernel, dim3(n_pb, n_frames), dim3(256), 0, ctx->stream, R);
  hipLaunchKernelGGL(dssim_sum_kernel, dim3(ns, n_frames), dim3(256), 0, ctx->stream, R);
  if ((rc = check_hip(ctx, hipGetLastError(), "dssim kernel launch"))) return rc;
  if (h_slots) return check_hip(ctx, hipMemcpyAsync(h_slots, d_slots, (size_t)n_frames * 3 * kDssimScales * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "dssim: scores D2H");
  std::vector<double> slots((size_t)n_frames * 3 * kDssimScales);
  if ((rc = check_hip(ctx, hipMemcpyAsync(slots.data(), d_slots, slots.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "dssim: scores D2H"))) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "dssim: sync"))) return rc;
  for (int f = 0; f < n_frames; f++) out[f] = dssim_value_of(a, slots.data() + (size_t)f * 3 * kDssimScales);
  return MI355_OK;
}

int dssim_compare_frames(mi355_ctx *ctx, const mi355_dssim_image *a, const uint8_t *const *d_frames, int n_frames, int stride, int width, int height,
                         int channels, double *out) {
  return dssim_compare_frames_impl(ctx, a, nullptr, d_frames, n_frames, stride, width, height, channels, out, nullptr);
}

// n (reference, frame) pairs of one geometry, enqueued on ctx->stream with their slots copied to `h_slots` (pinned); no host wait
int dssim_compare_pairs_enqueue(mi355_ctx *ctx, const uint8_t *const *d_refs, const uint8_t *const *d_frames, int n_pairs, int stride, int width, int height,
                                int channels, double *h_slots) {
  return dssim_compare_frames_impl(ctx, nullptr, d_refs, d_frames, n_pairs, stride, width, height, channels, nullptr, h_slots);
}

// ... and the values once the copy has landed
void dssim_scores_from_slots(int width, int height, const double *h_slots, int n_pairs, double *out) {
  mi355_dssim_image g;
  int ns = 0;
  for (int w = width, h = height; ns < kDssimScales; ) {
    g.s[ns].w = w; g.s[ns].h = h; ns++;
    if (w < 8 || h < 8) break;
    w /= 2; h /= 2;
  }
  g.n_scales = ns;
  for (int f = 0; f < n_pairs; f++) out[f] = dssim_value_of(&g, h_slots + (size_t)f * 3 * kDssimScales);
}

// every f32 with bits in [lo_bits, hi_bits]: the cube root with the trimmed division, scalar and packed, against the one with the compiler's
__global__ __launch_bounds__(256) void dssim_cbrt_selftest_kernel(uint32_t lo_bits, uint64_t count, unsigned long long *mismatches) {
  unsigned long long bad = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
    const float x = __uint_as_float(lo_bits + (uint32_t)i);
    const uint32_t lit = __float_as_uint(dssim_cbrt_poly<true>(x));
    bad += __float_as_uint(dssim_cbrt_poly<false>(x)) != lit;
    // the packed form: x in either element, an unrelated value in the other
    const dssim_f2 p0 = dssim_cbrt_poly2(dssim_f2{x, 0.731f}), p1 = dssim_cbrt_poly2(dssim_f2{0.0123f, x});
    bad += __float_as_uint(p0.x) != lit || __float_as_uint(p1.y) != lit;
  }
  if (bad) atomicAdd(mismatches, bad);
}

int dssim_cbrt_selftest(mi355_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches) {
  if (hi_bits < lo_bits) return set_error(ctx, MI355_ERR_INVALID_ARG, "selftest: empty range");
  unsigned long long *d = nullptr;
  int rc = check_hip(ctx, hipMalloc((void **)&d, 8), "hipMalloc(selftest)");
  if (rc) return rc;
  (void)hipMemsetAsync(d, 0, 8, ctx->stream);
  hipLaunchKernelGGL(dssim_cbrt_selftest_kernel, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream, lo_bits, (uint64_t)(hi_bits - lo_bits) + 1, d);
  unsigned long long h = 0;
  rc = check_hip(ctx, hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream), "selftest D2H");
  if (!rc) rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "selftest sync");
  (void)hipFree(d);
  *mismatches = h;
  return rc;
}

}  // namespace mi355
