// hsv_kernels.hip — gfx950 kernels for hsvfilter / hsvdetector.
//
// Reference loops replaced (gst-plugins-rs tree):
//   video/hsv/src/hsvutils.rs:44-128    from_rgb / from_bgr
//   video/hsv/src/hsvutils.rs:132-198   to_rgb / to_bgr
//   video/hsv/src/hsvfilter/imp.rs:96-118     per-pixel filter body
//   video/hsv/src/hsvdetector/imp.rs:126-156  per-pixel detector body
//
// Two arithmetic variants per kernel:
//   GENERIC — literal transliteration (IEEE `/`, fmodf, every branch incl. NaN/inf settings).
//   FAST    — strength-reduced but bit-identical for `hue_shift == 0 || 1e-30 <= |hue_shift| <= 360`
//             (finite): exact constant divisions (exact_math.hpp), rcp+residual quotients,
//             bounded-range fmod, sextant select as one LDS lookup + v_perm_b32.
//             Facts it relies on, all checked exhaustively over the 2^24 colours by the tests:
//             hue in [0,360) before the shift (so `hue % 360` is the identity), saturation and
//             value already inside [0,1], results of to_rgb inside [0,255.0001].
// The pixel filters are pure streaming kernels: 4 B read + 4 B written per pixel, HBM-bound
// once the VALU work per pixel is below ~80 instructions.
#include "internal.hpp"
#include "exact_math.hpp"

namespace mi355 {

struct HsvK {
  float hue_shift, sat_mul, sat_off, val_mul, val_off;
};

// ---------------------------------------------------------------- RGB -> HSV

// r8,g8,b8: integer-valued floats 0..255.
template <bool FAST>
__device__ __forceinline__ void hsv_from_rgb(float r8, float g8, float b8, float &hue, float &sat,
                                             float &val) {
  if constexpr (FAST) {
    const float r = div255_u8(r8), g = div255_u8(g8), b = div255_u8(b8);
    // x/255 is monotone, so max of the quotients == quotient of the max (hsvutils.rs:49-59).
    const float value = fmaxf(fmaxf(r, g), b);
    const float minv = fminf(fminf(r, g), b);
    const float chroma = value - minv;
    // |value - c| < 1e-5  <=>  c == value: distinct u8/255 quotients differ by > 3.9e-3.
    const bool is_r = (r == value);
    const bool is_g = (g == value);
    const float na = is_r ? g : (is_g ? b : r);
    const float nb = is_r ? b : (is_g ? r : g);
    const float add = is_r ? 0.0f : (is_g ? 2.0f : 4.0f);
    // chroma == 0 => all channels equal => numerator 0 => quotient 0 => hue 0 (hsvutils.rs:61-62).
    const float q = div_rcp_refine(na - nb, fmaxf(chroma, 1e-30f));
    float h = 60.0f * (add + q);
    h = (h < 0.0f) ? h + 360.0f : h;
    hue = h;  // h in [0,360): `% 360.0` is the identity
    // value == 0 => chroma == 0 => 0 (hsvutils.rs:77); quotient already within [0,1].
    sat = div_rcp_refine(chroma, fmaxf(value, 1e-30f));
    val = value;
  } else {
    const float r = r8 / 255.0f, g = g8 / 255.0f, b = b8 / 255.0f;
    const float value = fmaxf(fmaxf(r8, g8), b8) / 255.0f;
    const float chroma = value - (fminf(fminf(r8, g8), b8) / 255.0f);
    const float EPS = 0.00001f;
    float h;
    if (chroma == 0.0f) {
      h = 0.0f;
    } else if (fabsf(value - r) < EPS) {
      h = 60.0f * ((g - b) / chroma);
    } else if (fabsf(value - g) < EPS) {
      h = 60.0f * (2.0f + ((b - r) / chroma));
    } else if (fabsf(value - b) < EPS) {
      h = 60.0f * (4.0f + ((r - g) / chroma));
    } else {
      h = 0.0f;
    }
    if (h < 0.0f) h += 360.0f;
    const float s = (value == 0.0f) ? 0.0f : chroma / value;
    hue = fmodf(h, 360.0f);
    sat = rs_clamp(s, 0.0f, 1.0f);
    val = rs_clamp(value, 0.0f, 1.0f);
  }
}

// ---------------------------------------------------------------- HSV -> RGB (generic)

__device__ __forceinline__ void hsv_to_rgb_generic(float h, float s, float v, uint32_t &r, uint32_t &g,
                                                   uint32_t &b) {
  const float c = v * s;
  const float hp = h / 60.0f;
  const float x = c * (1.0f - fabsf(fmodf(hp, 2.0f) - 1.0f));
  float rp, gp, bp;
  if (hp < 0.0f) { rp = 0.0f; gp = 0.0f; bp = 0.0f; }
  else if (hp <= 1.0f) { rp = c; gp = x; bp = 0.0f; }
  else if (hp <= 2.0f) { rp = x; gp = c; bp = 0.0f; }
  else if (hp <= 3.0f) { rp = 0.0f; gp = c; bp = x; }
  else if (hp <= 4.0f) { rp = 0.0f; gp = x; bp = c; }
  else if (hp <= 5.0f) { rp = x; gp = 0.0f; bp = c; }
  else if (hp <= 6.0f) { rp = c; gp = 0.0f; bp = x; }
  else { rp = 0.0f; gp = 0.0f; bp = 0.0f; }
  const float m = v - c;
  r = rs_as_u8(rs_clamp((rp + m) * 255.0f, 0.0f, 255.0f));
  g = rs_as_u8(rs_clamp((gp + m) * 255.0f, 0.0f, 255.0f));
  b = rs_as_u8(rs_clamp((bp + m) * 255.0f, 0.0f, 255.0f));
}

// ---------------------------------------------------------------- sextant selector table
// FAST to_rgb produces three byte candidates A=(c+m), B=(x+m), C=(m) packed as bytes 0,1,2 of one
// dword; the sextant decides which candidate lands in which channel (hsvutils.rs:138-154):
//   k=ceil(h/60): 0,1 -> (R,G,B)=(A,B,C)  2 -> (B,A,C)  3 -> (C,A,B)  4 -> (C,B,A)  5 -> (B,C,A)  6 -> (A,C,B)
// v_perm_b32(orig_pixel, packedABC, sel): selector byte 0..3 picks a byte of packedABC, 4..7 a byte
// of the original pixel (the untouched x/alpha byte).
__host__ __device__ constexpr uint32_t hsv_sel_entry(int k, int rpos, int gpos, int bpos, int npos) {
  const int codes[7][3] = {{0, 1, 2}, {0, 1, 2}, {1, 0, 2}, {2, 0, 1}, {2, 1, 0}, {1, 2, 0}, {0, 2, 1}};
  uint32_t sel = 0;
  sel |= (uint32_t)codes[k][0] << (8 * rpos);
  sel |= (uint32_t)codes[k][1] << (8 * gpos);
  sel |= (uint32_t)codes[k][2] << (8 * bpos);
  sel |= (uint32_t)(4 + npos) << (8 * npos);
  return sel;
}

// One pixel, 4-byte formats. RPOS/GPOS/BPOS = byte index of each channel inside the little-endian
// dword, NPOS = the untouched byte.
template <bool FAST, int RPOS, int GPOS, int BPOS, int NPOS>
__device__ __forceinline__ uint32_t hsvfilter_px(uint32_t p, const HsvK &k, const uint32_t *sel_tab) {
  const float r8 = (float)((p >> (8 * RPOS)) & 0xffu);
  const float g8 = (float)((p >> (8 * GPOS)) & 0xffu);
  const float b8 = (float)((p >> (8 * BPOS)) & 0xffu);
  float h, s, v;
  hsv_from_rgb<FAST>(r8, g8, b8, h, s, v);
  if constexpr (FAST) {
    // (h + shift) % 360 with h in [0,360), |shift| <= 360: one conditional exact subtraction
    // (Sterbenz for t in [360,720]); t == -360 maps to +0 instead of fmod's -0 (indistinguishable
    // downstream). Then the reference's `if h < 0 { h += 360 }` (hsvfilter/imp.rs:102-105).
    float t = h + k.hue_shift;
    t = (t >= 360.0f) ? t - 360.0f : t;
    t = (t < 0.0f) ? t + 360.0f : t;
    // crate Clamp trait = max-then-min, NaN -> 0 (hsvfilter/imp.rs:106-115, hsvutils.rs:23-38)
    s = fminf(fmaxf(k.sat_mul * s + k.sat_off, 0.0f), 1.0f);
    v = fminf(fmaxf(k.val_mul * v + k.val_off, 0.0f), 1.0f);
    // to_rgb / to_bgr (hsvutils.rs:132-198), t in [0,360], s,v in [0,1]
    const float c = v * s;
    const float hp = div60_hue(t);
    // hp % 2 == 2*fract(hp/2) exactly for hp >= 0; (hp % 2) - 1 rounds once in the fma.
    const float w = __builtin_fmaf(2.0f, __builtin_amdgcn_fractf(hp * 0.5f), -1.0f);
    const float x = c * (1.0f - fabsf(w));
    const float m = v - c;
    const uint32_t a8 = (uint32_t)((c + m) * 255.0f);  // values in [0,255.0001]: trunc == `as u8`
    const uint32_t b8o = (uint32_t)((x + m) * 255.0f);
    const uint32_t c8 = (uint32_t)(m * 255.0f);
    const uint32_t packed = a8 | (b8o << 8) | (c8 << 16);
    const uint32_t sel = sel_tab[(uint32_t)ceilf(hp)];
    return __builtin_amdgcn_perm(p, packed, sel);
  } else {
    h = fmodf(h + k.hue_shift, 360.0f);
    if (h < 0.0f) h += 360.0f;
    s = fminf(fmaxf(k.sat_mul * s + k.sat_off, 0.0f), 1.0f);
    v = fminf(fmaxf(k.val_mul * v + k.val_off, 0.0f), 1.0f);
    uint32_t r, g, b;
    hsv_to_rgb_generic(h, s, v, r, g, b);
    const uint32_t keep = p & (0xffu << (8 * NPOS));
    return keep | (r << (8 * RPOS)) | (g << (8 * GPOS)) | (b << (8 * BPOS));
  }
}

// ---------------------------------------------------------------- FAST path, two pixels per call
//
// Measured gfx950 VALU issue costs (tools/valu_bench.hip, cycles per wave64 instruction per SIMD):
// v_add/sub/mul_f32 and v_ashrrev ~2.6; every other VALU op (v_fma, v_cndmask, v_cmp, v_cvt_*, v_perm,
// v_min/max, bit ops) ~4.3; v_pk_mul/add/fma_f32 ~4.8 for two results; v_rcp_f32 ~8.5. The filter is
// VALU-bound, so this version minimises the "slow" class:
//   * the max channel is rotated to byte 0 with one v_perm_b32 (selector picked by two SDWA byte
//     compares), which also delivers the hue sector constant (0/2/4) as byte 3 — no float selects;
//   * fused multiply-adds are issued pairwise over the two pixels (v_pk_fma_f32);
//   * sign fix-ups use v_ashrrev + v_and + v_add instead of compare+select;
//   * the sextant index is floor(h/60) (not ceil): at integer h/60 both neighbouring sextants give
//     identical triples (x == c or x == 0 there), so one v_cvt_u32 feeds the LDS selector lookup;
//   * the three output bytes are converted straight into their byte lanes (SDWA v_cvt_u32_f32);
//   * hue-shift sign and identity saturation/value settings are compile-time variants.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }

enum HsvShiftClass { HSV_SHIFT_ZERO = 0, HSV_SHIFT_POS = 1, HSV_SHIFT_NEG = 2 };

// selector table indexed by floor(h/60) in 0..6 (see hsv_sel_entry for the byte codes):
//   0:(A,B,C) 1:(B,A,C) 2:(C,A,B) 3:(C,B,A) 4:(B,C,A) 5,6:(A,C,B)
__host__ __device__ constexpr uint32_t hsv_sel_entry_floor(int k, int rpos, int gpos, int bpos, int npos) {
  return hsv_sel_entry(k < 6 ? k + 1 : 6, rpos, gpos, bpos, npos);
}

// x + 360 if x < 0 else x, without compare/select: (bits(x) >>s 31) & bits(360.0f).
__device__ __forceinline__ float add360_if_negative(float x) {
  const int m = __float_as_int(x) >> 31;
  return x + __int_as_float(m & 0x43b40000);
}

__device__ __forceinline__ void cvt_u8_into(uint32_t &packed, float v, int byte) {
  // `as u8` of a value known to lie in [0,255.0001]: truncating convert written into one byte lane
  if (byte == 0) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(v));
  else if (byte == 1) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(v));
  else asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(v));
}

// from_rgb / from_bgr (hsvutils.rs:44-128) for two pixels: hue in [0,360), saturation, value.
template <int RPOS, int GPOS, int BPOS>
__device__ __forceinline__ void hsv_from_rgb_pair_fast(uint32_t p0, uint32_t p1, f2 &h, f2 &sat, f2 &value) {
  // rotate so that byte0 = max channel M, bytes 1,2 = the other two in the cyclic order the hue
  // formula subtracts them, byte3 = 0/2/4
  constexpr uint32_t SEL_R = (uint32_t)RPOS | ((uint32_t)GPOS << 8) | ((uint32_t)BPOS << 16) | (4u << 24);
  constexpr uint32_t SEL_G = (uint32_t)GPOS | ((uint32_t)BPOS << 8) | ((uint32_t)RPOS << 16) | (5u << 24);
  constexpr uint32_t SEL_B = (uint32_t)BPOS | ((uint32_t)RPOS << 8) | ((uint32_t)GPOS << 16) | (6u << 24);
  uint32_t rot[2];
  {
    const uint32_t p[2] = {p0, p1};
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const uint32_t r = (p[j] >> (8 * RPOS)) & 0xffu, g = (p[j] >> (8 * GPOS)) & 0xffu, b = (p[j] >> (8 * BPOS)) & 0xffu;
      // first of R,G,B equal to the max (hsvutils.rs:63-68)
      const uint32_t sel = (r >= g && r >= b) ? SEL_R : (g >= b ? SEL_G : SEL_B);
      rot[j] = __builtin_amdgcn_perm(0x00040200u, p[j], sel);
    }
  }
  const f2 M8 = {(float)(rot[0] & 0xffu), (float)(rot[1] & 0xffu)};
  const f2 a8 = {(float)((rot[0] >> 8) & 0xffu), (float)((rot[1] >> 8) & 0xffu)};
  const f2 b8 = {(float)((rot[0] >> 16) & 0xffu), (float)((rot[1] >> 16) & 0xffu)};
  const f2 add = {(float)(rot[0] >> 24), (float)(rot[1] >> 24)};
  const f2 hi = splat2(MI355_INV255_HI), lo = splat2(MI355_INV255_LO);
  value = fma2(M8, hi, M8 * lo);  // RN(n/255), see div255_u8
  const f2 af = fma2(a8, hi, a8 * lo);
  const f2 bf = fma2(b8, hi, b8 * lo);
  const f2 minv = {fminf(af.x, bf.x), fminf(af.y, bf.y)};
  const f2 chroma = value - minv;
  const f2 num = af - bf;
  // q = num/chroma, sat = chroma/value: hardware reciprocal + one residual step (div_rcp_refine),
  // the four fmas issued as two packed ones. Zero denominators only occur with zero numerators.
  const f2 dq = {fmaxf(chroma.x, 1e-30f), fmaxf(chroma.y, 1e-30f)};
  const f2 ds = {fmaxf(value.x, 1e-30f), fmaxf(value.y, 1e-30f)};
  const f2 yq = {__builtin_amdgcn_rcpf(dq.x), __builtin_amdgcn_rcpf(dq.y)};
  const f2 ys = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
  f2 q = num * yq;
  sat = chroma * ys;
  q = fma2(fma2(-q, dq, num), yq, q);
  sat = fma2(fma2(-sat, ds, chroma), ys, sat);
  h = splat2(60.0f) * (add + q);
  h.x = add360_if_negative(h.x);
  h.y = add360_if_negative(h.y);
}

template <int RPOS, int GPOS, int BPOS, int NPOS, int SHIFT, bool SV_IDENT>
__device__ __forceinline__ void hsvfilter_px2_fast(uint32_t &p0, uint32_t &p1, const HsvK &k, const uint32_t *sel_tab) {
  f2 h, sat, value;
  hsv_from_rgb_pair_fast<RPOS, GPOS, BPOS>(p0, p1, h, sat, value);
  // ---- filter (hsvfilter/imp.rs:102-115)
  f2 t = h;
  if constexpr (SHIFT == HSV_SHIFT_POS) {
    // t in [0, 720): subtract 360 exactly when t >= 360 (bit-select on the sign of t-360)
    t = h + splat2(k.hue_shift);
    const f2 u = t - splat2(360.0f);
    const int mx = __float_as_int(u.x) >> 31, my = __float_as_int(u.y) >> 31;
    t.x = __int_as_float((__float_as_int(t.x) & mx) | (__float_as_int(u.x) & ~mx));
    t.y = __int_as_float((__float_as_int(t.y) & my) | (__float_as_int(u.y) & ~my));
  } else if constexpr (SHIFT == HSV_SHIFT_NEG) {
    // t in [-360, 360): fmod is the identity (t == -360 -> +0 instead of -0, indistinguishable),
    // then `if h < 0 { h += 360 }`
    t = h + splat2(k.hue_shift);
    t.x = add360_if_negative(t.x);
    t.y = add360_if_negative(t.y);
  }
  f2 s = sat, v = value;
  if constexpr (!SV_IDENT) {
    const f2 s1 = splat2(k.sat_mul) * sat + splat2(k.sat_off);
    const f2 v1 = splat2(k.val_mul) * value + splat2(k.val_off);
    s = f2{fminf(fmaxf(s1.x, 0.0f), 1.0f), fminf(fmaxf(s1.y, 0.0f), 1.0f)};
    v = f2{fminf(fmaxf(v1.x, 0.0f), 1.0f), fminf(fmaxf(v1.y, 0.0f), 1.0f)};
  }
  // ---- to_rgb / to_bgr (hsvutils.rs:132-198)
  const f2 c = v * s;
  const f2 hp = fma2(t, splat2(MI355_INV60_HI), t * splat2(MI355_INV60_LO));  // RN(t/60), see div60_hue
  const f2 hh = hp * splat2(0.5f);
  const f2 fr = {__builtin_amdgcn_fractf(hh.x), __builtin_amdgcn_fractf(hh.y)};
  const f2 w = fma2(splat2(2.0f), fr, splat2(-1.0f));
  const f2 x = c * (splat2(1.0f) - f2{fabsf(w.x), fabsf(w.y)});
  const f2 m = v - c;
  const f2 A = (c + m) * splat2(255.0f), B = (x + m) * splat2(255.0f), C = m * splat2(255.0f);
  uint32_t pk0 = 0, pk1 = 0;
  cvt_u8_into(pk0, A.x, 0); cvt_u8_into(pk0, B.x, 1); cvt_u8_into(pk0, C.x, 2);
  cvt_u8_into(pk1, A.y, 0); cvt_u8_into(pk1, B.y, 1); cvt_u8_into(pk1, C.y, 2);
  const uint32_t sel0 = sel_tab[(uint32_t)hp.x];
  const uint32_t sel1 = sel_tab[(uint32_t)hp.y];
  p0 = __builtin_amdgcn_perm(p0, pk0, sel0);
  p1 = __builtin_amdgcn_perm(p1, pk1, sel1);
}

// ---------------------------------------------------------------- hsvfilter kernels

// Flat streaming kernel for 4-byte formats on contiguous storage (stride == width*4 and frames
// back to back): each lane owns 16 B (4 pixels) per iteration -> global_load/store_dwordx4.
// VARIANT: -1 = GENERIC arithmetic; otherwise FAST with SHIFT = VARIANT & 3, SV_IDENT = VARIANT >> 2.
template <int VARIANT, int FIRST, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_flat_kernel(uint4 *__restrict__ data, size_t n_vec,
                                                             HsvK k) {
  constexpr int RPOS = FIRST + (BGR ? 2 : 0), GPOS = FIRST + 1, BPOS = FIRST + (BGR ? 0 : 2);
  constexpr int NPOS = FIRST == 0 ? 3 : 0;
  constexpr bool FAST = VARIANT >= 0;
  __shared__ uint32_t sel_tab[8];
  if (threadIdx.x < 7)
    sel_tab[threadIdx.x] = FAST ? hsv_sel_entry_floor(threadIdx.x, RPOS, GPOS, BPOS, NPOS)
                                : hsv_sel_entry(threadIdx.x, RPOS, GPOS, BPOS, NPOS);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto one = [&](uint4 &p) {
    if constexpr (FAST) {
      hsvfilter_px2_fast<RPOS, GPOS, BPOS, NPOS, VARIANT & 3, (VARIANT >> 2) != 0>(p.x, p.y, k, sel_tab);
      hsvfilter_px2_fast<RPOS, GPOS, BPOS, NPOS, VARIANT & 3, (VARIANT >> 2) != 0>(p.z, p.w, k, sel_tab);
    } else {
      p.x = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p.x, k, sel_tab);
      p.y = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p.y, k, sel_tab);
      p.z = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p.z, k, sel_tab);
      p.w = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p.w, k, sel_tab);
    }
  };
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // two independent 16 B loads in flight per lane (four measured slower: register pressure)
  for (; i + stride < n_vec; i += 2 * stride) {
    uint4 p = data[i];
    uint4 q = data[i + stride];
    one(p);
    data[i] = p;
    one(q);
    data[i + stride] = q;
  }
  if (i < n_vec) {
    uint4 p = data[i];
    one(p);
    data[i] = p;
  }
}

// 3-byte formats (RGB / BGR) on contiguous storage: one lane = 12 B = 4 pixels. The three dwords are
// split into four pixel words with v_alignbit-style shifts, filtered by the same pair routine as the
// 4-byte formats (byte 3 of each word is scratch) and re-packed.
struct Rgb24x4 { uint32_t d0, d1, d2; };
template <int VARIANT, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_rgb24_kernel(Rgb24x4 *__restrict__ data, size_t n_grp, HsvK k) {
  constexpr int RPOS = BGR ? 2 : 0, GPOS = 1, BPOS = BGR ? 0 : 2, NPOS = 3;
  constexpr bool FAST = VARIANT >= 0;
  __shared__ uint32_t sel_tab[8];
  if (threadIdx.x < 7)
    sel_tab[threadIdx.x] = FAST ? hsv_sel_entry_floor(threadIdx.x, RPOS, GPOS, BPOS, NPOS)
                                : hsv_sel_entry(threadIdx.x, RPOS, GPOS, BPOS, NPOS);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_grp; i += stride) {
    const Rgb24x4 v = data[i];
    uint32_t p0 = v.d0, p1 = (v.d0 >> 24) | (v.d1 << 8), p2 = (v.d1 >> 16) | (v.d2 << 16), p3 = v.d2 >> 8;
    if constexpr (FAST) {
      hsvfilter_px2_fast<RPOS, GPOS, BPOS, NPOS, VARIANT & 3, (VARIANT >> 2) != 0>(p0, p1, k, sel_tab);
      hsvfilter_px2_fast<RPOS, GPOS, BPOS, NPOS, VARIANT & 3, (VARIANT >> 2) != 0>(p2, p3, k, sel_tab);
    } else {
      p0 = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p0, k, sel_tab);
      p1 = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p1, k, sel_tab);
      p2 = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p2, k, sel_tab);
      p3 = hsvfilter_px<false, RPOS, GPOS, BPOS, NPOS>(p3, k, sel_tab);
    }
    Rgb24x4 o;
    o.d0 = (p0 & 0x00ffffffu) | (p1 << 24);
    o.d1 = ((p1 >> 8) & 0x0000ffffu) | (p2 << 16);
    o.d2 = ((p2 >> 16) & 0x000000ffu) | (p3 << 8);
    data[i] = o;
  }
}

template <int VARIANT>
static void launch_rgb24(mi355_ctx *ctx, Rgb24x4 *d, size_t n_grp, const HsvK &k, int bgr, int grid) {
  if (bgr) hipLaunchKernelGGL((hsvfilter_rgb24_kernel<VARIANT, true>), dim3(grid), dim3(256), 0, ctx->stream, d, n_grp, k);
  else hipLaunchKernelGGL((hsvfilter_rgb24_kernel<VARIANT, false>), dim3(grid), dim3(256), 0, ctx->stream, d, n_grp, k);
}

// General kernel: any stride / pixel stride (3 or 4) / triple offset, one pixel per lane, byte
// accesses. Only `line[..width*pixel_stride]` of each row is touched (hsvfilter/imp.rs:94-97).
template <bool FAST>
__global__ __launch_bounds__(256) void hsvfilter_rows_kernel(uint8_t *__restrict__ data, int n_frames,
                                                             size_t frame_pitch, int width, int height,
                                                             int stride, int pixel_stride, int first,
                                                             int bgr, HsvK k) {
  __shared__ uint32_t sel_tab[8];
  if (threadIdx.x < 7) sel_tab[threadIdx.x] = hsv_sel_entry(threadIdx.x, 0, 1, 2, 3);
  __syncthreads();
  const size_t per_frame = (size_t)width * (size_t)height;
  const size_t total = per_frame * (size_t)n_frames;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const size_t f = i / per_frame;
    const size_t r = i - f * per_frame;
    const size_t row = r / (size_t)width;
    const size_t col = r - row * (size_t)width;
    uint8_t *px = data + f * frame_pitch + row * (size_t)stride + col * (size_t)pixel_stride + first;
    const uint32_t c0 = px[0], c1 = px[1], c2 = px[2];
    const uint32_t p = bgr ? (c2 | (c1 << 8) | (c0 << 16)) : (c0 | (c1 << 8) | (c2 << 16));
    const uint32_t o = hsvfilter_px<FAST, 0, 1, 2, 3>(p, k, sel_tab);
    const uint8_t orr = (uint8_t)(o & 0xff), og = (uint8_t)((o >> 8) & 0xff), ob = (uint8_t)((o >> 16) & 0xff);
    px[0] = bgr ? ob : orr;
    px[1] = og;
    px[2] = bgr ? orr : ob;
  }
}

static bool hsv_fast_ok(const mi355_hsv_settings &s) {
  const float a = fabsf(s.hue_shift);
  // NaN fails every comparison -> generic path.
  return (s.hue_shift == 0.0f) || (a >= 1e-30f && a <= 360.0f);
}

static int grid_for(mi355_ctx *ctx, size_t work_items, int block, int blocks_per_cu) {
  size_t blocks = (work_items + (size_t)block - 1) / (size_t)block;
  size_t cap = (size_t)ctx->n_cu * (size_t)blocks_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <int VARIANT>
static void launch_flat(mi355_ctx *ctx, uint4 *d, size_t n_vec, const HsvK &k, int first, int bgr, int grid) {
  dim3 g(grid), b(256);
  if (first == 0 && !bgr) hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 0, false>), g, b, 0, ctx->stream, d, n_vec, k);
  else if (first == 0 && bgr) hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 0, true>), g, b, 0, ctx->stream, d, n_vec, k);
  else if (first == 1 && !bgr) hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 1, false>), g, b, 0, ctx->stream, d, n_vec, k);
  else hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 1, true>), g, b, 0, ctx->stream, d, n_vec, k);
}

int launch_hsvfilter(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width,
                     int height, int stride, const PixFmt &fmt, const mi355_hsv_settings &s) {
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;  // nothing to do
  const HsvK k{s.hue_shift, s.saturation_mul, s.saturation_off, s.value_mul, s.value_off};
  const bool fast = hsv_fast_ok(s) && !ctx->force_generic;
  const size_t row_bytes = (size_t)width * (size_t)fmt.pixel_stride;
  const bool contiguous = fmt.pixel_stride == 4 && (size_t)stride == row_bytes &&
                          (n_frames == 1 || frame_pitch == row_bytes * (size_t)height);
  const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
  if (contiguous && ((uintptr_t)d_data % 16 == 0) && (total_bytes % 16 == 0)) {
    const size_t n_vec = total_bytes / 16;
    const int grid = grid_for(ctx, n_vec, 256, ctx->hsv_blocks_per_cu);
    if (!fast) {
      launch_flat<-1>(ctx, (uint4 *)d_data, n_vec, k, fmt.first, fmt.bgr, grid);
    } else {
      // x*1+0 == x exactly for the x in [0,1] the conversion produces: identity settings skip the affine step
      const bool sv_ident = s.saturation_mul == 1.0f && s.saturation_off == 0.0f && s.value_mul == 1.0f && s.value_off == 0.0f;
      const int shift = s.hue_shift == 0.0f ? HSV_SHIFT_ZERO : (s.hue_shift > 0.0f ? HSV_SHIFT_POS : HSV_SHIFT_NEG);
      uint4 *d = (uint4 *)d_data;
      switch (shift | (sv_ident ? 4 : 0)) {
        case 0: launch_flat<0>(ctx, d, n_vec, k, fmt.first, fmt.bgr, grid); break;
        case 1: launch_flat<1>(ctx, d, n_vec, k, fmt.first, fmt.bgr, grid); break;
        case 2: launch_flat<2>(ctx, d, n_vec, k, fmt.first, fmt.bgr, grid); break;
        case 4: launch_flat<4>(ctx, d, n_vec, k, fmt.first, fmt.bgr, grid); break;
        case 5: launch_flat<5>(ctx, d, n_vec, k, fmt.first, fmt.bgr, grid); break;
        default: launch_flat<6>(ctx, d, n_vec, k, fmt.first, fmt.bgr, grid); break;
      }
    }
  } else if (fmt.pixel_stride == 3 && fmt.first == 0 && (size_t)stride == row_bytes &&
             (n_frames == 1 || frame_pitch == row_bytes * (size_t)height) && ((uintptr_t)d_data % 4 == 0) && (total_bytes % 12 == 0)) {
    const size_t n_grp = total_bytes / 12;
    const int grid = grid_for(ctx, n_grp, 256, ctx->hsv_blocks_per_cu);
    Rgb24x4 *d = (Rgb24x4 *)d_data;
    if (!fast) {
      launch_rgb24<-1>(ctx, d, n_grp, k, fmt.bgr, grid);
    } else {
      const bool sv_ident = s.saturation_mul == 1.0f && s.saturation_off == 0.0f && s.value_mul == 1.0f && s.value_off == 0.0f;
      const int shift = s.hue_shift == 0.0f ? HSV_SHIFT_ZERO : (s.hue_shift > 0.0f ? HSV_SHIFT_POS : HSV_SHIFT_NEG);
      switch (shift | (sv_ident ? 4 : 0)) {
        case 0: launch_rgb24<0>(ctx, d, n_grp, k, fmt.bgr, grid); break;
        case 1: launch_rgb24<1>(ctx, d, n_grp, k, fmt.bgr, grid); break;
        case 2: launch_rgb24<2>(ctx, d, n_grp, k, fmt.bgr, grid); break;
        case 4: launch_rgb24<4>(ctx, d, n_grp, k, fmt.bgr, grid); break;
        case 5: launch_rgb24<5>(ctx, d, n_grp, k, fmt.bgr, grid); break;
        default: launch_rgb24<6>(ctx, d, n_grp, k, fmt.bgr, grid); break;
      }
    }
  } else {
    const size_t total = (size_t)width * (size_t)height * (size_t)n_frames;
    const int grid = grid_for(ctx, total, 256, 32);
    if (fast)
      hipLaunchKernelGGL((hsvfilter_rows_kernel<true>), dim3(grid), dim3(256), 0, ctx->stream, d_data, n_frames,
                         frame_pitch, width, height, stride, fmt.pixel_stride, fmt.first, fmt.bgr, k);
    else
      hipLaunchKernelGGL((hsvfilter_rows_kernel<false>), dim3(grid), dim3(256), 0, ctx->stream, d_data, n_frames,
                         frame_pitch, width, height, stride, fmt.pixel_stride, fmt.first, fmt.bgr, k);
  }
  return check_hip(ctx, hipGetLastError(), "hsvfilter kernel launch");
}

// ---------------------------------------------------------------- hsvdetector

struct HsvDetK {
  float hue_ref, hue_var, sat_ref, sat_var, val_ref, val_var;
};

// One pixel per lane; input 3 or 4 bytes/pixel, output always 4 (hsvdetector/imp.rs:118-156).
// Arithmetic is the GENERIC conversion (IEEE ops) — the detector is a "next" row.
__global__ __launch_bounds__(256) void hsvdetect_rows_kernel(const uint8_t *__restrict__ src, size_t src_pitch,
                                                             int src_stride, int in_pixel_stride, int in_first,
                                                             int in_bgr, uint8_t *__restrict__ dst, size_t dst_pitch,
                                                             int dst_stride, int out_alpha_first, int out_bgr,
                                                             int n_frames, int width, int height, HsvDetK k) {
  const size_t per_frame = (size_t)width * (size_t)height;
  const size_t total = per_frame * (size_t)n_frames;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const size_t f = i / per_frame;
    const size_t r = i - f * per_frame;
    const size_t row = r / (size_t)width;
    const size_t col = r - row * (size_t)width;
    const uint8_t *ip = src + f * src_pitch + row * (size_t)src_stride + col * (size_t)in_pixel_stride + in_first;
    uint8_t *op = dst + f * dst_pitch + row * (size_t)dst_stride + col * 4;
    const uint32_t c0 = ip[0], c1 = ip[1], c2 = ip[2];
    const uint32_t r8 = in_bgr ? c2 : c0, g8 = c1, b8 = in_bgr ? c0 : c2;
    float h, s, v;
    hsv_from_rgb<false>((float)r8, (float)g8, (float)b8, h, s, v);
    const float ref_hue_offset = 180.0f - k.hue_ref;
    float sh = h + ref_hue_offset;
    if (sh < 0.0f) sh += 360.0f;
    sh = fmodf(sh, 360.0f);
    const bool hit = fabsf(sh - 180.0f) <= k.hue_var && fabsf(s - k.sat_ref) <= k.sat_var &&
                     fabsf(v - k.val_ref) <= k.val_var;
    const uint8_t alpha = hit ? 255 : 0;
    uint8_t *c = op + (out_alpha_first ? 1 : 0);
    if (out_bgr) { c[0] = (uint8_t)b8; c[1] = (uint8_t)g8; c[2] = (uint8_t)r8; }
    else { c[0] = (uint8_t)r8; c[1] = (uint8_t)g8; c[2] = (uint8_t)b8; }
    op[out_alpha_first ? 0 : 3] = alpha;
  }
}

// FAST detector kernel: 4-byte input formats on contiguous storage, 4 pixels per lane. Valid for
// hue_ref in [-180,180] (finite): shifted = hue + (180 - hue_ref) lies in [0,720), so the reference's
// `if shifted < 0` never fires and `% 360` is one conditional exact subtraction.
template <int IN_FIRST, bool IN_BGR>
__global__ __launch_bounds__(256) void hsvdetect_flat_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_vec,
                                                             HsvDetK k, uint32_t out_sel) {
  constexpr int RPOS = IN_FIRST + (IN_BGR ? 2 : 0), GPOS = IN_FIRST + 1, BPOS = IN_FIRST + (IN_BGR ? 0 : 2);
  const float off = 180.0f - k.hue_ref;  // ref_hue_offset (hsvdetector/imp.rs:140)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const uint4 p = src[i];
    const uint32_t in[4] = {p.x, p.y, p.z, p.w};
    uint32_t out[4];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      f2 h, s, v;
      hsv_from_rgb_pair_fast<RPOS, GPOS, BPOS>(in[j], in[j + 1], h, s, v);
      f2 sh = h + splat2(off);
      const f2 u = sh - splat2(360.0f);
      sh.x = (u.x >= 0.0f) ? u.x : sh.x;
      sh.y = (u.y >= 0.0f) ? u.y : sh.y;
      const f2 dh = sh - splat2(180.0f), dsat = s - splat2(k.sat_ref), dv = v - splat2(k.val_ref);
      const bool hit0 = fabsf(dh.x) <= k.hue_var && fabsf(dsat.x) <= k.sat_var && fabsf(dv.x) <= k.val_var;
      const bool hit1 = fabsf(dh.y) <= k.hue_var && fabsf(dsat.y) <= k.sat_var && fabsf(dv.y) <= k.val_var;
      // out_sel picks the colour bytes from the input pixel (selector 4..7) and the alpha byte from the
      // second operand (selector 0)
      out[j] = __builtin_amdgcn_perm(in[j], hit0 ? 255u : 0u, out_sel);
      out[j + 1] = __builtin_amdgcn_perm(in[j + 1], hit1 ? 255u : 0u, out_sel);
    }
    dst[i] = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

int launch_hsvdetect(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                     const PixFmt &sfmt, uint8_t *d_dst, size_t dst_pitch, int dst_stride,
                     int dst_alpha_first, int dst_bgr, int n_frames, int width, int height,
                     const mi355_hsvdetect_settings &s) {
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  const HsvDetK k{s.hue_ref, s.hue_var, s.saturation_ref, s.saturation_var, s.value_ref, s.value_var};
  const size_t total = (size_t)width * (size_t)height * (size_t)n_frames;
  const size_t row_bytes = (size_t)width * 4;
  const bool contiguous = sfmt.pixel_stride == 4 && (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                          (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
  const bool fast = !ctx->force_generic && s.hue_ref >= -180.0f && s.hue_ref <= 180.0f;
  if (fast && contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && ((total * 4) % 16 == 0)) {
    // output byte j: colour channel c sits at input byte in_pos(c); alpha comes from operand 1 byte 0
    const int in_pos[3] = {sfmt.first + (sfmt.bgr ? 2 : 0), sfmt.first + 1, sfmt.first + (sfmt.bgr ? 0 : 2)};  // R,G,B
    const int cbase = dst_alpha_first ? 1 : 0;
    uint32_t sel = 0;  // alpha byte selector stays 0 (= byte 0 of the 0/255 operand)
    for (int c = 0; c < 3; c++) {
      const int out_byte = cbase + (dst_bgr ? 2 - c : c);
      sel |= (uint32_t)(4 + in_pos[c]) << (8 * out_byte);
    }
    const size_t n_vec = total / 4;
    const int fgrid = grid_for(ctx, n_vec, 256, 64);
    dim3 g(fgrid), b(256);
    const uint4 *sp = (const uint4 *)d_src;
    uint4 *dp = (uint4 *)d_dst;
    if (sfmt.first == 0 && !sfmt.bgr) hipLaunchKernelGGL((hsvdetect_flat_kernel<0, false>), g, b, 0, ctx->stream, sp, dp, n_vec, k, sel);
    else if (sfmt.first == 0 && sfmt.bgr) hipLaunchKernelGGL((hsvdetect_flat_kernel<0, true>), g, b, 0, ctx->stream, sp, dp, n_vec, k, sel);
    else if (sfmt.first == 1 && !sfmt.bgr) hipLaunchKernelGGL((hsvdetect_flat_kernel<1, false>), g, b, 0, ctx->stream, sp, dp, n_vec, k, sel);
    else hipLaunchKernelGGL((hsvdetect_flat_kernel<1, true>), g, b, 0, ctx->stream, sp, dp, n_vec, k, sel);
    return check_hip(ctx, hipGetLastError(), "hsvdetect flat kernel launch");
  }
  const int grid = grid_for(ctx, total, 256, 32);
  hipLaunchKernelGGL(hsvdetect_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_src, src_pitch, src_stride,
                     sfmt.pixel_stride, sfmt.first, sfmt.bgr, d_dst, dst_pitch, dst_stride, dst_alpha_first, dst_bgr,
                     n_frames, width, height, k);
  return check_hip(ctx, hipGetLastError(), "hsvdetect kernel launch");
}

}  // namespace mi355
