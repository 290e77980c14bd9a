// hsv_device.hpp — device-side hsv arithmetic shared by hsv_kernels.hip and the fused hsvfilter+colorlut
// kernel in colorlut_kernels.hip. See hsv_kernels.hip for the reference citations and the FAST/GENERIC contract.
#pragma once
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

// Host side: arithmetic variant for one settings snapshot. -1 = GENERIC; otherwise FAST with
// SHIFT = v & 3 and SV_IDENT = v >> 2 (x*1+0 == x exactly for the x in [0,1] the conversion produces,
// so identity saturation/value settings skip the affine step).
inline int hsv_variant_for(const mi355_hsv_settings &s, bool force_generic) {
  const float a = fabsf(s.hue_shift);
  // NaN fails every comparison -> generic path.
  const bool fast = (s.hue_shift == 0.0f) || (a >= 1e-30f && a <= 360.0f);
  if (!fast || force_generic) return -1;
  const bool sv_ident = s.saturation_mul == 1.0f && s.saturation_off == 0.0f && s.value_mul == 1.0f && s.value_off == 0.0f;
  const int shift = s.hue_shift == 0.0f ? HSV_SHIFT_ZERO : (s.hue_shift > 0.0f ? HSV_SHIFT_POS : HSV_SHIFT_NEG);
  return shift | (sv_ident ? 4 : 0);
}

}  // namespace mi355
