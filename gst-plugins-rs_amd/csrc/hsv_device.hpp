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
  // hp % 2 == 2 * fract(hp / 2) exactly for hp >= 0 (v_fract_f32; both scalings are by powers of two); for hp < 0 and
  // NaN the arms below do not use x (hsvutils.rs:138-154)
  const float x = c * (1.0f - fabsf(2.0f * __builtin_amdgcn_fractf(hp * 0.5f) - 1.0f));
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

// selector table indexed by floor(h/60) in 0..6 (see hsv_sel_entry for the byte codes):
//   0:(A,B,C) 1:(B,A,C) 2:(C,A,B) 3:(C,B,A) 4:(B,C,A) 5,6:(A,C,B)
__host__ __device__ constexpr uint32_t hsv_sel_entry_floor(int k, int rpos, int gpos, int bpos, int npos) {
  return hsv_sel_entry(k < 6 ? k + 1 : 6, rpos, gpos, bpos, npos);
}

// min of two floats' bit patterns as UNSIGNED integers (v_min_u32). With a >= -0 or b >= -0 this is a select on the
// sign bit: a negative float is a huge unsigned number, two non-negative floats order like their values.
__device__ __forceinline__ float min_bits(float a, float b) {
  const uint32_t x = (uint32_t)__float_as_int(a), y = (uint32_t)__float_as_int(b);
  return __int_as_float((int)(x < y ? x : y));
}
// x < 0 ? x + 360 : x (hsvutils.rs:72-74, hsvfilter/imp.rs:103-105) for x in (-360, 360): the candidate x + 360 is
// computed for the pair (v_pk_add_f32), the choice is one v_min_u32 each: x < 0 -> bits(x) is the larger number.
__device__ __forceinline__ f2 add360_if_negative2(f2 x) {
  const f2 g = x + splat2(360.0f);
  return f2{min_bits(x.x, g.x), min_bits(x.y, g.y)};
}
// t >= 360 ? t - 360 : t for t in [0, 720): t - 360 is negative (huge as an unsigned number) exactly when t < 360 and
// below t otherwise. The subtraction is exact wherever it is chosen (Sterbenz).
__device__ __forceinline__ f2 sub360_if_reached2(f2 t) {
  const f2 u = t - splat2(360.0f);
  return f2{min_bits(t.x, u.x), min_bits(t.y, u.y)};
}
// fmodf(a, 360) for 0 <= a < 2^23, exactly: f0 = floor(a * RN(1/360)) is within one of floor(a/360); a - 360 f0 is a
// multiple of ulp(a) below 720 in magnitude, so the fma and the two wraps are exact.
__device__ __forceinline__ f2 fmod360_abs2(f2 a) {
  const f2 p = a * splat2(0x1.6c16c2p-9f);
  const f2 f0 = {__builtin_floorf(p.x), __builtin_floorf(p.y)};
  f2 r = fma2(-f0, splat2(360.0f), a);
  r = add360_if_negative2(r);
  return sub360_if_reached2(r);
}

__device__ __forceinline__ void cvt_u8_into(uint32_t &packed, float v, int byte) {
  // `as u8` of a value known to lie in [0,255.0001]: truncating convert written into one byte lane; lane 0 comes first
  // and clears the other three
  if (byte == 0) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(packed) : "v"(v));
  else if (byte == 1) asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(v));
  else asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(packed) : "v"(v));
}

// Block-level LDS image of the FAST pixel routine (3,848 B, filled by hsv_lds_fill):
//   value[n] = RN(n / 255)            the reference's `u8 as f32 / 255.0` (hsvutils.rs:45-59)
//   rcpv[n]  = RN(1 / value[n])       correctly rounded reciprocal of it (0 for n = 0: chroma is 0 there too)
//   sext[k]  = {v_perm selector, odd} by k = floor(h / 60): see hsv_sel_entry_floor; odd = 2 floor(k / 2) + 1 is what
//              `hp % 2 - 1` subtracts from hp in one rounding (hsvutils.rs:136)
// value and rcpv are read with the same address register; the gap keeps their distance off the multiples of 256 B (and
// above 1020 B) that ds_read2[st64]_b32 can encode: a fused two-dword read lands in one register PAIR, and the packed
// arithmetic wants {value of pixel 0, value of pixel 1} in a pair, which then costs two v_mov per pixel.
// sext[32 k]: the entry of sextant k sits at byte offset 256 k, which is what a convert that writes its result into
// byte lane 1 (v_cvt_u32_f32_sdwa dst_sel:BYTE_1) delivers without a shift.
struct HsvLds {
  float value[256];
  float gap[32];
  float rcpv[256];
  float gap2[32];
  uint2 sext[6 * 32 + 1];
};
struct HsvValueTab { float value[256]; float rcpv[256]; };
constexpr HsvValueTab hsv_make_value_tab() {
  HsvValueTab t{};
  for (int n = 0; n < 256; n++) {
    t.value[n] = (float)n / 255.0f;  // IEEE quotients, evaluated by the compiler (round to nearest even)
    t.rcpv[n] = n ? 1.0f / t.value[n] : 0.0f;
  }
  return t;
}
__device__ const HsvValueTab g_hsv_value_tab = hsv_make_value_tab();

// Fill the block's HsvLds from a block of NT threads (NT >= 8, compile time: 256 threads copy one entry each); the
// caller synchronises.
template <int RPOS, int GPOS, int BPOS, int NPOS, int NT = 256>
__device__ __forceinline__ void hsv_lds_fill(HsvLds *lds) {
#pragma unroll
  for (int i = threadIdx.x; i < 256; i += NT) {
    lds->value[i] = g_hsv_value_tab.value[i];
    lds->rcpv[i] = g_hsv_value_tab.rcpv[i];
  }
  if (threadIdx.x < 7) {
    const int k = (int)threadIdx.x;
    lds->sext[32 * k] = make_uint2(hsv_sel_entry_floor(k, RPOS, GPOS, BPOS, NPOS), (uint32_t)__float_as_int((float)(2 * (k >> 1) + 1)));
  }
}

// rotate so that byte0 = max channel M, bytes 1,2 = the other two in the cyclic order the hue formula subtracts them,
// byte3 = 0/2/4 (first of R,G,B equal to the max, hsvutils.rs:63-68)
template <int RPOS, int GPOS, int BPOS>
__device__ __forceinline__ uint32_t hsv_rotate_max_first(uint32_t p) {
  constexpr uint32_t SEL_R = (uint32_t)RPOS | ((uint32_t)GPOS << 8) | ((uint32_t)BPOS << 16) | (4u << 24);
  constexpr uint32_t SEL_G = (uint32_t)GPOS | ((uint32_t)BPOS << 8) | ((uint32_t)RPOS << 16) | (5u << 24);
  constexpr uint32_t SEL_B = (uint32_t)BPOS | ((uint32_t)RPOS << 8) | ((uint32_t)GPOS << 16) | (6u << 24);
  const uint32_t r = (p >> (8 * RPOS)) & 0xffu, g = (p >> (8 * GPOS)) & 0xffu, b = (p >> (8 * BPOS)) & 0xffu;
  const uint32_t sel = (r >= g && r >= b) ? SEL_R : (g >= b ? SEL_G : SEL_B);
  return __builtin_amdgcn_perm(0x00040200u, p, sel);
}

// from_rgb / from_bgr (hsvutils.rs:44-128) for two pixels: hue in [0,360), saturation, value.
// LDS >= 1: value and its reciprocal come from the block's HsvLds (one byte-lane shift + two ds_read_b32 instead of a
// convert, two constant-division ops and a quarter-rate v_rcp_f32).
template <int BYTE>
__device__ __forceinline__ uint32_t hsv_byte_times4(uint32_t v) {
  uint32_t o;
  const uint32_t two = 2;
  if constexpr (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(o) : "v"(two), "v"(v));
  else if constexpr (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(o) : "v"(two), "v"(v));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(o) : "v"(two), "v"(v));
  return o;
}
__device__ __forceinline__ float hsv_lds_at(const float *tab, uint32_t byte_off) { return *(const float *)((const char *)tab + byte_off); }
template <int RPOS, int GPOS, int BPOS, int LDS>
__device__ __forceinline__ void hsv_from_rgb_pair_fast(uint32_t p0, uint32_t p1, f2 &h, f2 &sat, f2 &value, const HsvLds *lds = nullptr) {
  const uint32_t rot[2] = {hsv_rotate_max_first<RPOS, GPOS, BPOS>(p0), hsv_rotate_max_first<RPOS, GPOS, BPOS>(p1)};
  const f2 add = {(float)(rot[0] >> 24), (float)(rot[1] >> 24)};
  const f2 hi = splat2(MI355_INV255_HI), lo = splat2(MI355_INV255_LO);
  f2 ys, ds, af, bf;
  if constexpr (LDS >= 1) {
    const uint32_t o0 = hsv_byte_times4<0>(rot[0]), o1 = hsv_byte_times4<0>(rot[1]);
    value = f2{hsv_lds_at(lds->value, o0), hsv_lds_at(lds->value, o1)};
    ys = f2{hsv_lds_at(lds->rcpv, o0), hsv_lds_at(lds->rcpv, o1)};
    ds = value;  // value == 0 only with chroma == 0: 0 * 0, fma(-0, 0, 0), fma(0, 0, 0) stay 0
  } else {
    const f2 M8 = {(float)(rot[0] & 0xffu), (float)(rot[1] & 0xffu)};
    value = fma2(M8, hi, M8 * lo);  // RN(n/255), see div255_u8
    ds = value + splat2(1e-30f);     // == value unless value == 0 (then chroma == 0 as well)
    ys = f2{__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)};
  }
  // (the other two channels' quotients through the same table measured no faster, r03: the kernel is no longer VALU-bound)
  const f2 a8 = {(float)((rot[0] >> 8) & 0xffu), (float)((rot[1] >> 8) & 0xffu)};
  const f2 b8 = {(float)((rot[0] >> 16) & 0xffu), (float)((rot[1] >> 16) & 0xffu)};
  af = fma2(a8, hi, a8 * lo);
  bf = fma2(b8, hi, b8 * lo);
  const f2 minv = {fminf(af.x, bf.x), fminf(af.y, bf.y)};
  const f2 chroma = value - minv;
  const f2 num = af - bf;
  // q = num/chroma, sat = chroma/value: reciprocal + one residual step (div_rcp_refine), the four fmas issued as two
  // packed ones. A zero chroma only occurs with a zero numerator; chroma + 1e-30 == chroma for every other chroma
  // (the smallest is 1/255 - 1 ulp).
  const f2 dq = chroma + splat2(1e-30f);
  const f2 yq = {__builtin_amdgcn_rcpf(dq.x), __builtin_amdgcn_rcpf(dq.y)};
  f2 q = num * yq;
  sat = chroma * ys;
  q = fma2(fma2(-q, dq, num), yq, q);
  sat = fma2(fma2(-sat, ds, chroma), ys, sat);
  h = add360_if_negative2(splat2(60.0f) * (add + q));
}

// SHIFT classes of the FAST filter: how (hue + hue_shift) % 360 and the `< 0 -> + 360` behind it are computed
// (hsvfilter/imp.rs:102-105), by the host-side classification of hue_shift (hsv_variant_for).
enum HsvShiftClass { HSV_SHIFT_ZERO = 0, HSV_SHIFT_POS = 1, HSV_SHIFT_NEG = 2, HSV_SHIFT_WIDE_POS = 3, HSV_SHIFT_WIDE_NEG = 4 };
// Kernel template argument VARIANT (-1 = GENERIC): bits 0-1 = ZERO / POS / NEG, bit 2 = identity saturation / value
// settings, bit 3 = wide shift (bit 0 then holds its sign).
constexpr int hsv_shift_of(int variant) { return (variant & 8) ? ((variant & 1) ? HSV_SHIFT_WIDE_NEG : HSV_SHIFT_WIDE_POS) : (variant & 3); }
constexpr bool hsv_sv_ident_of(int variant) { return (variant & 4) != 0; }

template <int RPOS, int GPOS, int BPOS, int NPOS, int SHIFT, bool SV_IDENT, int LDS>
__device__ __forceinline__ void hsvfilter_px2_fast_impl(uint32_t &p0, uint32_t &p1, const HsvK &k, const uint32_t *sel_tab, const HsvLds *lds) {
  f2 h, sat, value;
  hsv_from_rgb_pair_fast<RPOS, GPOS, BPOS, LDS>(p0, p1, h, sat, value, lds);
  // ---- filter (hsvfilter/imp.rs:102-115)
  f2 t = h;
  if constexpr (SHIFT == HSV_SHIFT_POS) {
    t = sub360_if_reached2(h + splat2(k.hue_shift));  // h + shift in [0, 720)
  } else if constexpr (SHIFT == HSV_SHIFT_NEG) {
    // h + shift in [-360, 360): fmod is the identity (-360 -> +0 instead of -0, indistinguishable), then `< 0 -> + 360`
    t = add360_if_negative2(h + splat2(k.hue_shift));
  } else if constexpr (SHIFT == HSV_SHIFT_WIDE_POS) {
    t = fmod360_abs2(h + splat2(k.hue_shift));  // 360 < shift <= 2^22: the sum is positive
  } else if constexpr (SHIFT == HSV_SHIFT_WIDE_NEG) {
    // -2^22 <= shift < -360: the sum is negative, fmod = -fmod(|sum|, 360), and `+ 360` rounds once like the reference's;
    // a zero remainder gives 360 where the reference keeps -0: the same pixel (x == 0 in sextant 0 and in sextant 6)
    t = splat2(360.0f) - fmod360_abs2(-(h + splat2(k.hue_shift)));
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
  uint32_t sel0, sel1;
  f2 x;
  if constexpr (LDS >= 1) {
    // byte offset of the sextant's entry: floor(hp) << 8, straight out of the convert
    uint32_t o0, o1;
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(o0) : "v"(hp.x));
    asm("v_cvt_u32_f32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(o1) : "v"(hp.y));
    const uint2 e0 = *(const uint2 *)((const char *)lds->sext + o0), e1 = *(const uint2 *)((const char *)lds->sext + o1);
    sel0 = e0.x;
    sel1 = e1.x;
    // hp % 2 - 1 == hp - odd (one rounding of the same number), then 1 - |w| with the abs as a source modifier. Written
    // as instructions: left to itself the compiler pairs the two pixels up for v_pk_add_f32, which costs two v_mov to
    // bring the odds into one register pair and two v_and for the abs.
    float w0, w1, u0, u1;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(w0) : "v"(hp.x), "v"(e0.y));
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(w1) : "v"(hp.y), "v"(e1.y));
    asm("v_sub_f32_e64 %0, 1.0, |%1|" : "=v"(u0) : "v"(w0));
    asm("v_sub_f32_e64 %0, 1.0, |%1|" : "=v"(u1) : "v"(w1));
    x = c * f2{u0, u1};
  } else {
    sel0 = sel_tab[(uint32_t)hp.x];
    sel1 = sel_tab[(uint32_t)hp.y];
    const f2 hh = hp * splat2(0.5f);
    const f2 fr = {__builtin_amdgcn_fractf(hh.x), __builtin_amdgcn_fractf(hh.y)};
    const f2 w = fma2(splat2(2.0f), fr, splat2(-1.0f));
    x = c * f2{1.0f - fabsf(w.x), 1.0f - fabsf(w.y)};
  }
  const f2 m = v - c;
  const f2 A = (c + m) * splat2(255.0f), B = (x + m) * splat2(255.0f), C = m * splat2(255.0f);
  uint32_t pk0, pk1;
  cvt_u8_into(pk0, A.x, 0); cvt_u8_into(pk0, B.x, 1); cvt_u8_into(pk0, C.x, 2);
  cvt_u8_into(pk1, A.y, 0); cvt_u8_into(pk1, B.y, 1); cvt_u8_into(pk1, C.y, 2);
  p0 = __builtin_amdgcn_perm(p0, pk0, sel0);
  p1 = __builtin_amdgcn_perm(p1, pk1, sel1);
}

// No block tables beyond the 7-entry selector table (the fused hsvfilter+colorlut kernels, whose LDS is spoken for).
template <int RPOS, int GPOS, int BPOS, int NPOS, int SHIFT, bool SV_IDENT>
__device__ __forceinline__ void hsvfilter_px2_fast(uint32_t &p0, uint32_t &p1, const HsvK &k, const uint32_t *sel_tab) {
  hsvfilter_px2_fast_impl<RPOS, GPOS, BPOS, NPOS, SHIFT, SV_IDENT, 0>(p0, p1, k, sel_tab, nullptr);
}
// With the block's HsvLds (the element's own kernels).
template <int RPOS, int GPOS, int BPOS, int NPOS, int SHIFT, bool SV_IDENT>
__device__ __forceinline__ void hsvfilter_px2_lds(uint32_t &p0, uint32_t &p1, const HsvK &k, const HsvLds *lds) {
  hsvfilter_px2_fast_impl<RPOS, GPOS, BPOS, NPOS, SHIFT, SV_IDENT, 1>(p0, p1, k, nullptr, lds);
}

// GENERIC settings (hsv_variant_for == -1), two pixels: from_rgb is independent of the settings, so the FAST pair routine
// serves it; the filter and to_rgb are the reference's operations one by one (fmodf with any operand, IEEE `/ 60`, every
// sextant arm incl. the NaN one, inherent clamp + saturating cast).
__device__ __forceinline__ float hsv_fmod360_any(float t) {
  // fmodf(t, 360): below 2^23 the exact one-fma form (sign of the dividend, -0 for a negative multiple), else the library's
  if (fabsf(t) < 8388608.0f) {
    const float a = fabsf(t);
    const float f0 = __builtin_floorf(a * 0x1.6c16c2p-9f);
    float r = __builtin_fmaf(-f0, 360.0f, a);
    r = (r < 0.0f) ? r + 360.0f : r;
    r = (r >= 360.0f) ? r - 360.0f : r;
    return __builtin_copysignf(r, t);
  }
  return fmodf(t, 360.0f);
}
template <int RPOS, int GPOS, int BPOS, int NPOS>
__device__ __forceinline__ void hsvfilter_px2_generic(uint32_t &p0, uint32_t &p1, const HsvK &k, const HsvLds *lds) {
  f2 h2, sat, value;
  hsv_from_rgb_pair_fast<RPOS, GPOS, BPOS, 1>(p0, p1, h2, sat, value, lds);
  uint32_t *px[2] = {&p0, &p1};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    float h = hsv_fmod360_any((j ? h2.y : h2.x) + k.hue_shift);
    if (h < 0.0f) h += 360.0f;
    const float s = fminf(fmaxf(k.sat_mul * (j ? sat.y : sat.x) + k.sat_off, 0.0f), 1.0f);
    const float v = fminf(fmaxf(k.val_mul * (j ? value.y : value.x) + k.val_off, 0.0f), 1.0f);
    uint32_t r, g, b;
    hsv_to_rgb_generic(h, s, v, r, g, b);
    const uint32_t keep = *px[j] & (0xffu << (8 * NPOS));
    *px[j] = keep | (r << (8 * RPOS)) | (g << (8 * GPOS)) | (b << (8 * BPOS));
  }
}

// Host side: arithmetic variant for one settings snapshot (see hsv_shift_of). -1 = GENERIC: non-finite hue_shift,
// 0 < |hue_shift| < 1e-30 (div60_hue's domain) or |hue_shift| > 2^22 (fmod360_abs2's). Identity saturation / value
// settings skip the affine step (x*1+0 == x exactly for the x in [0,1] the conversion produces). `allow_wide` = the
// caller has the 360 < |hue_shift| <= 2^22 instantiations (the element's own kernels; the fused ones take GENERIC).
inline int hsv_variant_for(const mi355_hsv_settings &s, bool force_generic, bool allow_wide = false) {
  const float a = fabsf(s.hue_shift);
  // NaN fails every comparison -> generic path.
  const bool narrow = (s.hue_shift == 0.0f) || (a >= 1e-30f && a <= 360.0f);
  const bool wide = allow_wide && a > 360.0f && a <= 4194304.0f;
  if (!(narrow || wide) || force_generic) return -1;
  const bool sv_ident = s.saturation_mul == 1.0f && s.saturation_off == 0.0f && s.value_mul == 1.0f && s.value_off == 0.0f;
  if (wide) return 8 | (s.hue_shift < 0.0f ? 1 : 0) | (sv_ident ? 4 : 0);
  const int shift = s.hue_shift == 0.0f ? HSV_SHIFT_ZERO : (s.hue_shift > 0.0f ? HSV_SHIFT_POS : HSV_SHIFT_NEG);
  return shift | (sv_ident ? 4 : 0);
}

}  // namespace mi355
