/* tests/cpu_replay/fast_path_replay.c — CPU replay of the strength-reduced (FAST) device arithmetic.
 *
 * Same operation sequences as gst-plugins-rs_amd/csrc/{exact_math.hpp,hsv_kernels.hip,colorlut_kernels.hip},
 * written with fmaf() where the device code writes __builtin_fmaf / v_pk_fma_f32, so the exactness
 * arguments can be checked without a GPU. The hardware reciprocal (v_rcp_f32, <= 1 ulp) is modelled by
 * `rcp_mode`: 0 = correctly rounded 1/x, +1 / -1 = one ulp above / below; the refinement must give the
 * IEEE quotient for all three, i.e. for ANY reciprocal within 1 ulp.
 * Build: gcc -O2 -mfma -ffp-contract=off -fno-fast-math -shared -fPIC.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define INV255_HI 0x1.010102p-8f
#define INV255_LO -0x1.fdfdfep-33f
#define INV65535_HI 0x1.0001p-16f
#define INV65535_LO 0x1.0001p-48f
#define INV60_HI 0x1.111112p-6f
#define INV60_LO -0x1.dddddep-31f

float replay_div255(float n) { return fmaf(n, INV255_HI, n * INV255_LO); }
float replay_div65535(float n) { return fmaf(n, INV65535_HI, n * INV65535_LO); }
float replay_div60(float h) { return fmaf(h, INV60_HI, h * INV60_LO); }

static float rcp_model(float b, int mode) {
  float y = 1.0f / b;
  if (mode > 0) y = nextafterf(y, INFINITY);
  if (mode < 0) y = nextafterf(y, 0.0f);
  return y;
}
static float div_rcp_refine(float a, float b, int mode) {
  float y = rcp_model(b, mode);
  float q = a * y;
  float r = fmaf(-q, b, a);
  return fmaf(r, y, q);
}
static float add360_if_negative(float x) {
  int32_t bits;
  memcpy(&bits, &x, 4);
  int32_t m = bits >> 31;
  int32_t k = m & 0x43b40000;
  float add;
  memcpy(&add, &k, 4);
  return x + add;
}
static float fractf_(float x) { return x - floorf(x); } /* v_fract_f32, exact for x >= 0 */

/* selector codes by floor(h/60): which of (A,B,C) goes to (R,G,B) */
static const int kFloorCodes[7][3] = {{0, 1, 2}, {1, 0, 2}, {2, 0, 1}, {2, 1, 0}, {1, 2, 0}, {0, 2, 1}, {0, 2, 1}};

/* FAST hsvfilter on packed RGBx pixels (r | g<<8 | b<<16 | x<<24). shift_class: 0 zero, 1 pos, 2 neg. */
void replay_hsvfilter_fast(uint32_t *px, size_t n, const float st[5], int rcp_mode) {
  const float hs = st[0], sm = st[1], so = st[2], vm = st[3], vo = st[4];
  const int shift_class = hs == 0.0f ? 0 : (hs > 0.0f ? 1 : 2);
  const int sv_ident = sm == 1.0f && so == 0.0f && vm == 1.0f && vo == 0.0f;
  for (size_t i = 0; i < n; i++) {
    const uint32_t p = px[i];
    const uint32_t r = p & 255, g = (p >> 8) & 255, b = (p >> 16) & 255;
    uint32_t M8, a8, b8, addi;
    if (r >= g && r >= b) { M8 = r; a8 = g; b8 = b; addi = 0; }
    else if (g >= b) { M8 = g; a8 = b; b8 = r; addi = 2; }
    else { M8 = b; a8 = r; b8 = g; addi = 4; }
    const float value = replay_div255((float)M8), af = replay_div255((float)a8), bf = replay_div255((float)b8);
    const float minv = fminf(af, bf);
    const float chroma = value - minv;
    const float num = af - bf;
    const float q = div_rcp_refine(num, fmaxf(chroma, 1e-30f), rcp_mode);
    const float sat = div_rcp_refine(chroma, fmaxf(value, 1e-30f), rcp_mode);
    float h = 60.0f * ((float)addi + q);
    h = add360_if_negative(h);
    float t = h;
    if (shift_class == 1) {
      t = h + hs;
      const float u = t - 360.0f;
      t = (u < 0.0f) ? t : u; /* bit-select on sign(u) */
      if (u == 0.0f && signbit(u)) t = h + hs; /* -0 cannot occur: t-360 == 0 gives +0 */
    } else if (shift_class == 2) {
      t = add360_if_negative(h + hs);
    }
    float s = sat, v = value;
    if (!sv_ident) {
      s = fminf(fmaxf(sm * sat + so, 0.0f), 1.0f);
      v = fminf(fmaxf(vm * value + vo, 0.0f), 1.0f);
    }
    const float c = v * s;
    const float hp = replay_div60(t);
    const float w = fmaf(2.0f, fractf_(hp * 0.5f), -1.0f);
    const float x = c * (1.0f - fabsf(w));
    const float m = v - c;
    const uint32_t cand[3] = {(uint32_t)((c + m) * 255.0f), (uint32_t)((x + m) * 255.0f), (uint32_t)(m * 255.0f)};
    const int k = (int)hp; /* floor */
    const uint32_t ro = cand[kFloorCodes[k][0]], go = cand[kFloorCodes[k][1]], bo = cand[kFloorCodes[k][2]];
    px[i] = (p & 0xff000000u) | ro | (go << 8) | (bo << 16);
  }
}

/* ---------------------------------------------------------------- round-3 form of the FAST pixel algorithm
 * (csrc/hsv_device.hpp: hsvfilter_px2_fast). Differences from the round-2 form replayed above, each of which must leave
 * every output byte unchanged:
 *   - value = n/255 and its reciprocal come from a 256-entry table (LDS on the device) holding the IEEE quotients, so
 *     chroma/value is refined from a CORRECTLY ROUNDED reciprocal; only 1/chroma still comes from v_rcp_f32 (rcp_mode);
 *   - zero denominators: chroma + 1e-30f (== chroma for every non-zero chroma, which is >= 1/255 - ulp) instead of fmaxf;
 *   - "x < 0 ? x + 360 : x" and "t >= 360 ? t - 360 : t" are ONE unsigned-integer minimum of the two candidates' bit
 *     patterns (a negative float is a huge unsigned number);
 *   - hp % 2 - 1 == hp - odd with odd = 2*floor(hp/2) + 1 read from the sextant table next to the byte selector
 *     (both are one rounding of the same real number);
 *   - 360 < |hue_shift| <= 2^22 ("wide" classes 3 / 4) no longer needs the literal fmodf kernel: fmod(|t|, 360) is
 *     computed exactly with one fma around floor(|t| * (1/360)) and two exact wraps. */
static uint32_t fbits(float x) { uint32_t b; memcpy(&b, &x, 4); return b; }
static float minu(float a, float b) { return fbits(a) < fbits(b) ? a : b; }
static float exact_fmod360_abs(float a) { /* fmodf(a, 360) for 0 <= a < 2^23 */
  const float f0 = floorf(a * 0x1.6c16c2p-9f); /* RN(1/360); f0 is within one of floor(a/360) */
  float r = fmaf(-f0, 360.0f, a);              /* exact: a multiple of ulp(a) below 720 in magnitude */
  r = minu(r, r + 360.0f);                     /* r in [-360,0) -> +360 (exact) */
  r = minu(r, r - 360.0f);                     /* r in [360,720) -> -360 (exact) */
  return r;
}

void replay_hsvfilter_fast2(uint32_t *px, size_t n, const float st[5], int rcp_mode) {
  const float hs = st[0], sm = st[1], so = st[2], vm = st[3], vo = st[4];
  const float ahs = fabsf(hs);
  const int shift_class = hs == 0.0f ? 0 : (ahs <= 360.0f ? (hs > 0.0f ? 1 : 2) : (hs > 0.0f ? 3 : 4));
  const int sv_ident = sm == 1.0f && so == 0.0f && vm == 1.0f && vo == 0.0f;
  float tab_v[256], tab_y[256];
  for (int i = 0; i < 256; i++) { tab_v[i] = (float)i / 255.0f; tab_y[i] = i ? 1.0f / tab_v[i] : 0.0f; }
  for (size_t i = 0; i < n; i++) {
    const uint32_t p = px[i];
    const uint32_t r = p & 255, g = (p >> 8) & 255, b = (p >> 16) & 255;
    uint32_t M8, a8, b8, addi;
    if (r >= g && r >= b) { M8 = r; a8 = g; b8 = b; addi = 0; }
    else if (g >= b) { M8 = g; a8 = b; b8 = r; addi = 2; }
    else { M8 = b; a8 = r; b8 = g; addi = 4; }
    const float value = tab_v[M8], yv = tab_y[M8];
    const float af = replay_div255((float)a8), bf = replay_div255((float)b8);
    const float minv = fminf(af, bf);
    const float chroma = value - minv;
    const float num = af - bf;
    const float dq = chroma + 1e-30f;
    const float yq = rcp_model(dq, rcp_mode);
    float q = num * yq, sat = chroma * yv;
    q = fmaf(fmaf(-q, dq, num), yq, q);
    sat = fmaf(fmaf(-sat, value, chroma), yv, sat);
    float h = 60.0f * ((float)addi + q);
    h = minu(h, h + 360.0f);
    float t = h;
    if (shift_class == 1) { t = h + hs; t = minu(t, t - 360.0f); }
    else if (shift_class == 2) { t = h + hs; t = minu(t, t + 360.0f); }
    else if (shift_class == 3) { t = exact_fmod360_abs(h + hs); }
    else if (shift_class == 4) { t = 360.0f - exact_fmod360_abs(-(h + hs)); }
    float s = sat, v = value;
    if (!sv_ident) {
      s = fminf(fmaxf(sm * sat + so, 0.0f), 1.0f);
      v = fminf(fmaxf(vm * value + vo, 0.0f), 1.0f);
    }
    const float c = v * s;
    const float hp = replay_div60(t);
    const int k = (int)hp; /* floor, 0..6 */
    const float odd = (float)(2 * (k >> 1) + 1);
    const float w = hp - odd;
    const float x = c * (1.0f - fabsf(w));
    const float m = v - c;
    const uint32_t cand[3] = {(uint32_t)((c + m) * 255.0f), (uint32_t)((x + m) * 255.0f), (uint32_t)(m * 255.0f)};
    const uint32_t ro = cand[kFloorCodes[k][0]], go = cand[kFloorCodes[k][1]], bo = cand[kFloorCodes[k][2]];
    px[i] = (p & 0xff000000u) | ro | (go << 8) | (bo << 16);
  }
}

/* colorlut FAST output conversion: v_cvt_rpi_i32_f32(clamp01(o)*255) = floor(y + 0.5) computed exactly */
uint32_t replay_float_to_u8_fast(float o) {
  float c = fminf(fmaxf(o, 0.0f), 1.0f);
  float y = c * 255.0f;
  return (uint32_t)floor((double)y + 0.5); /* exact in double */
}
uint32_t replay_float_to_u8_ref(float o) {
  float c = o < 0.0f ? 0.0f : (o > 1.0f ? 1.0f : o);
  return (uint32_t)roundf(c * 255.0f);
}
