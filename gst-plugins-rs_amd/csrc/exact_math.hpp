// exact_math.hpp — device-side building blocks that are BIT-EXACT re-expressions of the IEEE
// operations the reference performs (Rust: no contraction, correctly rounded `/`, C fmodf for `%`).
//
// Everything in the translation units that include this header is compiled with
// -ffp-contract=off; fused operations appear only where written explicitly (__builtin_fmaf) and
// only inside sequences proven to round identically to the unfused reference expression.
// The proofs are exhaustive over the finite operand domains involved (see tests/test_exact_math.py
// which replays them on the CPU with the same operation sequence, and the all-2^24-colours GPU
// parity tests which cover every operand pair the pixel filters can produce).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mi355 {

// 1/255, 1/65535 and 1/60 as an unevaluated sum hi + lo of two floats (hi = RN(1/d)).
#define MI355_INV255_HI 0x1.010102p-8f
#define MI355_INV255_LO -0x1.fdfdfep-33f
#define MI355_INV65535_HI 0x1.0001p-16f
#define MI355_INV65535_LO 0x1.0001p-48f
#define MI355_INV60_HI 0x1.111112p-6f
#define MI355_INV60_LO -0x1.dddddep-31f

// RN(n / 255.0f) for integer-valued n in [0,255]: fma(n, hi, RN(n*lo)).
// Exhaustively equal to the IEEE quotient for all 256 inputs (hsvutils.rs:45-47, colorlut/imp.rs:472).
__device__ __forceinline__ float div255_u8(float n) {
  return __builtin_fmaf(n, MI355_INV255_HI, n * MI355_INV255_LO);
}
// RN(n / 65535.0f) for integer-valued n in [0,65535] (colorlut/imp.rs:477); exhaustive over 65536 inputs.
__device__ __forceinline__ float div65535_u16(float n) {
  return __builtin_fmaf(n, MI355_INV65535_HI, n * MI355_INV65535_LO);
}
// RN(h / 60.0f) for h == 0 or 2^-116 <= h <= 360 (hsvutils.rs:134). Exhaustive over all 1.1e9
// floats in [0,360]: the only failures are below 1.1e-35 (underflow of h*lo), which the FAST
// hsvfilter path excludes on the host side.
__device__ __forceinline__ float div60_hue(float h) {
  return __builtin_fmaf(h, MI355_INV60_HI, h * MI355_INV60_LO);
}

// a / b for |a| <= 1, 2^-9 < b <= 1 via hardware reciprocal (1 ulp) + one residual correction.
// q0 = a*y carries <= ~1.5 ulp error; r = a - q0*b is exact in the fma; q0 + r*y then rounds to
// RN(a/b) unless a/b lies within ~2^-46 relative of a rounding boundary. With a correctly rounded
// reciprocal the sequence is exact for every operand pair the HSV conversion produces (CPU replay,
// tests/test_exact_math.py); with a reciprocal a full ulp off it mis-rounds ~60 of the 2^24 colours.
// gfx950's v_rcp_f32 is exact enough on this operand set: the all-colours GPU parity tests compare
// every (chroma, value, channel-difference) combination the filter can ever see against the oracle.
// The GENERIC kernels use the IEEE `/` sequence instead.
__device__ __forceinline__ float div_rcp_refine(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  float q = a * y;
  float r = __builtin_fmaf(-q, b, a);
  return __builtin_fmaf(r, y, q);
}

// Rust `x as u8` (truncate, saturate, NaN -> 0).
__device__ __forceinline__ uint32_t rs_as_u8(float x) {
  // v_cvt_u32_f32 truncates toward zero, clamps negatives to 0 and NaN to 0; clamp the top here.
  uint32_t u = (x >= 255.0f) ? 255u : (x > 0.0f ? (uint32_t)x : 0u);
  return u;
}
__device__ __forceinline__ uint32_t rs_as_u16(float x) {
  uint32_t u = (x >= 65535.0f) ? 65535u : (x > 0.0f ? (uint32_t)x : 0u);
  return u;
}
// inherent f32::clamp (NaN passes through)
__device__ __forceinline__ float rs_clamp(float x, float lo, float hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}

}  // namespace mi355
