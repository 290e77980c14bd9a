/* oracle/rust_sem.h — TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product path).
 *
 * C spellings of the Rust scalar semantics the reference's inner loops rely on
 * (SURVEY.md §8c "Rust→C semantic map"). Everything here must be compiled with
 * -ffp-contract=off -fno-fast-math: Rust never contracts a*b+c and has no fast-math.
 */
#ifndef ORACLE_RUST_SEM_H
#define ORACLE_RUST_SEM_H
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* inherent f32::clamp: `if x < lo {lo} else if x > hi {hi} else {x}` — NaN passes through. */
static inline float rs_f32_clamp(float x, float lo, float hi) {
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
/* f32::max / f32::min: IEEE maxNum/minNum (a NaN operand is ignored). */
static inline float rs_f32_max(float a, float b) { return fmaxf(a, b); }
static inline float rs_f32_min(float a, float b) { return fminf(a, b); }
/* `x as u8`: truncate toward zero, saturate, NaN -> 0. */
static inline uint8_t rs_f32_as_u8(float x) {
  if (!(x == x)) return 0;
  if (x <= 0.0f) return 0;
  if (x >= 255.0f) return 255;
  return (uint8_t)x;
}
static inline uint16_t rs_f32_as_u16(float x) {
  if (!(x == x)) return 0;
  if (x <= 0.0f) return 0;
  if (x >= 65535.0f) return 65535;
  return (uint16_t)x;
}
/* `x as usize` (64-bit target). */
static inline size_t rs_f32_as_usize(float x) {
  if (!(x == x)) return 0;
  if (x <= 0.0f) return 0;
  if (x >= 18446744073709551616.0f) return (size_t)UINT64_MAX;
  return (size_t)x;
}
static inline uint16_t rs_bswap16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }
#endif
