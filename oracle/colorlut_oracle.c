/* oracle/colorlut_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar) of the reference's colorlut element: the Adobe .cube
 * parser and the 1D / 3D LUT pixel loops on RGBA8 and RGBA64 (LE/BE). Checker for the HIP
 * path and "port" CPU baseline; never linked into the product path.
 *
 * Follows:
 *   video/colorlut/src/parser.rs:12-16     size limits                     -> LUT_*_SIZE
 *   video/colorlut/src/parser.rs:19-53     Lut3D / at()                    -> oracle_cube.table (x + y*S + z*S*S, [r,g,b,1])
 *   video/colorlut/src/parser.rs:110-281   CubeLut::parse                  -> oracle_cube_parse
 *   video/colorlut/src/parser.rs:284-375   ensure_header/validate/parse_*  -> static helpers below
 *   video/colorlut/src/colorlut/imp.rs:237-265  transform_rgba_1d          -> oracle_colorlut_rgba8
 *   video/colorlut/src/colorlut/imp.rs:267-294  transform_rgba_3d          -> oracle_colorlut_rgba8
 *   video/colorlut/src/colorlut/imp.rs:308-397  transform_rgba64_{1d,3d}<LE> -> oracle_colorlut_rgba64
 *   video/colorlut/src/colorlut/imp.rs:399-469  apply_1d/apply_3d[_u16]
 *   video/colorlut/src/colorlut/imp.rs:471-479  norm_comp[_u16]
 *   video/colorlut/src/colorlut/imp.rs:482-526  sample_1d / sample_3d
 *   video/colorlut/src/colorlut/imp.rs:528-543  lerp4 / float_to_u8 / float_to_u16
 *
 * Parity pinning: the parser known-answer tests of video/colorlut/src/parser.rs:381-473 are
 * replayed by tests/test_oracle_colorlut.py. Per-pixel LUT outputs are not pinned by any
 * reference test: pinned by source semantics + the independent numpy-f32
 * restatement (oracle/np_restate.py) + the identity-LUT pass-through property.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include "rust_sem.h"
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LUT_1D_MIN_SIZE 2
#define LUT_1D_MAX_SIZE 65536
#define LUT_3D_MIN_SIZE 2
#define LUT_3D_MAX_SIZE 256

typedef struct oracle_cube {
  int is3d;
  size_t size;
  float domain_scale[3];
  float domain_offset[3];
  /* 1D: three planes r[size], g[size], b[size] back to back (3*size floats).
   * 3D: size^3 cells of [r,g,b,1.0] (4*size^3 floats), index x + y*size + z*size*size. */
  float *table;
} oracle_cube;

/* ---------- text helpers mirroring str::lines / trim / split_whitespace ---------- */

/* Decode one UTF-8 scalar; returns its length or 0 on invalid input. */
static size_t utf8_decode(const unsigned char *s, size_t n, uint32_t *cp) {
  if (n == 0) return 0;
  unsigned char c = s[0];
  if (c < 0x80) { *cp = c; return 1; }
  if (c >= 0xC2 && c <= 0xDF && n >= 2 && (s[1] & 0xC0) == 0x80) {
    *cp = ((uint32_t)(c & 0x1F) << 6) | (s[1] & 0x3F); return 2;
  }
  if (c >= 0xE0 && c <= 0xEF && n >= 3 && (s[1] & 0xC0) == 0x80 && (s[2] & 0xC0) == 0x80) {
    uint32_t v = ((uint32_t)(c & 0x0F) << 12) | ((uint32_t)(s[1] & 0x3F) << 6) | (s[2] & 0x3F);
    if (v < 0x800 || (v >= 0xD800 && v <= 0xDFFF)) return 0;
    *cp = v; return 3;
  }
  if (c >= 0xF0 && c <= 0xF4 && n >= 4 && (s[1] & 0xC0) == 0x80 && (s[2] & 0xC0) == 0x80 &&
      (s[3] & 0xC0) == 0x80) {
    uint32_t v = ((uint32_t)(c & 0x07) << 18) | ((uint32_t)(s[1] & 0x3F) << 12) |
                 ((uint32_t)(s[2] & 0x3F) << 6) | (s[3] & 0x3F);
    if (v < 0x10000 || v > 0x10FFFF) return 0;
    *cp = v; return 4;
  }
  return 0;
}

/* char::is_whitespace (Unicode White_Space). */
static int is_unicode_ws(uint32_t c) {
  return (c >= 0x09 && c <= 0x0D) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 ||
         (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F ||
         c == 0x205F || c == 0x3000;
}

/* Next whitespace-delimited token in [*pos, end). Returns 0 when exhausted. */
static int next_token(const unsigned char *line, size_t end, size_t *pos, size_t *tok, size_t *tok_len) {
  size_t p = *pos;
  uint32_t cp = 0;
  while (p < end) {
    size_t l = utf8_decode(line + p, end - p, &cp);
    if (!is_unicode_ws(cp)) break;
    p += l;
  }
  if (p >= end) { *pos = p; return 0; }
  size_t start = p;
  while (p < end) {
    size_t l = utf8_decode(line + p, end - p, &cp);
    if (is_unicode_ws(cp)) break;
    p += l;
  }
  *tok = start; *tok_len = p - start; *pos = p;
  return 1;
}

static int ascii_ieq(const char *s, size_t n, const char *lit) {
  size_t m = strlen(lit);
  if (n != m) return 0;
  for (size_t i = 0; i < n; i++) {
    char c = s[i];
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    if (c != lit[i]) return 0;
  }
  return 1;
}

/* `str::parse::<f32>()`: [+-]? ( inf | infinity | nan | digits[.digits*] | .digits ) ([eE][+-]?digits)?
 * — no hex, no surrounding whitespace; value is the correctly rounded decimal (glibc strtof is). */
static int rust_parse_f32(const char *s, size_t n, float *out) {
  if (n == 0 || n > 4096) return 0;
  size_t i = 0;
  if (s[i] == '+' || s[i] == '-') i++;
  if (i >= n) return 0;
  const char *rest = s + i;
  size_t rn = n - i;
  int special = ascii_ieq(rest, rn, "inf") || ascii_ieq(rest, rn, "infinity") || ascii_ieq(rest, rn, "nan");
  if (!special) {
    size_t j = i, digits = 0;
    while (j < n && s[j] >= '0' && s[j] <= '9') { j++; digits++; }
    if (j < n && s[j] == '.') {
      j++;
      while (j < n && s[j] >= '0' && s[j] <= '9') { j++; digits++; }
    }
    if (digits == 0) return 0;
    if (j < n && (s[j] == 'e' || s[j] == 'E')) {
      j++;
      if (j < n && (s[j] == '+' || s[j] == '-')) j++;
      size_t ed = 0;
      while (j < n && s[j] >= '0' && s[j] <= '9') { j++; ed++; }
      if (ed == 0) return 0;
    }
    if (j != n) return 0;
  }
  char buf[4100];
  memcpy(buf, s, n);
  buf[n] = 0;
  char *endp = NULL;
  float v = strtof(buf, &endp);
  if (endp != buf + n) return 0;
  *out = v;
  return 1;
}

/* `str::parse::<usize>()`: optional '+', then ASCII digits; overflow is an error. */
static int rust_parse_usize(const char *s, size_t n, size_t *out) {
  size_t i = 0;
  if (n == 0) return 0;
  if (s[0] == '+') i = 1;
  if (i >= n) return 0;
  uint64_t v = 0;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return 0;
    uint64_t d = (uint64_t)(s[i] - '0');
    if (v > (UINT64_MAX - d) / 10) return 0;
    v = v * 10 + d;
  }
  *out = (size_t)v;
  return 1;
}

enum { ST_HEADER = 0, ST_1D = 1, ST_3D = 2 };

static int set_err(char *err, size_t errlen, const char *fmt, size_t line_no) {
  if (err && errlen) snprintf(err, errlen, fmt, line_no);
  return -1;
}

void oracle_cube_free(oracle_cube *c) {
  if (!c) return;
  free(c->table);
  free(c);
}

/* parser.rs:110-281. Returns NULL on error (message in err). */
oracle_cube *oracle_cube_parse(const char *text, size_t len, char *err, size_t errlen) {
  const unsigned char *t = (const unsigned char *)text;
  /* fs::read_to_string: invalid UTF-8 is an I/O error (parser.rs:106). */
  for (size_t p = 0; p < len;) {
    uint32_t cp;
    size_t l = utf8_decode(t + p, len - p, &cp);
    if (l == 0) { set_err(err, errlen, "IO error: invalid UTF-8 at byte %zu", p); return NULL; }
    p += l;
  }

  float domain_min[3] = {0.0f, 0.0f, 0.0f};
  float domain_max[3] = {1.0f, 1.0f, 1.0f};
  int state = ST_HEADER, have_data = 0;
  size_t size = 0;
  float *values = NULL; /* rows of 3 */
  size_t nvalues = 0, cap = 0;
  size_t line_no = 0;
  int rc = 0;

  size_t pos = 0;
  while (pos < len && rc == 0) {
    /* str::lines(): split on '\n', strip one trailing '\r'. */
    size_t eol = pos;
    while (eol < len && t[eol] != '\n') eol++;
    size_t lend = eol;
    if (lend > pos && t[lend - 1] == '\r') lend--;
    const unsigned char *line = t + pos;
    size_t llen = lend - pos;
    pos = eol < len ? eol + 1 : len;
    line_no++;

    /* trim() + is_empty()/starts_with('#') */
    size_t tp = 0, tok = 0, tlen = 0;
    if (!next_token(line, llen, &tp, &tok, &tlen)) continue; /* blank */
    if (line[tok] == '#') continue;

    const char *first = (const char *)line + tok;
    size_t first_len = tlen;
#define IS_KW(k) (first_len == strlen(k) && memcmp(first, k, first_len) == 0)
    int is_title = IS_KW("TITLE"), is_dmin = IS_KW("DOMAIN_MIN"), is_dmax = IS_KW("DOMAIN_MAX");
    int is_1d = IS_KW("LUT_1D_SIZE"), is_3d = IS_KW("LUT_3D_SIZE");
#undef IS_KW
    if (is_title || is_dmin || is_dmax || is_1d || is_3d) {
      /* ensure_header (parser.rs:284-303) */
      if (state != ST_HEADER && have_data) { rc = set_err(err, errlen, "Invalid LUT: Header found after LUT data at line %zu", line_no); break; }
      if (is_title) continue;
      if (is_dmin || is_dmax) {
        float v[3];
        int ok = 1;
        for (int k = 0; k < 3 && ok; k++) {
          if (!next_token(line, llen, &tp, &tok, &tlen)) { ok = 0; break; }
          if (!rust_parse_f32((const char *)line + tok, tlen, &v[k])) { rc = set_err(err, errlen, "Invalid LUT: Invalid float at line %zu", line_no); ok = 0; }
        }
        if (rc) break;
        if (!ok || next_token(line, llen, &tp, &tok, &tlen)) { rc = set_err(err, errlen, "Invalid LUT: Invalid line %zu", line_no); break; }
        memcpy(is_dmin ? domain_min : domain_max, v, sizeof v);
        continue;
      }
      /* LUT_1D_SIZE / LUT_3D_SIZE */
      if (state != ST_HEADER) { rc = set_err(err, errlen, "Invalid LUT: Invalid LUT size keyword at line %zu", line_no); break; }
      size_t sz;
      if (!next_token(line, llen, &tp, &tok, &tlen)) { rc = set_err(err, errlen, "Invalid LUT: Invalid line %zu", line_no); break; }
      if (!rust_parse_usize((const char *)line + tok, tlen, &sz)) { rc = set_err(err, errlen, "Invalid LUT: Invalid integer at line %zu", line_no); break; }
      if (next_token(line, llen, &tp, &tok, &tlen)) { rc = set_err(err, errlen, "Invalid LUT: Invalid line %zu", line_no); break; }
      size_t lo = is_1d ? LUT_1D_MIN_SIZE : LUT_3D_MIN_SIZE, hi = is_1d ? LUT_1D_MAX_SIZE : LUT_3D_MAX_SIZE;
      if (sz < lo || sz > hi) { rc = set_err(err, errlen, "Invalid LUT: Invalid LUT size at line %zu", line_no); break; }
      size = sz;
      state = is_1d ? ST_1D : ST_3D;
      have_data = 0;
      continue;
    }

    /* data row */
    if (state == ST_HEADER) { rc = set_err(err, errlen, "Invalid LUT: LUT data found before LUT size at line %zu", line_no); break; }
    have_data = 1;
    float v[3];
    if (!rust_parse_f32(first, first_len, &v[0])) { rc = set_err(err, errlen, "Invalid LUT: Invalid float at line %zu", line_no); break; }
    for (int k = 1; k < 3 && rc == 0; k++) {
      if (!next_token(line, llen, &tp, &tok, &tlen)) { rc = set_err(err, errlen, "Invalid LUT: Invalid line %zu", line_no); break; }
      if (!rust_parse_f32((const char *)line + tok, tlen, &v[k])) { rc = set_err(err, errlen, "Invalid LUT: Invalid float at line %zu", line_no); break; }
    }
    if (rc) break;
    if (next_token(line, llen, &tp, &tok, &tlen)) { rc = set_err(err, errlen, "Invalid LUT: Invalid line %zu", line_no); break; }
    if (nvalues == cap) {
      cap = cap ? cap * 2 : 1024;
      values = (float *)realloc(values, cap * 3 * sizeof(float));
    }
    memcpy(values + nvalues * 3, v, sizeof v);
    nvalues++;
  }
  if (rc) { free(values); return NULL; }

  /* parser.rs:205-212 (NaN compares false, i.e. passes — as in the reference) */
  if (domain_min[0] >= domain_max[0] || domain_min[1] >= domain_max[1] || domain_min[2] >= domain_max[2]) {
    set_err(err, errlen, "Invalid LUT: Invalid domain min/max (%zu)", (size_t)0);
    free(values); return NULL;
  }
  if (state == ST_HEADER) { set_err(err, errlen, "Invalid LUT: Missing LUT size (%zu)", (size_t)0); free(values); return NULL; }

  oracle_cube *c = (oracle_cube *)calloc(1, sizeof *c);
  c->size = size;
  if (state == ST_1D) {
    if (nvalues != size) { set_err(err, errlen, "Invalid LUT: Invalid 1D LUT value count, got %zu", nvalues); free(values); free(c); return NULL; }
    c->is3d = 0;
    c->table = (float *)malloc(3 * size * sizeof(float));
    for (size_t i = 0; i < size; i++) {
      c->table[i] = values[i * 3 + 0];
      c->table[size + i] = values[i * 3 + 1];
      c->table[2 * size + i] = values[i * 3 + 2];
    }
  } else {
    size_t expected = size * size * size;
    if (nvalues != expected) { set_err(err, errlen, "Invalid LUT: Invalid 3D LUT value count, got %zu", nvalues); free(values); free(c); return NULL; }
    c->is3d = 1;
    c->table = (float *)malloc(4 * expected * sizeof(float));
    for (size_t i = 0; i < expected; i++) {
      c->table[i * 4 + 0] = values[i * 3 + 0];
      c->table[i * 4 + 1] = values[i * 3 + 1];
      c->table[i * 4 + 2] = values[i * 3 + 2];
      c->table[i * 4 + 3] = 1.0f;
    }
  }
  free(values);
  for (int k = 0; k < 3; k++) { /* parser.rs:264-274 */
    c->domain_scale[k] = 1.0f / (domain_max[k] - domain_min[k]);
    c->domain_offset[k] = -domain_min[k] * c->domain_scale[k];
  }
  return c;
}

/* Build a cube directly from arrays (tests / bench: avoids printing+parsing text). */
oracle_cube *oracle_cube_from_table(int is3d, size_t size, const float *table, const float scale[3],
                                    const float offset[3]) {
  oracle_cube *c = (oracle_cube *)calloc(1, sizeof *c);
  c->is3d = is3d;
  c->size = size;
  size_t n = is3d ? 4 * size * size * size : 3 * size;
  c->table = (float *)malloc(n * sizeof(float));
  memcpy(c->table, table, n * sizeof(float));
  memcpy(c->domain_scale, scale, 3 * sizeof(float));
  memcpy(c->domain_offset, offset, 3 * sizeof(float));
  return c;
}

int oracle_cube_is3d(const oracle_cube *c) { return c->is3d; }
size_t oracle_cube_size(const oracle_cube *c) { return c->size; }
const float *oracle_cube_table(const oracle_cube *c) { return c->table; }
void oracle_cube_domain(const oracle_cube *c, float scale[3], float offset[3]) {
  memcpy(scale, c->domain_scale, sizeof c->domain_scale);
  memcpy(offset, c->domain_offset, sizeof c->domain_offset);
}

/* ---------- pixel math ---------- */

/* imp.rs:471-474 */
static inline float norm_comp(const oracle_cube *lut, int component, uint8_t value) {
  float v = (float)value / 255.0f;
  return rs_f32_clamp(v * lut->domain_scale[component] + lut->domain_offset[component], 0.0f, 1.0f);
}
/* imp.rs:476-479 */
static inline float norm_comp_u16(const oracle_cube *lut, int component, uint16_t value) {
  float v = (float)value / 65535.0f;
  return rs_f32_clamp(v * lut->domain_scale[component] + lut->domain_offset[component], 0.0f, 1.0f);
}
/* imp.rs:537-543 */
static inline uint8_t float_to_u8(float v) { return rs_f32_as_u8(roundf(rs_f32_clamp(v, 0.0f, 1.0f) * 255.0f)); }
static inline uint16_t float_to_u16(float v) { return rs_f32_as_u16(roundf(rs_f32_clamp(v, 0.0f, 1.0f) * 65535.0f)); }

static inline size_t usize_min(size_t a, size_t b) { return a < b ? a : b; }

/* imp.rs:482-490 */
static inline float sample_1d(const float *lut, size_t len, float x) {
  size_t max_idx = len - 1;
  size_t x0 = usize_min(rs_f32_as_usize(floorf(x)), max_idx);
  size_t x1 = usize_min(x0 + 1, max_idx);
  float t = x - (float)x0;
  return lut[x0] + (lut[x1] - lut[x0]) * t;
}

/* imp.rs:528-535 */
static inline void lerp4(const float a[4], const float b[4], float t, float out[4]) {
  out[0] = a[0] + (b[0] - a[0]) * t;
  out[1] = a[1] + (b[1] - a[1]) * t;
  out[2] = a[2] + (b[2] - a[2]) * t;
  out[3] = a[3] + (b[3] - a[3]) * t;
}

/* imp.rs:493-526 */
static inline void sample_3d(const oracle_cube *lut, float x, float y, float z, float out[4]) {
  size_t size = lut->size;
  size_t max_idx = size - 1;
  size_t x0 = usize_min(rs_f32_as_usize(floorf(x)), max_idx);
  size_t y0 = usize_min(rs_f32_as_usize(floorf(y)), max_idx);
  size_t z0 = usize_min(rs_f32_as_usize(floorf(z)), max_idx);
  size_t x1 = usize_min(x0 + 1, max_idx);
  size_t y1 = usize_min(y0 + 1, max_idx);
  size_t z1 = usize_min(z0 + 1, max_idx);
  float tx = x - (float)x0;
  float ty = y - (float)y0;
  float tz = z - (float)z0;
#define AT(X, Y, Z) (lut->table + 4 * ((X) + (Y) * size + (Z) * size * size)) /* parser.rs:43-53 */
  const float *c000 = AT(x0, y0, z0), *c100 = AT(x1, y0, z0);
  const float *c010 = AT(x0, y1, z0), *c110 = AT(x1, y1, z0);
  const float *c001 = AT(x0, y0, z1), *c101 = AT(x1, y0, z1);
  const float *c011 = AT(x0, y1, z1), *c111 = AT(x1, y1, z1);
#undef AT
  float c00[4], c10[4], c01[4], c11[4], c0[4], c1[4];
  lerp4(c000, c100, tx, c00);
  lerp4(c010, c110, tx, c10);
  lerp4(c001, c101, tx, c01);
  lerp4(c011, c111, tx, c11);
  lerp4(c00, c10, ty, c0);
  lerp4(c01, c11, ty, c1);
  lerp4(c0, c1, tz, out);
}

/* imp.rs:399-414 */
static inline uint8_t apply_1d(const oracle_cube *lut, int component, uint8_t value) {
  const float *table = lut->table + (size_t)component * lut->size;
  float x = norm_comp(lut, component, value) * ((float)lut->size - 1.0f);
  return float_to_u8(sample_1d(table, lut->size, x));
}
/* imp.rs:416-431 */
static inline uint16_t apply_1d_u16(const oracle_cube *lut, int component, uint16_t value) {
  const float *table = lut->table + (size_t)component * lut->size;
  float x = norm_comp_u16(lut, component, value) * ((float)lut->size - 1.0f);
  return float_to_u16(sample_1d(table, lut->size, x));
}
/* imp.rs:433-451 */
static inline void apply_3d(const oracle_cube *lut, uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]) {
  float sm1 = (float)lut->size - 1.0f;
  float x = norm_comp(lut, 0, r) * sm1;
  float y = norm_comp(lut, 1, g) * sm1;
  float z = norm_comp(lut, 2, b) * sm1;
  float o[4];
  sample_3d(lut, x, y, z, o);
  out[0] = float_to_u8(o[0]);
  out[1] = float_to_u8(o[1]);
  out[2] = float_to_u8(o[2]);
}
/* imp.rs:453-469 */
static inline void apply_3d_u16(const oracle_cube *lut, uint16_t r, uint16_t g, uint16_t b, uint16_t out[3]) {
  float sm1 = (float)lut->size - 1.0f;
  float x = norm_comp_u16(lut, 0, r) * sm1;
  float y = norm_comp_u16(lut, 1, g) * sm1;
  float z = norm_comp_u16(lut, 2, b) * sm1;
  float o[4];
  sample_3d(lut, x, y, z, o);
  out[0] = float_to_u16(o[0]);
  out[1] = float_to_u16(o[1]);
  out[2] = float_to_u16(o[2]);
}

/* imp.rs:226-294 — RGBA8, 1D or 3D. src/dst are plane 0 of two different frames with
 * independent strides; rows = chunks(stride).take(height). */
void oracle_colorlut_rgba8(const oracle_cube *lut, const uint8_t *src, int src_stride, uint8_t *dst,
                           int dst_stride, int width, int height) {
  for (int row = 0; row < height; row++) {
    const uint8_t *s = src + (size_t)row * (size_t)src_stride;
    uint8_t *d = dst + (size_t)row * (size_t)dst_stride;
    for (int i = 0; i < width; i++, s += 4, d += 4) {
      if (lut->is3d) {
        uint8_t o[3];
        apply_3d(lut, s[0], s[1], s[2], o);
        d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
      } else {
        for (int c = 0; c < 3; c++) d[c] = apply_1d(lut, c, s[c]);
      }
      d[3] = s[3];
    }
  }
}

void oracle_colorlut_rgba8_mt(const oracle_cube *lut, const uint8_t *src, int src_stride, uint8_t *dst,
                              int dst_stride, int width, int height, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int t = 0; t < nthreads; t++) {
    int r0 = (int)((long)height * t / nthreads), r1 = (int)((long)height * (t + 1) / nthreads);
    if (r1 > r0)
      oracle_colorlut_rgba8(lut, src + (size_t)r0 * (size_t)src_stride, src_stride,
                            dst + (size_t)r0 * (size_t)dst_stride, dst_stride, width, r1 - r0);
  }
}

/* imp.rs:296-397 — RGBA64; `le` selects RGBA64_LE (1) or RGBA64_BE (0). The host is
 * little-endian (x86-64), so from_le/to_le are no-ops and from_be/to_be swap bytes. */
void oracle_colorlut_rgba64(const oracle_cube *lut, const uint8_t *src, int src_stride, uint8_t *dst,
                            int dst_stride, int width, int height, int le) {
  for (int row = 0; row < height; row++) {
    const uint16_t *s = (const uint16_t *)(src + (size_t)row * (size_t)(src_stride / 2) * 2);
    uint16_t *d = (uint16_t *)(dst + (size_t)row * (size_t)(dst_stride / 2) * 2);
    for (int i = 0; i < width; i++, s += 4, d += 4) {
      uint16_t in[3];
      for (int c = 0; c < 3; c++) in[c] = le ? s[c] : rs_bswap16(s[c]);
      uint16_t o[3];
      if (lut->is3d) {
        apply_3d_u16(lut, in[0], in[1], in[2], o);
      } else {
        for (int c = 0; c < 3; c++) o[c] = apply_1d_u16(lut, c, in[c]);
      }
      for (int c = 0; c < 3; c++) d[c] = le ? o[c] : rs_bswap16(o[c]);
      d[3] = s[3]; /* alpha word copied raw */
    }
  }
}

/* CPU-baseline helper: `n_streams` independent 4-byte-pixel frames, each pushed through
 * hsvfilter (in place, RGBx layout) then colorlut (RGBA8) by ONE thread — the reference's only scaling
 * axis is more independent pipelines, each with a single streaming thread. */
void oracle_hsvfilter_frame(uint8_t *data, size_t data_len, int width, int stride, int pixel_stride,
                            int first, int bgr, const float settings[5]);
void oracle_chain_streams(const oracle_cube *lut, uint8_t *frames, uint8_t *outs, int n_streams, int width,
                          int height, const float settings[5], int nthreads) {
  const size_t fb = (size_t)width * 4 * (size_t)height;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
  for (int s = 0; s < n_streams; s++) {
    oracle_hsvfilter_frame(frames + (size_t)s * fb, fb, width, width * 4, 4, 0, 0, settings);
    oracle_colorlut_rgba8(lut, frames + (size_t)s * fb, width * 4, outs + (size_t)s * fb, width * 4, width, height);
  }
}
