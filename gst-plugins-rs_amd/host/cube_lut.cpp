// cube_lut.cpp — host-side mirror of the colorlut element's Adobe .cube reader
// (video/colorlut/src/parser.rs): same grammar, same acceptance/rejection behaviour, same
// resulting CubeLut {domain_scale, domain_offset, kind}. This is the part of the element that
// stays on the host: it runs once in start() (video/colorlut/src/colorlut/imp.rs:168-194) and its
// output is what mi355_colorlut_load() takes.
//
// Written as a line/token state machine over std::string_view; number conversion follows Rust's
// `str::parse::<f32>` / `<usize>` acceptance rules (no hex, optional sign, inf/infinity/nan,
// whole token must be consumed) with correctly rounded decimal conversion.
#include "mi355fx_host.h"

#include <cerrno>
#include <locale.h>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <string_view>
#include <vector>

namespace {

using sv = std::string_view;

// ---- Unicode helpers (str::trim / split_whitespace use char::is_whitespace)
struct Scalar { char32_t cp; size_t len; };

bool decode_utf8(sv s, size_t at, Scalar &out) {
  const auto b = [&](size_t i) { return static_cast<unsigned char>(s[at + i]); };
  const size_t left = s.size() - at;
  if (left == 0) return false;
  const unsigned char c0 = b(0);
  auto cont = [&](size_t i) { return i < left && (b(i) & 0xC0) == 0x80; };
  if (c0 < 0x80) { out = {c0, 1}; return true; }
  if (c0 >= 0xC2 && c0 <= 0xDF && cont(1)) { out = {char32_t((c0 & 0x1F) << 6 | (b(1) & 0x3F)), 2}; return true; }
  if (c0 >= 0xE0 && c0 <= 0xEF && cont(1) && cont(2)) {
    const char32_t v = char32_t((c0 & 0x0F) << 12 | (b(1) & 0x3F) << 6 | (b(2) & 0x3F));
    if (v < 0x800 || (v >= 0xD800 && v <= 0xDFFF)) return false;
    out = {v, 3};
    return true;
  }
  if (c0 >= 0xF0 && c0 <= 0xF4 && cont(1) && cont(2) && cont(3)) {
    const char32_t v = char32_t((c0 & 0x07) << 18 | (b(1) & 0x3F) << 12 | (b(2) & 0x3F) << 6 | (b(3) & 0x3F));
    if (v < 0x10000 || v > 0x10FFFF) return false;
    out = {v, 4};
    return true;
  }
  return false;
}

bool is_white_space(char32_t c) {
  switch (c) {
    case 0x09: case 0x0A: case 0x0B: case 0x0C: case 0x0D: case 0x20: case 0x85: case 0xA0: case 0x1680:
    case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000:
      return true;
    default:
      return c >= 0x2000 && c <= 0x200A;
  }
}

bool valid_utf8(sv s) {
  for (size_t i = 0; i < s.size();) {
    Scalar sc{};
    if (!decode_utf8(s, i, sc)) return false;
    i += sc.len;
  }
  return true;
}

// split_whitespace()
std::vector<sv> tokens_of(sv line) {
  std::vector<sv> toks;
  size_t i = 0, start = sv::npos;
  while (i < line.size()) {
    Scalar sc{};
    decode_utf8(line, i, sc);  // input was validated up front
    if (is_white_space(sc.cp)) {
      if (start != sv::npos) { toks.push_back(line.substr(start, i - start)); start = sv::npos; }
    } else if (start == sv::npos) {
      start = i;
    }
    i += sc.len;
  }
  if (start != sv::npos) toks.push_back(line.substr(start));
  return toks;
}

bool ieq(sv a, const char *lower) {
  const size_t n = std::strlen(lower);
  if (a.size() != n) return false;
  for (size_t i = 0; i < n; i++) {
    char c = a[i];
    if (c >= 'A' && c <= 'Z') c = char(c + 32);
    if (c != lower[i]) return false;
  }
  return true;
}

bool is_digit(char c) { return c >= '0' && c <= '9'; }

// Rust f32::from_str acceptance.
bool parse_f32_token(sv t, float &out) {
  if (t.empty() || t.size() > 2048) return false;
  sv body = t;
  if (body.front() == '+' || body.front() == '-') body.remove_prefix(1);
  if (body.empty()) return false;
  const bool named = ieq(body, "inf") || ieq(body, "infinity") || ieq(body, "nan");
  if (!named) {
    size_t i = 0, mant_digits = 0;
    while (i < body.size() && is_digit(body[i])) { i++; mant_digits++; }
    if (i < body.size() && body[i] == '.') {
      i++;
      while (i < body.size() && is_digit(body[i])) { i++; mant_digits++; }
    }
    if (mant_digits == 0) return false;
    if (i < body.size() && (body[i] == 'e' || body[i] == 'E')) {
      i++;
      if (i < body.size() && (body[i] == '+' || body[i] == '-')) i++;
      size_t exp_digits = 0;
      while (i < body.size() && is_digit(body[i])) { i++; exp_digits++; }
      if (exp_digits == 0) return false;
    }
    if (i != body.size()) return false;
  }
  const std::string z(t);
  char *end = nullptr;
  errno = 0;
  // Rust's str::parse::<f32> knows no locale; a GStreamer process has called setlocale(LC_ALL, "") (gst_init), and under a
  // comma-decimal LC_NUMERIC plain strtof stops at the '.' of every .cube value. strtof_l with the "C" locale: glibc's
  // correctly rounded conversion (overflow -> inf, underflow -> 0 / denormal, like Rust) whatever the process locale is.
  static const locale_t c_locale = newlocale(LC_ALL_MASK, "C", (locale_t)0);
  const float v = c_locale ? strtof_l(z.c_str(), &end, c_locale) : std::strtof(z.c_str(), &end);
  if (end != z.c_str() + z.size()) return false;
  out = v;
  return true;
}

// Rust usize::from_str acceptance.
bool parse_usize_token(sv t, size_t &out) {
  if (t.empty()) return false;
  if (t.front() == '+') t.remove_prefix(1);
  if (t.empty()) return false;
  unsigned long long acc = 0;
  for (char c : t) {
    if (!is_digit(c)) return false;
    const unsigned d = unsigned(c - '0');
    if (acc > (~0ull - d) / 10ull) return false;
    acc = acc * 10ull + d;
  }
  out = size_t(acc);
  return true;
}

constexpr size_t kLut1dMin = 2, kLut1dMax = 65536;  // parser.rs:12-13
constexpr size_t kLut3dMin = 2, kLut3dMax = 256;    // parser.rs:15-16

enum class Kind { Header, Lut1D, Lut3D };

struct ParseFail { std::string msg; };

[[noreturn]] void fail(const std::string &m) { throw ParseFail{"Invalid LUT: " + m}; }

std::string at_line(size_t n, sv line) { return " at line " + std::to_string(n) + ": " + std::string(line); }

}  // namespace

struct mi355h_cube {
  bool is3d = false;
  size_t size = 0;
  float domain_scale[3] = {1, 1, 1};
  float domain_offset[3] = {0, 0, 0};
  std::vector<float> table;  // 3D: size^3 x [r,g,b,1]; 1D: r | g | b planes
};

namespace {

mi355h_cube *parse_text(sv text) {
  if (!valid_utf8(text)) throw ParseFail{"IO error: stream did not contain valid UTF-8"};

  float dmin[3] = {0.f, 0.f, 0.f}, dmax[3] = {1.f, 1.f, 1.f};
  Kind kind = Kind::Header;
  bool have_data = false;
  size_t size = 0;
  std::vector<float> rows;  // 3 per data row

  size_t line_no = 0, cursor = 0;
  while (cursor < text.size()) {
    // str::lines(): '\n' terminated, one trailing '\r' stripped
    size_t nl = text.find('\n', cursor);
    sv raw = text.substr(cursor, (nl == sv::npos ? text.size() : nl) - cursor);
    cursor = (nl == sv::npos) ? text.size() : nl + 1;
    if (!raw.empty() && raw.back() == '\r') raw.remove_suffix(1);
    ++line_no;

    const std::vector<sv> tk = tokens_of(raw);
    if (tk.empty()) continue;             // blank after trim
    if (tk[0].front() == '#') continue;   // comment
    // `line` as the reference prints it = trimmed raw line
    const sv line = raw.substr(size_t(tk.front().data() - raw.data()),
                               size_t(tk.back().data() + tk.back().size() - tk.front().data()));

    const sv head = tk[0];
    const bool kw_title = head == "TITLE", kw_min = head == "DOMAIN_MIN", kw_max = head == "DOMAIN_MAX";
    const bool kw_1d = head == "LUT_1D_SIZE", kw_3d = head == "LUT_3D_SIZE";

    if (kw_title || kw_min || kw_max || kw_1d || kw_3d) {
      // ensure_header (parser.rs:284-303)
      if (kind != Kind::Header && have_data) fail("Header found after LUT data" + at_line(line_no, line));
      if (kw_title) continue;  // arguments ignored
      if (kw_min || kw_max) {
        float v[3];
        for (int k = 0; k < 3; k++) {
          if (tk.size() < size_t(k + 2)) fail("Invalid line " + std::to_string(line_no) + ": " + std::string(line));
          if (!parse_f32_token(tk[k + 1], v[k])) fail("Invalid float" + at_line(line_no, line));
        }
        if (tk.size() != 4) fail("Invalid line " + std::to_string(line_no) + ": " + std::string(line));
        std::memcpy(kw_min ? dmin : dmax, v, sizeof v);
        continue;
      }
      if (kind != Kind::Header) fail(std::string("Invalid ") + (kw_1d ? "LUT_1D_SIZE" : "LUT_3D_SIZE") + at_line(line_no, line));
      if (tk.size() < 2) fail("Invalid line " + std::to_string(line_no) + ": " + std::string(line));
      size_t sz = 0;
      if (!parse_usize_token(tk[1], sz)) fail("Invalid integer" + at_line(line_no, line));
      if (tk.size() != 2) fail("Invalid line " + std::to_string(line_no) + ": " + std::string(line));
      const size_t lo = kw_1d ? kLut1dMin : kLut3dMin, hi = kw_1d ? kLut1dMax : kLut3dMax;
      if (sz < lo || sz > hi)
        fail("Invalid LUT size " + std::to_string(sz) + " at line " + std::to_string(line_no) + ", expected " +
             std::to_string(lo) + "..=" + std::to_string(hi));
      size = sz;
      kind = kw_1d ? Kind::Lut1D : Kind::Lut3D;
      have_data = false;
      continue;
    }

    if (kind == Kind::Header) fail("LUT data found before LUT size" + at_line(line_no, line));
    have_data = true;
    float v[3];
    for (int k = 0; k < 3; k++) {
      if (tk.size() < size_t(k + 1)) fail("Invalid line " + std::to_string(line_no) + ": " + std::string(line));
      if (!parse_f32_token(tk[k], v[k])) fail("Invalid float" + at_line(line_no, line));
    }
    if (tk.size() != 3) fail("Invalid line " + std::to_string(line_no) + ": " + std::string(line));
    rows.insert(rows.end(), v, v + 3);
  }

  // parser.rs:205-212 — comparisons with NaN are false, so a NaN bound is accepted like in the reference
  if (dmin[0] >= dmax[0] || dmin[1] >= dmax[1] || dmin[2] >= dmax[2]) fail("Invalid domain min/max");
  if (kind == Kind::Header) fail("Missing LUT size");

  auto cube = new mi355h_cube();
  cube->size = size;
  const size_t n_rows = rows.size() / 3;
  if (kind == Kind::Lut1D) {
    if (n_rows != size) {
      delete cube;
      fail("Invalid 1D LUT value count, expected " + std::to_string(size) + ", got " + std::to_string(n_rows));
    }
    cube->is3d = false;
    cube->table.resize(3 * size);
    for (size_t i = 0; i < size; i++)
      for (int c = 0; c < 3; c++) cube->table[size_t(c) * size + i] = rows[i * 3 + c];
  } else {
    const size_t expected = size * size * size;
    if (n_rows != expected) {
      delete cube;
      fail("Invalid 3D LUT value count, expected " + std::to_string(expected) + ", got " + std::to_string(n_rows));
    }
    cube->is3d = true;
    cube->table.resize(4 * expected);
    for (size_t i = 0; i < expected; i++) {
      cube->table[4 * i + 0] = rows[3 * i + 0];
      cube->table[4 * i + 1] = rows[3 * i + 1];
      cube->table[4 * i + 2] = rows[3 * i + 2];
      cube->table[4 * i + 3] = 1.0f;  // parser.rs:253-256
    }
  }
  for (int c = 0; c < 3; c++) {  // parser.rs:264-274
    cube->domain_scale[c] = 1.0f / (dmax[c] - dmin[c]);
    cube->domain_offset[c] = -dmin[c] * cube->domain_scale[c];
  }
  return cube;
}

void put_err(char *err, size_t errlen, const std::string &m) {
  if (!err || errlen == 0) return;
  const size_t n = m.size() < errlen - 1 ? m.size() : errlen - 1;
  std::memcpy(err, m.data(), n);
  err[n] = 0;
}

}  // namespace

extern "C" {

mi355h_cube *mi355h_cube_parse(const char *text, size_t len, char *err, size_t errlen) {
  try {
    return parse_text(sv(text ? text : "", text ? len : 0));
  } catch (const ParseFail &f) {
    put_err(err, errlen, f.msg);
    return nullptr;
  } catch (const std::exception &e) {
    put_err(err, errlen, e.what());
    return nullptr;
  }
}

mi355h_cube *mi355h_cube_parse_file(const char *path, char *err, size_t errlen) {
  std::ifstream f(path ? path : "", std::ios::binary);
  if (!f) {
    put_err(err, errlen, std::string("IO error: cannot open ") + (path ? path : "(null)"));
    return nullptr;
  }
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return mi355h_cube_parse(text.data(), text.size(), err, errlen);
}

void mi355h_cube_free(mi355h_cube *c) { delete c; }
int mi355h_cube_is3d(const mi355h_cube *c) { return c && c->is3d ? 1 : 0; }
size_t mi355h_cube_size(const mi355h_cube *c) { return c ? c->size : 0; }
const float *mi355h_cube_table(const mi355h_cube *c) { return c ? c->table.data() : nullptr; }
size_t mi355h_cube_table_len(const mi355h_cube *c) { return c ? c->table.size() : 0; }
void mi355h_cube_domain(const mi355h_cube *c, float scale[3], float offset[3]) {
  if (!c) return;
  std::memcpy(scale, c->domain_scale, sizeof c->domain_scale);
  std::memcpy(offset, c->domain_offset, sizeof c->domain_offset);
}

}  // extern "C"
