/* oracle/hrtf_oracle.c — CPU restatement of the hrtfrender hot path. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party crate `hrtf` 0.8.1 (+ `rustfft` 6.4.1;
 * Cargo.lock:6939-6940,10713-10714), whose sources are not under /root/reference, and the reference's own
 * tests (audio/hrtf/tests/hrtfrender.rs) pin only buffer sizes and timestamps, no sample values. This file
 * restates the crate's published algorithm, anchored on the reference's call sites:
 *
 *   audio/hrtf/src/hrtf/imp.rs:84-94    HrirSphere::new(bytes, rate) / from_file
 *   audio/hrtf/src/hrtf/imp.rs:662-680  HrtfProcessor::new(sphere, interpolation_steps, block_length); per channel
 *                                       prev_left/right_samples, prev_sample_vector, prev_distance_gain
 *   audio/hrtf/src/hrtf/imp.rs:214-246  per block and channel: de-interleave, process_samples(HrtfContext{..}),
 *                                       remember new vector/gain as the previous ones
 *   audio/hrtf/src/hrtf/imp.rs:256-268  sum every channel's (l, r) scratch into the zero-filled stereo output
 *
 * Algorithm of the crate (as published):
 *   file:   "HRIR" | u32 rate | u32 len | u32 n_vertices | u32 n_indices | u32 idx[n_indices] |
 *           n_vertices x { f32 x, y, z | f32 left[len] | f32 right[len] }        (little endian)
 *   setup:  pad = len - 1; N = block_len + pad; HRTF(vertex, ear) = FFT_N(hrir zero-padded to N)
 *   block:  for step in 0..steps: t = (step+1)/steps; dir = lerp(prev_vec, new_vec, t);
 *           first face (file order) hit by the ray origin->10*dir, 0 < t_hit < 1, barycentric (u,v,w) inside with
 *           f32::EPSILON slack -> hrtf = u*A + v*B + w*C (kept from the previous call when nothing is hit);
 *           in[0..pad] = previous tail, in[pad..] = block samples, tail = last `pad` samples of in;
 *           in = IFFT_N(FFT_N(in) * hrtf) (unnormalised); gain = lerp(prev_gain, new_gain, t) / N;
 *           out[i] += (in_left[pad+i].re * gain, in_right[pad+i].re * gain)
 * i.e. streaming linear convolution of each source with a per-step barycentric blend of three HRIRs.
 * Sample-rate conversion of the sphere (rubato, when file rate != stream rate) is not restated: unsupported.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cpx;

/* ------------------------------------------------------------------ generic-size complex FFT (f32 data, twiddles
 * computed in f64 and rounded once, as rustfft does); mixed radix with O(p^2) butterflies for each prime factor. */
typedef struct {
  int n;
  int nfac, fac[32];
  cpx *tw;      /* e^{-2 pi i k / n}, k in [0, n) */
  cpx *tmp;
} fft_plan;

static fft_plan *fft_plan_new(int n) {
  fft_plan *p = (fft_plan *)calloc(1, sizeof(*p));
  p->n = n;
  int m = n;
  for (int f = 2; m > 1;) {
    if (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
    else { f++; if ((long)f * f > m) f = m; }
  }
  p->tw = (cpx *)malloc(sizeof(cpx) * (size_t)n);
  p->tmp = (cpx *)malloc(sizeof(cpx) * (size_t)n);
  for (int k = 0; k < n; k++) {
    const double a = -2.0 * M_PI * (double)k / (double)n;
    p->tw[k].re = (float)cos(a);
    p->tw[k].im = (float)sin(a);
  }
  return p;
}
static void fft_plan_free(fft_plan *p) { if (p) { free(p->tw); free(p->tmp); free(p); } }

/* out[0..n) = DFT of in[0], in[stride], ... (n = product of fac[level..]); decimation in time. */
static void fft_rec(const fft_plan *P, int level, int n, const cpx *in, int stride, cpx *out, int inverse) {
  if (n == 1) { out[0] = in[0]; return; }
  const int p = P->fac[level], m = n / p;
  for (int r = 0; r < p; r++) fft_rec(P, level + 1, m, in + (size_t)r * stride, stride * p, out + (size_t)r * m, inverse);
  const int tstep = P->n / n;  /* twiddle index scale: e^{-2 pi i k / n} = tw[k * tstep] */
  cpx col[64];
  cpx *colp = p <= 64 ? col : (cpx *)malloc(sizeof(cpx) * (size_t)p);
  for (int k = 0; k < m; k++) {
    for (int r = 0; r < p; r++) {
      cpx w = P->tw[((long)r * k % n) * tstep];
      if (inverse) w.im = -w.im;
      const cpx v = out[(size_t)r * m + k];
      colp[r].re = v.re * w.re - v.im * w.im;
      colp[r].im = v.re * w.im + v.im * w.re;
    }
    for (int q = 0; q < p; q++) {
      float sr = 0.0f, si = 0.0f;
      for (int r = 0; r < p; r++) {
        cpx w = P->tw[((long)q * r % p) * (P->n / p)];
        if (inverse) w.im = -w.im;
        sr += colp[r].re * w.re - colp[r].im * w.im;
        si += colp[r].re * w.im + colp[r].im * w.re;
      }
      P->tmp[(size_t)q * m + k].re = sr;
      P->tmp[(size_t)q * m + k].im = si;
    }
  }
  memcpy(out, P->tmp, sizeof(cpx) * (size_t)n);
  if (colp != col) free(colp);
}

static void fft_run(const fft_plan *P, cpx *buf, cpx *scratch, int inverse) {
  memcpy(scratch, buf, sizeof(cpx) * (size_t)P->n);
  fft_rec(P, 0, P->n, scratch, 1, buf, inverse);
}

/* ------------------------------------------------------------------ sphere */
typedef struct { float x, y, z; } vec3;
static vec3 v_sub(vec3 a, vec3 b) { return (vec3){a.x - b.x, a.y - b.y, a.z - b.z}; }
static vec3 v_add(vec3 a, vec3 b) { return (vec3){a.x + b.x, a.y + b.y, a.z + b.z}; }
static vec3 v_scale(vec3 a, float s) { return (vec3){a.x * s, a.y * s, a.z * s}; }
static float v_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static vec3 v_cross(vec3 a, vec3 b) { return (vec3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static vec3 v_normalize(vec3 a) { const float l = sqrtf(v_dot(a, a)); return (vec3){a.x / l, a.y / l, a.z / l}; }

typedef struct {
  uint32_t rate, len, n_vertices, n_indices;
  uint32_t *indices;
  vec3 *pos;
  float *hrir; /* [vertex][ear][len] */
} hrir_sphere;

static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static float rd_f32(const uint8_t *p) { uint32_t u = rd_u32(p); float f; memcpy(&f, &u, 4); return f; }

/* returns 0 ok, -1 bad magic / truncated, -2 zero length, -3 rate mismatch (resampling unsupported), -4 bad index */
int oracle_hrir_parse(const uint8_t *bytes, size_t n, uint32_t device_rate, hrir_sphere **out) {
  *out = NULL;
  if (n < 20 || memcmp(bytes, "HRIR", 4) != 0) return -1;
  hrir_sphere *s = (hrir_sphere *)calloc(1, sizeof(*s));
  s->rate = rd_u32(bytes + 4);
  s->len = rd_u32(bytes + 8);
  s->n_vertices = rd_u32(bytes + 12);
  s->n_indices = rd_u32(bytes + 16);
  if (s->len == 0) { free(s); return -2; }
  const size_t need = 20 + 4 * (size_t)s->n_indices + (size_t)s->n_vertices * (12 + 8 * (size_t)s->len);
  if (n < need) { free(s); return -1; }
  if (s->rate != device_rate) { free(s); return -3; }
  s->indices = (uint32_t *)malloc(4 * (size_t)s->n_indices + 4);
  s->pos = (vec3 *)malloc(sizeof(vec3) * (size_t)s->n_vertices + 4);
  s->hrir = (float *)malloc(sizeof(float) * 2 * (size_t)s->len * (size_t)s->n_vertices + 4);
  size_t off = 20;
  for (uint32_t i = 0; i < s->n_indices; i++, off += 4) {
    s->indices[i] = rd_u32(bytes + off);
    if (s->indices[i] >= s->n_vertices) { free(s->indices); free(s->pos); free(s->hrir); free(s); return -4; }
  }
  for (uint32_t v = 0; v < s->n_vertices; v++) {
    s->pos[v] = (vec3){rd_f32(bytes + off), rd_f32(bytes + off + 4), rd_f32(bytes + off + 8)};
    off += 12;
    for (uint32_t k = 0; k < 2 * s->len; k++, off += 4) s->hrir[(size_t)v * 2 * s->len + k] = rd_f32(bytes + off);
  }
  *out = s;
  return 0;
}
void oracle_hrir_free(hrir_sphere *s) { if (s) { free(s->indices); free(s->pos); free(s->hrir); free(s); } }
uint32_t oracle_hrir_len(const hrir_sphere *s) { return s->len; }
uint32_t oracle_hrir_vertices(const hrir_sphere *s) { return s->n_vertices; }
uint32_t oracle_hrir_faces(const hrir_sphere *s) { return s->n_indices / 3; }

/* ray from the origin along `dir` (already scaled) against triangle (a,b,c); barycentric weights on a hit */
static int ray_triangle(vec3 dir, vec3 a, vec3 b, vec3 c, float *u, float *v, float *w) {
  const vec3 origin = {0.0f, 0.0f, 0.0f};
  const vec3 ba = v_sub(b, a), ca = v_sub(c, a);
  const vec3 normal = v_normalize(v_cross(ba, ca));
  const float d = -v_dot(a, normal);
  const float num = -(v_dot(origin, normal) + d);
  const float den = v_dot(dir, normal);
  const float t = num / den;
  if (!(t > 0.0f && t < 1.0f)) return 0;
  const vec3 p = v_add(origin, v_scale(dir, t));
  const vec3 v0 = ba, v1 = ca, v2 = v_sub(p, a);
  const float d00 = v_dot(v0, v0), d01 = v_dot(v0, v1), d11 = v_dot(v1, v1), d20 = v_dot(v2, v0), d21 = v_dot(v2, v1);
  const float denom = d00 * d11 - d01 * d01;
  const float bv = (d11 * d20 - d01 * d21) / denom;
  const float bw = (d00 * d21 - d01 * d20) / denom;
  const float bu = 1.0f - bv - bw;
  const float eps = 1.1920929e-7f; /* f32::EPSILON */
  if (!(bu >= -eps && bv >= -eps && bu + bv <= 1.0f + eps)) return 0;
  *u = bu; *v = bv; *w = bw;
  return 1;
}

/* exported for the tests of the device-side search: first face hit, weights; returns face index or -1 */
int oracle_hrir_sample(const hrir_sphere *s, const float dir3[3], float uvw[3]) {
  const vec3 dir = v_scale((vec3){dir3[0], dir3[1], dir3[2]}, 10.0f);
  for (uint32_t f = 0; f + 2 < s->n_indices; f += 3) {
    float u, v, w;
    if (ray_triangle(dir, s->pos[s->indices[f]], s->pos[s->indices[f + 1]], s->pos[s->indices[f + 2]], &u, &v, &w)) {
      uvw[0] = u; uvw[1] = v; uvw[2] = w;
      return (int)(f / 3);
    }
  }
  return -1;
}

/* ------------------------------------------------------------------ processor (one per channel, imp.rs:662-680) */
typedef struct {
  const hrir_sphere *sphere;
  int steps, block_len, pad, n;
  fft_plan *plan;
  cpx *hrtf;          /* [vertex][ear][n] */
  cpx *cur[2];        /* interpolated hrtf of the last successful sample_bilinear, per ear */
  cpx *inbuf[2], *scratch;
} hrtf_processor;

typedef struct {
  int channels;
  hrtf_processor **proc;
  float *prev_tail;   /* [channel][ear][pad] */
  float *prev_vec;    /* [channel][3] */
  float *prev_gain;   /* [channel] */
  uint8_t *have_prev; /* [channel] */
} hrtf_render;

static hrtf_processor *processor_new(const hrir_sphere *s, int steps, int block_len) {
  hrtf_processor *p = (hrtf_processor *)calloc(1, sizeof(*p));
  p->sphere = s; p->steps = steps; p->block_len = block_len;
  p->pad = (int)s->len - 1;
  p->n = block_len + p->pad;
  p->plan = fft_plan_new(p->n);
  p->hrtf = (cpx *)calloc((size_t)s->n_vertices * 2 * (size_t)p->n, sizeof(cpx));
  p->scratch = (cpx *)calloc((size_t)p->n, sizeof(cpx));
  for (int e = 0; e < 2; e++) {
    p->cur[e] = (cpx *)calloc((size_t)p->n, sizeof(cpx));
    p->inbuf[e] = (cpx *)calloc((size_t)p->n, sizeof(cpx));
  }
  for (uint32_t v = 0; v < s->n_vertices; v++)
    for (int e = 0; e < 2; e++) {
      cpx *h = p->hrtf + ((size_t)v * 2 + e) * (size_t)p->n;
      for (uint32_t k = 0; k < s->len; k++) h[k].re = s->hrir[((size_t)v * 2 + e) * s->len + k];
      fft_run(p->plan, h, p->scratch, 0);
    }
  return p;
}
static void processor_free(hrtf_processor *p) {
  if (!p) return;
  fft_plan_free(p->plan); free(p->hrtf); free(p->scratch);
  for (int e = 0; e < 2; e++) { free(p->cur[e]); free(p->inbuf[e]); }
  free(p);
}

hrtf_render *oracle_hrtf_new(const hrir_sphere *s, int channels, int steps, int block_len) {
  hrtf_render *r = (hrtf_render *)calloc(1, sizeof(*r));
  r->channels = channels;
  r->proc = (hrtf_processor **)calloc((size_t)channels, sizeof(*r->proc));
  for (int c = 0; c < channels; c++) r->proc[c] = processor_new(s, steps, block_len);
  const int pad = (int)s->len - 1;
  r->prev_tail = (float *)calloc((size_t)channels * 2 * (size_t)(pad > 0 ? pad : 1), sizeof(float));
  r->prev_vec = (float *)calloc((size_t)channels * 3, sizeof(float));
  r->prev_gain = (float *)calloc((size_t)channels, sizeof(float));
  r->have_prev = (uint8_t *)calloc((size_t)channels, 1);
  return r;
}
void oracle_hrtf_free(hrtf_render *r) {
  if (!r) return;
  for (int c = 0; c < r->channels; c++) processor_free(r->proc[c]);
  free(r->proc); free(r->prev_tail); free(r->prev_vec); free(r->prev_gain); free(r->have_prev); free(r);
}
/* State::reset_processors (imp.rs:124-129): only the sample tails are cleared */
void oracle_hrtf_reset(hrtf_render *r) {
  const int pad = r->proc[0]->pad;
  memset(r->prev_tail, 0, sizeof(float) * (size_t)r->channels * 2 * (size_t)(pad > 0 ? pad : 1));
}

static float lerpf(float a, float b, float t) { return a + (b - a) * t; }

/* One block of block_len*steps frames: `in` interleaved [frames][channels], `out` stereo interleaved [frames][2]
 * (overwritten: zero fill + mix, imp.rs:186,256-268). positions: [channels][3] already in the right-handed
 * system handed to the crate (imp.rs:64-73); gains: [channels]. */
void oracle_hrtf_process_block(hrtf_render *r, const float *in, float *out, const float *positions, const float *gains) {
  const hrtf_processor *p0 = r->proc[0];
  const int steps = p0->steps, bl = p0->block_len, pad = p0->pad, n = p0->n, C = r->channels;
  const int frames = steps * bl;
  memset(out, 0, sizeof(float) * 2 * (size_t)frames);
  float *scratch_out = (float *)calloc(2 * (size_t)frames, sizeof(float));
  for (int c = 0; c < C; c++) {
    hrtf_processor *p = r->proc[c];
    const hrir_sphere *s = p->sphere;
    const float *nv = positions + 3 * c;
    const float ng = gains[c];
    const float *pv = r->have_prev[c] ? r->prev_vec + 3 * c : nv;
    const float pg = r->have_prev[c] ? r->prev_gain[c] : ng;
    memset(scratch_out, 0, sizeof(float) * 2 * (size_t)frames);
    for (int step = 0; step < steps; step++) {
      const float t = (float)(step + 1) / (float)steps;
      const float dir[3] = {lerpf(pv[0], nv[0], t), lerpf(pv[1], nv[1], t), lerpf(pv[2], nv[2], t)};
      float uvw[3];
      const int face = oracle_hrir_sample(s, dir, uvw);
      if (face >= 0) {
        for (int e = 0; e < 2; e++) {
          const cpx *A = p->hrtf + ((size_t)s->indices[3 * face] * 2 + e) * (size_t)n;
          const cpx *B = p->hrtf + ((size_t)s->indices[3 * face + 1] * 2 + e) * (size_t)n;
          const cpx *Cc = p->hrtf + ((size_t)s->indices[3 * face + 2] * 2 + e) * (size_t)n;
          for (int k = 0; k < n; k++) {
            p->cur[e][k].re = A[k].re * uvw[0] + B[k].re * uvw[1] + Cc[k].re * uvw[2];
            p->cur[e][k].im = A[k].im * uvw[0] + B[k].im * uvw[1] + Cc[k].im * uvw[2];
          }
        }
      }
      for (int e = 0; e < 2; e++) {
        cpx *buf = p->inbuf[e];
        float *tail = r->prev_tail + ((size_t)c * 2 + e) * (size_t)(pad > 0 ? pad : 1);
        for (int k = 0; k < pad; k++) { buf[k].re = tail[k]; buf[k].im = 0.0f; }
        for (int i = 0; i < bl; i++) { buf[pad + i].re = in[(size_t)(step * bl + i) * C + c]; buf[pad + i].im = 0.0f; }
        for (int k = 0; k < pad; k++) tail[k] = buf[n - pad + k].re;
        fft_run(p->plan, buf, p->scratch, 0);
        for (int k = 0; k < n; k++) {
          const cpx a = buf[k], h = p->cur[e][k];
          buf[k].re = a.re * h.re - a.im * h.im;
          buf[k].im = a.re * h.im + a.im * h.re;
        }
        fft_run(p->plan, buf, p->scratch, 1);
      }
      const float k_gain = lerpf(pg, ng, t) / (float)n;
      for (int i = 0; i < bl; i++) {
        scratch_out[2 * (size_t)(step * bl + i)] += p->inbuf[0][pad + i].re * k_gain;
        scratch_out[2 * (size_t)(step * bl + i) + 1] += p->inbuf[1][pad + i].re * k_gain;
      }
    }
    for (int i = 0; i < 2 * frames; i++) out[i] += scratch_out[i];
    memcpy(r->prev_vec + 3 * c, nv, 3 * sizeof(float));
    r->prev_gain[c] = ng;
    r->have_prev[c] = 1;
  }
  free(scratch_out);
}

/* f64 time-domain evaluation of the same mathematical result (streaming convolution with the per-step blended
 * HRIR), used by the tests to bound BOTH the FFT restatement above and the device kernel against the exact value.
 * State is kept separately by the caller: hist [channels][pad] f64 input history, prev vec/gain as above. */
void oracle_hrtf_block_exact(const hrir_sphere *s, int channels, int steps, int bl, const float *in, double *out,
                             const float *prev_pos, const float *prev_gain, const float *positions, const float *gains,
                             double *hist, double *last_taps /* [channels][2][len], persists across calls */) {
  const int L = (int)s->len, pad = L - 1, frames = steps * bl;
  memset(out, 0, sizeof(double) * 2 * (size_t)frames);
  double *x = (double *)malloc(sizeof(double) * (size_t)(pad + frames));
  for (int c = 0; c < channels; c++) {
    for (int k = 0; k < pad; k++) x[k] = hist[(size_t)c * (pad > 0 ? pad : 1) + k];
    for (int i = 0; i < frames; i++) x[pad + i] = in[(size_t)i * channels + c];
    for (int step = 0; step < steps; step++) {
      const float t = (float)(step + 1) / (float)steps;
      const float dir[3] = {lerpf(prev_pos[3 * c], positions[3 * c], t), lerpf(prev_pos[3 * c + 1], positions[3 * c + 1], t),
                            lerpf(prev_pos[3 * c + 2], positions[3 * c + 2], t)};
      float uvw[3];
      const int face = oracle_hrir_sample(s, dir, uvw);
      double *taps = last_taps + (size_t)c * 2 * L;
      if (face >= 0)
        for (int e = 0; e < 2; e++)
          for (int k = 0; k < L; k++)
            taps[e * L + k] = (double)s->hrir[((size_t)s->indices[3 * face] * 2 + e) * L + k] * uvw[0] +
                              (double)s->hrir[((size_t)s->indices[3 * face + 1] * 2 + e) * L + k] * uvw[1] +
                              (double)s->hrir[((size_t)s->indices[3 * face + 2] * 2 + e) * L + k] * uvw[2];
      const double g = (double)lerpf(prev_gain[c], gains[c], t);
      for (int i = 0; i < bl; i++) {
        const int nidx = step * bl + i;
        for (int e = 0; e < 2; e++) {
          double acc = 0.0;
          for (int k = 0; k < L; k++) acc += taps[e * L + k] * x[pad + nidx - k];
          out[2 * (size_t)nidx + e] += acc * g;
        }
      }
    }
    for (int k = 0; k < pad; k++) hist[(size_t)c * (pad > 0 ? pad : 1) + k] = x[frames + k];
  }
  free(x);
}
