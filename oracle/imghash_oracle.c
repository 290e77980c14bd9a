/* oracle/imghash_oracle.c — CPU restatement of videocompare's resize-based hashes. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: HashAlgorithm::{Mean, Gradient, VertGradient, DoubleGradient} (video/videofx/src/videocompare/mod.rs:60-100,
 * hashed_image.rs:24-40,83-95) run in the third-party crates image_hasher 3.1.1 and image 0.25.10 (Cargo.lock:7459-7460,
 * 7398-7399), whose sources are not under /root/reference; no reference test uses them. Published algorithm restated:
 *   HasherConfig::new(): 8x8 hash, FilterType::Lanczos3, no DCT, no gaussian pre-blur.
 *   to_grayscale: u8 luma = (2126*r + 7152*g + 722*b) / 10000 (integer; alpha ignored).
 *   resize to (w,h) = Mean (8,8), Gradient (9,8), VertGradient (8,9), DoubleGradient (5,5) with image::imageops::resize:
 *     vertical_sample to an f32 image of the new height, then horizontal_sample to the new width; per output sample
 *       ratio = in/out; sratio = max(ratio,1); support = 3*sratio; centre = (o+0.5)*ratio;
 *       left = clamp(floor(centre - support), 0, in-1); right = clamp(ceil(centre + support), left+1, in);
 *       w_i = lanczos3((i - (centre-0.5)) / sratio), normalised by their running f32 sum; value = sum_i p_i*w_i (f32, in order);
 *     the horizontal pass clamps to [0,255] and rounds half away from zero (FloatNearest) to u8.
 *   bits: Mean  v >= mean (mean = sum/len in u32, truncated to u8); Gradient  row-wise v[x] < v[x+1];
 *         VertGradient  v[y][x] < v[y+1][x]; DoubleGradient  the Gradient bits of the 5x5 image, then its VertGradient bits.
 *   distance = Hamming distance. lanczos3(x) = |x| < 3 ? sinc(x)*sinc(x/3) : 0, sinc(t) = t == 0 ? 1 : sin(pi t)/(pi t), f32 (sinf). */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float sincf_(float t) { const float a = t * 3.14159265358979323846f; return t == 0.0f ? 1.0f : sinf(a) / a; }
static float lanczos3(float x) { return fabsf(x) < 3.0f ? sincf_(x) * sincf_(x / 3.0f) : 0.0f; }

/* weights of one output sample; returns count, *left = first input index */
static int weights(int in, int out, int o, float *ws, int *left_out) {
  const float ratio = (float)in / (float)out;
  const float sratio = ratio < 1.0f ? 1.0f : ratio;
  const float support = 3.0f * sratio;
  float centre = ((float)o + 0.5f) * ratio;
  long left = (long)floorf(centre - support);
  if (left < 0) left = 0;
  if (left > in - 1) left = in - 1;
  long right = (long)ceilf(centre + support);
  if (right < left + 1) right = left + 1;
  if (right > in) right = in;
  centre = centre - 0.5f;
  float sum = 0.0f;
  int n = 0;
  for (long i = left; i < right; i++) { const float w = lanczos3(((float)i - centre) / sratio); ws[n++] = w; sum += w; }
  for (int i = 0; i < n; i++) ws[i] /= sum;
  *left_out = (int)left;
  return n;
}

/* algo: 0 Mean, 1 Gradient, 2 VertGradient, 3 DoubleGradient. Returns the number of hash bits (<= 64), bits in *hash
 * (bit i = i-th bool in the crate's iteration order); optionally the resized u8 image in `small`. */
int oracle_imghash(const uint8_t *data, int width, int height, int stride, int channels, int algo, uint64_t *hash, uint8_t *small) {
  int rw, rh;
  switch (algo) { case 0: rw = 8; rh = 8; break; case 1: rw = 9; rh = 8; break; case 2: rw = 8; rh = 9; break; default: rw = 5; rh = 5; break; }
  float *tmp = (float *)malloc(sizeof(float) * (size_t)width * rh);   /* vertical_sample output: rh rows x width */
  float *ws = (float *)malloc(sizeof(float) * (size_t)(height > width ? height : width + 8));
  for (int oy = 0; oy < rh; oy++) {
    int left;
    const int n = weights(height, rh, oy, ws, &left);
    for (int x = 0; x < width; x++) {
      float t = 0.0f;
      for (int i = 0; i < n; i++) {
        const uint8_t *p = data + (size_t)(left + i) * stride + (size_t)x * channels;
        const uint32_t l = (2126u * p[0] + 7152u * p[1] + 722u * p[2]) / 10000u;
        t += (float)l * ws[i];
      }
      tmp[(size_t)oy * width + x] = t;
    }
  }
  uint8_t img[81];
  for (int ox = 0; ox < rw; ox++) {
    int left;
    const int n = weights(width, rw, ox, ws, &left);
    for (int y = 0; y < rh; y++) {
      float t = 0.0f;
      for (int i = 0; i < n; i++) t += tmp[(size_t)y * width + left + i] * ws[i];
      t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
      img[y * rw + ox] = (uint8_t)roundf(t);
    }
  }
  free(tmp); free(ws);
  if (small) memcpy(small, img, (size_t)rw * rh);
  uint64_t h = 0; int nb = 0;
  if (algo == 0) {
    uint32_t sum = 0; for (int i = 0; i < 64; i++) sum += img[i];
    const uint8_t mean = (uint8_t)(sum / 64u);
    for (int i = 0; i < 64; i++) if (img[i] >= mean) h |= 1ull << nb, nb++; else nb++;
  } else if (algo == 1 || algo == 3) {
    for (int y = 0; y < rh; y++) for (int x = 0; x + 1 < rw; x++) { if (img[y * rw + x] < img[y * rw + x + 1]) h |= 1ull << nb; nb++; }
    if (algo == 3) for (int y = 0; y + 1 < rh; y++) for (int x = 0; x < rw; x++) { if (img[y * rw + x] < img[(y + 1) * rw + x]) h |= 1ull << nb; nb++; }
  } else {
    for (int y = 0; y + 1 < rh; y++) for (int x = 0; x < rw; x++) { if (img[y * rw + x] < img[(y + 1) * rw + x]) h |= 1ull << nb; nb++; }
  }
  *hash = h;
  return nb;
}
