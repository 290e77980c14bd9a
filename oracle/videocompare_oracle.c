/* oracle/videocompare_oracle.c — CPU restatement of videocompare's default hash. TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED for arbitrary content: the arithmetic lives in the third-party crate image_hasher 3.1.1
 * (Cargo.lock:7459-7460; call sites video/videofx/src/videocompare/hashed_image.rs:24-79,110-130), whose sources are
 * not under /root/reference. The reference's own tests (video/videofx/tests/videocompare.rs) pin: identical frames
 * -> distance 0 with Blockhash (and Dssim), "snow" vs "red" -> distance > 0.
 *
 * Restated: HashAlg::Blockhash with HasherConfig::new() defaults (8x8 hash) on the tightly packed RGB / RGBA frame
 * (hashed_image.rs:110-130 packs rows; the hasher is handed image::RgbImage / RgbaImage, i.e. ALL channels):
 *   fast path (width % 8 == 0 && height % 8 == 0): u32 block sums of sum_px over (width/8) x (height/8) pixel blocks,
 *     sum_px(RGB) = r+g+b; sum_px(RGBA) = a == 0 ? 765 : r+g+b
 *   bands of hash_width*4 = 32 blocks; median = element len/2 of the sorted band (quick-select, upper median)
 *   bit = block > median || (block == median && median > 765 * block_area / 2)
 *   distance = Hamming distance of the two 64-bit hashes (ImageHash::dist), as f64.
 * The floating-point slow path for other sizes: oracle_blockhash_slow below. Mean/Gradient/VertGradient/DoubleGradient
 * (grayscale + Lanczos3 resize in the `image` crate): imghash_oracle.c; HashAlgorithm::Dssim (dssim-core 3.4.0): dssim_restate.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_u32(const void *a, const void *b) {
  const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}

static int cmp_f32(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return x < y ? -1 : x > y;
}

/* The crate's floating-point path (blockhash_slow), taken when the frame does not divide into 8 x 8 whole blocks. PARITY
 * UNPINNED, restated from memory of image_hasher 3.1.1's src/alg/blockhash.rs like the fast path:
 *   block_width = iwidth as f32 / 8.0, block_height likewise; blocks: [f32; 64] = 0
 *   foreach_pixel8 (row-major): px_sum = sum_px(px) as f32; block_x = x / block_width; x_mod = x + 1. % block_width
 *     [sic: `%` binds tighter than `+`, so x_mod = x + 1 for every block wider than a pixel]; weight_left = fract(x_mod) = 0,
 *     weight_right = 1, block_right = block_left = floor(block_x) (x_mod.trunc() is never 0); the same in y. Of the four
 *     `blocks[..] += px_sum * w * w` updates three add 0.0 and one adds px_sum: every pixel goes WHOLE to block
 *     (floor(x / block_width), floor(y / block_height)), all in f32, accumulated in pixel order - which is what decides the
 *     low bits of a block sum once it passes 2^24 (a 4K block holds up to 99 M).
 *   bands of 32 blocks; median = element 16 of the sorted band; half_block_value = 765 * block_width * block_height / 2
 *   bit = block > median || (|block - median| < 1 && median > half_block_value). */
static int oracle_blockhash_slow(const uint8_t *data, int width, int height, int stride, int channels, uint64_t *hash) {
  const float bw = (float)width / 8.0f, bh = (float)height / 8.0f;
  float blocks[64];
  memset(blocks, 0, sizeof blocks);
  for (int y = 0; y < height; y++) {
    const uint8_t *row = data + (size_t)y * (size_t)stride;
    const int by = (int)floorf((float)y / bh);
    for (int x = 0; x < width; x++) {
      const uint8_t *p = row + (size_t)x * channels;
      uint32_t s = (uint32_t)p[0] + p[1] + p[2];
      if (channels == 4 && p[3] == 0) s = 765;
      const int bx = (int)floorf((float)x / bw);
      blocks[by * 8 + bx] += (float)s;
    }
  }
  const float half = 765.0f * bw * bh / 2.0f;
  uint64_t h = 0;
  for (int g = 0; g < 2; g++) {
    float sorted[32];
    memcpy(sorted, blocks + 32 * g, sizeof sorted);
    qsort(sorted, 32, sizeof(float), cmp_f32);
    const float median = sorted[16];
    for (int i = 0; i < 32; i++) {
      const float b = blocks[32 * g + i];
      if (b > median || (fabsf(b - median) < 1.0f && median > half)) h |= 1ull << (32 * g + i);
    }
  }
  *hash = h;
  return 0;
}

/* returns 0 and the 64-bit hash (bit i = block i, row-major), -1 on bad arguments */
int oracle_blockhash(const uint8_t *data, int width, int height, int stride, int channels, uint64_t *hash) {
  if (width <= 0 || height <= 0 || (channels != 3 && channels != 4)) return -1;
  if (width % 8 || height % 8) return width >= 8 && height >= 8 ? oracle_blockhash_slow(data, width, height, stride, channels, hash) : -1;
  const int bw = width / 8, bh = height / 8;
  uint32_t blocks[64];
  memset(blocks, 0, sizeof blocks);
  for (int y = 0; y < height; y++) {
    const uint8_t *row = data + (size_t)y * (size_t)stride;
    for (int x = 0; x < width; x++) {
      const uint8_t *p = row + (size_t)x * channels;
      uint32_t s = (uint32_t)p[0] + p[1] + p[2];
      if (channels == 4 && p[3] == 0) s = 765;
      blocks[(y / bh) * 8 + x / bw] += s;
    }
  }
  const uint32_t cmp_factor = 765u * (uint32_t)(bw * bh) / 2u;
  uint64_t h = 0;
  for (int g = 0; g < 2; g++) {
    uint32_t sorted[32];
    memcpy(sorted, blocks + 32 * g, sizeof sorted);
    qsort(sorted, 32, sizeof(uint32_t), cmp_u32);
    const uint32_t median = sorted[16];
    for (int i = 0; i < 32; i++) {
      const uint32_t b = blocks[32 * g + i];
      if (b > median || (b == median && median > cmp_factor)) h |= 1ull << (32 * g + i);
    }
  }
  *hash = h;
  return 0;
}

double oracle_hash_distance(uint64_t a, uint64_t b) { return (double)__builtin_popcountll(a ^ b); }
