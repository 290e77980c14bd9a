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
 * The floating-point slow path for other sizes and the Mean/Gradient/VertGradient/DoubleGradient algorithms
 * (grayscale + Lanczos3 resize in the `image` crate) and HashAlgorithm::Dssim (dssim-core 3.4.0) are not restated. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cmp_u32(const void *a, const void *b) {
  const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}

/* returns 0 and the 64-bit hash (bit i = block i, row-major) or -1 when the fast path does not apply */
int oracle_blockhash(const uint8_t *data, int width, int height, int stride, int channels, uint64_t *hash) {
  if (width <= 0 || height <= 0 || width % 8 || height % 8 || (channels != 3 && channels != 4)) return -1;
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
