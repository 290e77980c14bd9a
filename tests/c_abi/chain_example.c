/* tests/c_abi/chain_example.c — a plain C99 consumer of include/mi355fx.h, the way a reference-side shim binds the
 * library: hsvfilter in place, then colorlut, through the host-buffer entry points, on a frame and a 17^3
 * LUT generated here with a fixed LCG. Writes the output frame to argv[1]; tests/test_gpu_c_abi.py regenerates the same
 * inputs with numpy, runs the CPU oracle and compares byte for byte. Also checks the "No LUT configured" error path.
 * Build: gcc -std=c99 -Iinclude tests/c_abi/chain_example.c -Lgst-plugins-rs_amd -lmi355fx -Wl,-rpath,... */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355fx.h"

#define W 640
#define H 360
#define S 17

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <out.bin>\n", argv[0]); return 2; }
  uint32_t seed = 12345u;
  size_t n = (size_t)W * H * 4, i;
  uint8_t *frame = (uint8_t *)malloc(n), *out = (uint8_t *)malloc(n);
  float *table = (float *)malloc(sizeof(float) * 4 * S * S * S);
  const float scale[3] = {1.0f, 1.0f, 1.0f}, offset[3] = {0.0f, 0.0f, 0.0f};
  mi355_hsv_settings hs;
  mi355_ctx *ctx;
  int rc, x, y, z;
  FILE *f;
  if (!frame || !out || !table) return 3;
  for (i = 0; i < n; i++) frame[i] = (uint8_t)(lcg(&seed) & 255u);
  for (z = 0; z < S; z++)
    for (y = 0; y < S; y++)
      for (x = 0; x < S; x++) {
        float *c = table + 4 * ((size_t)x + (size_t)y * S + (size_t)z * S * S);
        /* identity with the channels swapped pairwise towards each other by a quarter: exact in f32 */
        float r = (float)x / (float)(S - 1), g = (float)y / (float)(S - 1), b = (float)z / (float)(S - 1);
        c[0] = 0.75f * r + 0.25f * g;
        c[1] = 0.75f * g + 0.25f * b;
        c[2] = 0.75f * b + 0.25f * r;
        c[3] = 1.0f;
      }
  memset(&hs, 0, sizeof hs);
  hs.hue_shift = 45.0f; hs.saturation_mul = 1.25f; hs.saturation_off = -0.05f; hs.value_mul = 0.9f; hs.value_off = 0.02f;

  ctx = mi355_ctx_create(0, &rc);
  if (!ctx) { fprintf(stderr, "ctx_create: %d\n", rc); return 4; }
  /* colorlut without a LUT must fail the way the element does */
  rc = mi355_colorlut_frame(ctx, frame, W * 4, out, W * 4, W, H, MI355_FMT_RGBA);
  if (rc != MI355_ERR_NOT_CONFIGURED) { fprintf(stderr, "expected NOT_CONFIGURED, got %d\n", rc); return 5; }
  rc = mi355_colorlut_load(ctx, 1, S, table, scale, offset);
  if (rc != MI355_OK) { fprintf(stderr, "colorlut_load: %d %s\n", rc, mi355_ctx_last_error(ctx)); return 6; }
  rc = mi355_hsvfilter_frame_ip(ctx, frame, n, W, W * 4, MI355_FMT_RGBA, &hs);
  if (rc != MI355_OK) { fprintf(stderr, "hsvfilter: %d %s\n", rc, mi355_ctx_last_error(ctx)); return 7; }
  rc = mi355_colorlut_frame(ctx, frame, W * 4, out, W * 4, W, H, MI355_FMT_RGBA);
  if (rc != MI355_OK) { fprintf(stderr, "colorlut: %d %s\n", rc, mi355_ctx_last_error(ctx)); return 8; }
  mi355_ctx_destroy(ctx);
  f = fopen(argv[1], "wb");
  if (!f || fwrite(out, 1, n, f) != n) return 9;
  fclose(f);
  free(frame); free(out); free(table);
  puts("ok");
  return 0;
}
