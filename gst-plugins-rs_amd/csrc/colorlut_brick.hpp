// colorlut_brick.hpp — host-side handle of the brick-cache colorlut kernel (colorlut_brick.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "brickwatch.hpp"
#include <cstddef>
#include <cstdint>

struct mi355_ctx;
struct mi355_hsv_settings;

namespace mi355 {

constexpr int kBrickMaxSize = 65;  // size^3 lines of 128 B: 35 MB at 65 (the tag field itself would allow 128)

// Device-side brick form of a loaded 3D LUT (built by brick_upload at mi355_colorlut_load).
struct BrickLut {
  float *d_bricks = nullptr;               // size^3 x 32 floats: {c, d} x 4 corner rows (24 floats) + 8 pad
  uint32_t *d_axis = nullptr;              // 2 geometries x 3 x 256 x {t, set byte offset | tag contribution << 16}
  uint32_t *d_cellnum = nullptr;           // 3 x 256: per-axis contribution to the cell number x0 + S*y0 + S*S*z0 (miss path)
  unsigned long long *d_counters = nullptr;  // 1024 slots x {256-pixel steps with a cache miss, steps that ended on the slow path}
  int size = 0;
  float scale[3] = {1, 1, 1}, offset[3] = {0, 0, 0};  // domain (the RGBA64 kernels compute coordinates from it)
  int fold_axis = 2;                       // axis that has 2 residues in the 32-set geometry (2 = z)
  bool ok = false;                         // kernel applicable to this LUT (3D, size <= kBrickMaxSize, finite domain)
  // content watch (brick_choose / brick_after_launch): the miss counters are copied to pinned host memory every few
  // launches behind an event that is only ever polled, never waited for
  unsigned long long *h_counters = nullptr;
  hipEvent_t ev = nullptr;
  bool pending = false;                    // a snapshot copy is in flight
  unsigned long long px_since = 0;         // pixels launched through the brick kernel since the last snapshot
  unsigned long long px_snapshot = 0;      // ... covered by the snapshot in flight
  unsigned launches_since = 0;
  int level_since = -1;                    // level of the launches counted in px_since (a change of level restarts the count)
  int level_snapshot = 0;                  // level of the snapshot in flight
  bool shared_snapshot = false;            // ... and whether level 1 is the block-shared cache for the launches it covers
  BrickWatch watch;                        // the policy (brickwatch.hpp)
};

// Content watch for the interpolating path: level for this launch (0 / 1 = brick kernel with 32 / 64 sets, 2 = three-pass).
int brick_choose(BrickLut &B, int min_level = 0);
// before / after a brick launch of `pixels` pixels at `level` on ctx's stream: count it and, every few launches, start a non-blocking snapshot
int brick_before_launch(mi355_ctx *ctx, BrickLut &B, int level);
void brick_mark_unwatched(BrickLut &B);
int brick_after_launch(mi355_ctx *ctx, BrickLut &B, unsigned long long pixels, int level, bool shared1 = false);

int brick_upload(mi355_ctx *ctx, BrickLut &B, int S, const float *cells, const float scale[3], const float offset[3]);
void brick_release(BrickLut &B);
bool brick_applicable(const BrickLut &B, const uint8_t *d_src, size_t src_pitch, int src_stride, const uint8_t *d_dst, size_t dst_pitch,
                      int dst_stride, int n_frames, int width, int height, int bytes_per_pixel = 4);
// hs == nullptr: colorlut alone; otherwise the fused hsvfilter -> colorlut chain. src == dst is allowed.
int brick_launch(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int n_frames, int width, int height,
                 const mi355_hsv_settings *hs, int sets, int fmt64 = 0);  // sets: 32 or 64; fmt64: 0 RGBA8, 1 RGBA64_LE, 2 RGBA64_BE
// the block-shared brick cache (colorlut3d_shared_kernel): RGBA8, plain colorlut; same applicability as the brick kernel plus
// a launch large enough for a cache to warm up
bool shared_applicable(const mi355_ctx *ctx, int width, int dst_stride, int n_frames, int height, bool *small = nullptr);
int shared_launch(mi355_ctx *ctx, const BrickLut &B, const uint8_t *d_src, int src_stride, uint8_t *d_dst, int dst_stride, int n_frames, int width, int height);
// synchronous read (and optional reset) of the miss counters
int brick_read_counters(mi355_ctx *ctx, const BrickLut &B, unsigned long long out[2], bool reset);

}  // namespace mi355
