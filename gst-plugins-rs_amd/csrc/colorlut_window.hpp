// colorlut_window.hpp — launcher of the kernel that reads a memoised 2^24-entry table through a block-shared LDS cache of
// table bricks (colorlut_window.hip).
//
// The table is Morton-indexed (colorlut_kernels.hip: layout 1, bit k of r / g / b at bit 3k / 3k+1 / 3k+2 of the slot):
//
//   bits  0..5   the colour's place inside its BRICK, a 4 x 4 x 4 colour cube (the two low bits of r, g, b interleaved)
//   bits  6..13  the brick's place inside a box of 8 x 8 x 4 bricks (bits 2..4 of r and g, bits 2..3 of b) = the SET of
//                the LDS cache
//   bits 14..23  which box = the TAG
//
// so a brick is 64 consecutive entries (256 B, two cache lines), the slot is X[r] | Y[g] | Z[b] for three 256-entry
// tables whose bits do not overlap (one v_add3), and `slot & 0x3fff` is the entry's place in a way of the LDS cache. One
// table serves this kernel and the gather kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

struct mi355_ctx;

namespace mi355 {

// bit k of an 8-bit value -> bit 3k + axis: the contribution of channel `axis` (0 = r, 1 = g, 2 = b) with value v to the slot
__host__ __device__ constexpr uint32_t window_axis_entry(int axis, uint32_t v) {
  uint32_t o = 0;
  for (int k = 0; k < 8; k++) o |= ((v >> k) & 1u) << (3 * k + axis);
  return o;
}
static_assert((window_axis_entry(0, 255) | window_axis_entry(1, 255) | window_axis_entry(2, 255)) == 0xffffffu &&
              (window_axis_entry(0, 255) & window_axis_entry(1, 255)) == 0u && window_axis_entry(2, 0x10) == (1u << 14), "Morton");

// steps (256 x 32 pixel tiles) a block must have ahead of it for the LDS cache to pay: its first step runs on a cold cache
// (default of MI355_FLAG_WINDOW_MIN_STEPS)
constexpr int kWindowMinStepsPerBlock = 3;

// packed RGBA8 rows of w4 16-byte groups (row strides sw4 / dw4 groups) through the Morton-indexed `table`. The kernel wants
// width % 4 == 0; anything else is the caller's business (colorlut_kernels.hip: launch_table_raw).
bool window_applicable(const mi355_ctx *ctx, unsigned w4, unsigned dw4, size_t rows);
int launch_window_table(mi355_ctx *ctx, const uint32_t *table, const uint8_t *d_src, uint8_t *d_dst, unsigned w4, unsigned sw4, unsigned dw4, size_t rows);
// the same over n_frames SEPARATE packed frames (rows of w4 groups, frame_rows rows each) in one launch on `stream` (group.hip)
bool window_multi_applicable(const mi355_ctx *ctx, unsigned w4, size_t frame_rows, int n_frames);
int launch_window_table_multi(mi355_ctx *ctx, hipStream_t stream, const uint32_t *table, uint8_t *const *srcs, uint8_t *const *dsts, int n_frames, unsigned w4,
                              size_t frame_rows);
// {pixels looked up, pixels served from the table in global memory, bricks installed} since the last reset (diagnostics; synchronous)
int window_read_counters(mi355_ctx *ctx, unsigned long long out[3], bool reset);
void window_release(mi355_ctx *ctx);

}  // namespace mi355
