// hsv_kernels.hip — gfx950 kernels for hsvfilter / hsvdetector.
//
// Reference loops replaced (gst-plugins-rs tree):
//   video/hsv/src/hsvutils.rs:44-128    from_rgb / from_bgr
//   video/hsv/src/hsvutils.rs:132-198   to_rgb / to_bgr
//   video/hsv/src/hsvfilter/imp.rs:96-118     per-pixel filter body
//   video/hsv/src/hsvdetector/imp.rs:126-156  per-pixel detector body
//
// Two arithmetic variants per kernel:
//   GENERIC — literal transliteration (IEEE `/`, fmodf, every branch incl. NaN/inf settings).
//   FAST    — strength-reduced but bit-identical for `hue_shift == 0 || 1e-30 <= |hue_shift| <= 360`
//             (finite): exact constant divisions (exact_math.hpp), rcp+residual quotients,
//             bounded-range fmod, sextant select as one LDS lookup + v_perm_b32.
//             Facts it relies on, all checked exhaustively over the 2^24 colours by the tests:
//             hue in [0,360) before the shift (so `hue % 360` is the identity), saturation and
//             value already inside [0,1], results of to_rgb inside [0,255.0001].
// The pixel filters are pure streaming kernels: 4 B read + 4 B written per pixel, HBM-bound
// once the VALU work per pixel is below ~80 instructions.
#include "internal.hpp"
#include "exact_math.hpp"
#include "hsv_device.hpp"

namespace mi355 {

// ---------------------------------------------------------------- hsvfilter kernels

// Four pixels of one lane through the filter. FAST variants: two calls of the pair routine with the block's HsvLds;
// GENERIC: the pair routine's settings-independent from_rgb (exact for every colour, all-colours tests) followed by the
// literal fmodf / `/ 60` / sextant chain, which is what non-finite and out-of-range hue shifts need.
template <int VARIANT, int RPOS, int GPOS, int BPOS, int NPOS>
__device__ __forceinline__ void hsvfilter_quad(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, const HsvK &k, const HsvLds *lds) {
  if constexpr (VARIANT >= 0) {
    hsvfilter_px2_lds<RPOS, GPOS, BPOS, NPOS, hsv_shift_of(VARIANT), hsv_sv_ident_of(VARIANT)>(a, b, k, lds);
    hsvfilter_px2_lds<RPOS, GPOS, BPOS, NPOS, hsv_shift_of(VARIANT), hsv_sv_ident_of(VARIANT)>(c, d, k, lds);
  } else {
    hsvfilter_px2_generic<RPOS, GPOS, BPOS, NPOS>(a, b, k, lds);
    hsvfilter_px2_generic<RPOS, GPOS, BPOS, NPOS>(c, d, k, lds);
  }
}

// Flat streaming kernel for 4-byte formats on contiguous storage (stride == width*4 and frames
// back to back): each lane owns 16 B (4 pixels) per iteration -> global_load/store_dwordx4.
// VARIANT: -1 = GENERIC arithmetic; otherwise FAST, see hsv_shift_of / hsv_sv_ident_of.
// Plain loads and stores on purpose: with the non-temporal hint on both, this kernel alone is 5 % faster (0.086 against
// 0.091 ms for the bare read-modify-write of 8 x 4K frames, tools/hsv_mem_probe.hip), but its output then bypasses the
// Infinity Cache and the element behind it reads from HBM: colorlut 0.115 instead of 0.087 ms, the chain 38.5 k instead of
// 44.4 k frames/s (r03, bench.py). One load in flight per lane: two measured the same or slower once the arithmetic had
// shrunk (63 VGPRs, 8 waves per SIMD cover the latency).
template <int VARIANT, int FIRST, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_flat_kernel(uint4 *__restrict__ data, size_t n_vec,
                                                             HsvK k, int nt) {
  constexpr int RPOS = FIRST + (BGR ? 2 : 0), GPOS = FIRST + 1, BPOS = FIRST + (BGR ? 0 : 2);
  constexpr int NPOS = FIRST == 0 ? 3 : 0;
  __shared__ HsvLds lds;
  hsv_lds_fill<RPOS, GPOS, BPOS, NPOS>(&lds);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (nt) {  // MI355_FLAG_HSV_NT (the A/B of bench.py: what the chain costs when this launch's output bypasses the Infinity Cache)
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t *d4 = (u32x4_t *)data;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
      const u32x4_t v = __builtin_nontemporal_load(d4 + i);
      uint4 p = {v.x, v.y, v.z, v.w};
      hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p.x, p.y, p.z, p.w, k, &lds);
      const u32x4_t o = {p.x, p.y, p.z, p.w};
      __builtin_nontemporal_store(o, d4 + i);
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    uint4 p = data[i];
    hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p.x, p.y, p.z, p.w, k, &lds);
    data[i] = p;
  }
}

// The flat kernel over up to kMultiFrames SEPARATE frames of one size (blockIdx.y = frame): the frames of different streams
// batched into one launch by the group dispatcher (group.hip), so that a frame does not pay the ramp and the tail of a launch
// of its own. Same arithmetic, same per-lane work.
template <int VARIANT, int FIRST, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_flat_multi_kernel(MultiFramePtrs frames, size_t n_vec, HsvK k) {
  constexpr int RPOS = FIRST + (BGR ? 2 : 0), GPOS = FIRST + 1, BPOS = FIRST + (BGR ? 0 : 2);
  constexpr int NPOS = FIRST == 0 ? 3 : 0;
  __shared__ HsvLds lds;
  hsv_lds_fill<RPOS, GPOS, BPOS, NPOS>(&lds);
  __syncthreads();
  uint4 *__restrict__ data = (uint4 *)frames.p[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    uint4 p = data[i];
    hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p.x, p.y, p.z, p.w, k, &lds);
    data[i] = p;
  }
}

// 4-byte formats with padded rows and / or frame pitches (what real caps negotiate when upstream aligns strides):
// the same 16 B per lane, addressed as (frame, row, 4-pixel group). Needs base, stride and pitch to be multiples of 16;
// the last group of a row may hold 1-3 pixels: its padding bytes are read, filtered as pixels and NOT written back
// (only `line[..width*4]` of a row is touched, hsvfilter/imp.rs:94-97).
template <int VARIANT, int FIRST, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_strided_kernel(uint8_t *__restrict__ data, int n_frames, size_t frame_pitch, int width,
                                                                int height, int stride, HsvK k) {
  constexpr int RPOS = FIRST + (BGR ? 2 : 0), GPOS = FIRST + 1, BPOS = FIRST + (BGR ? 0 : 2);
  constexpr int NPOS = FIRST == 0 ? 3 : 0;
  __shared__ HsvLds lds;
  hsv_lds_fill<RPOS, GPOS, BPOS, NPOS>(&lds);
  __syncthreads();
  const uint32_t groups = ((uint32_t)width + 3u) >> 2;  // per row
  const size_t rows = (size_t)height * (size_t)n_frames;
  const size_t total = rows * groups;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const size_t rowi = i / groups;
    const uint32_t g = (uint32_t)(i - rowi * groups);
    const size_t f = rowi / (size_t)height;
    const size_t row = rowi - f * (size_t)height;
    uint8_t *at = data + f * frame_pitch + row * (size_t)stride + (size_t)g * 16;
    const int left = width - (int)(g * 4);  // pixels of this group inside the row
    if (left >= 4) {
      uint4 p = *(const uint4 *)at;
      hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p.x, p.y, p.z, p.w, k, &lds);
      *(uint4 *)at = p;
    } else {
      // the row's bytes end inside this group: read only what belongs to the row
      uint32_t px[4] = {0, 0, 0, 0};
      for (int j = 0; j < left; j++) px[j] = ((const uint32_t *)at)[j];
      hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(px[0], px[1], px[2], px[3], k, &lds);
      for (int j = 0; j < left; j++) ((uint32_t *)at)[j] = px[j];
    }
  }
}

// The same for the common case of padded ROWS only (frame_pitch == stride * height: one tall picture of rows_total rows), without
// the two 64-bit divisions per lane and iteration of the kernel above (~200 instructions next to the 146 of the four pixels): a
// block is 64 lanes x 4 rows, blockIdx.x picks 64 groups of the row (1 KB contiguous per wave), blockIdx.y strides over the rows.
// Round 5: 65 % -> see profiles/r05_configs_elements.txt.
template <int VARIANT, int FIRST, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_rowpad_kernel(uint8_t *__restrict__ data, unsigned rows_total, int width, int stride, HsvK k) {
  constexpr int RPOS = FIRST + (BGR ? 2 : 0), GPOS = FIRST + 1, BPOS = FIRST + (BGR ? 0 : 2);
  constexpr int NPOS = FIRST == 0 ? 3 : 0;
  __shared__ HsvLds lds;
  hsv_lds_fill<RPOS, GPOS, BPOS, NPOS>(&lds);
  __syncthreads();
  const uint32_t groups = ((uint32_t)width + 3u) >> 2;
  const uint32_t g = blockIdx.x * 64u + (threadIdx.x & 63u);
  if (g >= groups) return;
  const int left = width - (int)(g * 4);  // pixels of this group inside the row
  for (uint32_t r = blockIdx.y * 4u + (threadIdx.x >> 6); r < rows_total; r += gridDim.y * 4u) {
    uint8_t *at = data + (size_t)r * (size_t)stride + (size_t)g * 16;
    if (left >= 4) {
      uint4 p = *(const uint4 *)at;
      hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p.x, p.y, p.z, p.w, k, &lds);
      *(uint4 *)at = p;
    } else {
      uint32_t px[4] = {0, 0, 0, 0};
      for (int j = 0; j < left; j++) px[j] = ((const uint32_t *)at)[j];
      hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(px[0], px[1], px[2], px[3], k, &lds);
      for (int j = 0; j < left; j++) ((uint32_t *)at)[j] = px[j];
    }
  }
}

// 3-byte formats (RGB / BGR) on contiguous storage: one lane = 12 B = 4 pixels. The three dwords are
// split into four pixel words with v_alignbit-style shifts, filtered by the same pair routine as the
// 4-byte formats (byte 3 of each word is scratch) and re-packed.
struct Rgb24x4 { uint32_t d0, d1, d2; };
template <int VARIANT, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_rgb24_kernel(Rgb24x4 *__restrict__ data, size_t n_grp, HsvK k) {
  constexpr int RPOS = BGR ? 2 : 0, GPOS = 1, BPOS = BGR ? 0 : 2, NPOS = 3;
  __shared__ HsvLds lds;
  hsv_lds_fill<RPOS, GPOS, BPOS, NPOS>(&lds);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_grp; i += stride) {
    const Rgb24x4 v = data[i];
    uint32_t p0 = v.d0, p1 = (v.d0 >> 24) | (v.d1 << 8), p2 = (v.d1 >> 16) | (v.d2 << 16), p3 = v.d2 >> 8;
    hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p0, p1, p2, p3, k, &lds);
    Rgb24x4 o;
    o.d0 = (p0 & 0x00ffffffu) | (p1 << 24);
    o.d1 = ((p1 >> 8) & 0x0000ffffu) | (p2 << 16);
    o.d2 = ((p2 >> 16) & 0x000000ffu) | (p3 << 8);
    data[i] = o;
  }
}

// The same formats with fully coalesced memory accesses (round 3; the 12-byte-per-lane form above moves 42-57 % of the
// HBM peak, its dwordx3 accesses leave every fourth dword slot of the memory pipeline empty): a wave takes 3 KB = 1024 pixels
// per iteration as three 16-byte loads per lane (lane l: bytes [16 l, 16 l + 16) of each 1 KB third), passes them through
// a wave-private 3 KB LDS strip so that lane l owns the 48 contiguous bytes of pixels 16 l .. 16 l + 15 (ds_read_b128 at a
// 48-byte lane stride: the 16 lanes of a pass start on 16 different 4-bank groups), filters them with the pair routine
// and sends them back the same way. The caller gives whole 3 KB chunks; the remainder goes to the kernel above.
template <int VARIANT, bool BGR>
__global__ __launch_bounds__(256) void hsvfilter_rgb24x16_kernel(uint4 *__restrict__ data, size_t n_chunks, HsvK k) {
  constexpr int RPOS = BGR ? 2 : 0, GPOS = 1, BPOS = BGR ? 0 : 2, NPOS = 3;
  __shared__ HsvLds lds;
  __shared__ uint4 strip[4][192];
  hsv_lds_fill<RPOS, GPOS, BPOS, NPOS>(&lds);
  __syncthreads();
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint4 *x = strip[wave];
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  const size_t n_waves = (size_t)gridDim.x * 4;
  size_t c = (size_t)blockIdx.x * 4 + wave;
  if (c >= n_chunks) return;
  uint4 a0 = data[c * 192 + lane], a1 = data[c * 192 + 64 + lane], a2 = data[c * 192 + 128 + lane];
  for (; c < n_chunks; c += n_waves) {
    uint4 *base = data + c * 192;
    // the next chunk's loads are in flight while this one is worked on (16 pixels per lane are ~600 VALU instructions:
    // without the prefetch a wave has nothing outstanding for four fifths of its time)
    const size_t cn = c + n_waves < n_chunks ? c + n_waves : c;
    const uint4 n0 = data[cn * 192 + lane], n1 = data[cn * 192 + 64 + lane], n2 = data[cn * 192 + 128 + lane];
    x[lane] = a0;
    x[64 + lane] = a1;
    x[128 + lane] = a2;
    wave_sync();
    uint4 v[3] = {x[3 * lane], x[3 * lane + 1], x[3 * lane + 2]};  // 12 dwords = 16 pixels
    wave_sync();
    uint32_t *d = (uint32_t *)v;
#pragma unroll
    for (int q = 0; q < 4; q++) {  // four pixels = three dwords at a time
      const uint32_t d0 = d[3 * q], d1 = d[3 * q + 1], d2 = d[3 * q + 2];
      uint32_t p0 = d0, p1 = (d0 >> 24) | (d1 << 8), p2 = (d1 >> 16) | (d2 << 16), p3 = d2 >> 8;
      hsvfilter_quad<VARIANT, RPOS, GPOS, BPOS, NPOS>(p0, p1, p2, p3, k, &lds);
      d[3 * q] = (p0 & 0x00ffffffu) | (p1 << 24);
      d[3 * q + 1] = ((p1 >> 8) & 0x0000ffffu) | (p2 << 16);
      d[3 * q + 2] = ((p2 >> 16) & 0x000000ffu) | (p3 << 8);
    }
    x[3 * lane] = v[0];
    x[3 * lane + 1] = v[1];
    x[3 * lane + 2] = v[2];
    wave_sync();
    base[lane] = x[lane];
    base[64 + lane] = x[64 + lane];
    base[128 + lane] = x[128 + lane];
    wave_sync();
    a0 = n0; a1 = n1; a2 = n2;
  }
}

template <int VARIANT>
static void launch_rgb24x16(mi355_ctx *ctx, uint4 *d, size_t n_chunks, const HsvK &k, int bgr, int grid) {
  if (bgr) hipLaunchKernelGGL((hsvfilter_rgb24x16_kernel<VARIANT, true>), dim3(grid), dim3(256), 0, ctx->stream, d, n_chunks, k);
  else hipLaunchKernelGGL((hsvfilter_rgb24x16_kernel<VARIANT, false>), dim3(grid), dim3(256), 0, ctx->stream, d, n_chunks, k);
}

template <int VARIANT>
static void launch_rgb24(mi355_ctx *ctx, Rgb24x4 *d, size_t n_grp, const HsvK &k, int bgr, int grid) {
  if (bgr) hipLaunchKernelGGL((hsvfilter_rgb24_kernel<VARIANT, true>), dim3(grid), dim3(256), 0, ctx->stream, d, n_grp, k);
  else hipLaunchKernelGGL((hsvfilter_rgb24_kernel<VARIANT, false>), dim3(grid), dim3(256), 0, ctx->stream, d, n_grp, k);
}

// General kernel: any stride / pixel stride (3 or 4) / triple offset / alignment, one pixel per lane, byte
// accesses. Only `line[..width*pixel_stride]` of each row is touched (hsvfilter/imp.rs:94-97).
template <bool FAST>
__global__ __launch_bounds__(256) void hsvfilter_rows_kernel(uint8_t *__restrict__ data, int n_frames,
                                                             size_t frame_pitch, int width, int height,
                                                             int stride, int pixel_stride, int first,
                                                             int bgr, HsvK k) {
  __shared__ uint32_t sel_tab[8];
  if (threadIdx.x < 7) sel_tab[threadIdx.x] = hsv_sel_entry(threadIdx.x, 0, 1, 2, 3);
  __syncthreads();
  const size_t per_frame = (size_t)width * (size_t)height;
  const size_t total = per_frame * (size_t)n_frames;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const size_t f = i / per_frame;
    const size_t r = i - f * per_frame;
    const size_t row = r / (size_t)width;
    const size_t col = r - row * (size_t)width;
    uint8_t *px = data + f * frame_pitch + row * (size_t)stride + col * (size_t)pixel_stride + first;
    const uint32_t c0 = px[0], c1 = px[1], c2 = px[2];
    const uint32_t p = bgr ? (c2 | (c1 << 8) | (c0 << 16)) : (c0 | (c1 << 8) | (c2 << 16));
    const uint32_t o = hsvfilter_px<FAST, 0, 1, 2, 3>(p, k, sel_tab);
    const uint8_t orr = (uint8_t)(o & 0xff), og = (uint8_t)((o >> 8) & 0xff), ob = (uint8_t)((o >> 16) & 0xff);
    px[0] = bgr ? ob : orr;
    px[1] = og;
    px[2] = bgr ? orr : ob;
  }
}

static int grid_for(mi355_ctx *ctx, size_t work_items, int block, int blocks_per_cu) {
  size_t blocks = (work_items + (size_t)block - 1) / (size_t)block;
  size_t cap = (size_t)ctx->n_cu * (size_t)blocks_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <int VARIANT>
static void launch_flat(mi355_ctx *ctx, uint4 *d, size_t n_vec, const HsvK &k, int first, int bgr, int grid) {
  dim3 g(grid), b(256);
  const int nt = ctx->hsv_nt;
  if (first == 0 && !bgr) hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 0, false>), g, b, 0, ctx->stream, d, n_vec, k, nt);
  else if (first == 0 && bgr) hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 0, true>), g, b, 0, ctx->stream, d, n_vec, k, nt);
  else if (first == 1 && !bgr) hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 1, false>), g, b, 0, ctx->stream, d, n_vec, k, nt);
  else hipLaunchKernelGGL((hsvfilter_flat_kernel<VARIANT, 1, true>), g, b, 0, ctx->stream, d, n_vec, k, nt);
}

template <int VARIANT>
static void launch_flat_multi(hipStream_t stream, const MultiFramePtrs &frames, int n_frames, size_t n_vec, const HsvK &k, int first, int bgr, int grid_x) {
  dim3 g(grid_x, n_frames), b(256);
  if (first == 0 && !bgr) hipLaunchKernelGGL((hsvfilter_flat_multi_kernel<VARIANT, 0, false>), g, b, 0, stream, frames, n_vec, k);
  else if (first == 0 && bgr) hipLaunchKernelGGL((hsvfilter_flat_multi_kernel<VARIANT, 0, true>), g, b, 0, stream, frames, n_vec, k);
  else if (first == 1 && !bgr) hipLaunchKernelGGL((hsvfilter_flat_multi_kernel<VARIANT, 1, false>), g, b, 0, stream, frames, n_vec, k);
  else hipLaunchKernelGGL((hsvfilter_flat_multi_kernel<VARIANT, 1, true>), g, b, 0, stream, frames, n_vec, k);
}

template <int VARIANT>
static void launch_strided(mi355_ctx *ctx, uint8_t *d, int n_frames, size_t frame_pitch, int width, int height, int stride, const HsvK &k,
                           int first, int bgr, int grid) {
  if ((n_frames == 1 || frame_pitch == (size_t)stride * (size_t)height) && (size_t)n_frames * (size_t)height < (1u << 31)) {
    // padded rows only: the batch is one tall picture
    const unsigned rows_total = (unsigned)n_frames * (unsigned)height, gx = ((unsigned)(width + 3) / 4 + 63) / 64;
    unsigned gy = (rows_total + 3) / 4;
    const unsigned cap = (unsigned)ctx->n_cu * (unsigned)ctx->hsv_blocks_per_cu / (gx ? gx : 1) + 1;
    if (gy > cap) gy = cap;
    if (gy > 65535) gy = 65535;
    dim3 g2(gx, gy), b2(256);
    if (first == 0 && !bgr) hipLaunchKernelGGL((hsvfilter_rowpad_kernel<VARIANT, 0, false>), g2, b2, 0, ctx->stream, d, rows_total, width, stride, k);
    else if (first == 0 && bgr) hipLaunchKernelGGL((hsvfilter_rowpad_kernel<VARIANT, 0, true>), g2, b2, 0, ctx->stream, d, rows_total, width, stride, k);
    else if (first == 1 && !bgr) hipLaunchKernelGGL((hsvfilter_rowpad_kernel<VARIANT, 1, false>), g2, b2, 0, ctx->stream, d, rows_total, width, stride, k);
    else hipLaunchKernelGGL((hsvfilter_rowpad_kernel<VARIANT, 1, true>), g2, b2, 0, ctx->stream, d, rows_total, width, stride, k);
    return;
  }
  dim3 g(grid), b(256);
  if (first == 0 && !bgr) hipLaunchKernelGGL((hsvfilter_strided_kernel<VARIANT, 0, false>), g, b, 0, ctx->stream, d, n_frames, frame_pitch, width, height, stride, k);
  else if (first == 0 && bgr) hipLaunchKernelGGL((hsvfilter_strided_kernel<VARIANT, 0, true>), g, b, 0, ctx->stream, d, n_frames, frame_pitch, width, height, stride, k);
  else if (first == 1 && !bgr) hipLaunchKernelGGL((hsvfilter_strided_kernel<VARIANT, 1, false>), g, b, 0, ctx->stream, d, n_frames, frame_pitch, width, height, stride, k);
  else hipLaunchKernelGGL((hsvfilter_strided_kernel<VARIANT, 1, true>), g, b, 0, ctx->stream, d, n_frames, frame_pitch, width, height, stride, k);
}

// Calls F<VARIANT>(args...) for the run-time variant number (hsv_variant_for).
#define MI355_HSV_VARIANT_SWITCH(variant, F, ...)   \
  switch (variant) {                                \
    case -1: F<-1>(__VA_ARGS__); break;             \
    case 0: F<0>(__VA_ARGS__); break;               \
    case 1: F<1>(__VA_ARGS__); break;               \
    case 2: F<2>(__VA_ARGS__); break;               \
    case 4: F<4>(__VA_ARGS__); break;               \
    case 5: F<5>(__VA_ARGS__); break;               \
    case 6: F<6>(__VA_ARGS__); break;               \
    case 8: F<8>(__VA_ARGS__); break;               \
    case 9: F<9>(__VA_ARGS__); break;               \
    case 12: F<12>(__VA_ARGS__); break;             \
    default: F<13>(__VA_ARGS__); break;             \
  }

int launch_hsvfilter_compute(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width,
                     int height, int stride, const PixFmt &fmt, const mi355_hsv_settings &s) {
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;  // nothing to do
  const HsvK k{s.hue_shift, s.saturation_mul, s.saturation_off, s.value_mul, s.value_off};
  const int variant = hsv_variant_for(s, ctx->force_generic, true);
  const size_t row_bytes = (size_t)width * (size_t)fmt.pixel_stride;
  const bool packed_rows = (size_t)stride == row_bytes && (n_frames == 1 || frame_pitch == row_bytes * (size_t)height);
  const size_t total_bytes = row_bytes * (size_t)height * (size_t)n_frames;
  if (fmt.pixel_stride == 4 && packed_rows && ((uintptr_t)d_data % 16 == 0) && (total_bytes % 16 == 0)) {
    const size_t n_vec = total_bytes / 16;
    // at least two 16-byte groups per lane: a block pays its LDS tables and its ramp once, and launches of one or two frames
    // (one stream's buffer) are 10-15 % faster than with one group per lane (tools/r03_single_frame_launch.py)
    const int grid = grid_for(ctx, (n_vec + 1) / 2, 256, ctx->hsv_blocks_per_cu);
    uint4 *d = (uint4 *)d_data;
    MI355_HSV_VARIANT_SWITCH(variant, launch_flat, ctx, d, n_vec, k, fmt.first, fmt.bgr, grid)
  } else if (fmt.pixel_stride == 4 && ((uintptr_t)d_data % 16 == 0) && stride % 16 == 0 && (n_frames == 1 || frame_pitch % 16 == 0)) {
    const size_t n_grp = (size_t)((width + 3) / 4) * (size_t)height * (size_t)n_frames;
    const int grid = grid_for(ctx, (n_grp + 1) / 2, 256, ctx->hsv_blocks_per_cu);
    MI355_HSV_VARIANT_SWITCH(variant, launch_strided, ctx, d_data, n_frames, frame_pitch, width, height, stride, k, fmt.first, fmt.bgr, grid)
  } else if (fmt.pixel_stride == 3 && fmt.first == 0 && packed_rows && ((uintptr_t)d_data % 4 == 0) && (total_bytes % 12 == 0)) {
    // whole 3 KB chunks (1024 pixels) through the coalescing kernel when the base is 16-byte aligned, the rest 12 bytes per lane
    const size_t n_chunks = ((uintptr_t)d_data % 16 == 0) ? total_bytes / 3072 : 0;
    if (n_chunks) {
      const int grid = grid_for(ctx, n_chunks * 64, 256, ctx->hsv_blocks_per_cu / 2);
      MI355_HSV_VARIANT_SWITCH(variant, launch_rgb24x16, ctx, (uint4 *)d_data, n_chunks, k, fmt.bgr, grid)
    }
    const size_t n_grp = (total_bytes - n_chunks * 3072) / 12;
    if (n_grp) {
      const int grid = grid_for(ctx, n_grp, 256, ctx->hsv_blocks_per_cu);
      Rgb24x4 *d = (Rgb24x4 *)(d_data + n_chunks * 3072);
      MI355_HSV_VARIANT_SWITCH(variant, launch_rgb24, ctx, d, n_grp, k, fmt.bgr, grid)
    }
  } else {
    const size_t total = (size_t)width * (size_t)height * (size_t)n_frames;
    const int grid = grid_for(ctx, total, 256, 32);
    // one pixel per lane: the single-pixel FAST routine knows the narrow shift classes only
    if (hsv_variant_for(s, ctx->force_generic, false) >= 0)
      hipLaunchKernelGGL((hsvfilter_rows_kernel<true>), dim3(grid), dim3(256), 0, ctx->stream, d_data, n_frames,
                         frame_pitch, width, height, stride, fmt.pixel_stride, fmt.first, fmt.bgr, k);
    else
      hipLaunchKernelGGL((hsvfilter_rows_kernel<false>), dim3(grid), dim3(256), 0, ctx->stream, d_data, n_frames,
                         frame_pitch, width, height, stride, fmt.pixel_stride, fmt.first, fmt.bgr, k);
  }
  return check_hip(ctx, hipGetLastError(), "hsvfilter kernel launch");
}

// hsvfilter in place on n separate packed frames of one size and format with one set of settings, ONE launch on `stream`
// (group.hip). false if the geometry is not the flat kernel's (the caller then goes frame by frame through launch_hsvfilter).
bool hsvfilter_multi_applicable(const uint8_t *const *frames, int n_frames, int width, int height, int stride, const PixFmt &fmt) {
  if (n_frames < 1 || n_frames > kMultiFrames || width <= 0 || height <= 0 || fmt.pixel_stride != 4 || (size_t)stride != (size_t)width * 4) return false;
  if (((size_t)width * 4 * (size_t)height) % 16 != 0) return false;
  for (int f = 0; f < n_frames; f++)
    if (!frames[f] || (uintptr_t)frames[f] % 16 != 0) return false;
  return true;
}
int launch_hsvfilter_multi(mi355_ctx *ctx, hipStream_t stream, uint8_t *const *frames, int n_frames, int width, int height, const PixFmt &fmt,
                           const mi355_hsv_settings &s) {
  const HsvK k{s.hue_shift, s.saturation_mul, s.saturation_off, s.value_mul, s.value_off};
  const int variant = hsv_variant_for(s, ctx->force_generic, true);
  const size_t n_vec = (size_t)width * 4 * (size_t)height / 16;
  MultiFramePtrs ptrs{};
  for (int f = 0; f < n_frames; f++) ptrs.p[f] = frames[f];
  // the same total grid as one launch over n contiguous frames would get, split evenly over the frames
  int gx = grid_for(ctx, ((size_t)n_frames * n_vec + 1) / 2, 256, ctx->hsv_blocks_per_cu) / n_frames;
  if (gx < 1) gx = 1;
  MI355_HSV_VARIANT_SWITCH(variant, launch_flat_multi, stream, ptrs, n_frames, n_vec, k, fmt.first, fmt.bgr, gx)
  return check_hip(ctx, hipGetLastError(), "hsvfilter multi-frame kernel launch");
}

// ---------------------------------------------------------------- hsvdetector

struct HsvDetK {
  float hue_ref, hue_var, sat_ref, sat_var, val_ref, val_var;
};

// One pixel per lane; input 3 or 4 bytes/pixel, output always 4 (hsvdetector/imp.rs:118-156).
// Arithmetic is the GENERIC conversion (IEEE ops) — the detector is a "next" row.
__global__ __launch_bounds__(256) void hsvdetect_rows_kernel(const uint8_t *__restrict__ src, size_t src_pitch,
                                                             int src_stride, int in_pixel_stride, int in_first,
                                                             int in_bgr, uint8_t *__restrict__ dst, size_t dst_pitch,
                                                             int dst_stride, int out_alpha_first, int out_bgr,
                                                             int n_frames, int width, int height, HsvDetK k) {
  const size_t per_frame = (size_t)width * (size_t)height;
  const size_t total = per_frame * (size_t)n_frames;
  const size_t gstride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const size_t f = i / per_frame;
    const size_t r = i - f * per_frame;
    const size_t row = r / (size_t)width;
    const size_t col = r - row * (size_t)width;
    const uint8_t *ip = src + f * src_pitch + row * (size_t)src_stride + col * (size_t)in_pixel_stride + in_first;
    uint8_t *op = dst + f * dst_pitch + row * (size_t)dst_stride + col * 4;
    const uint32_t c0 = ip[0], c1 = ip[1], c2 = ip[2];
    const uint32_t r8 = in_bgr ? c2 : c0, g8 = c1, b8 = in_bgr ? c0 : c2;
    float h, s, v;
    hsv_from_rgb<false>((float)r8, (float)g8, (float)b8, h, s, v);
    const float ref_hue_offset = 180.0f - k.hue_ref;
    float sh = h + ref_hue_offset;
    if (sh < 0.0f) sh += 360.0f;
    sh = fmodf(sh, 360.0f);
    const bool hit = fabsf(sh - 180.0f) <= k.hue_var && fabsf(s - k.sat_ref) <= k.sat_var &&
                     fabsf(v - k.val_ref) <= k.val_var;
    const uint8_t alpha = hit ? 255 : 0;
    uint8_t *c = op + (out_alpha_first ? 1 : 0);
    if (out_bgr) { c[0] = (uint8_t)b8; c[1] = (uint8_t)g8; c[2] = (uint8_t)r8; }
    else { c[0] = (uint8_t)r8; c[1] = (uint8_t)g8; c[2] = (uint8_t)b8; }
    op[out_alpha_first ? 0 : 3] = alpha;
  }
}

// FAST detector kernel: 4-byte input formats on contiguous storage, 4 pixels per lane. Two classes of the offset
// off = 180 - hue_ref (hsvdetector/imp.rs:140-144: shifted = hue + off; if shifted < 0 { shifted += 360 }; shifted %= 360):
//   NEG = false, off in [0, 360] (hue_ref in [-180, 180]): shifted lies in [0, 720), the `< 0` test never fires and `% 360`
//         is one conditional exact subtraction;
//   NEG = true, off in [-360, 0) (hue_ref in (180, 540], i.e. every hue on the usual 0..360 dial): shifted lies in
//         [-360, 360); negative values get + 360 (one rounding, as in the reference) and land in [0, 360], where `% 360` only
//         turns an exact 360 into 0; non-negative values are below 360 and `% 360` keeps them.
__device__ __forceinline__ f2 hsvdetect_shifted(f2 h, float off, bool neg) {
  const f2 t = h + splat2(off);
  return sub360_if_reached2(neg ? add360_if_negative2(t) : t);
}

template <int IN_FIRST, bool IN_BGR, bool NEG = false>
__global__ __launch_bounds__(256) void hsvdetect_flat_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_vec,
                                                             HsvDetK k, uint32_t out_sel) {
  constexpr int RPOS = IN_FIRST + (IN_BGR ? 2 : 0), GPOS = IN_FIRST + 1, BPOS = IN_FIRST + (IN_BGR ? 0 : 2);
  const float off = 180.0f - k.hue_ref;  // ref_hue_offset (hsvdetector/imp.rs:140)
  __shared__ HsvLds lds;  // value / reciprocal tables of the pair routine (the sextant entries are not used here)
  hsv_lds_fill<0, 1, 2, 3>(&lds);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const uint4 p = src[i];
    const uint32_t in[4] = {p.x, p.y, p.z, p.w};
    uint32_t out[4];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      f2 h, s, v;
      hsv_from_rgb_pair_fast<RPOS, GPOS, BPOS, 1>(in[j], in[j + 1], h, s, v, &lds);
      const f2 sh = hsvdetect_shifted(h, off, NEG);
      const f2 dh = sh - splat2(180.0f), dsat = s - splat2(k.sat_ref), dv = v - splat2(k.val_ref);
      const bool hit0 = fabsf(dh.x) <= k.hue_var && fabsf(dsat.x) <= k.sat_var && fabsf(dv.x) <= k.val_var;
      const bool hit1 = fabsf(dh.y) <= k.hue_var && fabsf(dsat.y) <= k.sat_var && fabsf(dv.y) <= k.val_var;
      // out_sel picks the colour bytes from the input pixel (selector 4..7) and the alpha byte from the
      // second operand (selector 0)
      out[j] = __builtin_amdgcn_perm(in[j], hit0 ? 255u : 0u, out_sel);
      out[j + 1] = __builtin_amdgcn_perm(in[j + 1], hit1 ? 255u : 0u, out_sel);
    }
    dst[i] = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

// Same for the 3-byte input formats (RGB / BGR): one lane reads 12 B = 4 pixels as three dwords, splits them into four
// pixel words (byte 3 = scratch) and writes 16 B.
template <bool IN_BGR, bool NEG = false>
__global__ __launch_bounds__(256) void hsvdetect_rgb24_kernel(const Rgb24x4 *__restrict__ src, uint4 *__restrict__ dst, size_t n_grp, HsvDetK k,
                                                              uint32_t out_sel) {
  constexpr int RPOS = IN_BGR ? 2 : 0, GPOS = 1, BPOS = IN_BGR ? 0 : 2;
  const float off = 180.0f - k.hue_ref;
  __shared__ HsvLds lds;
  hsv_lds_fill<0, 1, 2, 3>(&lds);
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_grp; i += stride) {
    const Rgb24x4 v = src[i];
    const uint32_t in[4] = {v.d0, (v.d0 >> 24) | (v.d1 << 8), (v.d1 >> 16) | (v.d2 << 16), v.d2 >> 8};
    uint32_t out[4];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      f2 h, s, vv;
      hsv_from_rgb_pair_fast<RPOS, GPOS, BPOS, 1>(in[j], in[j + 1], h, s, vv, &lds);
      const f2 sh = hsvdetect_shifted(h, off, NEG);
      const f2 dh = sh - splat2(180.0f), dsat = s - splat2(k.sat_ref), dv = vv - splat2(k.val_ref);
      const bool hit0 = fabsf(dh.x) <= k.hue_var && fabsf(dsat.x) <= k.sat_var && fabsf(dv.x) <= k.val_var;
      const bool hit1 = fabsf(dh.y) <= k.hue_var && fabsf(dsat.y) <= k.sat_var && fabsf(dv.y) <= k.val_var;
      out[j] = __builtin_amdgcn_perm(in[j], hit0 ? 255u : 0u, out_sel);
      out[j + 1] = __builtin_amdgcn_perm(in[j + 1], hit1 ? 255u : 0u, out_sel);
    }
    dst[i] = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

int launch_hsvdetect(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride,
                     const PixFmt &sfmt, uint8_t *d_dst, size_t dst_pitch, int dst_stride,
                     int dst_alpha_first, int dst_bgr, int n_frames, int width, int height,
                     const mi355_hsvdetect_settings &s) {
  if (n_frames <= 0 || width <= 0 || height <= 0) return MI355_OK;
  const HsvDetK k{s.hue_ref, s.hue_var, s.saturation_ref, s.saturation_var, s.value_ref, s.value_var};
  const size_t total = (size_t)width * (size_t)height * (size_t)n_frames;
  const size_t row_bytes = (size_t)width * 4;
  const bool contiguous = sfmt.pixel_stride == 4 && (size_t)src_stride == row_bytes && (size_t)dst_stride == row_bytes &&
                          (n_frames == 1 || (src_pitch == row_bytes * (size_t)height && dst_pitch == row_bytes * (size_t)height));
  const float off = 180.0f - s.hue_ref;   // as the kernels (and the reference) compute it
  const bool fast = !ctx->force_generic && off >= -360.0f && off <= 360.0f;
  const bool neg = off < 0.0f;
  if (fast && contiguous && ((uintptr_t)d_src % 16 == 0) && ((uintptr_t)d_dst % 16 == 0) && ((total * 4) % 16 == 0)) {
    // output byte j: colour channel c sits at input byte in_pos(c); alpha comes from operand 1 byte 0
    const int in_pos[3] = {sfmt.first + (sfmt.bgr ? 2 : 0), sfmt.first + 1, sfmt.first + (sfmt.bgr ? 0 : 2)};  // R,G,B
    const int cbase = dst_alpha_first ? 1 : 0;
    uint32_t sel = 0;  // alpha byte selector stays 0 (= byte 0 of the 0/255 operand)
    for (int c = 0; c < 3; c++) {
      const int out_byte = cbase + (dst_bgr ? 2 - c : c);
      sel |= (uint32_t)(4 + in_pos[c]) << (8 * out_byte);
    }
    const size_t n_vec = total / 4;
    const int fgrid = grid_for(ctx, (n_vec + 1) / 2, 256, 64);   // two groups per lane, as the hsvfilter kernels (one-frame launches)
    dim3 g(fgrid), b(256);
    const uint4 *sp = (const uint4 *)d_src;
    uint4 *dp = (uint4 *)d_dst;
#define MI355_DET_LAUNCH(F, B)                                                                                              \
    do {                                                                                                                    \
      if (neg) hipLaunchKernelGGL((hsvdetect_flat_kernel<F, B, true>), g, b, 0, ctx->stream, sp, dp, n_vec, k, sel);       \
      else hipLaunchKernelGGL((hsvdetect_flat_kernel<F, B, false>), g, b, 0, ctx->stream, sp, dp, n_vec, k, sel);          \
    } while (0)
    if (sfmt.first == 0 && !sfmt.bgr) MI355_DET_LAUNCH(0, false);
    else if (sfmt.first == 0 && sfmt.bgr) MI355_DET_LAUNCH(0, true);
    else if (sfmt.first == 1 && !sfmt.bgr) MI355_DET_LAUNCH(1, false);
    else MI355_DET_LAUNCH(1, true);
#undef MI355_DET_LAUNCH
    return check_hip(ctx, hipGetLastError(), "hsvdetect flat kernel launch");
  }
  const size_t row3 = (size_t)width * 3;
  const bool contiguous3 = sfmt.pixel_stride == 3 && sfmt.first == 0 && (size_t)src_stride == row3 && (size_t)dst_stride == row_bytes &&
                           (n_frames == 1 || (src_pitch == row3 * (size_t)height && dst_pitch == row_bytes * (size_t)height));
  if (fast && contiguous3 && ((uintptr_t)d_src % 4 == 0) && ((uintptr_t)d_dst % 16 == 0) && (total % 4 == 0)) {
    const int in_pos[3] = {sfmt.bgr ? 2 : 0, 1, sfmt.bgr ? 0 : 2};
    const int cbase = dst_alpha_first ? 1 : 0;
    uint32_t sel = 0;
    for (int c = 0; c < 3; c++) {
      const int out_byte = cbase + (dst_bgr ? 2 - c : c);
      sel |= (uint32_t)(4 + in_pos[c]) << (8 * out_byte);
    }
    const size_t n_grp = total / 4;
    const int fgrid = grid_for(ctx, n_grp, 256, 64);
    if (sfmt.bgr && neg) hipLaunchKernelGGL((hsvdetect_rgb24_kernel<true, true>), dim3(fgrid), dim3(256), 0, ctx->stream, (const Rgb24x4 *)d_src, (uint4 *)d_dst, n_grp, k, sel);
    else if (sfmt.bgr) hipLaunchKernelGGL((hsvdetect_rgb24_kernel<true, false>), dim3(fgrid), dim3(256), 0, ctx->stream, (const Rgb24x4 *)d_src, (uint4 *)d_dst, n_grp, k, sel);
    else if (neg) hipLaunchKernelGGL((hsvdetect_rgb24_kernel<false, true>), dim3(fgrid), dim3(256), 0, ctx->stream, (const Rgb24x4 *)d_src, (uint4 *)d_dst, n_grp, k, sel);
    else hipLaunchKernelGGL((hsvdetect_rgb24_kernel<false, false>), dim3(fgrid), dim3(256), 0, ctx->stream, (const Rgb24x4 *)d_src, (uint4 *)d_dst, n_grp, k, sel);
    return check_hip(ctx, hipGetLastError(), "hsvdetect rgb24 kernel launch");
  }
  const int grid = grid_for(ctx, total, 256, 32);
  hipLaunchKernelGGL(hsvdetect_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_src, src_pitch, src_stride,
                     sfmt.pixel_stride, sfmt.first, sfmt.bgr, d_dst, dst_pitch, dst_stride, dst_alpha_first, dst_bgr,
                     n_frames, width, height, k);
  return check_hip(ctx, hipGetLastError(), "hsvdetect kernel launch");
}

}  // namespace mi355
