// videocompare_kernels.hip — gfx950 kernels for videocompare's default hash (BASELINE config 5 shape:
// many concurrent 4K streams, one comparison per stream per frame).
//
// Reference path replaced: HasherEngine::hash_image / compare (video/videofx/src/videocompare/hashed_image.rs:24-79)
// with HashAlgorithm::Blockhash, the element default (videocompare/imp.rs:31) -> crate image_hasher 3.1.1
// `blockhash` (sources not in the reference tree; algorithm restated in oracle/videocompare_oracle.c):
//   64 u32 block sums over (width/8) x (height/8) pixel blocks of sum_px (r+g+b; RGBA with a == 0 counts 765),
//   two bands of 32 blocks, upper median per band, bit = block > median || (block == median && median > half scale),
//   distance = Hamming distance of the 64-bit hashes.
// The frame is read once (4 B/px RGBA, 3 B/px RGB) and nothing but 64 counters is written: a pure HBM-bound
// reduction. Lanes of a wavefront that fall into the same hash block (the common case: a block is hundreds of
// pixels wide) are combined with a cross-lane shuffle reduction before one LDS atomic; the 64 LDS counters of
// a workgroup are flushed with one global atomic each. Integer adds: the result is order-independent, bit-exact.
#include "internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace mi355 {

__device__ __forceinline__ uint32_t sum_px_rgba(uint32_t p) {
  const uint32_t s = (p & 0xffu) + ((p >> 8) & 0xffu) + ((p >> 16) & 0xffu);
  return (p >> 24) == 0u ? 765u : s;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// add `v` to LDS counter `bin`; lanes of the wave sharing one bin are reduced first
__device__ __forceinline__ void bin_add(uint32_t *s_bins, int bin, uint32_t v) {
  const int first = __shfl(bin, 0);
  if (__all(bin == first)) {
    const uint32_t t = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && bin >= 0) atomicAdd(&s_bins[bin], t);
  } else if (bin >= 0) {
    atomicAdd(&s_bins[bin], v);
  }
}

// per-lane add (no wave-collective part): used for the rare second block column of a straddling pixel group
__device__ __forceinline__ void bin_add_lane(uint32_t *s_bins, int bin, uint32_t v) {
  if (bin >= 0 && v) atomicAdd(&s_bins[bin], v);
}

// RGBA, rows 16 B aligned, width % 4 == 0, block width >= 4: one lane owns a 4-pixel column group (uint4) over a
// group of kRowGroup rows, so its pixels stay in the same hash-block column; the row loop keeps several independent
// 16 B loads in flight and the cross-lane reduction happens once per row group instead of once per row.
// Work item = (frame, row group, 1024-pixel segment).
constexpr int kRowGroup = 32;
__global__ __launch_bounds__(256) void blockhash_rgba_kernel(const uint8_t *__restrict__ frames, size_t frame_pitch, int stride,
                                                             int n_frames, int width, int height, uint32_t *__restrict__ sums) {
  __shared__ uint32_t s_bins[64];
  const int bw = width / 8, bh = height / 8;
  const int segs = (width + 1023) / 1024;
  const int groups = (height + kRowGroup - 1) / kRowGroup;
  const size_t items_per_frame = (size_t)groups * segs;
  const size_t total = items_per_frame * (size_t)n_frames;
  // contiguous range of items per workgroup so that it flushes its LDS counters once per frame it touches
  const size_t per_wg = (total + gridDim.x - 1) / gridDim.x;
  size_t it = (size_t)blockIdx.x * per_wg;
  const size_t it_end = it + per_wg < total ? it + per_wg : total;
  int cur_frame = -1;
  if (threadIdx.x < 64) s_bins[threadIdx.x] = 0;
  __syncthreads();
  for (; it < it_end; it++) {
    const int f = (int)(it / items_per_frame);
    if (f != cur_frame) {
      if (cur_frame >= 0) {
        __syncthreads();
        if (threadIdx.x < 64) { const uint32_t v = s_bins[threadIdx.x]; if (v) atomicAdd(&sums[(size_t)cur_frame * 64 + threadIdx.x], v); s_bins[threadIdx.x] = 0; }
        __syncthreads();
      }
      cur_frame = f;
    }
    const size_t r = it - (size_t)f * items_per_frame;
    const int g = (int)(r / segs), seg = (int)(r - (size_t)g * segs);
    const int x0 = seg * 1024 + (int)threadIdx.x * 4;
    const bool live = x0 < width;
    // the 4 pixels cover at most two block columns (bw >= 4): column of pixel 0 and of pixel 3
    const int bxa = live ? x0 / bw : 0, bxb = live ? (x0 + 3) / bw : 0;
    const int split = (bxa + 1) * bw - x0;  // pixels [0, split) belong to bxa (split >= 4 when bxa == bxb)
    const uint8_t *col = frames + (size_t)f * frame_pitch + (size_t)x0 * 4;
    const int y0 = g * kRowGroup, y1 = y0 + kRowGroup < height ? y0 + kRowGroup : height;
    uint32_t acca = 0, accb = 0;
    int by = y0 / bh;
    for (int yb = y0; yb < y1; yb += 8) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        v[k] = make_uint4(0xff000000u, 0xff000000u, 0xff000000u, 0xff000000u);  // opaque black: adds 0
        if (live && yb + k < y1) v[k] = *(const uint4 *)(col + (size_t)(yb + k) * (size_t)stride);
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int y = yb + k;
        if (y < y1 && y / bh != by) {  // row group crosses a block-row edge: flush (wave-uniform condition)
          bin_add(s_bins, live ? by * 8 + bxa : -1, acca);
          if (bxb != bxa) bin_add_lane(s_bins, live ? by * 8 + bxb : -1, accb);
          acca = accb = 0;
          by = y / bh;
        }
        const uint32_t s0 = sum_px_rgba(v[k].x), s1 = sum_px_rgba(v[k].y), s2 = sum_px_rgba(v[k].z), s3 = sum_px_rgba(v[k].w);
        acca += s0 + (split > 1 ? s1 : 0u) + (split > 2 ? s2 : 0u) + (split > 3 ? s3 : 0u);
        accb += (split > 1 ? 0u : s1) + (split > 2 ? 0u : s2) + (split > 3 ? 0u : s3);
      }
    }
    bin_add(s_bins, live ? by * 8 + bxa : -1, acca);
    if (bxb != bxa) bin_add_lane(s_bins, live ? by * 8 + bxb : -1, accb);
  }
  __syncthreads();
  if (cur_frame >= 0 && threadIdx.x < 64) { const uint32_t v = s_bins[threadIdx.x]; if (v) atomicAdd(&sums[(size_t)cur_frame * 64 + threadIdx.x], v); }
}

// Any packed layout (RGB, unaligned RGBA): one lane = one pixel, byte loads.
__global__ __launch_bounds__(256) void blockhash_bytes_kernel(const uint8_t *__restrict__ frames, size_t frame_pitch, int stride,
                                                              int n_frames, int width, int height, int channels,
                                                              uint32_t *__restrict__ sums) {
  __shared__ uint32_t s_bins[64];
  const int bw = width / 8, bh = height / 8;
  const int segs = (width + 255) / 256;
  const size_t items_per_frame = (size_t)height * segs;
  const size_t total = items_per_frame * (size_t)n_frames;
  const size_t per_wg = (total + gridDim.x - 1) / gridDim.x;
  size_t it = (size_t)blockIdx.x * per_wg;
  const size_t it_end = it + per_wg < total ? it + per_wg : total;
  int cur_frame = -1;
  if (threadIdx.x < 64) s_bins[threadIdx.x] = 0;
  __syncthreads();
  for (; it < it_end; it++) {
    const int f = (int)(it / items_per_frame);
    if (f != cur_frame) {
      if (cur_frame >= 0) {
        __syncthreads();
        if (threadIdx.x < 64) { const uint32_t v = s_bins[threadIdx.x]; if (v) atomicAdd(&sums[(size_t)cur_frame * 64 + threadIdx.x], v); s_bins[threadIdx.x] = 0; }
        __syncthreads();
      }
      cur_frame = f;
    }
    const size_t r = it - (size_t)f * items_per_frame;
    const int y = (int)(r / segs), seg = (int)(r - (size_t)y * segs);
    const int x = seg * 256 + (int)threadIdx.x;
    uint32_t acc = 0;
    int bin = -1;
    if (x < width) {
      const uint8_t *p = frames + (size_t)f * frame_pitch + (size_t)y * (size_t)stride + (size_t)x * channels;
      acc = (uint32_t)p[0] + p[1] + p[2];
      if (channels == 4 && p[3] == 0) acc = 765u;
      bin = (y / bh) * 8 + x / bw;
    }
    bin_add(s_bins, bin, acc);
  }
  __syncthreads();
  if (cur_frame >= 0 && threadIdx.x < 64) { const uint32_t v = s_bins[threadIdx.x]; if (v) atomicAdd(&sums[(size_t)cur_frame * 64 + threadIdx.x], v); }
}

// one wave per frame: lanes 0..31 own band 0, 32..63 band 1; rank-based upper median, then the bits
__global__ __launch_bounds__(64) void blockhash_finish_kernel(const uint32_t *__restrict__ sums, int n_frames, uint32_t cmp_factor,
                                                              unsigned long long *__restrict__ hashes) {
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const int lane = threadIdx.x, band = lane >> 5;
  const uint32_t v = sums[(size_t)f * 64 + lane];
  // rank = number of band elements smaller than v (ties broken by lane) -> the element of rank 16 is sorted[16]
  int rank = 0;
  for (int j = 0; j < 32; j++) {
    const int src = band * 32 + j;
    const uint32_t o = __shfl(v, src);
    rank += (o < v) || (o == v && src < lane);
  }
  const unsigned long long is_med = __ballot(rank == 16);
  const int med_lane0 = __ffsll((long long)(is_med & 0xffffffffull)) - 1;
  const int med_lane1 = __ffsll((long long)(is_med >> 32)) - 1 + 32;
  const uint32_t median = __shfl(v, band ? med_lane1 : med_lane0);
  const bool bit = v > median || (v == median && median > cmp_factor);
  const unsigned long long h = __ballot(bit);
  if (lane == 0) hashes[f] = h;
}

// The crate's floating-point path for frames that do not divide into 8 x 8 whole blocks (blockhash_slow; restated in
// oracle/videocompare_oracle.c, parity unpinned like the rest of the hash). Every pixel goes whole to block
// (floor(x / (w/8)), floor(y / (h/8))) - f32 quotients - and a block's f32 sum is accumulated in PIXEL ORDER, which decides its
// low bits once it passes 2^24. So one lane owns one block and walks its pixels row by row: 64 lanes per frame, no reduction
// tree, no atomics - sequential by definition, 1-2 ms per 4K frame. Off by default (MI355_FLAG_BLOCKHASH_ANY_SIZE).
__global__ __launch_bounds__(64) void blockhash_slow_kernel(const uint8_t *__restrict__ frames, size_t frame_pitch, int stride, int width, int height,
                                                            int channels, float *__restrict__ sums) {
  const uint8_t *frame = frames + (size_t)blockIdx.x * frame_pitch;
  const int b = threadIdx.x, bx = b & 7, by = b >> 3;
  const float bw = (float)width / 8.0f, bh = (float)height / 8.0f;
  // this block's pixel range: the x with floorf(x / bw) == bx are contiguous; found with the very expression that assigns them
  int x0 = width, x1 = 0, y0 = height, y1 = 0;
  for (int x = 0; x < width; x++)
    if ((int)floorf((float)x / bw) == bx) { x0 = x < x0 ? x : x0; x1 = x + 1; }
  for (int y = 0; y < height; y++)
    if ((int)floorf((float)y / bh) == by) { y0 = y < y0 ? y : y0; y1 = y + 1; }
  float acc = 0.0f;
  for (int y = y0; y < y1; y++) {
    const uint8_t *row = frame + (size_t)y * (size_t)stride;
    for (int x = x0; x < x1; x++) {
      const uint8_t *p = row + (size_t)x * channels;
      uint32_t s = (uint32_t)p[0] + p[1] + p[2];
      if (channels == 4 && p[3] == 0) s = 765u;
      acc += (float)s;
    }
  }
  sums[(size_t)blockIdx.x * 64 + b] = acc;
}

static int launch_blockhash_slow(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch, int stride, int n_frames, int width, int height, int channels,
                                 unsigned long long *hashes) {
  const size_t need = (size_t)n_frames * 64 * sizeof(float);
  if (ctx->d_stage_bytes[1] < need) {
    if (ctx->d_stage[1]) (void)hipFree(ctx->d_stage[1]);
    ctx->d_stage[1] = nullptr;
    ctx->d_stage_bytes[1] = 0;
    int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[1], need), "hipMalloc(blockhash scratch)");
    if (rc) return rc;
    ctx->d_stage_bytes[1] = need;
  }
  float *d_sums = (float *)ctx->d_stage[1];
  hipLaunchKernelGGL(blockhash_slow_kernel, dim3(n_frames), dim3(64), 0, ctx->stream, d_frames, frame_pitch, stride, width, height, channels, d_sums);
  int rc = check_hip(ctx, hipGetLastError(), "blockhash (any size) kernel launch");
  if (rc) return rc;
  std::vector<float> sums((size_t)n_frames * 64);
  if ((rc = check_hip(ctx, hipMemcpyAsync(sums.data(), d_sums, need, hipMemcpyDeviceToHost, ctx->stream), "blockhash D2H"))) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "blockhash sync"))) return rc;
  // the 64 bits per frame: bands of 32 blocks, upper median, the crate's float comparison (host: 64 values per frame)
  const float bw = (float)width / 8.0f, bh = (float)height / 8.0f;
  const float half = 765.0f * bw * bh / 2.0f;
  for (int f = 0; f < n_frames; f++) {
    const float *blocks = sums.data() + (size_t)f * 64;
    unsigned long long h = 0;
    for (int g = 0; g < 2; g++) {
      float sorted[32];
      std::memcpy(sorted, blocks + 32 * g, sizeof sorted);
      std::sort(sorted, sorted + 32);
      const float median = sorted[16];
      for (int i = 0; i < 32; i++) {
        const float v = blocks[32 * g + i];
        if (v > median || (std::fabs(v - median) < 1.0f && median > half)) h |= 1ull << (32 * g + i);
      }
    }
    hashes[f] = h;
  }
  return MI355_OK;
}

// Hashes `n_frames` device-resident frames; `hashes` is a HOST array of n_frames u64.
int launch_blockhash(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch, int stride, int n_frames, int width, int height,
                     int channels, unsigned long long *hashes) {
  if (width % 8 != 0 || height % 8 != 0) {
    if (ctx->blockhash_any_size && width >= 8 && height >= 8) return launch_blockhash_slow(ctx, d_frames, frame_pitch, stride, n_frames, width, height, channels, hashes);
    return set_error(ctx, MI355_ERR_UNSUPPORTED, "videocompare: blockhash on a frame that does not divide into 8 x 8 whole blocks takes the crate's floating-point path: "
                                                  "set MI355_FLAG_BLOCKHASH_ANY_SIZE (restated from memory, sequential per block)");
  }
  // scratch: [n_frames][64] u32 sums + [n_frames] u64 hashes in staging slot 1 (slot 0 holds the host entry's frame)
  const size_t need = (size_t)n_frames * (64 * 4 + 8);
  if (ctx->d_stage_bytes[1] < need) {
    if (ctx->d_stage[1]) (void)hipFree(ctx->d_stage[1]);
    ctx->d_stage[1] = nullptr;
    ctx->d_stage_bytes[1] = 0;
    int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[1], need), "hipMalloc(blockhash scratch)");
    if (rc) return rc;
    ctx->d_stage_bytes[1] = need;
  }
  unsigned long long *d_hashes = (unsigned long long *)ctx->d_stage[1];
  uint32_t *d_sums = (uint32_t *)(d_hashes + n_frames);
  int rc = check_hip(ctx, hipMemsetAsync(d_sums, 0, (size_t)n_frames * 64 * 4, ctx->stream), "hipMemset(blockhash sums)");
  if (rc) return rc;
  const bool vec = channels == 4 && stride % 16 == 0 && frame_pitch % 16 == 0 && ((uintptr_t)d_frames % 16 == 0) && width >= 32;
  const size_t seg_px = vec ? 1024 : 256;
  const size_t rows = vec ? ((size_t)height + kRowGroup - 1) / kRowGroup : (size_t)height;
  const size_t items = (size_t)n_frames * rows * (((size_t)width + seg_px - 1) / seg_px);
  size_t grid = (size_t)ctx->n_cu * 8;
  if (grid > items) grid = items;
  if (grid < 1) grid = 1;
  if (vec)
    hipLaunchKernelGGL(blockhash_rgba_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, d_frames, frame_pitch, stride, n_frames, width, height, d_sums);
  else
    hipLaunchKernelGGL(blockhash_bytes_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, d_frames, frame_pitch, stride, n_frames, width, height, channels, d_sums);
  const uint32_t cmp_factor = 765u * (uint32_t)((width / 8) * (height / 8)) / 2u;
  hipLaunchKernelGGL(blockhash_finish_kernel, dim3(n_frames), dim3(64), 0, ctx->stream, (const uint32_t *)d_sums, n_frames, cmp_factor, d_hashes);
  rc = check_hip(ctx, hipGetLastError(), "blockhash kernel launch");
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpyAsync(hashes, d_hashes, (size_t)n_frames * 8, hipMemcpyDeviceToHost, ctx->stream), "blockhash D2H");
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "blockhash sync");
}

// group.hip (mi355_group_submit_compare with Blockhash): n SEPARATE frames (any pointers, one geometry, width and height multiples
// of 8) hashed on ctx->stream into d_hashes[n] with NO host wait - one kernel per frame (a 4K frame is 33 MB: ~8 us of reads), one
// finish launch for all of them. d_sums: n x 64 u32 of scratch. Same kernels, same bits as launch_blockhash.
int blockhash_enqueue(mi355_ctx *ctx, const uint8_t *const *d_frames, int n, int stride, int width, int height, int channels, uint32_t *d_sums,
                      unsigned long long *d_hashes) {
  if (width % 8 != 0 || height % 8 != 0 || width < 8 || height < 8) return set_error(ctx, MI355_ERR_UNSUPPORTED, "videocompare: blockhash batches take frames of 8 x 8 whole blocks");
  int rc = check_hip(ctx, hipMemsetAsync(d_sums, 0, (size_t)n * 64 * 4, ctx->stream), "hipMemset(blockhash sums)");
  if (rc) return rc;
  for (int f = 0; f < n; f++) {
    const bool vec = channels == 4 && stride % 16 == 0 && ((uintptr_t)d_frames[f] % 16 == 0) && width >= 32;
    const size_t seg_px = vec ? 1024 : 256;
    const size_t rows = vec ? ((size_t)height + kRowGroup - 1) / kRowGroup : (size_t)height;
    const size_t items = rows * (((size_t)width + seg_px - 1) / seg_px);
    size_t grid = (size_t)ctx->n_cu * 8;
    if (grid > items) grid = items;
    if (grid < 1) grid = 1;
    if (vec)
      hipLaunchKernelGGL(blockhash_rgba_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, d_frames[f], (size_t)0, stride, 1, width, height, d_sums + (size_t)f * 64);
    else
      hipLaunchKernelGGL(blockhash_bytes_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->stream, d_frames[f], (size_t)0, stride, 1, width, height, channels, d_sums + (size_t)f * 64);
  }
  const uint32_t cmp_factor = 765u * (uint32_t)((width / 8) * (height / 8)) / 2u;
  hipLaunchKernelGGL(blockhash_finish_kernel, dim3(n), dim3(64), 0, ctx->stream, (const uint32_t *)d_sums, n, cmp_factor, d_hashes);
  return check_hip(ctx, hipGetLastError(), "blockhash kernel launch");
}


// ------------------------------------------------------------------ resize-based hashes (Mean / Gradient / VertGradient / DoubleGradient)
// image_hasher 3.1.1 with HasherConfig::new(): grayscale (integer luma), image::imageops::resize with Lanczos3 to
// 8x8 / 9x8 / 8x9 / 5x5 (vertical_sample to f32, then horizontal_sample with clamp + round), then the bit rule
// (restated in oracle/imghash_oracle.c; parity unpinned). The filter weights are computed on the host exactly as
// the crate does (f32, sinf) and the two sampling passes accumulate in the crate's order, one lane per output sample:
//   imghash_vertical_kernel   one lane per (frame, output row, column): sum over ~3*height/rows input rows of luma*w
//   imghash_finish_kernel     one lane per output pixel: horizontal sum, clamp, round; lane 0 packs the bits
struct ImgHashW {            // per output sample: first input index, tap count, offset into the weight array
  int left[9], count[9], offset[9];
};

__global__ __launch_bounds__(256) void imghash_vertical_kernel(const uint8_t *__restrict__ frames, size_t frame_pitch, int stride, int channels, int width,
                                                               ImgHashW vw, const float *__restrict__ weights, int rh, float *__restrict__ tmp) {
  const int x = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, f = blockIdx.z;
  if (x >= width) return;
  const float *w = weights + vw.offset[oy];
  const int n = vw.count[oy];
  const uint8_t *p = frames + (size_t)f * frame_pitch + (size_t)vw.left[oy] * stride + (size_t)x * channels;
  float t = 0.0f;
  for (int i = 0; i < n; i++, p += stride) {
    const uint32_t l = (2126u * p[0] + 7152u * p[1] + 722u * p[2]) / 10000u;
    t += (float)l * w[i];
  }
  tmp[((size_t)f * rh + oy) * width + x] = t;
}

__global__ __launch_bounds__(128) void imghash_finish_kernel(const float *__restrict__ tmp, int width, ImgHashW hw, const float *__restrict__ weights, int rw,
                                                             int rh, int algo, unsigned long long *__restrict__ hashes) {
  __shared__ uint8_t img[81];
  const int f = blockIdx.x, e = threadIdx.x;
  if (e < rw * rh) {
    const int y = e / rw, ox = e - y * rw;
    const float *w = weights + hw.offset[ox];
    const float *row = tmp + ((size_t)f * rh + y) * width + hw.left[ox];
    float t = 0.0f;
    for (int i = 0; i < hw.count[ox]; i++) t += row[i] * w[i];
    t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);
    img[e] = (uint8_t)roundf(t);
  }
  __syncthreads();
  if (e == 0) {
    unsigned long long h = 0;
    int nb = 0;
    if (algo == 0) {
      uint32_t sum = 0;
      for (int i = 0; i < 64; i++) sum += img[i];
      const uint8_t mean = (uint8_t)(sum / 64u);
      for (int i = 0; i < 64; i++, nb++) if (img[i] >= mean) h |= 1ull << nb;
    } else if (algo == 1 || algo == 3) {
      for (int y = 0; y < rh; y++) for (int x = 0; x + 1 < rw; x++, nb++) if (img[y * rw + x] < img[y * rw + x + 1]) h |= 1ull << nb;
      if (algo == 3) for (int y = 0; y + 1 < rh; y++) for (int x = 0; x < rw; x++, nb++) if (img[y * rw + x] < img[(y + 1) * rw + x]) h |= 1ull << nb;
    } else {
      for (int y = 0; y + 1 < rh; y++) for (int x = 0; x < rw; x++, nb++) if (img[y * rw + x] < img[(y + 1) * rw + x]) h |= 1ull << nb;
    }
    hashes[f] = h;
  }
}

static float ih_sinc(float t) { const float a = t * 3.14159265358979323846f; return t == 0.0f ? 1.0f : sinf(a) / a; }
static float ih_lanczos3(float x) { return fabsf(x) < 3.0f ? ih_sinc(x) * ih_sinc(x / 3.0f) : 0.0f; }

// image::imageops::sample weights for `out` samples over `in` inputs (same f32 expressions as the crate)
static void ih_weights(int in, int out, ImgHashW *W, std::vector<float> *all) {
  const float ratio = (float)in / (float)out;
  const float sratio = ratio < 1.0f ? 1.0f : ratio;
  const float support = 3.0f * sratio;
  for (int o = 0; o < out; o++) {
    float centre = ((float)o + 0.5f) * ratio;
    long left = (long)floorf(centre - support);
    if (left < 0) left = 0;
    if (left > in - 1) left = in - 1;
    long right = (long)ceilf(centre + support);
    if (right < left + 1) right = left + 1;
    if (right > in) right = in;
    centre = centre - 0.5f;
    const size_t off = all->size();
    float sum = 0.0f;
    for (long i = left; i < right; i++) { const float w = ih_lanczos3(((float)i - centre) / sratio); all->push_back(w); sum += w; }
    for (size_t i = off; i < all->size(); i++) (*all)[i] /= sum;
    W->left[o] = (int)left; W->count[o] = (int)(right - left); W->offset[o] = (int)off;
  }
}

// algo 0..3 as GstVideoCompareHashAlgorithm; `hashes` is a HOST array of n_frames u64
int launch_imghash(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch, int stride, int n_frames, int width, int height, int channels,
                   int algo, unsigned long long *hashes) {
  int rw, rh;
  switch (algo) { case 0: rw = 8; rh = 8; break; case 1: rw = 9; rh = 8; break; case 2: rw = 8; rh = 9; break; default: rw = 5; rh = 5; break; }
  ImgHashW vw, hw;
  std::vector<float> wts;
  ih_weights(height, rh, &vw, &wts);
  ih_weights(width, rw, &hw, &wts);
  // scratch (slot 1): weights | tmp [n_frames][rh][width] f32 | hashes
  const size_t wbytes = (wts.size() * 4 + 15) & ~(size_t)15, tbytes = (size_t)n_frames * rh * width * 4;
  const size_t need = wbytes + tbytes + (size_t)n_frames * 8;
  if (ctx->d_stage_bytes[1] < need) {
    if (ctx->d_stage[1]) (void)hipFree(ctx->d_stage[1]);
    ctx->d_stage[1] = nullptr; ctx->d_stage_bytes[1] = 0;
    int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[1], need), "hipMalloc(imghash scratch)");
    if (rc) return rc;
    ctx->d_stage_bytes[1] = need;
  }
  float *d_w = (float *)ctx->d_stage[1];
  float *d_tmp = (float *)((char *)ctx->d_stage[1] + wbytes);
  unsigned long long *d_h = (unsigned long long *)((char *)d_tmp + tbytes);
  int rc = check_hip(ctx, hipMemcpyAsync(d_w, wts.data(), wts.size() * 4, hipMemcpyHostToDevice, ctx->stream), "imghash: weights H2D");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "imghash: sync"))) return rc;  // wts is a local vector
  hipLaunchKernelGGL(imghash_vertical_kernel, dim3((width + 255) / 256, rh, n_frames), dim3(256), 0, ctx->stream, d_frames, frame_pitch, stride, channels, width,
                     vw, (const float *)d_w, rh, d_tmp);
  hipLaunchKernelGGL(imghash_finish_kernel, dim3(n_frames), dim3(128), 0, ctx->stream, (const float *)d_tmp, width, hw, (const float *)d_w, rw, rh, algo, d_h);
  if ((rc = check_hip(ctx, hipGetLastError(), "imghash kernel launch"))) return rc;
  if ((rc = check_hip(ctx, hipMemcpyAsync(hashes, d_h, (size_t)n_frames * 8, hipMemcpyDeviceToHost, ctx->stream), "imghash D2H"))) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "imghash sync");
}

}  // namespace mi355
