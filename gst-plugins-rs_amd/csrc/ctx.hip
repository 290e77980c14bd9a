// ctx.hip — the extern "C" layer of libmi355fx.so (declared in include/mi355fx.h): context
// lifetime, staging of host buffers, argument validation, dispatch to the kernel launchers.
#include "internal.hpp"
#include <mutex>

#include <cstdio>
#include <cstring>

namespace mi355 {

bool pixfmt_of(int format, PixFmt *out) {
  switch (format) {
    case MI355_FMT_RGBX: *out = {4, 0, 0, 0}; return true;
    case MI355_FMT_RGBA: *out = {4, 0, 0, 1}; return true;
    case MI355_FMT_XRGB: *out = {4, 1, 0, 0}; return true;
    case MI355_FMT_ARGB: *out = {4, 1, 0, 1}; return true;
    case MI355_FMT_BGRX: *out = {4, 0, 1, 0}; return true;
    case MI355_FMT_BGRA: *out = {4, 0, 1, 1}; return true;
    case MI355_FMT_XBGR: *out = {4, 1, 1, 0}; return true;
    case MI355_FMT_ABGR: *out = {4, 1, 1, 1}; return true;
    case MI355_FMT_RGB: *out = {3, 0, 0, 0}; return true;
    case MI355_FMT_BGR: *out = {3, 0, 1, 0}; return true;
    default: return false;
  }
}

int set_error(mi355_ctx *ctx, int status, const std::string &msg) {
  if (ctx) ctx->last_error = msg;
  return status;
}

int check_hip(mi355_ctx *ctx, hipError_t e, const char *what) {
  if (e == hipSuccess) return MI355_OK;
  std::string msg = std::string(what) + ": " + hipGetErrorString(e);
  const int st = (e == hipErrorOutOfMemory) ? MI355_ERR_OUT_OF_MEMORY : MI355_ERR_HIP;
  return set_error(ctx, st, msg);
}

static int ensure_stage(mi355_ctx *ctx, int slot, size_t bytes) {
  if (ctx->d_stage_bytes[slot] >= bytes && ctx->d_stage[slot]) return MI355_OK;
  if (ctx->d_stage[slot]) (void)hipFree(ctx->d_stage[slot]);
  ctx->d_stage[slot] = nullptr;
  ctx->d_stage_bytes[slot] = 0;
  int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[slot], bytes ? bytes : 16), "hipMalloc(staging)");
  if (rc) return rc;
  ctx->d_stage_bytes[slot] = bytes ? bytes : 16;
  return MI355_OK;
}

}  // namespace mi355

using namespace mi355;

template <typename F>
static int time_launches(mi355_ctx *ctx, int iters, float *ms_per_launch, F &&launch) {
  if (iters <= 0 || !ms_per_launch) return set_error(ctx, MI355_ERR_INVALID_ARG, "timing: bad iters/output");
  hipEvent_t e0, e1;
  int rc = check_hip(ctx, hipEventCreate(&e0), "hipEventCreate");
  if (rc) return rc;
  rc = check_hip(ctx, hipEventCreate(&e1), "hipEventCreate");
  if (rc) { (void)hipEventDestroy(e0); return rc; }
  rc = check_hip(ctx, hipEventRecord(e0, ctx->stream), "hipEventRecord");
  for (int i = 0; i < iters && rc == MI355_OK; i++) rc = launch();
  if (rc == MI355_OK) rc = check_hip(ctx, hipEventRecord(e1, ctx->stream), "hipEventRecord");
  if (rc == MI355_OK) rc = check_hip(ctx, hipEventSynchronize(e1), "hipEventSynchronize");
  float ms = 0.0f;
  if (rc == MI355_OK) rc = check_hip(ctx, hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc == MI355_OK) *ms_per_launch = ms / (float)iters;
  return rc;
}


#define REQUIRE_CTX(ctx) \
  do { if (!(ctx)) return MI355_ERR_INVALID_ARG; } while (0)
#define BIND_DEVICE(ctx)                                                                  \
  do { int rc__ = check_hip((ctx), hipSetDevice((ctx)->device), "hipSetDevice");          \
       if (rc__) return rc__; } while (0)

extern "C" {

int mi355_abi_version(void) { return MI355FX_ABI_VERSION; }

int mi355_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *mi355_status_string(int status) {
  switch (status) {
    case MI355_OK: return "ok";
    case MI355_ERR_INVALID_ARG: return "invalid argument";
    case MI355_ERR_NO_DEVICE: return "no usable gfx950 device";
    case MI355_ERR_HIP: return "HIP runtime error";
    case MI355_ERR_NOT_CONFIGURED: return "element not configured";
    case MI355_ERR_OUT_OF_MEMORY: return "out of device memory";
    case MI355_ERR_UNSUPPORTED: return "unsupported";
    case MI355_ERR_TIMEOUT: return "timeout";
    default: return "unknown status";
  }
}

}  // extern "C"

namespace mi355 {
namespace {
// the last two launches per device that wrote frames: {begin, end}
struct Written { uintptr_t lo = 0, hi = 0; };
std::mutex g_written_mu;
Written g_written[16][2];
// An eight-frame 4K batch is 265 MB and the launch behind it still finds about half of it on-die (the gather kernel
// then beats the LDS-cached one, 0.086-0.089 against 0.093-0.099 ms); twice that has been pushed out by its own tail.
constexpr size_t kOnDieBytes = (size_t)320 << 20;
}  // namespace

void note_written(int device, const void *p, size_t bytes) {
  if (device < 0 || device >= 16 || !p || !bytes) return;
  std::lock_guard<std::mutex> lk(g_written_mu);
  g_written[device][1] = g_written[device][0];
  g_written[device][0].lo = (uintptr_t)p;
  g_written[device][0].hi = (uintptr_t)p + bytes;
}

// a copy from the host has replaced [p, p + bytes): whatever a launch wrote there is no longer what a reader finds (and a DMA
// upload lands in HBM, not in the cache)
void forget_written(int device, const void *p, size_t bytes) {
  if (device < 0 || device >= 16 || !p || !bytes) return;
  std::lock_guard<std::mutex> lk(g_written_mu);
  for (Written &w : g_written[device])
    if (w.hi > w.lo && (uintptr_t)p < w.hi && (uintptr_t)p + bytes > w.lo) w = Written{};
}

bool recently_written(int device, const void *p, size_t bytes) {
  if (device < 0 || device >= 16 || !p || !bytes || bytes > kOnDieBytes) return false;
  std::lock_guard<std::mutex> lk(g_written_mu);
  for (const Written &w : g_written[device])
    if (w.hi > w.lo && (uintptr_t)p >= w.lo && (uintptr_t)p + bytes <= w.hi && w.hi - w.lo <= kOnDieBytes) return true;
  return false;
}
}  // namespace mi355

extern "C" {

mi355_ctx *mi355_ctx_create(int device, int *status) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
    if (status) *status = MI355_ERR_NO_DEVICE;
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    if (status) *status = MI355_ERR_NO_DEVICE;
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    if (status) *status = MI355_ERR_NO_DEVICE;
    return nullptr;
  }
  // The code objects in this library are gfx950-only: refuse anything else loudly.
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "mi355fx: device %d is %s, this library only carries gfx950 code\n", device, prop.gcnArchName);
    if (status) *status = MI355_ERR_NO_DEVICE;
    return nullptr;
  }
  mi355_ctx *ctx = new mi355_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    if (status) *status = MI355_ERR_HIP;
    return nullptr;
  }
  ctx->own_stream = true;
  if (status) *status = MI355_OK;
  return ctx;
}

void mi355_ctx_destroy(mi355_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  lut_release(ctx);
  hsv_table_release(ctx);
  window_release(ctx);
  loudnorm_release(ctx);
  loudnorm_batch_release(ctx);
  dssim_release(ctx);
  roundedcorners_release(ctx);
  ebur128_release(ctx);
  hrtf_release(ctx);
  sofa_release(ctx);
  echo_release(ctx);
  if (ctx->host_copy_ev) (void)hipEventDestroy(ctx->host_copy_ev);
  for (int i = 0; i < 2; i++)
    if (ctx->d_stage[i]) (void)hipFree(ctx->d_stage[i]);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *mi355_ctx_last_error(const mi355_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

void *mi355_ctx_stream(mi355_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int mi355_ctx_set_stream(mi355_ctx *ctx, void *hip_stream) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  if (ctx->own_stream && ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  return MI355_OK;
}

int mi355_ctx_synchronize(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int mi355_ctx_set_flag(mi355_ctx *ctx, int flag, int value) {
  REQUIRE_CTX(ctx);
  if (flag == MI355_FLAG_FORCE_GENERIC) { ctx->force_generic = value != 0; return MI355_OK; }
  if (flag == MI355_FLAG_LUT_STAGGER && value >= 0 && value <= 4096) { ctx->lut_stagger = value; return MI355_OK; }
  if (flag == MI355_FLAG_LUT_VARIANT && value >= 0 && value <= 9) { ctx->lut_variant = value; return MI355_OK; }
  if (flag == MI355_FLAG_BRICK_TILES_PER_RUN && value >= 0 && value <= 4096) { ctx->brick_tiles_per_run = value; return MI355_OK; }
  if (flag == MI355_FLAG_BRICK_FOLD_AXIS && value >= 0 && value <= 2) { ctx->brick_fold_axis = value; return MI355_OK; }
  if (flag == MI355_FLAG_DSSIM_TRANSLUCENT && (value == 0 || value == 1)) { ctx->dssim_translucent = value; return MI355_OK; }
  if (flag == MI355_FLAG_BRICK_PRIO && value >= 0 && value <= 3) { ctx->brick_prio = value; return MI355_OK; }
  if (flag == MI355_FLAG_BRICK_SETS && (value == 0 || value == 32 || value == 48 || value == 64 || value == 512)) { ctx->brick_sets = value; return MI355_OK; }
  if (flag == MI355_FLAG_HSV_TABLE && value >= 0 && value <= 3) { ctx->hsv_table_mode = value; return MI355_OK; }
  if (flag == MI355_FLAG_FUSED_VARIANT && value >= 0 && value <= 1) { ctx->fused_variant = value; return MI355_OK; }
  if (flag == MI355_FLAG_HSV_BLOCKS_PER_CU && value >= 1 && value <= 4096) { ctx->hsv_blocks_per_cu = value; return MI355_OK; }
  if (flag == MI355_FLAG_HRTF_METHOD && value >= 0 && value <= 2) { ctx->hrtf_method = value; return MI355_OK; }
  if (flag == MI355_FLAG_BLOCKHASH_ANY_SIZE && (value == 0 || value == 1)) { ctx->blockhash_any_size = value; return MI355_OK; }
  if (flag == MI355_FLAG_HSV_NT && (value == 0 || value == 1)) { ctx->hsv_nt = value; return MI355_OK; }
  if (flag == MI355_FLAG_WINDOW_ORDER && (value == 0 || value == 1)) { ctx->window_order = value; return MI355_OK; }
  if (flag == MI355_FLAG_WINDOW_STATS && (value == 0 || value == 1)) { ctx->window_stats_on = value != 0; return MI355_OK; }
  if (flag == MI355_FLAG_WINDOW_MIN_STEPS && value >= 0 && value <= 4096) { ctx->window_min_steps = value; return MI355_OK; }
  return set_error(ctx, MI355_ERR_INVALID_ARG, "unknown flag");
}

void *mi355_device_alloc(mi355_ctx *ctx, size_t bytes) {
  if (!ctx) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  void *p = nullptr;
  if (check_hip(ctx, hipMalloc(&p, bytes ? bytes : 16), "hipMalloc")) return nullptr;
  return p;
}

int mi355_device_free(mi355_ctx *ctx, void *dptr) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return check_hip(ctx, hipFree(dptr), "hipFree");
}

int mi355_memcpy_h2d(mi355_ctx *ctx, void *dptr, const void *host, size_t bytes) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  int rc = check_hip(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D)");
  forget_written(ctx->device, dptr, bytes);
  __atomic_fetch_add(&ctx->n_h2d, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int mi355_memcpy_d2h(mi355_ctx *ctx, void *host, const void *dptr, size_t bytes) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  int rc = check_hip(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(D2H)");
  __atomic_fetch_add(&ctx->n_d2h, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

/* ------------------------------------------------------------------ hsvfilter */

int mi355_hsvfilter_frames_device(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width,
                                  int height, int stride, int format, const mi355_hsv_settings *settings) {
  REQUIRE_CTX(ctx);
  PixFmt fmt;
  if (!settings || !pixfmt_of(format, &fmt)) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: bad settings/format");
  if (n_frames < 0 || width < 0 || height < 0 || stride < 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: negative size");
  if (n_frames == 0 || width == 0 || height == 0) return MI355_OK;
  if (!d_data) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: null data");
  if ((size_t)stride < (size_t)width * fmt.pixel_stride)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: stride smaller than width*pixel_stride");
  if (n_frames > 1 && frame_pitch < (size_t)(height - 1) * (size_t)stride + (size_t)width * fmt.pixel_stride)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: frame_pitch smaller than one frame (frames would overlap)");
  BIND_DEVICE(ctx);
  const int rc = launch_hsvfilter(ctx, d_data, n_frames, frame_pitch, width, height, stride, fmt, *settings);
  if (rc == MI355_OK) note_written(ctx->device, d_data, (size_t)(n_frames - 1) * frame_pitch + (size_t)stride * (size_t)height);   // what the element behind finds on-die
  return rc;
}

int mi355_hsvfilter_frame_ip(mi355_ctx *ctx, uint8_t *data, size_t data_len, int width, int stride, int format,
                             const mi355_hsv_settings *settings) {
  REQUIRE_CTX(ctx);
  PixFmt fmt;
  if (!settings || !pixfmt_of(format, &fmt)) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: bad settings/format");
  if (width < 0 || stride <= 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: bad width/stride");
  // assert_eq!(data.len() % nb_channels, 0) (hsvfilter/imp.rs:92)
  if (data_len % (size_t)fmt.pixel_stride != 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: plane length not a multiple of pixel stride");
  const size_t rows = data_len / (size_t)stride;  // chunks_exact_mut(stride)
  if (rows == 0 || width == 0) return MI355_OK;
  if (!data) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: null data");
  // line[..line_bytes] panics in the reference when line_bytes > stride
  if ((size_t)width * fmt.pixel_stride > (size_t)stride) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: width*pixel_stride exceeds stride");
  if (rows > 0x7fffffffu) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: too many rows");
  BIND_DEVICE(ctx);
  const size_t bytes = rows * (size_t)stride;
  int rc = ensure_stage(ctx, 0, bytes);
  if (rc) return rc;
  uint8_t *d = (uint8_t *)ctx->d_stage[0];
  rc = check_hip(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D frame)");
  __atomic_fetch_add(&ctx->n_h2d, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  rc = launch_hsvfilter(ctx, d, 1, bytes, width, (int)rows, stride, fmt, *settings);
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(D2H frame)");
  __atomic_fetch_add(&ctx->n_d2h, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "hsvfilter: stream synchronize");
}

/* ------------------------------------------------------------------ hsvdetector */

static bool detect_out_fmt(int fmt, int *alpha_first, int *bgr) {
  switch (fmt) {
    case MI355_FMT_RGBA: *alpha_first = 0; *bgr = 0; return true;
    case MI355_FMT_ARGB: *alpha_first = 1; *bgr = 0; return true;
    case MI355_FMT_BGRA: *alpha_first = 0; *bgr = 1; return true;
    case MI355_FMT_ABGR: *alpha_first = 1; *bgr = 1; return true;
    default: return false;
  }
}
static bool detect_in_fmt(int fmt, PixFmt *p) {
  // hsvdetector/imp.rs:78-87: Rgbx, Xrgb, Bgrx, Xbgr, Rgb, Bgr
  switch (fmt) {
    case MI355_FMT_RGBX: case MI355_FMT_XRGB: case MI355_FMT_BGRX: case MI355_FMT_XBGR:
    case MI355_FMT_RGB: case MI355_FMT_BGR: return pixfmt_of(fmt, p);
    default: return false;
  }
}

int mi355_hsvdetect_frames_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, int src_format,
                                  uint8_t *d_dst, size_t dst_pitch, int dst_stride, int dst_format, int n_frames,
                                  int width, int height, const mi355_hsvdetect_settings *settings) {
  REQUIRE_CTX(ctx);
  PixFmt sfmt;
  int af = 0, bgr = 0;
  if (!settings || !detect_in_fmt(src_format, &sfmt) || !detect_out_fmt(dst_format, &af, &bgr))
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: bad settings/format");
  if (n_frames < 0 || width < 0 || height < 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: negative size");
  if (n_frames == 0 || width == 0 || height == 0) return MI355_OK;
  if (!d_src || !d_dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: null data");
  if ((size_t)src_stride < (size_t)width * sfmt.pixel_stride || (size_t)dst_stride < (size_t)width * 4)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: line bytes exceed stride");
  BIND_DEVICE(ctx);
  const int rc = launch_hsvdetect(ctx, d_src, src_pitch, src_stride, sfmt, d_dst, dst_pitch, dst_stride, af, bgr, n_frames, width, height, *settings);
  if (rc == MI355_OK) note_written(ctx->device, d_dst, (size_t)(n_frames - 1) * dst_pitch + (size_t)dst_stride * (size_t)height);
  return rc;
}

int mi355_hsvdetect_frame(mi355_ctx *ctx, const uint8_t *src, size_t src_len, int src_stride, int src_format, uint8_t *dst,
                          size_t dst_len, int dst_stride, int dst_format, int width, const mi355_hsvdetect_settings *settings) {
  REQUIRE_CTX(ctx);
  PixFmt sfmt;
  int af = 0, bgr = 0;
  if (!settings || !detect_in_fmt(src_format, &sfmt) || !detect_out_fmt(dst_format, &af, &bgr))
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: bad settings/format");
  if (width < 0 || src_stride <= 0 || dst_stride <= 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: bad width/stride");
  const size_t rows_in = src_len / (size_t)src_stride, rows_out = dst_len / (size_t)dst_stride;
  // assert_eq!(out_data.len() / out_stride, in_data.len() / in_stride) (hsvdetector/imp.rs:123)
  if (rows_in != rows_out) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: input/output row count mismatch");
  if (src_len % (size_t)sfmt.pixel_stride != 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: plane length not a multiple of pixel stride");
  if ((size_t)width * sfmt.pixel_stride > (size_t)src_stride || (size_t)width * 4 > (size_t)dst_stride)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: line bytes exceed stride");
  if (rows_in == 0 || width == 0) return MI355_OK;
  if (!src || !dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvdetector: null data");
  BIND_DEVICE(ctx);
  const size_t sb = rows_in * (size_t)src_stride, db = rows_out * (size_t)dst_stride;
  int rc = ensure_stage(ctx, 0, sb);
  if (rc) return rc;
  rc = ensure_stage(ctx, 1, db);
  if (rc) return rc;
  uint8_t *ds = (uint8_t *)ctx->d_stage[0], *dd = (uint8_t *)ctx->d_stage[1];
  rc = check_hip(ctx, hipMemcpyAsync(ds, src, sb, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D src)");
  __atomic_fetch_add(&ctx->n_h2d, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  // only out_line[..width*4] is written by the reference: bring the rest of dst over unchanged
  rc = check_hip(ctx, hipMemcpyAsync(dd, dst, db, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D dst)");
  __atomic_fetch_add(&ctx->n_h2d, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  rc = launch_hsvdetect(ctx, ds, sb, src_stride, sfmt, dd, db, dst_stride, af, bgr, 1, width, (int)rows_in, *settings);
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpyAsync(dst, dd, db, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(D2H dst)");
  __atomic_fetch_add(&ctx->n_d2h, 1ull, __ATOMIC_RELAXED);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "hsvdetector: stream synchronize");
}

/* ------------------------------------------------------------------ colorlut */

int mi355_colorlut_load(mi355_ctx *ctx, int is3d, size_t size, const float *table, const float domain_scale[3],
                        const float domain_offset[3]) {
  REQUIRE_CTX(ctx);
  if (!table || !domain_scale || !domain_offset) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: null LUT data");
  // validate_lut_size (parser.rs:305-318, limits :12-16)
  if (is3d ? (size < 2 || size > 256) : (size < 2 || size > 65536))
    return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: LUT size out of range");
  BIND_DEVICE(ctx);
  int rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "colorlut: stream synchronize");
  if (rc) return rc;
  return lut_upload(ctx, is3d ? 1 : 0, size, table, domain_scale, domain_offset);
}

int mi355_colorlut_unload(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  lut_release(ctx);
  return MI355_OK;
}

int mi355_colorlut_kernel_choice(mi355_ctx *ctx, int fused, int *table_in_use, double *ms_per_mpx_compute, double *ms_per_mpx_table) {
  REQUIRE_CTX(ctx);
  if (fused == 10 || fused == 11) {  // inside the table path: which kernel read the table last (a rule on the input's provenance: no times)
    if (table_in_use) *table_in_use = ctx->lut.last_sub[fused - 10];
    if (ms_per_mpx_compute) *ms_per_mpx_compute = 0.0;
    if (ms_per_mpx_table) *ms_per_mpx_table = 0.0;
    return MI355_OK;
  }
  const AutoPick &A = fused == 2 ? ctx->hsv_table.pick : ctx->lut.pick[fused ? 1 : 0];
  // times are kept per 16-byte group = 4 pixels
  if (table_in_use)
    *table_in_use = fused == 2 ? (ctx->hsv_table.last_table ? 1 : 0)
                               : (ctx->lut_variant == 4 || ctx->lut_variant == 5 || ctx->lut_variant == 8 || ctx->lut_variant == 9 || (ctx->lut_variant == 0 && A.t_table > 0.0 && A.table));
  if (ms_per_mpx_compute) *ms_per_mpx_compute = A.t_compute * 250000.0;
  if (ms_per_mpx_table) *ms_per_mpx_table = A.t_table * 250000.0;
  return MI355_OK;
}

const char *mi355_colorlut_last_kernel(mi355_ctx *ctx) {
  if (!ctx) return "";
  return ctx->lut.last_kernel ? ctx->lut.last_kernel : "";
}

int mi355_colorlut_brick_stats(mi355_ctx *ctx, uint64_t counters[2], double *last_miss_fraction, int *hostile, int reset) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  unsigned long long c[2] = {0, 0};
  int rc = brick_read_counters(ctx, ctx->lut.brick, c, reset != 0);
  if (rc) return rc;
  if (counters) { counters[0] = c[0]; counters[1] = c[1]; }
  if (last_miss_fraction) *last_miss_fraction = ctx->lut.brick.watch.last_miss;
  if (hostile) *hostile = ctx->lut.brick.watch.home;
  return MI355_OK;
}

int mi355_colorlut_window_stats(mi355_ctx *ctx, uint64_t counters[3], int reset) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  if (!counters) return set_error(ctx, MI355_ERR_INVALID_ARG, "null counters");
  unsigned long long c[3] = {0, 0, 0};
  int rc = window_read_counters(ctx, c, reset != 0);
  if (rc) return rc;
  for (int i = 0; i < 3; i++) counters[i] = c[i];
  return MI355_OK;
}

static int colorlut_bpp(int format) {
  if (format == MI355_FMT_RGBA) return 4;
  if (format == MI355_FMT_RGBA64_LE || format == MI355_FMT_RGBA64_BE) return 8;
  return 0;
}

int mi355_colorlut_frames_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                                 size_t dst_pitch, int dst_stride, int n_frames, int width, int height, int format) {
  REQUIRE_CTX(ctx);
  const int bpp = colorlut_bpp(format);
  if (!bpp) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: format must be RGBA, RGBA64_LE or RGBA64_BE");
  if (!ctx->lut.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames < 0 || width < 0 || height < 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: negative size");
  if (n_frames == 0 || width == 0 || height == 0) return MI355_OK;
  if (!d_src || !d_dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: null data");
  if ((size_t)src_stride < (size_t)width * bpp || (size_t)dst_stride < (size_t)width * bpp)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: stride smaller than row bytes");
  if (n_frames > 1 && (src_pitch < (size_t)(height - 1) * (size_t)src_stride + (size_t)width * bpp ||
                       dst_pitch < (size_t)(height - 1) * (size_t)dst_stride + (size_t)width * bpp))
    return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: frame pitch smaller than one frame (frames would overlap)");
  BIND_DEVICE(ctx);
  const int rc = launch_colorlut(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, format);
  if (rc == MI355_OK) note_written(ctx->device, d_dst, (size_t)(n_frames - 1) * dst_pitch + (size_t)dst_stride * (size_t)height);
  return rc;
}

int mi355_hsv_colorlut_frames_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                                     size_t dst_pitch, int dst_stride, int n_frames, int width, int height,
                                     const mi355_hsv_settings *settings) {
  REQUIRE_CTX(ctx);
  if (!settings) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: null settings");
  if (!ctx->lut.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (n_frames < 0 || width < 0 || height < 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: negative size");
  if (n_frames == 0 || width == 0 || height == 0) return MI355_OK;
  if (!d_src || !d_dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: null data");
  if ((size_t)src_stride < (size_t)width * 4 || (size_t)dst_stride < (size_t)width * 4)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: stride smaller than row bytes");
  if (n_frames > 1 && (src_pitch < (size_t)(height - 1) * (size_t)src_stride + (size_t)width * 4 ||
                       dst_pitch < (size_t)(height - 1) * (size_t)dst_stride + (size_t)width * 4))
    return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: frame pitch smaller than one frame (frames would overlap)");
  BIND_DEVICE(ctx);
  const int rc = launch_hsv_colorlut(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, *settings);
  if (rc == MI355_OK) note_written(ctx->device, d_dst, (size_t)(n_frames - 1) * dst_pitch + (size_t)dst_stride * (size_t)height);
  return rc;
}

int mi355_colorlut_frame(mi355_ctx *ctx, const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int width,
                         int height, int format) {
  REQUIRE_CTX(ctx);
  const int bpp = colorlut_bpp(format);
  if (!bpp) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: format must be RGBA, RGBA64_LE or RGBA64_BE");
  if (!ctx->lut.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (width < 0 || height < 0 || src_stride <= 0 || dst_stride <= 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: bad size/stride");
  if (width == 0 || height == 0) return MI355_OK;
  if (!src || !dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: null data");
  // RGBA64 rows are addressed in u16 units: plane_stride / 2 (imp.rs:358-359)
  const size_t s_stride = bpp == 8 ? (size_t)(src_stride / 2) * 2 : (size_t)src_stride;
  const size_t d_stride = bpp == 8 ? (size_t)(dst_stride / 2) * 2 : (size_t)dst_stride;
  const size_t row_bytes = (size_t)width * bpp;
  if (row_bytes > s_stride || row_bytes > d_stride) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: row bytes exceed stride");
  BIND_DEVICE(ctx);
  // Stage both planes tightly packed on the device (pitch = row_bytes): only the `width` pixels of
  // each row are read and written, exactly what the reference touches.
  const size_t packed = row_bytes * (size_t)height;
  int rc = ensure_stage(ctx, 0, packed);
  if (rc) return rc;
  rc = ensure_stage(ctx, 1, packed);
  if (rc) return rc;
  uint8_t *ds = (uint8_t *)ctx->d_stage[0], *dd = (uint8_t *)ctx->d_stage[1];
  rc = check_hip(ctx, hipMemcpy2DAsync(ds, row_bytes, src, s_stride, row_bytes, (size_t)height, hipMemcpyHostToDevice, ctx->stream),
                 "hipMemcpy2DAsync(H2D src)");
  if (rc) return rc;
  rc = launch_colorlut(ctx, ds, packed, (int)row_bytes, dd, packed, (int)row_bytes, 1, width, height, format);
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpy2DAsync(dst, d_stride, dd, row_bytes, row_bytes, (size_t)height, hipMemcpyDeviceToHost, ctx->stream),
                 "hipMemcpy2DAsync(D2H dst)");
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "colorlut: stream synchronize");
}

/* ------------------------------------------------------------------ rsaudioecho */

int mi355_echo_setup(mi355_ctx *ctx, size_t ring_len) { return mi355_echo_setup_batch(ctx, 1, ring_len); }

int mi355_echo_setup_batch(mi355_ctx *ctx, int n_streams, size_t ring_len) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return echo_setup(ctx, n_streams, ring_len);
}

int mi355_echo_reset(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  echo_release(ctx);
  return MI355_OK;
}

int mi355_echo_process_batch_device(mi355_ctx *ctx, void *d_data, size_t stream_stride, size_t n, int is_f64, const size_t *delay_samples,
                                    const double *intensity, const double *feedback) {
  REQUIRE_CTX(ctx);
  if (!delay_samples || !intensity || !feedback) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: null parameter array");
  if (n && !d_data) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: null data");
  BIND_DEVICE(ctx);
  return launch_echo_batch(ctx, d_data, stream_stride, n, is_f64, delay_samples, intensity, feedback);
}

int mi355_echo_process_device(mi355_ctx *ctx, void *d_data, size_t n, int is_f64, size_t delay_samples, double intensity,
                              double feedback) {
  REQUIRE_CTX(ctx);
  if (n && !d_data) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: null data");
  BIND_DEVICE(ctx);
  return launch_echo(ctx, d_data, n, is_f64, delay_samples, intensity, feedback);
}

static int echo_host(mi355_ctx *ctx, void *data, size_t n, int is_f64, size_t delay, double intensity, double feedback) {
  REQUIRE_CTX(ctx);
  if (n && !data) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: null data");
  if (!ctx->echo.configured) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "rsaudioecho: not negotiated (setup not called)");
  BIND_DEVICE(ctx);
  const size_t bytes = n * (is_f64 ? sizeof(double) : sizeof(float));
  int rc = ensure_stage(ctx, 0, bytes);
  if (rc) return rc;
  if (n) {
    rc = check_hip(ctx, hipMemcpyAsync(ctx->d_stage[0], data, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D audio)");
    if (rc) return rc;
  }
  rc = launch_echo(ctx, ctx->d_stage[0], n, is_f64, delay, intensity, feedback);
  if (rc) return rc;
  if (n) {
    rc = check_hip(ctx, hipMemcpyAsync(data, ctx->d_stage[0], bytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(D2H audio)");
    if (rc) return rc;
  }
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "rsaudioecho: stream synchronize");
}

int mi355_echo_process_f32(mi355_ctx *ctx, float *data, size_t n, size_t delay_samples, double intensity, double feedback) {
  return echo_host(ctx, data, n, 0, delay_samples, intensity, feedback);
}
int mi355_echo_process_f64(mi355_ctx *ctx, double *data, size_t n, size_t delay_samples, double intensity, double feedback) {
  return echo_host(ctx, data, n, 1, delay_samples, intensity, feedback);
}

int mi355_echo_get_state(mi355_ctx *ctx, double *ring_out, size_t ring_len, size_t *pos_out) {
  return mi355_echo_get_state_batch(ctx, 0, ring_out, ring_len, pos_out);
}

int mi355_echo_get_state_batch(mi355_ctx *ctx, int stream, double *ring_out, size_t ring_len, size_t *pos_out) {
  REQUIRE_CTX(ctx);
  if (!ctx->echo.configured) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "rsaudioecho: not negotiated (setup not called)");
  if (stream < 0 || stream >= ctx->echo.n_streams) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: no such stream in the batch");
  BIND_DEVICE(ctx);
  if (pos_out) *pos_out = ctx->echo.pos;
  if (ring_out) {
    const size_t n = ring_len < ctx->echo.ring_len ? ring_len : ctx->echo.ring_len;
    int rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    if (rc) return rc;
    if (n) return check_hip(ctx, hipMemcpy(ring_out, ctx->echo.d_ring + (size_t)stream * ctx->echo.ring_len, n * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(echo ring)");
  }
  return MI355_OK;
}

/* ------------------------------------------------------------------ ebur128 (loudness meter) */

int mi355_ebur128_setup(mi355_ctx *ctx, unsigned channels, unsigned rate, unsigned mode, const int *channel_class) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_setup(ctx, channels, rate, mode, channel_class);
}
int mi355_ebur128_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, unsigned rate, unsigned mode, const int *channel_class) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_setup_batch(ctx, n_streams, channels, rate, mode, channel_class);
}
int mi355_ebur128_add_frames_batch(mi355_ctx *ctx, const void *data, size_t frames, int sample_format) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_add_frames_batch(ctx, data, frames, sample_format, 0);
}
int mi355_ebur128_add_frames_batch_device(mi355_ctx *ctx, const void *d_data, size_t frames, int sample_format) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_add_frames_batch(ctx, d_data, frames, sample_format, 1);
}
int mi355_ebur128_loudness_batch(mi355_ctx *ctx, int what, double *out) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return ebur128_query_batch(ctx, what, out); }
int mi355_ebur128_peak_batch(mi355_ctx *ctx, int true_peak, double *out) { REQUIRE_CTX(ctx); return ebur128_peak_batch(ctx, true_peak, out); }
int mi355_ebur128_reset(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_reset(ctx);
}
int mi355_ebur128_teardown(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  ebur128_release(ctx);
  return MI355_OK;
}
int mi355_ebur128_add_frames(mi355_ctx *ctx, const void *data, size_t frames, int sample_format) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_add_frames(ctx, data, nullptr, frames, sample_format);
}
int mi355_ebur128_add_frames_planar(mi355_ctx *ctx, const void *const *planes, size_t frames, int sample_format) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return ebur128_add_frames(ctx, nullptr, planes, frames, sample_format);
}
int mi355_ebur128_loudness_momentary(mi355_ctx *ctx, double *out) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return ebur128_query(ctx, 0, out); }
int mi355_ebur128_loudness_shortterm(mi355_ctx *ctx, double *out) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return ebur128_query(ctx, 1, out); }
int mi355_ebur128_loudness_global(mi355_ctx *ctx, double *out) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return ebur128_query(ctx, 2, out); }
int mi355_ebur128_relative_threshold(mi355_ctx *ctx, double *out) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return ebur128_query(ctx, 3, out); }
int mi355_ebur128_loudness_range(mi355_ctx *ctx, double *out) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return ebur128_query(ctx, 4, out); }
int mi355_ebur128_sample_peak(mi355_ctx *ctx, unsigned channel, double *out) { REQUIRE_CTX(ctx); return ebur128_peak(ctx, 0, channel, out); }
int mi355_ebur128_true_peak(mi355_ctx *ctx, unsigned channel, double *out) { REQUIRE_CTX(ctx); return ebur128_peak(ctx, 1, channel, out); }

/* ------------------------------------------------------------------ audioloudnorm */

int mi355_loudnorm_setup(mi355_ctx *ctx, unsigned channels, double loudness_target, double loudness_range_target, double max_true_peak,
                         double offset) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return loudnorm_setup(ctx, channels, loudness_target, loudness_range_target, max_true_peak, offset);
}
int mi355_loudnorm_push(mi355_ctx *ctx, const double *data, size_t frames, double *out, size_t out_capacity_frames, size_t *out_frames) {
  REQUIRE_CTX(ctx);
  if ((frames && !data) || !out || !out_frames) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: null argument");
  BIND_DEVICE(ctx);
  return loudnorm_push(ctx, data, frames, out, out_capacity_frames, out_frames);
}
int mi355_loudnorm_drain(mi355_ctx *ctx, double *out, size_t out_capacity_frames, size_t *out_frames, int *eos) {
  REQUIRE_CTX(ctx);
  if (!out || !out_frames || !eos) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: null argument");
  BIND_DEVICE(ctx);
  return loudnorm_drain(ctx, out, out_capacity_frames, out_frames, eos);
}
int mi355_loudnorm_teardown(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  loudnorm_release(ctx);
  loudnorm_batch_release(ctx);
  return MI355_OK;
}

int mi355_loudnorm_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, double loudness_target, double loudness_range_target,
                               double max_true_peak, double offset) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return loudnorm_setup_batch(ctx, n_streams, channels, loudness_target, loudness_range_target, max_true_peak, offset);
}
size_t mi355_loudnorm_batch_frame_size(mi355_ctx *ctx) { return ctx ? loudnorm_batch_frame_size(ctx) : 0; }
int mi355_loudnorm_process_batch(mi355_ctx *ctx, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stream_stride,
                                 size_t out_capacity_frames, size_t *out_frames, int final_frame, int device_data) {
  REQUIRE_CTX(ctx);
  if (!out_frames) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: null argument");
  BIND_DEVICE(ctx);
  return loudnorm_process_batch(ctx, data, stream_stride, frames, out, out_stream_stride, out_capacity_frames, out_frames, device_data, final_frame);
}

/* ------------------------------------------------------------------ videocompare */

static int videocompare_channels(mi355_ctx *ctx, int format, int algo, int *channels) {
  if (algo < MI355_HASH_MEAN || algo > MI355_HASH_DSSIM) return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: unknown hash-algo");
  if (algo == MI355_HASH_DSSIM)
    return set_error(ctx, MI355_ERR_UNSUPPORTED, "videocompare: hash-algo=dssim has no 64-bit hash: use mi355_dssim_create_image / mi355_dssim_compare");
  if (format == MI355_FMT_RGBA) *channels = 4;
  else if (format == MI355_FMT_RGB) *channels = 3;
  else return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: format must be RGB or RGBA");  // pad template caps (imp.rs:167-172)
  return MI355_OK;
}

int mi355_videocompare_hash_frames_device(mi355_ctx *ctx, const uint8_t *d_frames, size_t frame_pitch, int stride, int n_frames,
                                          int width, int height, int format, int algo, uint64_t *hashes) {
  REQUIRE_CTX(ctx);
  int channels = 0;
  int rc = videocompare_channels(ctx, format, algo, &channels);
  if (rc) return rc;
  if (n_frames < 0 || width <= 0 || height <= 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: bad frame size");
  if (n_frames == 0) return MI355_OK;
  if (!d_frames || !hashes) return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: null data");
  if ((size_t)stride < (size_t)width * channels) return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: stride smaller than row bytes");
  BIND_DEVICE(ctx);
  if (algo != MI355_HASH_BLOCKHASH)
    return launch_imghash(ctx, d_frames, frame_pitch, stride, n_frames, width, height, channels, algo, (unsigned long long *)hashes);
  return launch_blockhash(ctx, d_frames, frame_pitch, stride, n_frames, width, height, channels, (unsigned long long *)hashes);
}

int mi355_videocompare_hash_frame(mi355_ctx *ctx, const uint8_t *data, int stride, int width, int height, int format, int algo,
                                  uint64_t *hash) {
  REQUIRE_CTX(ctx);
  int channels = 0;
  int rc = videocompare_channels(ctx, format, algo, &channels);
  if (rc) return rc;
  if (width <= 0 || height <= 0 || !data || !hash) return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: bad frame");
  if ((size_t)stride < (size_t)width * channels) return set_error(ctx, MI355_ERR_INVALID_ARG, "videocompare: stride smaller than row bytes");
  BIND_DEVICE(ctx);
  // pack while copying: only the first width*channels bytes of every row travel (tightly_packed_framebuffer)
  const size_t row = (size_t)width * channels, pitch = (row + 15) & ~(size_t)15;
  rc = ensure_stage(ctx, 0, pitch * (size_t)height);
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpy2DAsync(ctx->d_stage[0], pitch, data, (size_t)stride, row, (size_t)height, hipMemcpyHostToDevice, ctx->stream),
                 "videocompare H2D");
  if (rc) return rc;
  if (algo != MI355_HASH_BLOCKHASH)
    return launch_imghash(ctx, (const uint8_t *)ctx->d_stage[0], pitch * (size_t)height, (int)pitch, 1, width, height, channels, algo,
                          (unsigned long long *)hash);
  return launch_blockhash(ctx, (const uint8_t *)ctx->d_stage[0], pitch * (size_t)height, (int)pitch, 1, width, height, channels,
                          (unsigned long long *)hash);
}

double mi355_videocompare_distance(int algo, uint64_t reference_hash, uint64_t frame_hash) {
  if (algo < MI355_HASH_MEAN || algo > MI355_HASH_BLOCKHASH) return -1.0;
  return (double)__builtin_popcountll(reference_hash ^ frame_hash);
}

/* ------------------------------------------------------------------ videocompare: Dssim engine */

static int dssim_channels(mi355_ctx *ctx, int format, int *channels) {
  if (format == MI355_FMT_RGBA) *channels = 4;
  else if (format == MI355_FMT_RGB) *channels = 3;
  else return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: format must be RGB or RGBA");
  return MI355_OK;
}

int mi355_dssim_create_image_device(mi355_ctx *ctx, const uint8_t *d_frame, int stride, int width, int height, int format,
                                    mi355_dssim_image **out) {
  REQUIRE_CTX(ctx);
  int channels = 0;
  int rc = dssim_channels(ctx, format, &channels);
  if (rc) return rc;
  if (!d_frame || !out || width <= 0 || height <= 0 || (size_t)stride < (size_t)width * channels)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: bad frame");
  BIND_DEVICE(ctx);
  return dssim_create_image(ctx, d_frame, stride, width, height, channels, out);
}

int mi355_dssim_create_image(mi355_ctx *ctx, const uint8_t *data, int stride, int width, int height, int format, mi355_dssim_image **out) {
  REQUIRE_CTX(ctx);
  int channels = 0;
  int rc = dssim_channels(ctx, format, &channels);
  if (rc) return rc;
  if (!data || !out || width <= 0 || height <= 0 || (size_t)stride < (size_t)width * channels)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: bad frame");
  BIND_DEVICE(ctx);
  const size_t row = (size_t)width * channels;
  rc = ensure_stage(ctx, 0, row * (size_t)height);
  if (rc) return rc;
  rc = check_hip(ctx, hipMemcpy2DAsync(ctx->d_stage[0], row, data, (size_t)stride, row, (size_t)height, hipMemcpyHostToDevice, ctx->stream), "dssim H2D");
  if (rc) return rc;
  // The caller may reuse `data` as soon as this returns, and an "async" copy from pageable memory is not always finished with
  // the host buffer when the call returns (seen with 2D copies of rows that are no multiple of 4 bytes): wait for the COPY -
  // the kernels queued behind it stay asynchronous.
  if (!ctx->host_copy_ev && (rc = check_hip(ctx, hipEventCreateWithFlags(&ctx->host_copy_ev, hipEventDisableTiming), "hipEventCreate"))) return rc;
  if ((rc = check_hip(ctx, hipEventRecord(ctx->host_copy_ev, ctx->stream), "hipEventRecord"))) return rc;
  if ((rc = check_hip(ctx, hipEventSynchronize(ctx->host_copy_ev), "dssim: wait for the frame upload"))) return rc;
  return dssim_create_image(ctx, (const uint8_t *)ctx->d_stage[0], (int)row, width, height, channels, out);
}

void mi355_dssim_free_image(mi355_ctx *ctx, mi355_dssim_image *image) {
  if (!ctx || !image) return;
  (void)hipSetDevice(ctx->device);
  dssim_free_image(ctx, image);
}

int mi355_dssim_compare(mi355_ctx *ctx, const mi355_dssim_image *original, const mi355_dssim_image *modified, double *dssim) {
  REQUIRE_CTX(ctx);
  if (!original || !modified || !dssim) return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: null argument");
  BIND_DEVICE(ctx);
  return dssim_compare(ctx, original, modified, dssim);
}

int mi355_issue_streams_round(mi355_ctx *const *ctxs, int n_streams, uint8_t *const *d_src, uint8_t *const *d_dst, int width, int height, int stride,
                              int format, const mi355_hsv_settings *settings) {
  if (!ctxs || !d_src || !d_dst || !settings || n_streams < 0) return MI355_ERR_INVALID_ARG;
  const size_t pitch = (size_t)stride * (size_t)(height > 0 ? height : 0);
  for (int i = 0; i < n_streams; i++) {
    int rc = mi355_hsvfilter_frames_device(ctxs[i], d_src[i], 1, pitch, width, height, stride, format, settings);
    if (rc) return rc;
    rc = mi355_colorlut_frames_device(ctxs[i], d_src[i], pitch, stride, d_dst[i], pitch, stride, 1, width, height, format);
    if (rc) return rc;
  }
  return MI355_OK;
}

// hsvfilter (in place) then colorlut for n independent batches from ONE native call, on the context's stream, as n pairs of the two
// element calls would be. `lanes`: 1; 2 is accepted and runs as 1. Round 5 put odd batches on a side stream (one lane's launch
// boundary under the other lane's kernel): measured 12 % SLOWER (two memory-bound launches side by side push each other's
// intermediate batch out of the Infinity Cache, profiles/r05_lanes_probe.txt) and not safe in auto mode - a table build that the
// choice policy starts at batch k > 0 was enqueued on one lane and read by the other without an event in between.
// Removed in round 6 rather than ordered: with every batch waiting for the other lane's builds there is nothing left to overlap.
int mi355_hsv_colorlut_chain_batches_device(mi355_ctx *ctx, uint8_t *const *d_src, uint8_t *const *d_dst, int n_batches, int n_frames, size_t frame_pitch,
                                            int stride, int width, int height, int format, const mi355_hsv_settings *settings, int lanes) {
  REQUIRE_CTX(ctx);
  if (!d_src || !d_dst || !settings || n_batches < 0 || lanes < 1 || lanes > 2) return set_error(ctx, MI355_ERR_INVALID_ARG, "chain batches: bad argument");
  BIND_DEVICE(ctx);
  int rc = MI355_OK;
  for (int k = 0; k < n_batches && !rc; k++) {
    if (!d_src[k] || !d_dst[k]) { rc = set_error(ctx, MI355_ERR_INVALID_ARG, "chain batches: null batch"); break; }
    rc = mi355_hsvfilter_frames_device(ctx, d_src[k], n_frames, frame_pitch, width, height, stride, format, settings);
    if (!rc) rc = mi355_colorlut_frames_device(ctx, d_src[k], frame_pitch, stride, d_dst[k], frame_pitch, stride, n_frames, width, height, format);
  }
  return rc;
}

int mi355_dssim_compare_frames_device(mi355_ctx *ctx, const mi355_dssim_image *original, const uint8_t *const *d_frames, int n_frames, int stride,
                                      int width, int height, int format, double *dssim) {
  REQUIRE_CTX(ctx);
  int channels = 0;
  int rc = dssim_channels(ctx, format, &channels);
  if (rc) return rc;
  if (!original || !d_frames || !dssim || n_frames < 0 || width <= 0 || height <= 0 || (size_t)stride < (size_t)width * channels)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: bad frames");
  for (int f = 0; f < n_frames; f++)
    if (!d_frames[f]) return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: null frame");
  if (n_frames == 0) return MI355_OK;
  BIND_DEVICE(ctx);
  return dssim_compare_frames(ctx, original, d_frames, n_frames, stride, width, height, channels, dssim);
}

int mi355_dssim_compare_frames(mi355_ctx *ctx, const mi355_dssim_image *original, const uint8_t *const *frames, int n_frames, int stride, int width,
                               int height, int format, double *dssim) {
  REQUIRE_CTX(ctx);
  int channels = 0;
  int rc = dssim_channels(ctx, format, &channels);
  if (rc) return rc;
  if (!original || !frames || !dssim || n_frames < 0 || n_frames > 64 || width <= 0 || height <= 0 || (size_t)stride < (size_t)width * channels)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: bad frames");
  for (int f = 0; f < n_frames; f++)
    if (!frames[f]) return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: null frame");
  if (n_frames == 0) return MI355_OK;
  BIND_DEVICE(ctx);
  const size_t row = (size_t)width * channels, frame_bytes = (row * (size_t)height + 255) & ~(size_t)255;
  rc = ensure_stage(ctx, 0, frame_bytes * (size_t)n_frames);
  if (rc) return rc;
  const uint8_t *d_frames[64];
  for (int f = 0; f < n_frames; f++) {
    uint8_t *d = (uint8_t *)ctx->d_stage[0] + frame_bytes * (size_t)f;
    rc = check_hip(ctx, hipMemcpy2DAsync(d, row, frames[f], (size_t)stride, row, (size_t)height, hipMemcpyHostToDevice, ctx->stream), "dssim H2D");
    if (rc) return rc;
    d_frames[f] = d;
  }
  return dssim_compare_frames(ctx, original, d_frames, n_frames, (int)row, width, height, channels, dssim);
}

int mi355_selftest_dssim_cbrt(mi355_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t *mismatches) {
  REQUIRE_CTX(ctx);
  if (!mismatches) return set_error(ctx, MI355_ERR_INVALID_ARG, "selftest: null argument");
  BIND_DEVICE(ctx);
  return dssim_cbrt_selftest(ctx, lo_bits, hi_bits, mismatches);
}

int mi355_dssim_image_plane(mi355_ctx *ctx, const mi355_dssim_image *image, int scale, int channel, int kind, float *out, int *width, int *height) {
  REQUIRE_CTX(ctx);
  if (!image) return set_error(ctx, MI355_ERR_INVALID_ARG, "dssim: null image");
  BIND_DEVICE(ctx);
  return dssim_image_plane(ctx, image, scale, channel, kind, out, width, height);
}

/* ------------------------------------------------------------------ sofalizer */

int mi355_sofa_setup(mi355_ctx *ctx, int channels, int filter_len, int partition_length, int block_length) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return sofa_setup(ctx, channels, filter_len, partition_length, block_length);
}
int mi355_sofa_set_filter(mi355_ctx *ctx, int channel, const float *left, const float *right, int delay_left, int delay_right) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return sofa_set_filter(ctx, channel, left, right, delay_left, delay_right);
}
int mi355_sofa_set_drop(mi355_ctx *ctx, int channel, int drop) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return sofa_set_drop(ctx, channel, drop); }
int mi355_sofa_reset(mi355_ctx *ctx) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return sofa_reset(ctx); }
int mi355_sofa_teardown(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  sofa_release(ctx);
  return MI355_OK;
}
int mi355_sofa_process_block(mi355_ctx *ctx, const float *in, float *out, const float *distance_gains) {
  REQUIRE_CTX(ctx);
  if (!in || !out || !distance_gains) return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: null argument");
  BIND_DEVICE(ctx);
  return sofa_process_block_host(ctx, in, out, distance_gains);
}
int mi355_sofa_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out, const float *distance_gains) {
  REQUIRE_CTX(ctx);
  if (!d_in || !d_out || !distance_gains) return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: null argument");
  BIND_DEVICE(ctx);
  return sofa_process_block_device(ctx, d_in, d_out, distance_gains);
}

/* ------------------------------------------------------------------ hrtfrender */

int mi355_hrtf_load_sphere(mi355_ctx *ctx, const void *bytes, size_t len, uint32_t device_rate) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return hrtf_load_sphere(ctx, (const unsigned char *)bytes, len, device_rate);
}
int mi355_hrtf_setup(mi355_ctx *ctx, int channels, int block_length, int interpolation_steps) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return hrtf_setup(ctx, channels, block_length, interpolation_steps);
}
int mi355_hrtf_reset(mi355_ctx *ctx) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return hrtf_reset(ctx); }
int mi355_hrtf_teardown(mi355_ctx *ctx) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  hrtf_release(ctx);
  return MI355_OK;
}
int mi355_hrtf_process_block(mi355_ctx *ctx, const float *in, float *out, const float *positions_xyz, const float *distance_gains) {
  REQUIRE_CTX(ctx);
  if (!in || !out || !positions_xyz || !distance_gains) return set_error(ctx, MI355_ERR_INVALID_ARG, "hrtfrender: null argument");
  BIND_DEVICE(ctx);
  return hrtf_process_block_host(ctx, in, out, positions_xyz, distance_gains);
}
int mi355_hrtf_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out, const float *positions_xyz, const float *distance_gains) {
  REQUIRE_CTX(ctx);
  if (!d_in || !d_out || !positions_xyz || !distance_gains) return set_error(ctx, MI355_ERR_INVALID_ARG, "hrtfrender: null argument");
  BIND_DEVICE(ctx);
  return hrtf_process_block_device(ctx, d_in, d_out, positions_xyz, distance_gains);
}
int mi355_hrtf_sphere_info(mi355_ctx *ctx, uint32_t *hrir_len, uint32_t *n_vertices, uint32_t *n_faces) {
  REQUIRE_CTX(ctx);
  return hrtf_info(ctx, hrir_len, n_vertices, n_faces);
}
int mi355_hrtf_last_lookup(mi355_ctx *ctx, int *faces, float *uvw) { REQUIRE_CTX(ctx); BIND_DEVICE(ctx); return hrtf_last_lookup(ctx, faces, uvw); }

/* ------------------------------------------------------------------ measurement helpers */

int mi355_time_hsvfilter_device(mi355_ctx *ctx, uint8_t *d_data, int n_frames, size_t frame_pitch, int width, int height,
                                int stride, int format, const mi355_hsv_settings *settings, int iters, float *ms_per_launch) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return time_launches(ctx, iters, ms_per_launch, [&]() {
    return mi355_hsvfilter_frames_device(ctx, d_data, n_frames, frame_pitch, width, height, stride, format, settings);
  });
}

int mi355_time_hsv_colorlut_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                                   size_t dst_pitch, int dst_stride, int n_frames, int width, int height,
                                   const mi355_hsv_settings *settings, int iters, float *ms_per_launch) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return time_launches(ctx, iters, ms_per_launch, [&]() {
    return mi355_hsv_colorlut_frames_device(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, settings);
  });
}

int mi355_time_colorlut_device(mi355_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int src_stride, uint8_t *d_dst,
                               size_t dst_pitch, int dst_stride, int n_frames, int width, int height, int format, int iters,
                               float *ms_per_launch) {
  REQUIRE_CTX(ctx);
  BIND_DEVICE(ctx);
  return time_launches(ctx, iters, ms_per_launch, [&]() {
    return mi355_colorlut_frames_device(ctx, d_src, src_pitch, src_stride, d_dst, dst_pitch, dst_stride, n_frames, width, height, format);
  });
}

}  // extern "C"
