// pipeline.hip — pinned host memory and an asynchronous host-buffer pipeline over the element kernels.
//
// The synchronous host entry points (mi355_hsvfilter_frame_ip, mi355_colorlut_frame, ...) do H2D -> kernel -> D2H
// on one stream and block: per 4K RGBA frame that is two PCIe crossings back to back (SURVEY.md §0 last row,
// §8(f) rank 1). A GStreamer shim can do better without changing the elements' semantics:
//   * mi355_host_alloc / mi355_host_free: pinned (page-locked) host memory for the shim's GstAllocator /
//     buffer pool (propose_allocation / decide_allocation), so that hipMemcpyAsync is a true DMA;
//   * mi355_pipe_*: a ring of `depth` device slots; submit() enqueues H2D on an upload stream, the element
//     kernels on the context's compute stream and D2H on a download stream, chained with events, and returns a
//     ticket at once; wait(ticket) blocks until that frame's output bytes are in the caller's buffer. With
//     depth >= 2 frame n+1's upload and frame n-1's download overlap frame n's kernels (BaseTransform's
//     submit_input_buffer / generate_output give the shim exactly this one-frame latency).
// The kernels, the arithmetic and the error behaviour are those of the synchronous entry points; buffers are
// borrowed until wait() returns. Pageable host buffers still work (the runtime stages them), only slower.
#include "internal.hpp"

#include <string>
#include <vector>

struct mi355_pipe {
  mi355_ctx *ctx = nullptr;
  int depth = 0;
  size_t cap = 0;  // bytes per slot buffer
  struct Slot {
    uint8_t *d_in = nullptr, *d_out = nullptr;
    hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
    bool busy = false;
    uint64_t ticket = 0;
    // group mode (mi355_pipe_set_group): the frame is with the group's dispatcher; its download is enqueued when somebody asks
    // for it (wait / slot reuse), behind the batch that carries it
    bool deferred = false;
    int failed = 0;  // group mode: the frame's launch or download could not be enqueued; reported (once) to whoever waits for the frame
    uint64_t group_ticket = 0;
    uint8_t *host_dst = nullptr;
    size_t host_stride = 0, row_bytes = 0, rows = 0;
  };
  std::vector<Slot> slots;
  hipStream_t s_up = nullptr, s_down = nullptr;
  uint64_t next_ticket = 1;
  mi355_group *group = nullptr;  // not owned
};

using namespace mi355;

// group mode: the frame of this slot is (or will be, the call flushes up to it) in a batch; order this context's stream behind
// it and enqueue the download
static int pipe_finish_deferred(mi355_pipe *p, mi355_pipe::Slot *s) {
  if (s->failed) return s->failed;
  if (!s->deferred) return MI355_OK;
  int rc = mi355_group_order_after(p->group, p->ctx, s->group_ticket);
  if (rc) set_error(p->ctx, rc, std::string("pipeline: ") + mi355_group_last_error(p->group));
  if (!rc) rc = check_hip(p->ctx, hipEventRecord(s->computed, p->ctx->stream), "pipeline: record compute");
  if (!rc) rc = check_hip(p->ctx, hipStreamWaitEvent(p->s_down, s->computed, 0), "pipeline: download waits for compute");
  if (!rc) rc = check_hip(p->ctx, hipMemcpy2DAsync(s->host_dst, s->host_stride, s->d_out, s->row_bytes, s->row_bytes, s->rows, hipMemcpyDeviceToHost, p->s_down), "pipeline: D2H");
  if (!rc) rc = check_hip(p->ctx, hipEventRecord(s->downloaded, p->s_down), "pipeline: record download");
  if (rc) {
    // No download will happen. The slot stays busy and remembers why, so that the frame's own wait reports it instead of
    // synchronising a stale event and returning OK; the group is made to let go of the slot's buffers first (its dispatcher may
    // still hold the frame, launched or not).
    const std::string why = p->ctx->last_error;
    (void)mi355_group_wait(p->group, s->group_ticket);
    p->ctx->last_error = why;
    s->failed = rc;
  }
  s->deferred = false;
  return rc;
}

// the slot's frame is over (waited for, or reported as failed): free for the next one
static int pipe_retire_slot(mi355_pipe *p, mi355_pipe::Slot *s, const char *what) {
  int rc = pipe_finish_deferred(p, s);
  if (!rc) rc = check_hip(p->ctx, hipEventSynchronize(s->downloaded), what);
  s->failed = 0;
  s->busy = false;
  return rc;
}

static int pipe_take_slot(mi355_pipe *p, mi355_pipe::Slot **out) {
  mi355_pipe::Slot &s = p->slots[(size_t)(p->next_ticket % (uint64_t)p->depth)];
  if (s.busy) {  // back-pressure: the ring is full, finish the oldest frame first
    int rc = pipe_retire_slot(p, &s, "pipeline: wait for a free slot");
    if (rc) return rc;
  }
  *out = &s;
  return MI355_OK;
}

// upload `rows` rows of `row_bytes` from a host plane with `stride` into the slot (tightly packed)
static int pipe_upload(mi355_pipe *p, mi355_pipe::Slot *s, const uint8_t *src, size_t stride, size_t row_bytes, size_t rows) {
  int rc = check_hip(p->ctx, hipMemcpy2DAsync(s->d_in, row_bytes, src, stride, row_bytes, rows, hipMemcpyHostToDevice, p->s_up), "pipeline: H2D");
  forget_written(p->ctx->device, s->d_in, row_bytes * rows);
  if (rc) return rc;
  rc = check_hip(p->ctx, hipEventRecord(s->uploaded, p->s_up), "pipeline: record upload");
  if (rc) return rc;
  return check_hip(p->ctx, hipStreamWaitEvent(p->ctx->stream, s->uploaded, 0), "pipeline: compute waits for upload");
}

static int pipe_download(mi355_pipe *p, mi355_pipe::Slot *s, const uint8_t *d_from, uint8_t *dst, size_t stride, size_t row_bytes, size_t rows,
                         uint64_t *ticket) {
  int rc = check_hip(p->ctx, hipEventRecord(s->computed, p->ctx->stream), "pipeline: record compute");
  if (rc) return rc;
  rc = check_hip(p->ctx, hipStreamWaitEvent(p->s_down, s->computed, 0), "pipeline: download waits for compute");
  if (rc) return rc;
  rc = check_hip(p->ctx, hipMemcpy2DAsync(dst, stride, d_from, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, p->s_down), "pipeline: D2H");
  if (rc) return rc;
  rc = check_hip(p->ctx, hipEventRecord(s->downloaded, p->s_down), "pipeline: record download");
  if (rc) return rc;
  // the next upload into this slot must not start before this download has read it: enforced by `busy` + wait
  s->busy = true;
  s->ticket = p->next_ticket++;
  if (ticket) *ticket = s->ticket;
  return MI355_OK;
}

extern "C" {

void *mi355_host_alloc(mi355_ctx *ctx, size_t bytes) {
  if (!ctx) return nullptr;
  if (check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
  void *p = nullptr;
  if (check_hip(ctx, hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
  return p;
}

int mi355_host_free(mi355_ctx *ctx, void *p) {
  if (!ctx) return MI355_ERR_INVALID_ARG;
  if (!p) return MI355_OK;
  return check_hip(ctx, hipHostFree(p), "hipHostFree");
}

mi355_pipe *mi355_pipe_create(mi355_ctx *ctx, int depth, size_t max_frame_bytes) {
  if (!ctx || depth < 1 || depth > 64 || max_frame_bytes == 0) {
    if (ctx) set_error(ctx, MI355_ERR_INVALID_ARG, "pipeline: depth must be 1..64 and max_frame_bytes > 0");
    return nullptr;
  }
  if (check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
  mi355_pipe *p = new mi355_pipe();
  p->ctx = ctx; p->depth = depth; p->cap = (max_frame_bytes + 15) & ~(size_t)15;
  p->slots.resize((size_t)depth);
  bool ok = !check_hip(ctx, hipStreamCreateWithFlags(&p->s_up, hipStreamNonBlocking), "pipeline: upload stream") &&
            !check_hip(ctx, hipStreamCreateWithFlags(&p->s_down, hipStreamNonBlocking), "pipeline: download stream");
  for (auto &s : p->slots) {
    if (!ok) break;
    ok = !check_hip(ctx, hipMalloc((void **)&s.d_in, p->cap), "pipeline: slot input") && !check_hip(ctx, hipMalloc((void **)&s.d_out, p->cap), "pipeline: slot output") &&
         !check_hip(ctx, hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming), "pipeline: event") &&
         !check_hip(ctx, hipEventCreateWithFlags(&s.computed, hipEventDisableTiming), "pipeline: event") &&
         !check_hip(ctx, hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming), "pipeline: event");
  }
  if (!ok) { mi355_pipe_destroy(p); return nullptr; }
  return p;
}

void mi355_pipe_destroy(mi355_pipe *p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  // group mode: frames handed to the dispatcher may still be waiting there for a batch - a later flush by any stream would run
  // kernels on the slot buffers freed below. Make the group finish (or give up) every frame of this pipeline first.
  for (auto &s : p->slots)
    if (s.busy && s.deferred && p->group) (void)mi355_group_wait(p->group, s.group_ticket);
  if (p->s_up) (void)hipStreamSynchronize(p->s_up);
  (void)hipStreamSynchronize(p->ctx->stream);
  if (p->s_down) (void)hipStreamSynchronize(p->s_down);
  for (auto &s : p->slots) {
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    if (s.computed) (void)hipEventDestroy(s.computed);
    if (s.downloaded) (void)hipEventDestroy(s.downloaded);
  }
  if (p->s_up) (void)hipStreamDestroy(p->s_up);
  if (p->s_down) (void)hipStreamDestroy(p->s_down);
  delete p;
}

int mi355_pipe_set_group(mi355_pipe *p, mi355_group *group) {
  if (!p) return MI355_ERR_INVALID_ARG;
  int rc = mi355_pipe_wait_all(p);  // frames in flight finish the way they started
  if (rc) return rc;
  p->group = group;
  return MI355_OK;
}

int mi355_pipe_wait(mi355_pipe *p, uint64_t ticket) {
  if (!p) return MI355_ERR_INVALID_ARG;
  if (ticket == 0 || ticket >= p->next_ticket) return set_error(p->ctx, MI355_ERR_INVALID_ARG, "pipeline: unknown ticket");
  mi355_pipe::Slot &s = p->slots[(size_t)(ticket % (uint64_t)p->depth)];
  if (!s.busy || s.ticket != ticket) return MI355_OK;  // already completed (its slot was reclaimed or waited on)
  return pipe_retire_slot(p, &s, "pipeline: wait");
}

int mi355_pipe_wait_all(mi355_pipe *p) {
  if (!p) return MI355_ERR_INVALID_ARG;
  int first = MI355_OK;
  for (auto &s : p->slots)
    if (s.busy) {
      int rc = pipe_retire_slot(p, &s, "pipeline: wait");
      if (rc && !first) first = rc;
    }
  return first;
}

// hsvfilter, in place on the host plane (same argument meaning and checks as mi355_hsvfilter_frame_ip)
int mi355_pipe_submit_hsvfilter(mi355_pipe *p, uint8_t *data, size_t data_len, int width, int stride, int format,
                                const mi355_hsv_settings *settings, uint64_t *ticket) {
  if (!p) return MI355_ERR_INVALID_ARG;
  mi355_ctx *ctx = p->ctx;
  PixFmt fmt;
  if (!settings || !pixfmt_of(format, &fmt) || fmt.pixel_stride > 4) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: bad settings/format");
  if (width <= 0 || stride <= 0 || !data) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: bad width/stride/data");
  if (data_len % (size_t)fmt.pixel_stride != 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: plane length not a multiple of pixel stride");
  const size_t rows = data_len / (size_t)stride, row_bytes = (size_t)width * fmt.pixel_stride;
  if (rows == 0 || rows > 0x7fffffffu) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: bad row count");
  if (row_bytes > (size_t)stride) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsvfilter: width*pixel_stride exceeds stride");
  if (row_bytes * rows > p->cap) return set_error(ctx, MI355_ERR_INVALID_ARG, "pipeline: frame larger than max_frame_bytes");
  int rc = check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  if (rc) return rc;
  mi355_pipe::Slot *s = nullptr;
  if ((rc = pipe_take_slot(p, &s))) return rc;
  if ((rc = pipe_upload(p, s, data, (size_t)stride, row_bytes, rows))) return rc;
  if ((rc = launch_hsvfilter(ctx, s->d_in, 1, row_bytes * rows, width, (int)rows, (int)row_bytes, fmt, *settings))) return rc;
  return pipe_download(p, s, s->d_in, data, (size_t)stride, row_bytes, rows, ticket);
}

// colorlut (RGBA / RGBA64), src -> dst host planes (as mi355_colorlut_frame)
int mi355_pipe_submit_colorlut(mi355_pipe *p, const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int width, int height,
                               int format, uint64_t *ticket) {
  if (!p) return MI355_ERR_INVALID_ARG;
  mi355_ctx *ctx = p->ctx;
  const int bpp = format == MI355_FMT_RGBA ? 4 : ((format == MI355_FMT_RGBA64_LE || format == MI355_FMT_RGBA64_BE) ? 8 : 0);
  if (!bpp) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: format must be RGBA, RGBA64_LE or RGBA64_BE");
  if (!ctx->lut.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (width <= 0 || height <= 0 || src_stride <= 0 || dst_stride <= 0 || !src || !dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: bad size/stride/data");
  const size_t row_bytes = (size_t)width * bpp;
  if (row_bytes > (size_t)src_stride || row_bytes > (size_t)dst_stride) return set_error(ctx, MI355_ERR_INVALID_ARG, "colorlut: row bytes exceed stride");
  if (row_bytes * (size_t)height > p->cap) return set_error(ctx, MI355_ERR_INVALID_ARG, "pipeline: frame larger than max_frame_bytes");
  int rc = check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  if (rc) return rc;
  mi355_pipe::Slot *s = nullptr;
  if ((rc = pipe_take_slot(p, &s))) return rc;
  if ((rc = pipe_upload(p, s, src, (size_t)src_stride, row_bytes, (size_t)height))) return rc;
  const size_t packed = row_bytes * (size_t)height;
  if ((rc = launch_colorlut(ctx, s->d_in, packed, (int)row_bytes, s->d_out, packed, (int)row_bytes, 1, width, height, format))) return rc;
  return pipe_download(p, s, s->d_out, dst, (size_t)dst_stride, row_bytes, (size_t)height, ticket);
}

// hsvfilter ! colorlut on RGBA, src -> dst host planes (as mi355_hsv_colorlut_frames_device on one frame)
int mi355_pipe_submit_hsv_colorlut(mi355_pipe *p, const uint8_t *src, int src_stride, uint8_t *dst, int dst_stride, int width, int height,
                                   const mi355_hsv_settings *settings, uint64_t *ticket) {
  if (!p) return MI355_ERR_INVALID_ARG;
  mi355_ctx *ctx = p->ctx;
  if (!settings) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: null settings");
  if (!ctx->lut.loaded) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (width <= 0 || height <= 0 || src_stride <= 0 || dst_stride <= 0 || !src || !dst) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: bad size/stride/data");
  const size_t row_bytes = (size_t)width * 4;
  if (row_bytes > (size_t)src_stride || row_bytes > (size_t)dst_stride) return set_error(ctx, MI355_ERR_INVALID_ARG, "hsv+colorlut: row bytes exceed stride");
  if (row_bytes * (size_t)height > p->cap) return set_error(ctx, MI355_ERR_INVALID_ARG, "pipeline: frame larger than max_frame_bytes");
  int rc = check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  if (rc) return rc;
  mi355_pipe::Slot *s = nullptr;
  if ((rc = pipe_take_slot(p, &s))) return rc;
  if ((rc = pipe_upload(p, s, src, (size_t)src_stride, row_bytes, (size_t)height))) return rc;
  const size_t packed = row_bytes * (size_t)height;
  if (p->group) {
    // the frame joins the other streams' frames in the group's next launch (the fused pair: one launch per batch through the
    // composed table once the settings have stayed for eight frames, the context's own fused path until then); the download
    // follows when the frame is asked for
    uint64_t gt = 0;
    if ((rc = mi355_group_submit_fused(p->group, ctx, s->d_in, s->d_out, width, height, (int)row_bytes, MI355_FMT_RGBA, settings, &gt)))
      return set_error(ctx, rc, std::string("pipeline: ") + mi355_group_last_error(p->group));
    s->deferred = true; s->group_ticket = gt; s->host_dst = dst; s->host_stride = (size_t)dst_stride; s->row_bytes = row_bytes; s->rows = (size_t)height;
    s->busy = true;
    s->ticket = p->next_ticket++;
    if (ticket) *ticket = s->ticket;
    return MI355_OK;
  }
  if ((rc = launch_hsv_colorlut(ctx, s->d_in, packed, (int)row_bytes, s->d_out, packed, (int)row_bytes, 1, width, height, *settings))) return rc;
  return pipe_download(p, s, s->d_out, dst, (size_t)dst_stride, row_bytes, (size_t)height, ticket);
}

}  // extern "C"
