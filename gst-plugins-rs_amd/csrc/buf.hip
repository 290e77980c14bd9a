// buf.hip — mi355_buf: a device-resident buffer that a GstMemory can wrap, so that adjacent mi355 elements hand frames over in
// HBM instead of crossing PCIe twice per element.
//
// Precedent in the reference: video/colorlut/src/d3d12colorlut/imp.rs:385-492 - propose_allocation / decide_allocation offer a
// pool of GPU memory, and an element whose input memory is "ours" works on the GPU resource directly; anything else maps the
// memory and gets bytes. The same contract here:
//   * mi355_buf_device_ptr(buf, ctx, flags): what the `_device` entry points take. If the host side holds newer bytes
//     (somebody mapped the memory for writing) they are uploaded first, on ctx's stream; the stream is ordered behind the
//     last work another context committed on the buffer (one hipStreamWaitEvent, no host wait).
//   * mi355_buf_commit(buf, ctx): "what ctx's stream holds now includes the last use of this buffer" - the element calls it
//     after enqueuing its kernels, like unmapping a GPU resource.
//   * mi355_buf_map_host / unmap_host: GstMemory's mem_map / mem_unmap - a pinned host shadow, downloaded lazily (only if the
//     device side is newer), marked newer than the device by a WRITE map.
// Dirty tracking is three states (in sync / host newer / device newer); every copy is counted on the owning context
// (mi355_ctx_transfer_counts): tests/test_gpu_buf.py runs hsvdetector -> colorlut -> videocompare on ONE upload and ONE download.
#include "internal.hpp"

#include <atomic>
#include <mutex>

struct mi355_buf {
  mi355_ctx *owner = nullptr;
  uint8_t *d = nullptr;
  uint8_t *h = nullptr;  // pinned shadow, allocated at the first host map
  size_t size = 0;
  enum State { kInSync = 0, kHostNewer = 1, kDeviceNewer = 2 } state = kInSync;
  hipEvent_t committed = nullptr;   // last commit (any context)
  hipStream_t committed_on = nullptr;
  bool have_commit = false;
  hipEvent_t uploaded = nullptr;    // last H2D from the shadow: a host WRITE map waits for it
  bool upload_pending = false;
  int map_flags = 0, map_count = 0;
  std::atomic<int> refs{1};
  std::mutex mu;
};

using namespace mi355;

extern "C" {

mi355_buf *mi355_buf_alloc(mi355_ctx *ctx, size_t size) {
  if (!ctx) return nullptr;
  if (check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
  mi355_buf *b = new mi355_buf();
  b->owner = ctx;
  b->size = size;
  if (check_hip(ctx, hipMalloc((void **)&b->d, size ? (size + 15) & ~(size_t)15 : 16), "hipMalloc(mi355_buf)") ||
      check_hip(ctx, hipEventCreateWithFlags(&b->committed, hipEventDisableTiming), "hipEventCreate(mi355_buf)") ||
      check_hip(ctx, hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming), "hipEventCreate(mi355_buf)")) {
    if (b->d) (void)hipFree(b->d);
    if (b->committed) (void)hipEventDestroy(b->committed);
    delete b;
    return nullptr;
  }
  return b;
}

mi355_buf *mi355_buf_ref(mi355_buf *b) {
  if (b) b->refs.fetch_add(1, std::memory_order_relaxed);
  return b;
}

void mi355_buf_unref(mi355_buf *b) {
  if (!b || b->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  (void)hipSetDevice(b->owner->device);
  if (b->have_commit) (void)hipEventSynchronize(b->committed);  // nothing reads or writes it any more
  if (b->upload_pending) (void)hipEventSynchronize(b->uploaded);
  if (b->d) (void)hipFree(b->d);
  if (b->h) (void)hipHostFree(b->h);
  (void)hipEventDestroy(b->committed);
  (void)hipEventDestroy(b->uploaded);
  delete b;
}

size_t mi355_buf_size(const mi355_buf *b) { return b ? b->size : 0; }

void *mi355_buf_device_ptr(mi355_buf *b, mi355_ctx *ctx, int flags) {
  if (!b || !ctx || !(flags & (MI355_MAP_READ | MI355_MAP_WRITE))) return nullptr;
  std::lock_guard<std::mutex> g(b->mu);
  if (ctx->device != b->owner->device) { set_error(ctx, MI355_ERR_INVALID_ARG, "mi355_buf: context of another device"); return nullptr; }
  // (a memory somebody has mapped for READING may be read on the device too - a GstVideoAggregator maps its pads' frames before
  // the element sees them; anything involving a write on either side is refused)
  if (b->map_count && ((b->map_flags | flags) & MI355_MAP_WRITE)) { set_error(ctx, MI355_ERR_INVALID_ARG, "mi355_buf: mapped on the host"); return nullptr; }
  if (check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
  // behind whatever another context's stream still does with the buffer
  if (b->have_commit && b->committed_on != ctx->stream && check_hip(ctx, hipStreamWaitEvent(ctx->stream, b->committed, 0), "hipStreamWaitEvent(mi355_buf)")) return nullptr;
  if (b->state == mi355_buf::kHostNewer) {
    // (a kernel that overwrites the whole buffer asks for WRITE alone and skips the upload)
    if (flags & MI355_MAP_READ) {
      if (check_hip(ctx, hipMemcpyAsync(b->d, b->h, b->size, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(mi355_buf H2D)")) return nullptr;
      if (check_hip(ctx, hipEventRecord(b->uploaded, ctx->stream), "hipEventRecord(mi355_buf)")) return nullptr;
      b->upload_pending = true;
      b->owner->n_h2d++;
    }
    b->state = mi355_buf::kInSync;
  }
  if (flags & MI355_MAP_WRITE) b->state = mi355_buf::kDeviceNewer;
  return b->d;
}

int mi355_buf_commit(mi355_buf *b, mi355_ctx *ctx) {
  if (!b || !ctx) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(b->mu);
  int rc = check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipEventRecord(b->committed, ctx->stream), "hipEventRecord(mi355_buf commit)"))) return rc;
  b->committed_on = ctx->stream;
  b->have_commit = true;
  return MI355_OK;
}

void *mi355_buf_map_host(mi355_buf *b, int flags) {
  if (!b || !(flags & (MI355_MAP_READ | MI355_MAP_WRITE))) return nullptr;
  std::lock_guard<std::mutex> g(b->mu);
  mi355_ctx *ctx = b->owner;
  if (check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
  if (!b->h && check_hip(ctx, hipHostMalloc((void **)&b->h, b->size ? b->size : 1, hipHostMallocDefault), "hipHostMalloc(mi355_buf)")) return nullptr;
  if (b->upload_pending) {  // the shadow is still being read by an upload
    if (check_hip(ctx, hipEventSynchronize(b->uploaded), "hipEventSynchronize(mi355_buf)")) return nullptr;
    b->upload_pending = false;
  }
  if (b->state == mi355_buf::kDeviceNewer) {
    // A WRITE-only map downloads too: the mapper may write part of the memory and the rest must stay what it was.
    if (b->have_commit && check_hip(ctx, hipStreamWaitEvent(ctx->stream, b->committed, 0), "hipStreamWaitEvent(mi355_buf)")) return nullptr;
    if (check_hip(ctx, hipMemcpyAsync(b->h, b->d, b->size, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(mi355_buf D2H)")) return nullptr;
    if (check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize(mi355_buf)")) return nullptr;
    ctx->n_d2h++;
    b->state = mi355_buf::kInSync;
  } else if (b->have_commit) {
    // in sync or host newer: nothing to fetch, but device work that READS the buffer must not be overtaken by a host write
    if ((flags & MI355_MAP_WRITE) && check_hip(ctx, hipEventSynchronize(b->committed), "hipEventSynchronize(mi355_buf)")) return nullptr;
  }
  b->map_count++;
  b->map_flags |= flags;
  return b->h;
}

int mi355_buf_unmap_host(mi355_buf *b) {
  if (!b) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(b->mu);
  if (b->map_count <= 0) return MI355_ERR_INVALID_ARG;
  if (--b->map_count == 0) {
    if (b->map_flags & MI355_MAP_WRITE) b->state = mi355_buf::kHostNewer;
    b->map_flags = 0;
  }
  return MI355_OK;
}

int mi355_buf_state(mi355_buf *b) {
  if (!b) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> g(b->mu);
  return (int)b->state;
}

int mi355_ctx_transfer_counts(mi355_ctx *ctx, uint64_t *h2d, uint64_t *d2h) {
  if (!ctx) return MI355_ERR_INVALID_ARG;
  if (h2d) *h2d = ctx->n_h2d;
  if (d2h) *d2h = ctx->n_d2h;
  return MI355_OK;
}

}  // extern "C"
