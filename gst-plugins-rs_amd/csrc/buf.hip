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

// Who still uses the buffer on the device: one entry per distinct stream since the last time everybody was waited for. Two
// contexts may READ one buffer concurrently (a tee into two mi355 branches): a single "last commit" event would let the second
// commit hide the first reader from a later writer or from the free.
struct BufUse {
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  bool recorded = false;   // ev marks the end of this stream's last committed use
  bool writes = false;     // that use (or one since) wrote the buffer
};
constexpr int kBufUses = 4;

struct mi355_buf {
  mi355_ctx *owner = nullptr;
  uint8_t *d = nullptr;
  uint8_t *h = nullptr;  // pinned shadow, allocated at the first host map
  size_t size = 0;
  enum State { kInSync = 0, kHostNewer = 1, kDeviceNewer = 2 } state = kInSync;
  BufUse use[kBufUses];
  hipEvent_t uploaded = nullptr;    // last H2D from the shadow: a host WRITE map waits for it
  bool upload_pending = false;
  int map_flags = 0, map_count = 0;
  std::atomic<int> refs{1};
  std::mutex mu;
};

namespace {

// a HIP call made on behalf of a buffer from a thread that may not be the owner context's: never writes owner->last_error
bool buf_ok(hipError_t e) {
  if (e == hipSuccess) return true;
  (void)hipGetLastError();
  return false;
}

// `stream` is ordered (on the device) behind the committed uses of the other streams: all of them before a write, the writers
// before a read. b->mu held.
bool order_behind(mi355_buf *b, hipStream_t stream, bool for_write) {
  for (BufUse &u : b->use)
    if (u.recorded && u.stream != stream && (for_write || u.writes) && !buf_ok(hipStreamWaitEvent(stream, u.ev, 0))) return false;
  return true;
}

// the calling thread waits for the committed uses (all, or the writers only); entries that have been waited for are cleared
bool host_wait(mi355_buf *b, bool all) {
  for (BufUse &u : b->use)
    if (u.recorded && (all || u.writes)) {
      if (!buf_ok(hipEventSynchronize(u.ev))) return false;
      u.recorded = false; u.writes = false; u.stream = nullptr;
    }
  return true;
}

BufUse *use_of(mi355_buf *b, hipStream_t stream) {
  for (BufUse &u : b->use)
    if (u.stream == stream) return &u;
  for (BufUse &u : b->use)
    if (!u.stream) { u.stream = stream; return &u; }
  // more distinct streams than slots: the oldest entry is folded into this stream (it waits for that use, then takes the slot)
  BufUse &u = b->use[0];
  if (u.recorded && !buf_ok(hipStreamWaitEvent(stream, u.ev, 0))) return nullptr;
  u.stream = stream; u.recorded = false;   // (writes stays: the chain behind this stream now includes that write)
  return &u;
}

}  // namespace

using namespace mi355;

extern "C" {

mi355_buf *mi355_buf_alloc(mi355_ctx *ctx, size_t size) {
  if (!ctx) return nullptr;
  if (check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
  mi355_buf *b = new mi355_buf();
  b->owner = ctx;
  b->size = size;
  bool bad = check_hip(ctx, hipMalloc((void **)&b->d, size ? (size + 15) & ~(size_t)15 : 16), "hipMalloc(mi355_buf)") ||
             check_hip(ctx, hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming), "hipEventCreate(mi355_buf)");
  for (BufUse &u : b->use) bad = bad || check_hip(ctx, hipEventCreateWithFlags(&u.ev, hipEventDisableTiming), "hipEventCreate(mi355_buf)");
  if (bad) {
    if (b->d) (void)hipFree(b->d);
    if (b->uploaded) (void)hipEventDestroy(b->uploaded);
    for (BufUse &u : b->use) if (u.ev) (void)hipEventDestroy(u.ev);
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
  (void)host_wait(b, true);  // EVERY stream that used it has finished: nothing reads or writes it any more
  if (b->upload_pending) (void)hipEventSynchronize(b->uploaded);
  if (b->d) (void)hipFree(b->d);
  if (b->h) (void)hipHostFree(b->h);
  for (BufUse &u : b->use) (void)hipEventDestroy(u.ev);
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
  // behind whatever the other contexts' streams still do with the buffer: behind all of them for a write (and for the upload
  // below, which writes it), behind the writers for a read
  const bool writes = (flags & MI355_MAP_WRITE) || (b->state == mi355_buf::kHostNewer && (flags & MI355_MAP_READ));
  if (!order_behind(b, ctx->stream, writes)) { set_error(ctx, MI355_ERR_HIP, "hipStreamWaitEvent(mi355_buf)"); return nullptr; }
  BufUse *mine = use_of(b, ctx->stream);
  if (!mine) { set_error(ctx, MI355_ERR_HIP, "hipStreamWaitEvent(mi355_buf)"); return nullptr; }
  if (writes) mine->writes = true;
  if (b->state == mi355_buf::kHostNewer) {
    // (a kernel that overwrites the whole buffer asks for WRITE alone and skips the upload)
    if (flags & MI355_MAP_READ) {
      if (check_hip(ctx, hipMemcpyAsync(b->d, b->h, b->size, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(mi355_buf H2D)")) return nullptr;
      forget_written(ctx->device, b->d, b->size);
      if (check_hip(ctx, hipEventRecord(b->uploaded, ctx->stream), "hipEventRecord(mi355_buf)")) return nullptr;
      b->upload_pending = true;
      __atomic_fetch_add(&b->owner->n_h2d, 1ull, __ATOMIC_RELAXED);
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
  BufUse *mine = use_of(b, ctx->stream);
  if (!mine) return set_error(ctx, MI355_ERR_HIP, "hipStreamWaitEvent(mi355_buf)");
  if ((rc = check_hip(ctx, hipEventRecord(mine->ev, ctx->stream), "hipEventRecord(mi355_buf commit)"))) return rc;
  mine->recorded = true;
  return MI355_OK;
}

void *mi355_buf_map_host(mi355_buf *b, int flags) {
  if (!b || !(flags & (MI355_MAP_READ | MI355_MAP_WRITE))) return nullptr;
  std::lock_guard<std::mutex> g(b->mu);
  // (any thread maps a GstMemory: nothing here writes the owner context's error string; the copy runs on the owner's stream, which
  // HIP allows from any thread, and the counters are atomic)
  mi355_ctx *ctx = b->owner;
  if (!buf_ok(hipSetDevice(ctx->device))) return nullptr;
  if (!b->h && !buf_ok(hipHostMalloc((void **)&b->h, b->size ? b->size : 1, hipHostMallocDefault))) return nullptr;
  if (b->upload_pending) {  // the shadow is still being read by an upload
    if (!buf_ok(hipEventSynchronize(b->uploaded))) return nullptr;
    b->upload_pending = false;
  }
  if (b->state == mi355_buf::kDeviceNewer) {
    // A WRITE-only map downloads too: the mapper may write part of the memory and the rest must stay what it was.
    if (!order_behind(b, ctx->stream, (flags & MI355_MAP_WRITE) != 0)) return nullptr;
    if (!buf_ok(hipMemcpyAsync(b->h, b->d, b->size, hipMemcpyDeviceToHost, ctx->stream))) return nullptr;
    if (!buf_ok(hipStreamSynchronize(ctx->stream))) return nullptr;
    __atomic_fetch_add(&ctx->n_d2h, 1ull, __ATOMIC_RELAXED);
    b->state = mi355_buf::kInSync;
    if (flags & MI355_MAP_WRITE) {   // the owner's stream waited for everybody and has drained: nobody uses the buffer any more
      for (BufUse &u : b->use) { u.recorded = false; u.writes = false; u.stream = nullptr; }
    }
  } else {
    // in sync or host newer: nothing to fetch, but device work that READS the buffer must not be overtaken by a host write
    if ((flags & MI355_MAP_WRITE) && !host_wait(b, true)) return nullptr;
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
  if (h2d) *h2d = __atomic_load_n(&ctx->n_h2d, __ATOMIC_RELAXED);
  if (d2h) *d2h = __atomic_load_n(&ctx->n_d2h, __ATOMIC_RELAXED);
  return MI355_OK;
}

}  // extern "C"
