// group.hip — frames of MANY streams in few launches: the dispatcher behind mi355_group_*.
//
// GstBaseTransform hands every element ONE buffer per call (video/hsv/src/hsvfilter/imp.rs:323-376,
// video/colorlut/src/colorlut/imp.rs:203-223), so N streams through `hsvfilter ! colorlut` are 2 N launches per frame period,
// each with its own ramp and tail (a one-frame hsvfilter launch: 16.5 us against 10.7 us for an eighth of an eight-frame
// launch; 32 streams: 34 k frames/s against 45 k batched, profiles/r03_launch_size.txt). A group collects what the streams
// submit - {stream's context, frame in, frame out, geometry, hsv settings of that frame} - and issues it as multi-frame
// launches: the frames of up to `max_batch` streams that agree in geometry, format, settings and LUT go to ONE hsvfilter launch
// and ONE colorlut launch (blockIdx -> frame -> base pointer from the kernel arguments; hsv_kernels.hip, colorlut_kernels.hip).
//   * submit never blocks and never launches on its own unless max_batch frames are pending;
//   * mi355_group_wait(ticket) flushes what is pending if that ticket has not been launched yet, then waits for its batch -
//     an element that works one frame deep (submit frame n, wait for frame n-1: gst/gstcolorlut.c) therefore fills batches
//     with the frames the other streams submitted in between, and a lone stream degenerates to today's two launches;
//   * order: batches run in submission order on the group's own HIP stream and hold at most one frame per stream, so frames of
//     one stream run in order; a frame starts after the work its context's stream held at submit time (an event, only if the
//     stream held any); what the context's stream does NEXT is ordered after a frame by mi355_group_order_after (one stream
//     wait, asked for by who needs it: a download) or by the host wait - not by default: eight barrier packets behind every
//     batch cost more than the batching saves (32 streams: 31 k frames/s with them, see DESIGN);
//   * results are the two element launches', bit for bit (same arithmetic, same table);
//   * frames the batched kernels do not take (padded rows, 3-byte formats, no table) run through their context's own path,
//     in order.
// No persistent kernel: nothing here can hang the GPU waiting for the host, a launch is a launch.
#include "internal.hpp"

#include <cstring>
#include <deque>
#include <mutex>
#include <unordered_map>
#include <vector>

using namespace mi355;

namespace {

struct Desc {
  mi355_ctx *ctx;
  uint8_t *src, *dst;
  int width, height, stride, format;
  mi355_hsv_settings hs;
  const uint32_t *table;  // the context's memoised colorlut table (nullptr: not batchable)
  // ... and a reference to it, held until the frame's batch has retired: the context may move on to other settings (or another
  // LUT) before this frame has been launched or has finished, and a table somebody else still refers to is never rebuilt in place
  // (colorlut_kernels.hip: shared_table_acquire takes that branch only for use_count() == 1) nor freed
  std::shared_ptr<void> table_ref;
  bool batchable;
  bool fused;             // one launch through the composed hsv+lut table, source left untouched (mi355_group_submit_fused)
  uint64_t ticket;
  hipEvent_t ready;       // recorded on ctx->stream at submit
};

struct Batch {
  uint64_t seq;                   // launch order on the group's stream: batch n is done => every batch before it is
  std::vector<uint64_t> tickets;  // the frames it carries (not a contiguous range: an incompatible frame waits for the next batch)
  hipEvent_t done;
  int waiters;  // threads inside hipEventSynchronize(done) right now: the event is not recycled under them
  std::shared_ptr<void> table_ref;  // the table its launches read (see Desc)
};

}  // namespace

struct mi355_group {
  int device = 0;
  int max_batch = 8;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::vector<Desc> pending;
  std::deque<Batch> batches;       // launched, oldest first
  std::vector<hipEvent_t> events;  // free list
  std::unordered_map<uint64_t, uint64_t> where;  // ticket -> seq of its batch, for launched batches not yet retired
  std::unordered_map<uint64_t, int> failed;      // ticket -> status of the launch that did not happen (reported by wait / order_after)
  uint64_t next_ticket = 1, next_seq = 1;
  uint64_t n_frames = 0, n_batched_launch_pairs = 0, n_single = 0;
  std::string last_error;
  // table references of retired batches: dropping the last reference to a table frees 64 MiB behind a device-wide wait, which
  // does not belong under `mu` - the entry points empty this list into a local one that dies after they have unlocked (Locked)
  std::vector<std::shared_ptr<void>> dead_refs;
};

namespace {

// g->mu for the scope, and behind it the table references the scope retired (destroyed last, i.e. outside the lock)
struct Locked {
  std::vector<std::shared_ptr<void>> reap;
  mi355_group *g;
  std::unique_lock<std::mutex> lk;
  explicit Locked(mi355_group *g_) : g(g_), lk(g_->mu) {}
  ~Locked() {
    if (!lk.owns_lock()) lk.lock();
    reap.swap(g->dead_refs);
    lk.unlock();
  }
};

void retire_front(mi355_group *g) {
  Batch &b = g->batches.front();
  for (uint64_t t : b.tickets) g->where.erase(t);
  g->events.push_back(b.done);
  if (b.table_ref) g->dead_refs.push_back(std::move(b.table_ref));
  g->batches.pop_front();
}

// batches that have finished leave without anybody waiting for them (a stream that only ever orders its own stream behind its
// frames never waits here; its batches, and the table references they hold, must not pile up)
void retire_done(mi355_group *g) {
  while (!g->batches.empty() && g->batches.front().waiters == 0) {
    if (hipEventQuery(g->batches.front().done) != hipSuccess) { (void)hipGetLastError(); break; }
    retire_front(g);
  }
}

hipEvent_t take_event(mi355_group *g) {
  if (!g->events.empty()) { hipEvent_t e = g->events.back(); g->events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}

int fail(mi355_group *g, int status, const std::string &msg) {
  g->last_error = msg;
  return status;
}

bool same_class(const Desc &a, const Desc &b) {
  return a.batchable && b.batchable && a.fused == b.fused && a.width == b.width && a.height == b.height && a.stride == b.stride && a.format == b.format && a.table == b.table &&
         std::memcmp(&a.hs, &b.hs, sizeof(a.hs)) == 0 && a.ctx->force_generic == b.ctx->force_generic;
}

// launches what is pending, batch by batch - all of it, or (until != 0) only up to the batch that carries ticket `until`: what a
// waiter needs; the frames behind it stay and meet the other streams' next frames in a fuller batch. g->mu held.
int flush_locked(mi355_group *g, uint64_t until = 0) {
  bool reached = false;
  while (!g->pending.empty() && !reached) {
    std::vector<Desc> take, keep;
    std::vector<mi355_ctx *> blocked;  // streams with a frame left behind: their later frames must not overtake it
    const Desc first = g->pending.front();
    for (const Desc &d : g->pending) {
      bool stream_blocked = false;
      for (mi355_ctx *c : blocked) stream_blocked |= c == d.ctx;
      const bool fits = take.empty() || (!stream_blocked && (int)take.size() < g->max_batch && same_class(first, d));
      if (fits) take.push_back(d);
      else keep.push_back(d);
      // ONE frame per stream and batch: the hsvfilter launch of a batch runs over all its frames before the colorlut launch
      // does, so a frame that reads what the stream's previous frame wrote (or simply comes after it) belongs to a later batch
      if (!stream_blocked) blocked.push_back(d.ctx);
    }
    g->pending.swap(keep);
    PixFmt fmt;
    (void)pixfmt_of(first.format, &fmt);
    hipEvent_t done = take_event(g);
    if (!done) return fail(g, MI355_ERR_HIP, "hipEventCreate");
    int rc = MI355_OK;
    if (first.batchable) {
      uint8_t *srcs[kMultiFrames], *dsts[kMultiFrames];
      for (size_t i = 0; i < take.size(); i++) {
        srcs[i] = take[i].src;
        dsts[i] = take[i].dst;
        if (take[i].ready && hipStreamWaitEvent(g->stream, take[i].ready, 0) != hipSuccess) rc = MI355_ERR_HIP;
      }
      // (fused: `table` is the composed hsvfilter -> colorlut table of these settings: the gather IS the chain, one launch)
      if (!rc && !first.fused) rc = launch_hsvfilter_multi(first.ctx, g->stream, srcs, (int)take.size(), first.width, first.height, fmt, first.hs);
      if (!rc) rc = launch_colorlut_multi(first.ctx, g->stream, first.table, srcs, dsts, (int)take.size(), first.width, first.height);
      if (!rc && hipEventRecord(done, g->stream) != hipSuccess) rc = MI355_ERR_HIP;
      g->n_batched_launch_pairs++;
    } else {
      // not the batched kernels' geometry: this frame through its context's own path, on its own stream, after everything
      // launched so far (order), and the group's stream after it
      const Desc &d = take[0];
      const size_t pitch = (size_t)d.stride * (size_t)d.height;
      if (!g->batches.empty() && hipStreamWaitEvent(d.ctx->stream, g->batches.back().done, 0) != hipSuccess) rc = MI355_ERR_HIP;
      if (!rc && d.fused) rc = launch_hsv_colorlut(d.ctx, d.src, pitch, d.stride, d.dst, pitch, d.stride, 1, d.width, d.height, d.hs);
      if (!rc && !d.fused) rc = launch_hsvfilter(d.ctx, d.src, 1, pitch, d.width, d.height, d.stride, fmt, d.hs);
      if (!rc && !d.fused) rc = launch_colorlut(d.ctx, d.src, pitch, d.stride, d.dst, pitch, d.stride, 1, d.width, d.height, d.format);
      if (!rc && hipEventRecord(done, d.ctx->stream) != hipSuccess) rc = MI355_ERR_HIP;
      if (!rc && hipStreamWaitEvent(g->stream, done, 0) != hipSuccess) rc = MI355_ERR_HIP;
      if (rc && rc != MI355_ERR_HIP) g->last_error = d.ctx->last_error;
      g->n_single++;
    }
    for (const Desc &d : take)
      if (d.ready) g->events.push_back(d.ready);
    if (rc) {
      // the frames of this batch have left `pending` and will never be in `where`: whoever waits for one of them is told
      (void)hipGetLastError();
      g->events.push_back(done);
      if (g->failed.size() > 65536) g->failed.clear();  // (tickets nobody ever waited for)
      for (const Desc &d : take) g->failed[d.ticket] = rc;
      if (g->last_error.empty()) g->last_error = "group launch failed";
      return rc;
    }
    Batch b{g->next_seq++, {}, done, 0, first.batchable ? first.table_ref : std::shared_ptr<void>()};
    for (const Desc &d : take) { b.tickets.push_back(d.ticket); g->where[d.ticket] = b.seq; reached |= until != 0 && d.ticket == until; }
    g->batches.push_back(std::move(b));
    g->n_frames += take.size();
  }
  return MI355_OK;
}

// Waits (on the host) for the batch that holds `ticket` - and with it, the stream being in order, for every earlier one. The
// lock is NOT held while waiting: other streams' threads keep submitting. `lk` owns g->mu on entry and on return.
int wait_unlocking(mi355_group *g, std::unique_lock<std::mutex> &lk, uint64_t ticket) {
  auto bad = g->failed.find(ticket);
  if (bad != g->failed.end()) {
    const int rc = bad->second;
    g->failed.erase(bad);
    return fail(g, rc, "group: the launch that carried this frame failed");
  }
  auto it = g->where.find(ticket);
  if (it == g->where.end()) return MI355_OK;  // launched and already retired (by this or another waiter)
  const uint64_t seq = it->second;
  Batch *mine = nullptr;
  for (Batch &b : g->batches)
    if (b.seq == seq) { mine = &b; break; }
  if (!mine) return MI355_OK;
  const hipEvent_t ev = mine->done;
  mine->waiters++;
  lk.unlock();
  const hipError_t e = hipEventSynchronize(ev);
  lk.lock();
  for (Batch &b : g->batches)
    if (b.seq == seq) { b.waiters--; break; }
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipEventSynchronize(group batch)"); }
  // everything up to that batch is done: retire from the front (batches somebody still waits in stay until they leave)
  while (!g->batches.empty() && g->batches.front().seq <= seq && g->batches.front().waiters == 0) retire_front(g);
  return MI355_OK;
}

// waits for everything launched so far
int wait_all_unlocking(mi355_group *g, std::unique_lock<std::mutex> &lk) {
  while (!g->batches.empty()) {
    const uint64_t t = g->batches.back().tickets.back();
    const size_t before = g->batches.size();
    int rc = wait_unlocking(g, lk, t);
    if (rc) return rc;
    if (!g->batches.empty() && g->batches.size() >= before && g->batches.back().tickets.back() == t) break;  // held by other waiters: done anyway
  }
  return MI355_OK;
}

}  // namespace

extern "C" {

mi355_group *mi355_group_create(int device, int max_batch, int *status) {
  if (max_batch < 0 || max_batch > kMultiFrames) { if (status) *status = MI355_ERR_INVALID_ARG; return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); if (status) *status = MI355_ERR_NO_DEVICE; return nullptr; }
  mi355_group *g = new mi355_group();
  g->device = device;
  g->max_batch = max_batch ? max_batch : 8;  // eight 4K frames: what one launch should hold to stay inside the Infinity Cache
  if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    delete g;
    if (status) *status = MI355_ERR_HIP;
    return nullptr;
  }
  if (status) *status = MI355_OK;
  return g;
}

void mi355_group_destroy(mi355_group *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  {
    Locked L(g);
    (void)flush_locked(g);
    (void)wait_all_unlocking(g, L.lk);
  }
  (void)hipStreamSynchronize(g->stream);
  for (Batch &b : g->batches) (void)hipEventDestroy(b.done);   // (normally none left: wait_all retired them)
  for (Desc &d : g->pending)
    if (d.ready) (void)hipEventDestroy(d.ready);
  g->pending.clear();
  g->batches.clear();  // their table references go here, after the stream has drained
  g->dead_refs.clear();
  for (hipEvent_t e : g->events) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(g->stream);
  delete g;
}

const char *mi355_group_last_error(mi355_group *g) { return g ? g->last_error.c_str() : "null group"; }

static int submit_frame(mi355_group *g, mi355_ctx *ctx, uint8_t *d_src, uint8_t *d_dst, int width, int height, int stride, int format,
                        const mi355_hsv_settings *settings, uint64_t *ticket, bool fused) {
  if (!g) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  PixFmt fmt;
  if (!ctx || !d_src || !d_dst || !settings || width <= 0 || height <= 0 || !pixfmt_of(format, &fmt) || (size_t)stride < (size_t)width * fmt.pixel_stride)
    return fail(g, MI355_ERR_INVALID_ARG, "group: bad frame");
  // the chain exists for the one format both elements accept (hsvfilter/imp.rs:252-266, colorlut/imp.rs:125-137): refused here,
  // not as a failed launch of a whole batch later
  if (format != MI355_FMT_RGBA) return fail(g, MI355_ERR_INVALID_ARG, "group: hsvfilter ! colorlut takes RGBA frames only");
  if (ctx->device != g->device) return fail(g, MI355_ERR_INVALID_ARG, "group: context of another device");
  if (!ctx->lut.loaded) return fail(g, MI355_ERR_NOT_CONFIGURED, "No LUT configured");
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  Desc d{};
  d.ctx = ctx; d.src = d_src; d.dst = d_dst; d.width = width; d.height = height; d.stride = stride; d.format = format; d.hs = *settings;
  d.fused = fused;
  retire_done(g);
  // a composed table is 64 MiB built by two launches over 2^24 colours: worth it for settings that stay (the fused entry point
  // has the same rule: eight calls), not for a hue shift animated frame by frame - those frames take their context's own path
  bool settled = true;
  if (fused) {
    if (std::memcmp(&ctx->group_fused_hs, settings, sizeof(*settings)) == 0) { if (ctx->group_fused_stable < 1000000u) ctx->group_fused_stable++; }
    else { ctx->group_fused_hs = *settings; ctx->group_fused_stable = 1; }
    settled = ctx->group_fused_stable >= 8u;
  }
  const uint8_t *one[1] = {d_src};
  // (the two-launch form filters d_src in place and then reads it: not onto itself; the fused form reads a tile and writes the same tile)
  d.batchable = settled && format == MI355_FMT_RGBA && (fused || d_src != d_dst) && hsvfilter_multi_applicable(one, 1, width, height, stride, fmt) && width % 4 == 0 &&
                width >= 128 && (uintptr_t)d_dst % 16 == 0 && ctx->lut_variant == 0 && ctx->hsv_table_mode != 2 && (!fused || ctx->lut.is3d);
  if (d.batchable && (fused ? colorlut_multi_fused_table(ctx, settings, &d.table) : colorlut_multi_table(ctx, &d.table)) != MI355_OK) {  // (the build, if any, is on ctx->stream: before `ready`)
    (void)hipGetLastError();
    d.batchable = false;
    d.table = nullptr;
  }
  // (this thread is the one that drives ctx, and the reference is copied from ctx's own: the registry's "sole user" test cannot
  // run concurrently with this copy)
  if (d.batchable) d.table_ref = ctx->lut.table_ref[fused ? 1 : 0];
  // The frame starts after what ctx's stream holds now (an upload, the table build). A stream that holds nothing - the common
  // case for a stream that only ever submits here - needs no event: every cross-stream wait is a barrier packet the command
  // processor resolves in microseconds, eight of them in front of a 90 us launch are a bubble.
  d.ready = nullptr;
  if (hipStreamQuery(ctx->stream) != hipSuccess) {
    (void)hipGetLastError();
    d.ready = take_event(g);
    if (!d.ready || hipEventRecord(d.ready, ctx->stream) != hipSuccess) {
      (void)hipGetLastError();
      if (d.ready) g->events.push_back(d.ready);
      return fail(g, MI355_ERR_HIP, "group: hipEventRecord(ready)");
    }
  }
  d.ticket = g->next_ticket++;
  if (ticket) *ticket = d.ticket;
  g->pending.push_back(d);
  // enough for a full launch: the batch of the oldest pending frame goes now (the rest keeps collecting). This frame has been
  // accepted whatever that launch does: a failed batch is reported to the frames it carried (g->failed: their wait / order_after),
  // which may or may not include this one - never as a refusal of this submit, whose caller would then reuse buffers the group
  // still refers to.
  if ((int)g->pending.size() >= g->max_batch) (void)flush_locked(g, g->pending.front().ticket);
  return MI355_OK;
}

int mi355_group_submit_chain(mi355_group *g, mi355_ctx *ctx, uint8_t *d_src, uint8_t *d_dst, int width, int height, int stride, int format,
                             const mi355_hsv_settings *settings, uint64_t *ticket) {
  return submit_frame(g, ctx, d_src, d_dst, width, height, stride, format, settings, ticket, false);
}

int mi355_group_submit_fused(mi355_group *g, mi355_ctx *ctx, uint8_t *d_src, uint8_t *d_dst, int width, int height, int stride, int format,
                             const mi355_hsv_settings *settings, uint64_t *ticket) {
  return submit_frame(g, ctx, d_src, d_dst, width, height, stride, format, settings, ticket, true);
}

int mi355_group_flush(mi355_group *g) {
  if (!g) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  return flush_locked(g);
}

int mi355_group_wait(mi355_group *g, uint64_t ticket) {
  if (!g) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  std::unique_lock<std::mutex> &lk = L.lk;
  if (ticket == 0 || ticket >= g->next_ticket) return fail(g, MI355_ERR_INVALID_ARG, "group: unknown ticket");
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  bool is_pending = false;
  for (const Desc &d : g->pending) is_pending |= d.ticket == ticket;
  if (is_pending) {
    int rc = flush_locked(g, ticket);
    if (rc && g->failed.find(ticket) == g->failed.end()) return rc;  // (a failure of this frame's own batch is reported - once - below)
  }
  return wait_unlocking(g, lk, ticket);
}

int mi355_group_order_after(mi355_group *g, mi355_ctx *ctx, uint64_t ticket) {
  if (!g || !ctx) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  if (ticket == 0 || ticket >= g->next_ticket) return fail(g, MI355_ERR_INVALID_ARG, "group: unknown ticket");
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  bool is_pending = false;
  for (const Desc &d : g->pending) is_pending |= d.ticket == ticket;
  if (is_pending) {
    int rc = flush_locked(g, ticket);
    if (rc) return rc;
  }
  auto bad = g->failed.find(ticket);
  if (bad != g->failed.end()) return fail(g, bad->second, "group: the launch that carried this frame failed");
  auto it = g->where.find(ticket);
  if (it == g->where.end()) return MI355_OK;  // retired: the frame is done, nothing to order
  for (Batch &b : g->batches)
    if (b.seq == it->second) {
      if (hipStreamWaitEvent(ctx->stream, b.done, 0) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipStreamWaitEvent(group batch)"); }
      break;
    }
  return MI355_OK;
}

int mi355_group_wait_all(mi355_group *g) {
  if (!g) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  std::unique_lock<std::mutex> &lk = L.lk;
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  int rc = flush_locked(g);
  if (rc) return rc;
  return wait_all_unlocking(g, lk);
}

int mi355_group_submit_round(mi355_group *g, mi355_ctx *const *ctxs, int n_streams, uint8_t *const *d_src, uint8_t *const *d_dst, int width, int height,
                             int stride, int format, const mi355_hsv_settings *settings) {
  if (!g || !ctxs || !d_src || !d_dst || n_streams < 0) return MI355_ERR_INVALID_ARG;
  for (int i = 0; i < n_streams; i++) {
    int rc = mi355_group_submit_chain(g, ctxs[i], d_src[i], d_dst[i], width, height, stride, format, settings, nullptr);
    if (rc) return rc;
  }
  return mi355_group_flush(g);
}

int mi355_group_submit_round_fused(mi355_group *g, mi355_ctx *const *ctxs, int n_streams, uint8_t *const *d_src, uint8_t *const *d_dst, int width,
                                   int height, int stride, int format, const mi355_hsv_settings *settings) {
  if (!g || !ctxs || !d_src || !d_dst || n_streams < 0) return MI355_ERR_INVALID_ARG;
  for (int i = 0; i < n_streams; i++) {
    int rc = mi355_group_submit_fused(g, ctxs[i], d_src[i], d_dst[i], width, height, stride, format, settings, nullptr);
    if (rc) return rc;
  }
  return mi355_group_flush(g);
}

int mi355_group_stats(mi355_group *g, uint64_t stats[3]) {
  if (!g || !stats) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  stats[0] = g->n_frames;
  stats[1] = g->n_batched_launch_pairs;
  stats[2] = g->n_single;
  return MI355_OK;
}

}  // extern "C"
