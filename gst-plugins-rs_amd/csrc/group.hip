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

#include <chrono>
#include <condition_variable>
#include <algorithm>
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

// ---- videocompare across independent element instances (mi355_group_submit_compare): one (reference frame, frame) pair per submit
struct CmpDesc {
  mi355_ctx *ctx;
  const uint8_t *ref, *frame;
  int width, height, stride, format, algo, translucent;
  uint64_t ticket;
  hipEvent_t ready;  // recorded on ctx->stream at submit (nullptr: the stream held nothing)
};
struct CmpResult { int status; double distance; uint64_t hashes[2]; };
struct CmpBatch {
  uint64_t seq;
  std::vector<uint64_t> tickets;
  int algo, width, height;
  int lane;        // which of the compare queue's streams carries it (batches of one lane finish in seq order)
  hipEvent_t done;
  int waiters;
  void *h_block;  // pinned: Dssim [n][15] doubles / Blockhash [2 n] u64 (reference hashes, then frame hashes)
  bool collected = false;  // its values are in cmp_results (the event and the block are recycled once nobody waits inside it)
};
constexpr int kCmpMaxBatch = 32;                      // pairs per launch set (a Dssim pair keeps 44 MB of maps until its batch is reduced)
constexpr int kCmpMaxLanes = 16;                      // HIP streams the pairs of one launch set are dealt out to
constexpr size_t kCmpBlockBytes = 64 * 15 * 8 + 1024; // one pinned result block

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
  // ---- compare queue (videocompare's Dssim / Blockhash): its own context + stream, independent of the filter batches above
  // The pairs of a launch set are dealt out to `n_lanes` contexts (streams): the Dssim kernels are VALU-bound at ~80 % busy when
  // one runs alone, and a second stream's launches fill its tails and launch boundaries (32 4K pairs: 1.48 k comparisons/s on one
  // stream, 1.84 k over eight) - what the dispatcher removes is the per-comparison host round trip, not the overlap.
  mi355_ctx *alane[kCmpMaxLanes] = {nullptr};
  int n_lanes = 8;
  mi355_ctx *actx = nullptr;                 // = alane[0], created at the first compare submit
  std::vector<CmpDesc> cmp_pending;
  std::deque<CmpBatch> cmp_batches;          // launched, oldest first
  std::unordered_map<uint64_t, uint64_t> cmp_where;      // ticket -> seq
  std::unordered_map<uint64_t, CmpResult> cmp_results;   // finished, not yet collected by mi355_group_wait_compare
  std::vector<void *> cmp_blocks;            // free pinned result blocks
  uint32_t *d_hash_sums[kCmpMaxLanes] = {nullptr};   // Blockhash scratch per lane: [2 kCmpMaxBatch][64] u32 + [2 kCmpMaxBatch] u64
  uint64_t next_cmp_seq = 1;
  uint64_t n_cmp_pairs = 0, n_cmp_batches = 0, n_cmp_largest = 0;
  int expected_streams = 0;                  // rendezvous: a waiter lingers until this many pairs are pending ...
  unsigned linger_us = 0;                    // ... or this long (mi355_group_set_rendezvous)
  std::condition_variable cv;                // "a compare batch has been launched"
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
      if (!rc) rc = launch_colorlut_multi(first.ctx, g->stream, first.table, srcs, dsts, (int)take.size(), first.width, first.height, first.fused);
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

// ------------------------------------------------------------------ compare queue

void *cmp_take_block(mi355_group *g) {
  if (!g->cmp_blocks.empty()) { void *b = g->cmp_blocks.back(); g->cmp_blocks.pop_back(); return b; }
  void *b = nullptr;
  if (hipHostMalloc(&b, kCmpBlockBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return b;
}

bool cmp_same_class(const CmpDesc &a, const CmpDesc &b) {
  return a.algo == b.algo && a.width == b.width && a.height == b.height && a.stride == b.stride && a.format == b.format && a.translucent == b.translucent;
}

// a finished batch: values from its pinned block into cmp_results (once). g->mu held.
void cmp_collect(mi355_group *g, CmpBatch &b) {
  if (b.collected) return;
  b.collected = true;
  const int n = (int)b.tickets.size();
  if (b.algo == MI355_HASH_DSSIM) {
    std::vector<double> v((size_t)n);
    dssim_scores_from_slots(b.width, b.height, (const double *)b.h_block, n, v.data());
    for (int i = 0; i < n; i++) g->cmp_results[b.tickets[i]] = CmpResult{MI355_OK, v[i], {0, 0}};
  } else {
    const uint64_t *h = (const uint64_t *)b.h_block;
    for (int i = 0; i < n; i++)
      g->cmp_results[b.tickets[i]] = CmpResult{MI355_OK, mi355_videocompare_distance(MI355_HASH_BLOCKHASH, h[i], h[n + i]), {h[i], h[n + i]}};
  }
  for (uint64_t t : b.tickets) g->cmp_where.erase(t);
}

// collected batches nobody waits inside leave from the front: event and pinned block back to their free lists. g->mu held.
void cmp_retire(mi355_group *g) {
  for (auto it = g->cmp_batches.begin(); it != g->cmp_batches.end();) {
    if (it->collected && it->waiters == 0) {
      g->cmp_blocks.push_back(it->h_block);
      g->events.push_back(it->done);
      it = g->cmp_batches.erase(it);
    } else {
      ++it;
    }
  }
}

// launches the pending pairs, class by class (all of them, or up to the batch that carries `until`): the pairs of a class are
// dealt out to the lanes in contiguous shares (pairs that share a reference frame stay together). g->mu held.
int cmp_flush_locked(mi355_group *g, uint64_t until = 0) {
  bool reached = false;
  int first_rc = MI355_OK;
  while (!g->cmp_pending.empty() && !reached) {
    std::vector<CmpDesc> all, keep;
    const CmpDesc first = g->cmp_pending.front();
    for (const CmpDesc &d : g->cmp_pending) {
      if ((int)all.size() < kCmpMaxBatch && cmp_same_class(first, d)) all.push_back(d);
      else keep.push_back(d);
    }
    g->cmp_pending.swap(keep);
    // pairs that share a reference frame next to each other (videocompare with several pads: the reference is hashed once)
    std::stable_sort(all.begin(), all.end(), [](const CmpDesc &a, const CmpDesc &b) { return (uintptr_t)a.ref < (uintptr_t)b.ref; });
    const int total = (int)all.size();
    const int lanes = total < g->n_lanes ? total : g->n_lanes;
    PixFmt fmt;
    (void)pixfmt_of(first.format, &fmt);
    int begin = 0;
    for (int lane = 0; lane < lanes; lane++) {
      int end = (int)((long long)total * (lane + 1) / lanes);
      // a reference frame is not split over two lanes
      while (end < total && end > begin && all[end].ref == all[end - 1].ref) end++;
      if (end <= begin) continue;
      std::vector<CmpDesc> take(all.begin() + begin, all.begin() + end);
      begin = end;
      const int n = (int)take.size();
      mi355_ctx *a = g->alane[lane];
      int rc = MI355_OK;
      hipEvent_t done = take_event(g);
      void *block = cmp_take_block(g);
      if (!done || !block) rc = MI355_ERR_HIP;
      for (const CmpDesc &d : take)
        if (!rc && d.ready && hipStreamWaitEvent(a->stream, d.ready, 0) != hipSuccess) rc = MI355_ERR_HIP;
      std::vector<const uint8_t *> refs((size_t)n), frames((size_t)n);
      for (int i = 0; i < n; i++) { refs[i] = take[i].ref; frames[i] = take[i].frame; }
      if (!rc && first.algo == MI355_HASH_DSSIM) {
        a->dssim_translucent = first.translucent;
        rc = dssim_compare_pairs_enqueue(a, refs.data(), frames.data(), n, first.stride, first.width, first.height, fmt.pixel_stride, (double *)block);
      } else if (!rc) {
        std::vector<const uint8_t *> both(refs);
        both.insert(both.end(), frames.begin(), frames.end());
        unsigned long long *d_hashes = (unsigned long long *)(g->d_hash_sums[lane] + (size_t)2 * kCmpMaxBatch * 64);
        rc = blockhash_enqueue(a, both.data(), 2 * n, first.stride, first.width, first.height, fmt.pixel_stride, g->d_hash_sums[lane], d_hashes);
        if (!rc && hipMemcpyAsync(block, d_hashes, (size_t)2 * n * 8, hipMemcpyDeviceToHost, a->stream) != hipSuccess) rc = MI355_ERR_HIP;
      }
      if (!rc && hipEventRecord(done, a->stream) != hipSuccess) rc = MI355_ERR_HIP;
      if (rc) {
        (void)hipGetLastError();
        if (done) g->events.push_back(done);
        if (block) g->cmp_blocks.push_back(block);
        g->last_error = a->last_error.empty() ? "group: compare launch failed" : a->last_error;
        if (g->cmp_results.size() > 65536) g->cmp_results.clear();   // (results nobody ever collected)
        for (const CmpDesc &d : take) g->cmp_results[d.ticket] = CmpResult{rc, 0.0, {0, 0}};   // told to the pair's own wait, once
        if (!first_rc) first_rc = rc;
        continue;
      }
      CmpBatch b{g->next_cmp_seq++, {}, first.algo, first.width, first.height, lane, done, 0, block};
      for (const CmpDesc &d : take) { b.tickets.push_back(d.ticket); g->cmp_where[d.ticket] = b.seq; reached |= until != 0 && d.ticket == until; }
      g->cmp_batches.push_back(std::move(b));
    }
    for (const CmpDesc &d : all)
      if (d.ready) g->events.push_back(d.ready);
    g->n_cmp_pairs += (uint64_t)total;
    g->n_cmp_batches++;
    if ((uint64_t)total > g->n_cmp_largest) g->n_cmp_largest = (uint64_t)total;
  }
  g->cv.notify_all();
  return first_rc;
}

// finished batches are collected without a waiter (their results stay in cmp_results until asked for)
void cmp_retire_done(mi355_group *g) {
  for (CmpBatch &b : g->cmp_batches) {
    if (b.collected) continue;
    if (hipEventQuery(b.done) != hipSuccess) { (void)hipGetLastError(); continue; }   // (the lanes finish independently)
    cmp_collect(g, b);
  }
  cmp_retire(g);
}

// host wait for the batch of `ticket`; `lk` owns g->mu on entry and on return, not while waiting
int cmp_wait_unlocking(mi355_group *g, std::unique_lock<std::mutex> &lk, uint64_t ticket) {
  auto it = g->cmp_where.find(ticket);
  if (it == g->cmp_where.end()) return MI355_OK;  // collected already (or failed: cmp_results has it)
  const uint64_t seq = it->second;
  CmpBatch *mine = nullptr;
  for (CmpBatch &b : g->cmp_batches)
    if (b.seq == seq) { mine = &b; break; }
  if (!mine) return MI355_OK;
  const hipEvent_t ev = mine->done;
  const int lane = mine->lane;
  mine->waiters++;
  lk.unlock();
  const hipError_t e = hipEventSynchronize(ev);
  lk.lock();
  for (CmpBatch &b : g->cmp_batches)
    if (b.seq == seq) { b.waiters--; break; }
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipEventSynchronize(group compare batch)"); }
  // a lane's stream is in order: everything of that lane up to this batch is done
  for (CmpBatch &b : g->cmp_batches)
    if (b.lane == lane && b.seq <= seq) cmp_collect(g, b);
  cmp_retire(g);
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
  if (g->actx) {
    // pairs still pending are launched and waited for: their frames belong to callers who may free them once this returns
    {
      Locked L(g);
      (void)cmp_flush_locked(g);
    }
    for (int l = 0; l < kCmpMaxLanes; l++)
      if (g->alane[l]) (void)hipStreamSynchronize(g->alane[l]->stream);
    for (CmpBatch &b : g->cmp_batches) { (void)hipEventDestroy(b.done); (void)hipHostFree(b.h_block); }
    g->cmp_batches.clear();
    for (void *b : g->cmp_blocks) (void)hipHostFree(b);
    for (int l = 0; l < kCmpMaxLanes; l++) {
      if (g->d_hash_sums[l]) (void)hipFree(g->d_hash_sums[l]);
      if (g->alane[l]) mi355_ctx_destroy(g->alane[l]);
    }
  }
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
  const int rc = flush_locked(g);
  const int rc2 = g->actx ? cmp_flush_locked(g) : MI355_OK;
  return rc ? rc : rc2;
}

// ---------------------------------------------------------------- videocompare pairs (Dssim / Blockhash) of independent elements

int mi355_group_set_rendezvous(mi355_group *g, int expected_streams, unsigned linger_us) {
  if (!g || expected_streams < 0) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  g->expected_streams = expected_streams;
  g->linger_us = linger_us;
  return MI355_OK;
}

int mi355_group_set_compare_lanes(mi355_group *g, int lanes) {
  if (!g || lanes < 1 || lanes > kCmpMaxLanes) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  if (g->actx) return fail(g, MI355_ERR_INVALID_ARG, "group: the compare queue's streams exist already (set the lanes before the first submit_compare)");
  g->n_lanes = lanes;
  return MI355_OK;
}

int mi355_group_submit_compare(mi355_group *g, mi355_ctx *ctx, const uint8_t *d_ref, const uint8_t *d_frame, int stride, int width, int height, int format,
                               int algo, uint64_t *ticket) {
  if (!g) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  PixFmt fmt;
  if (!ctx || !d_ref || !d_frame || width <= 0 || height <= 0 || !pixfmt_of(format, &fmt) || (format != MI355_FMT_RGB && format != MI355_FMT_RGBA) ||
      (size_t)stride < (size_t)width * fmt.pixel_stride)
    return fail(g, MI355_ERR_INVALID_ARG, "group: bad frame pair (videocompare's engines take packed RGB / RGBA)");
  if (algo != MI355_HASH_DSSIM && algo != MI355_HASH_BLOCKHASH)
    return fail(g, MI355_ERR_UNSUPPORTED, "group: pairs are batched for hash-algorithm dssim and blockhash; the resize hashes go through their context");
  if (algo == MI355_HASH_BLOCKHASH && (width % 8 != 0 || height % 8 != 0))
    return fail(g, MI355_ERR_UNSUPPORTED, "group: blockhash batches take frames of 8 x 8 whole blocks (the any-size path goes through its context)");
  if (ctx->device != g->device) return fail(g, MI355_ERR_INVALID_ARG, "group: context of another device");
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  if (!g->actx) {
    int st = MI355_OK;
    for (int l = 0; l < g->n_lanes && !st; l++) {
      g->alane[l] = mi355_ctx_create(g->device, &st);
      if (!g->alane[l] && !st) st = MI355_ERR_HIP;
      if (!st && hipMalloc((void **)&g->d_hash_sums[l], (size_t)2 * kCmpMaxBatch * (64 * 4 + 8)) != hipSuccess) { (void)hipGetLastError(); st = MI355_ERR_OUT_OF_MEMORY; }
    }
    if (st) {
      for (int l = 0; l < kCmpMaxLanes; l++) {
        if (g->d_hash_sums[l]) (void)hipFree(g->d_hash_sums[l]);
        if (g->alane[l]) mi355_ctx_destroy(g->alane[l]);
        g->d_hash_sums[l] = nullptr; g->alane[l] = nullptr;
      }
      return fail(g, st, "group: no contexts for the compare queue");
    }
    g->actx = g->alane[0];
  }
  cmp_retire_done(g);
  CmpDesc d{};
  d.ctx = ctx; d.ref = d_ref; d.frame = d_frame; d.width = width; d.height = height; d.stride = stride; d.format = format; d.algo = algo;
  d.translucent = ctx->dssim_translucent;
  d.ready = nullptr;
  if (hipStreamQuery(ctx->stream) != hipSuccess) {   // the pair starts after what the stream's own context holds now (an upload)
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
  g->cmp_pending.push_back(d);
  // everybody is here (rendezvous), or a launch set is full: go. The pair has been accepted whatever that launch does (a failure
  // is told to the waits of the pairs it carried).
  const int full = g->expected_streams > 0 && g->expected_streams < kCmpMaxBatch ? g->expected_streams : kCmpMaxBatch;
  if ((int)g->cmp_pending.size() >= full) (void)cmp_flush_locked(g);
  return MI355_OK;
}

int mi355_group_wait_compare(mi355_group *g, uint64_t ticket, double *distance, uint64_t hashes[2]) {
  if (!g) return MI355_ERR_INVALID_ARG;
  Locked L(g);
  std::unique_lock<std::mutex> &lk = L.lk;
  if (ticket == 0 || ticket >= g->next_ticket) return fail(g, MI355_ERR_INVALID_ARG, "group: unknown ticket");
  if (hipSetDevice(g->device) != hipSuccess) { (void)hipGetLastError(); return fail(g, MI355_ERR_HIP, "hipSetDevice"); }
  auto is_pending = [&]() { for (const CmpDesc &d : g->cmp_pending) if (d.ticket == ticket) return true; return false; };
  if (is_pending()) {
    // rendezvous: the other streams of this interval are about to submit - linger for them (bounded), then launch what is there
    if (g->expected_streams > 0 && g->linger_us > 0) {
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(g->linger_us);
      while (is_pending() && (int)g->cmp_pending.size() < g->expected_streams) {
        if (g->cv.wait_until(lk, deadline) == std::cv_status::timeout) break;
      }
    }
    if (is_pending()) (void)cmp_flush_locked(g, ticket);   // (a failure of this pair's own launch is in cmp_results)
  }
  int rc = cmp_wait_unlocking(g, lk, ticket);
  if (rc) return rc;
  auto r = g->cmp_results.find(ticket);
  if (r == g->cmp_results.end()) return fail(g, MI355_ERR_INVALID_ARG, "group: this pair's result has been collected already (or the ticket is not a pair's)");
  const CmpResult res = r->second;
  g->cmp_results.erase(r);
  if (res.status) return fail(g, res.status, "group: the launch that carried this pair failed");
  if (distance) *distance = res.distance;
  if (hashes) { hashes[0] = res.hashes[0]; hashes[1] = res.hashes[1]; }
  return MI355_OK;
}

int mi355_group_compare_stats(mi355_group *g, uint64_t stats[3]) {
  if (!g || !stats) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  stats[0] = g->n_cmp_pairs;
  stats[1] = g->n_cmp_batches;
  stats[2] = g->n_cmp_largest;
  return MI355_OK;
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

// ---------------------------------------------------------------- the process's group per device
// Elements of independent pipelines share nothing but the process: mi355_group_shared(device) is THE dispatcher of that device
// (created at first use, reference-counted: every call is paired with one mi355_group_release; the last release destroys it).
namespace {
std::mutex g_pg_mu;
struct ProcessGroup { int device; mi355_group *g; int refs; };
std::vector<ProcessGroup> g_pg;
}  // namespace

extern "C" {

mi355_group *mi355_group_shared(int device, int *status) {
  std::lock_guard<std::mutex> lk(g_pg_mu);
  for (ProcessGroup &p : g_pg)
    if (p.device == device) { p.refs++; if (status) *status = MI355_OK; return p.g; }
  mi355_group *g = mi355_group_create(device, 0, status);
  if (g) g_pg.push_back(ProcessGroup{device, g, 1});
  return g;
}

void mi355_group_release(mi355_group *g) {
  if (!g) return;
  bool last = false;
  {
    std::lock_guard<std::mutex> lk(g_pg_mu);
    for (size_t i = 0; i < g_pg.size(); i++)
      if (g_pg[i].g == g) {
        last = --g_pg[i].refs == 0;
        if (last) g_pg.erase(g_pg.begin() + (std::ptrdiff_t)i);
        break;
      }
  }
  if (last) mi355_group_destroy(g);
}

}  // extern "C"
