// agroup.hip — independent AUDIO element instances that share launches: the dispatcher behind mi355_agroup_*.
//
// The reference runs one element instance per stream and hands it ONE buffer per call: rsaudioecho's transform_ip
// (audio/audiofx/src/audioecho/imp.rs:205-227), ebur128level's (audio/audiofx/src/ebur128level/imp.rs:682-745), audioloudnorm's
// sink_chain -> drain_full_frames -> State::process per 100 ms frame (audio/audiofx/src/audioloudnorm/imp.rs:1545-1586, :226-268).
// One such buffer is a few thousand samples: on a GPU a single instance is launch- and round-trip-bound (one context per
// instance: echo 0.22 ms per 32-instance interval where one CPU core needs 0.17; audioloudnorm 54x real time against 86x), while
// the `_batch` kernels of this library advance hundreds of streams per launch (echo 0.014 ms, loudnorm 1,658x aggregate). Those
// entry points need one caller that owns all streams; a process full of independent elements has none. An agroup is that caller:
//   * members = element instances of ONE kind and configuration on one device (a transcoding farm's N identical pipelines);
//   * each member submits its buffer of the interval from its own streaming thread and waits for its ticket;
//   * the batch runs when every attached member has submitted (whoever completes the set runs it), ONE launch set for all;
//   * members are independent: rsaudioecho - own ring, position, buffer size and parameters per submit (a job table);
//     ebur128level - own buffer size, 100 ms phase and `reset` (per-stream rounds in ebur128_kernels.hip); audioloudnorm - own frame
//     type and ring positions (loudnorm.hip: a launch sequence per CLASS of members that stand at the same frame type and size:
//     streams that started together are one class). A waiter that has lingered `linger_us` launches whoever is there: a member that
//     is late, paused or gone costs the others one linger, never a hang, and is never fed silence;
//   * per-member results are those of a single-instance context fed the same buffers, bit for bit (tests/test_gpu_agroup.py).
// No persistent kernel, nothing on the device waits for the host.
#include "internal.hpp"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

using namespace mi355;

namespace mi355 {

// ---------------------------------------------------------------- rsaudioecho through a job table
// The arithmetic is echo_kernels.hip's (AudioEcho::process, audioecho/imp.rs:69-85 with RingBufferIter, ring_buffer.rs:37-82:
// e = ring[read]; out = inp + intensity * e; ring[write] = inp + feedback * e, f64, unfused), per job instead of per batch slot:
// every job carries its own buffer, ring, position, length and parameters.
struct EchoJob {
  void *data;
  double *w, *ring;
  unsigned long long n, size, pos, D;
  double intensity, feedback;
  int is_f64, pad;
};

template <typename T>
__device__ __forceinline__ void echo_job_widen(const EchoJob &J) {
  const T *data = (const T *)J.data;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < J.n; i += gs) J.w[i] = (double)data[i];
}

__global__ __launch_bounds__(256) void echo_jobs_widen_kernel(const EchoJob *__restrict__ jobs) {
  const EchoJob J = jobs[blockIdx.y];
  if (J.feedback != 0.0) return;  // the chain form widens as it goes
  if (J.is_f64) echo_job_widen<double>(J); else echo_job_widen<float>(J);
}

template <typename T>
__device__ __forceinline__ void echo_job_main(const EchoJob &J) {
  T *data = (T *)J.data;
  const size_t n = J.n, size = J.size, pos = J.pos, D = J.D;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  if (J.feedback == 0.0) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
      const double e = (i < D) ? J.ring[(pos + i + size - D) % size] : J.w[i - D];
      const double out = J.w[i] + J.intensity * e;
      data[i] = (T)out;
    }
  } else {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < D && t < n; t += gs) {
      double e = J.ring[(pos + t + size - D) % size];
      for (size_t i = t; i < n; i += D) {
        const double inp = (double)data[i];
        const double out = inp + J.intensity * e;
        const double wv = inp + J.feedback * e;
        data[i] = (T)out;
        J.w[i] = wv;
        e = wv;
      }
    }
  }
}

__global__ __launch_bounds__(256) void echo_jobs_main_kernel(const EchoJob *__restrict__ jobs) {
  const EchoJob J = jobs[blockIdx.y];
  if (J.is_f64) echo_job_main<double>(J); else echo_job_main<float>(J);
}

__global__ __launch_bounds__(256) void echo_jobs_commit_kernel(const EchoJob *__restrict__ jobs) {
  const EchoJob J = jobs[blockIdx.y];
  const size_t first = J.n > J.size ? J.n - J.size : 0;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < J.n; i += gs) J.ring[(J.pos + i) % J.size] = J.w[i];
}

}  // namespace mi355

namespace {

enum { KIND_ECHO = 1, KIND_EBUR128 = 2, KIND_LOUDNORM = 3 };

struct Sub {   // one member's submission of the interval being collected
  bool have = false;
  bool device = false;
  void *data = nullptr;      // caller's buffer (echo: in place; ebur128: input; loudnorm: input)
  void *out = nullptr;       // loudnorm: caller's output buffer
  size_t n = 0;              // echo: interleaved samples; ebur128 / loudnorm: frames
  size_t out_cap = 0;        // loudnorm: capacity of `out` in frames
  int fmt = 0;               // echo: is_f64; ebur128: sample format
  int final_frame = 0;       // loudnorm
  size_t delay = 0;
  double intensity = 0, feedback = 0;
  uint64_t interval = 0;     // the interval this submission belongs to
};

}  // namespace

struct mi355_agroup {
  int device = 0, kind = 0, n_members = 0;
  mi355_ctx *ctx = nullptr;   // the batch context: its stream carries every launch of the group
  std::mutex mu;
  std::condition_variable cv;
  std::string last_error;
  std::vector<char> attached;
  std::vector<Sub> sub;
  uint64_t interval = 1;      // the interval being collected; intervals < this one are complete
  unsigned linger_us = 0;     // echo: how long a waiter lingers for the missing members before it launches without them
  unsigned timeout_ms = 0;    // (accepted and ignored since every kind's members are independent: nobody waits for a member that does not come)
  // results of each member's last completed interval
  std::vector<int> res_status;
  std::vector<size_t> res_frames;
  std::vector<uint64_t> res_interval;
  uint64_t n_batches = 0, n_buffers = 0, n_largest = 0;
  int copying = 0;            // members that are copying between their buffer and their staging slot right now (outside the lock)
  std::vector<char> res_pending;   // [member]: a host member's result of a launch set that ran sits in the slabs and has not been collected
  // staging: per-member slots of `cap_bytes` bytes, pinned host + device (input and, for loudnorm, output)
  size_t cap_bytes = 0, out_cap_bytes = 0;
  char *h_in = nullptr, *d_in = nullptr, *h_out = nullptr, *d_out = nullptr;
  // ---- echo
  size_t ring_len = 0;
  double *d_rings = nullptr, *d_w = nullptr;
  size_t w_cap = 0;                 // doubles per member in d_w
  std::vector<size_t> pos;          // per-member ring position
  EchoJob *h_jobs = nullptr, *d_jobs = nullptr;
  hipEvent_t jobs_ev = nullptr;     // the job table of the previous launch set has left the pinned block
  // ---- ebur128
  unsigned channels = 0;
  uint64_t query_interval[5] = {0, 0, 0, 0, 0};   // ebur128: the interval the cached answers below belong to
  std::vector<double> query_cache[5];
  uint64_t peak_interval[2] = {0, 0};
  std::vector<double> peak_cache[2];
  // ---- loudnorm
  std::vector<size_t> ln_out;                 // [member]: frames the member's last frame produced
  std::vector<std::vector<double>> adapter;   // mi355_agroup_loudnorm_push: what a member has pushed and not yet handed over as a whole frame
  // ---- process-wide registry (mi355_agroup_shared_*): what the group was made from, members handed out, members released
  std::string shared_key;
  int handed_out = 0, released = 0;
};

namespace {

int afail(mi355_agroup *g, int status, const std::string &msg) {
  g->last_error = msg;
  return status;
}

int ahip(mi355_agroup *g, hipError_t e, const char *what) {
  if (e == hipSuccess) return MI355_OK;
  (void)hipGetLastError();
  g->last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? MI355_ERR_OUT_OF_MEMORY : MI355_ERR_HIP;
}

// staging slots of at least `need` bytes per member (and `out_need` for the output side). A slab is replaced while no copy runs on it and
// no launch is in flight; what it holds of members' business - submissions copied in, results not collected yet - moves along.
int ensure_staging(mi355_agroup *g, std::unique_lock<std::mutex> &lk, size_t need, size_t out_need) {
  if (need > g->cap_bytes || out_need > g->out_cap_bytes) g->cv.wait(lk, [g] { return g->copying == 0; });   // nobody is writing into the old slots
  if (need > g->cap_bytes) {
    (void)hipStreamSynchronize(g->ctx->stream);
    size_t cap = 4096;
    while (cap < need && cap < ((size_t)1 << 20)) cap *= 2;
    if (cap < need) cap = (need + 4095) & ~(size_t)4095;   // (large slots - a 3 s first frame - are sized exactly: they are pinned memory)
    char *h = nullptr, *d = nullptr;
    int rc = ahip(g, hipHostMalloc((void **)&h, cap * (size_t)g->n_members, hipHostMallocDefault), "hipHostMalloc(agroup staging)");
    if (rc) return rc;
    if ((rc = ahip(g, hipMalloc((void **)&d, cap * (size_t)g->n_members), "hipMalloc(agroup staging)"))) { (void)hipHostFree(h); return rc; }
    // submissions already copied into the old slots move along, and so do results that their members have not collected yet
    // (rsaudioecho works in place: the result of a member sits in its input slot)
    for (int m = 0; m < g->n_members; m++)
      if (g->h_in && ((g->sub[m].have && !g->sub[m].device) || g->res_pending[(size_t)m])) std::memcpy(h + (size_t)m * cap, g->h_in + (size_t)m * g->cap_bytes, g->cap_bytes);
    if (g->h_in) (void)hipHostFree(g->h_in);
    if (g->d_in) (void)hipFree(g->d_in);
    g->h_in = h; g->d_in = d; g->cap_bytes = cap;
  }
  if (out_need > g->out_cap_bytes) {
    (void)hipStreamSynchronize(g->ctx->stream);
    size_t cap = 4096;
    while (cap < out_need && cap < ((size_t)1 << 20)) cap *= 2;
    if (cap < out_need) cap = (out_need + 4095) & ~(size_t)4095;
    char *h = nullptr, *d = nullptr;
    int rc = ahip(g, hipHostMalloc((void **)&h, cap * (size_t)g->n_members, hipHostMallocDefault), "hipHostMalloc(agroup output staging)");
    if (rc) return rc;
    if ((rc = ahip(g, hipMalloc((void **)&d, cap * (size_t)g->n_members), "hipMalloc(agroup output staging)"))) { (void)hipHostFree(h); return rc; }
    // results that their members have not collected yet move along (a row of the output slab is out_cap_bytes wide, rounded down to
    // whole frames: wait() computes the row the same way)
    if (g->h_out && g->channels) {
      const size_t fb = (size_t)g->channels * 8, old_row = g->out_cap_bytes / fb * fb, new_row = cap / fb * fb;
      for (int m = 0; m < g->n_members; m++)
        if (g->res_pending[(size_t)m]) std::memcpy(h + (size_t)m * new_row, g->h_out + (size_t)m * old_row, old_row);
    }
    if (g->h_out) (void)hipHostFree(g->h_out);
    if (g->d_out) (void)hipFree(g->d_out);
    g->h_out = h; g->d_out = d; g->out_cap_bytes = cap;
  }
  return MI355_OK;
}

unsigned blocks_for(size_t n, int n_cu, int share) {
  size_t b = (n + 255) / 256;
  size_t cap = (size_t)n_cu * 8 / (size_t)(share > 0 ? share : 1);
  if (cap < 1) cap = 1;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- one launch set for the echo members that have submitted. g->mu held. Fills res_* of those members.
int run_echo(mi355_agroup *g) {
  std::vector<int> who;
  size_t max_host = 0, max_n = 0;
  for (int m = 0; m < g->n_members; m++)
    if (g->sub[m].have) {
      who.push_back(m);
      const size_t bytes = g->sub[m].n * (g->sub[m].fmt ? 8 : 4);
      if (!g->sub[m].device && bytes > max_host) max_host = bytes;
      if (g->sub[m].n > max_n) max_n = g->sub[m].n;
    }
  if (who.empty()) return MI355_OK;
  hipStream_t st = g->ctx->stream;
  int rc = MI355_OK;
  if (max_n > g->w_cap) {   // the scratch W[member][n]
    (void)hipStreamSynchronize(st);
    if (g->d_w) (void)hipFree(g->d_w);
    g->d_w = nullptr; g->w_cap = 0;
    size_t cap = 1024;
    while (cap < max_n) cap *= 2;
    if ((rc = ahip(g, hipMalloc((void **)&g->d_w, cap * 8 * (size_t)g->n_members), "hipMalloc(agroup echo scratch)"))) return rc;
    g->w_cap = cap;
  }
  // host members: their samples are in the pinned slots already (submit copied them): ONE strided copy per run of consecutive
  // participating host members - normally one for all of them. A copy never spans the slot of a member that is not part of this
  // launch set: that member may be filling its slot for the next one at this very moment.
  std::vector<std::pair<int, int>> runs;   // [first, last] member of each run
  for (int m : who) {
    if (g->sub[m].device) continue;
    if (!runs.empty() && runs.back().second == m - 1) runs.back().second = m;
    else runs.push_back({m, m});
  }
  if (max_host > 0)
    for (const auto &r : runs)
      if ((rc = ahip(g, hipMemcpy2DAsync(g->d_in + (size_t)r.first * g->cap_bytes, g->cap_bytes, g->h_in + (size_t)r.first * g->cap_bytes, g->cap_bytes, max_host,
                                         (size_t)(r.second - r.first + 1), hipMemcpyHostToDevice, st), "agroup echo: upload"))) return rc;
  if ((rc = ahip(g, hipEventSynchronize(g->jobs_ev), "hipEventSynchronize(agroup jobs)"))) return rc;
  size_t widest = 1, widest_chain = 1;
  bool any_nofb = false;
  for (size_t j = 0; j < who.size(); j++) {
    const int m = who[j];
    const Sub &s = g->sub[m];
    EchoJob &J = g->h_jobs[j];
    J.data = s.device ? s.data : (void *)(g->d_in + (size_t)m * g->cap_bytes);
    J.w = g->d_w + (size_t)m * g->w_cap;
    J.ring = g->d_rings + (size_t)m * g->ring_len;
    J.n = s.n; J.size = g->ring_len; J.pos = g->pos[m];
    J.D = s.delay == 0 ? g->ring_len : s.delay;   // read == write index: the slot written `size` samples ago (ring_buffer.rs:44-45)
    J.intensity = s.intensity; J.feedback = s.feedback; J.is_f64 = s.fmt; J.pad = 0;
    if (s.feedback == 0.0) { any_nofb = true; if (s.n > widest_chain) widest_chain = s.n; }
    else { const size_t chains = J.D < s.n ? (size_t)J.D : s.n; if (chains > widest_chain) widest_chain = chains; }
    if (s.n > widest) widest = s.n;
  }
  const unsigned J = (unsigned)who.size();
  if ((rc = ahip(g, hipMemcpyAsync(g->d_jobs, g->h_jobs, J * sizeof(EchoJob), hipMemcpyHostToDevice, st), "agroup echo: job table"))) return rc;
  if ((rc = ahip(g, hipEventRecord(g->jobs_ev, st), "hipEventRecord(agroup jobs)"))) return rc;
  const unsigned gb = blocks_for(widest, g->ctx->n_cu, (int)J), mb = blocks_for(widest_chain, g->ctx->n_cu, (int)J);
  if (any_nofb) hipLaunchKernelGGL(echo_jobs_widen_kernel, dim3(gb, J), dim3(256), 0, st, (const EchoJob *)g->d_jobs);
  hipLaunchKernelGGL(echo_jobs_main_kernel, dim3(mb, J), dim3(256), 0, st, (const EchoJob *)g->d_jobs);
  hipLaunchKernelGGL(echo_jobs_commit_kernel, dim3(gb, J), dim3(256), 0, st, (const EchoJob *)g->d_jobs);
  if ((rc = ahip(g, hipGetLastError(), "agroup echo kernel launch"))) return rc;
  if (max_host > 0)
    for (const auto &r : runs)
      if ((rc = ahip(g, hipMemcpy2DAsync(g->h_in + (size_t)r.first * g->cap_bytes, g->cap_bytes, g->d_in + (size_t)r.first * g->cap_bytes, g->cap_bytes, max_host,
                                         (size_t)(r.second - r.first + 1), hipMemcpyDeviceToHost, st), "agroup echo: download"))) return rc;
  if ((rc = ahip(g, hipStreamSynchronize(st), "agroup echo: sync"))) return rc;
  for (int m : who) g->pos[m] = (g->pos[m] + g->sub[m].n) % g->ring_len;   // RingBufferIter::drop (ring_buffer.rs:78-82)
  return MI355_OK;
}

// ---- ebur128level: the members that have submitted advance, each by its own buffer size and with its own 100 ms phase (the engine
// walks per-stream rounds, ebur128_kernels.hip); the others - late, detached, paused - do not move. One sample format per launch set.
int run_ebur128(mi355_agroup *g) {
  static const size_t esz[4] = {2, 4, 4, 8};
  std::vector<size_t> frames_per((size_t)g->n_members, 0);
  int fmt = -1;
  size_t max_host = 0;
  for (int m = 0; m < g->n_members; m++)
    if (g->sub[m].have) {
      fmt = g->sub[m].fmt;
      frames_per[(size_t)m] = g->sub[m].n;
      const size_t bytes = g->sub[m].n * g->channels * esz[fmt];
      if (!g->sub[m].device && bytes > max_host) max_host = bytes;
    }
  if (fmt < 0) return MI355_OK;
  hipStream_t st = g->ctx->stream;
  int rc;
  // host members: one strided copy per run of consecutive participating host members (never across the slot of a member that is
  // not part of this launch set: it may be filling it for the next one); device members: one D2D copy each
  std::vector<std::pair<int, int>> runs;
  for (int m = 0; m < g->n_members; m++) {
    if (!g->sub[m].have || g->sub[m].device) continue;
    if (!runs.empty() && runs.back().second == m - 1) runs.back().second = m;
    else runs.push_back({m, m});
  }
  for (const auto &r : runs)
    if ((rc = ahip(g, hipMemcpy2DAsync(g->d_in + (size_t)r.first * g->cap_bytes, g->cap_bytes, g->h_in + (size_t)r.first * g->cap_bytes, g->cap_bytes, max_host,
                                       (size_t)(r.second - r.first + 1), hipMemcpyHostToDevice, st), "agroup ebur128: upload"))) return rc;
  for (int m = 0; m < g->n_members; m++)
    if (g->sub[m].have && g->sub[m].device)
      if ((rc = ahip(g, hipMemcpyAsync(g->d_in + (size_t)m * g->cap_bytes, g->sub[m].data, g->sub[m].n * g->channels * esz[fmt], hipMemcpyDeviceToDevice, st),
                     "agroup ebur128: gather"))) return rc;
  rc = ebur128_add_frames_streams(g->ctx, g->d_in, g->cap_bytes / esz[fmt], frames_per.data(), fmt, 1);
  if (rc) g->last_error = g->ctx->last_error;
  return rc;
}

// ---- audioloudnorm: the members that have submitted advance, class by class - a class = the members that stand at the same frame
// type and hand over the same number of frames (streams that started together; a late starter is a class of its own until it has
// caught up with the 100 ms frames of the others). One launch sequence per class (loudnorm.hip: loudnorm_process_members), the
// other members do not move.
int run_loudnorm(mi355_agroup *g) {
  const size_t ch = g->channels, N = (size_t)g->n_members;
  hipStream_t st = g->ctx->stream;
  std::vector<char> todo(N, 0);
  bool any = false;
  for (size_t m = 0; m < N; m++) if (g->sub[m].have) { todo[m] = 1; any = true; }
  if (!any) return MI355_OK;
  const size_t in_stride = g->cap_bytes / 8, cap_frames = g->out_cap_bytes / (ch * 8);
  size_t max_out = 0;
  int rc;
  for (size_t m0 = 0; m0 < N; m0++) {
    if (!todo[m0]) continue;
    // the class of member m0
    const size_t frames = g->sub[m0].n;
    const int final_frame = g->sub[m0].final_frame, ft = loudnorm_member_frame_type(g->ctx, (unsigned)m0);
    const size_t fs = loudnorm_member_frame_size(g->ctx, (unsigned)m0);
    std::vector<unsigned char> cls(N, 0);
    for (size_t m = m0; m < N; m++)
      if (todo[m] && g->sub[m].n == frames && g->sub[m].final_frame == final_frame && loudnorm_member_frame_type(g->ctx, (unsigned)m) == ft &&
          loudnorm_member_frame_size(g->ctx, (unsigned)m) == fs) { cls[m] = 1; todo[m] = 0; }
    const size_t bytes = frames * ch * 8;
    if (bytes) {
      // host members of the class: one strided copy per run of consecutive ones; device members: one D2D copy each
      for (size_t m = 0; m < N;) {
        if (!(cls[m] && !g->sub[m].device)) { m++; continue; }
        size_t e = m;
        while (e + 1 < N && cls[e + 1] && !g->sub[e + 1].device) e++;
        if ((rc = ahip(g, hipMemcpy2DAsync(g->d_in + m * g->cap_bytes, g->cap_bytes, g->h_in + m * g->cap_bytes, g->cap_bytes, bytes, e - m + 1, hipMemcpyHostToDevice, st),
                       "agroup loudnorm: upload"))) return rc;
        m = e + 1;
      }
      for (size_t m = 0; m < N; m++)
        if (cls[m] && g->sub[m].device)
          if ((rc = ahip(g, hipMemcpyAsync(g->d_in + m * g->cap_bytes, g->sub[m].data, bytes, hipMemcpyDeviceToDevice, st), "agroup loudnorm: gather"))) return rc;
    }
    size_t out_frames = 0;
    rc = loudnorm_process_members(g->ctx, cls.data(), (const double *)g->d_in, in_stride, frames, (double *)g->d_out, cap_frames * ch, cap_frames, &out_frames, 1, final_frame);
    if (rc) { g->last_error = g->ctx->last_error; return rc; }
    for (size_t m = 0; m < N; m++) if (cls[m]) g->ln_out[m] = out_frames;
    if (out_frames > max_out) max_out = out_frames;
    if (out_frames)
      for (size_t m = 0; m < N; m++)
        if (cls[m] && g->sub[m].device && g->sub[m].out)
          if ((rc = ahip(g, hipMemcpyAsync(g->sub[m].out, g->d_out + m * cap_frames * ch * 8, out_frames * ch * 8, hipMemcpyDeviceToDevice, st), "agroup loudnorm: scatter"))) return rc;
  }
  // output: packed [member][cap frames] in the device slab -> one copy to the pinned slab for the host members (rows of members that
  // did not take part are copied along and never read)
  if (max_out)
    if ((rc = ahip(g, hipMemcpy2DAsync(g->h_out, cap_frames * ch * 8, g->d_out, cap_frames * ch * 8, max_out * ch * 8, N, hipMemcpyDeviceToHost, st), "agroup loudnorm: download"))) return rc;
  return ahip(g, hipStreamSynchronize(st), "agroup loudnorm: sync");
}

// runs the collected interval. g->mu held (the members are blocked on it or on the condition variable anyway).
void run_interval(mi355_agroup *g) {
  (void)hipSetDevice(g->device);
  int rc = MI355_OK;
  if (g->kind == KIND_ECHO) rc = run_echo(g);
  else if (g->kind == KIND_EBUR128) rc = run_ebur128(g);
  else rc = run_loudnorm(g);
  uint64_t carried = 0;
  for (int m = 0; m < g->n_members; m++) {
    Sub &s = g->sub[m];
    if (!s.have) continue;
    carried++;
    g->res_status[m] = rc;
    g->res_frames[m] = g->kind == KIND_LOUDNORM ? g->ln_out[(size_t)m] : s.n;
    g->res_interval[m] = s.interval;
    g->res_pending[(size_t)m] = !s.device && rc == MI355_OK && g->kind != KIND_EBUR128;
    s.have = false;   // (data / out stay: wait copies the member's result out)
  }
  g->n_batches++;
  g->n_buffers += carried;
  if (carried > g->n_largest) g->n_largest = carried;
  g->interval++;
  g->cv.notify_all();
}

bool everybody_here(const mi355_agroup *g) {
  bool any = false;
  for (int m = 0; m < g->n_members; m++) {
    if (!g->attached[m]) continue;
    if (!g->sub[m].have) return false;
    any = true;
  }
  return any;
}

mi355_agroup *agroup_new(int device, int kind, int n_members, int *status) {
  if (n_members < 1 || n_members > 4096) { if (status) *status = MI355_ERR_INVALID_ARG; return nullptr; }
  int st = MI355_OK;
  mi355_ctx *ctx = mi355_ctx_create(device, &st);
  if (!ctx) { if (status) *status = st ? st : MI355_ERR_NO_DEVICE; return nullptr; }
  mi355_agroup *g = new mi355_agroup();
  g->device = device; g->kind = kind; g->n_members = n_members; g->ctx = ctx;
  g->attached.assign((size_t)n_members, 1);
  g->sub.assign((size_t)n_members, Sub{});
  g->res_status.assign((size_t)n_members, MI355_OK);
  g->res_frames.assign((size_t)n_members, 0);
  g->res_interval.assign((size_t)n_members, 0);
  g->ln_out.assign((size_t)n_members, 0);
  g->res_pending.assign((size_t)n_members, 0);
  return g;
}

uint64_t ticket_of(const mi355_agroup *g, uint64_t interval, int member) { return interval * (uint64_t)g->n_members + (uint64_t)member + 1; }

// common tail of every submit: the member's host buffer goes into its staging slot OUTSIDE the lock (32 members copying 9 MB
// first frames one after the other would be the longest thing in the interval), then the slot counts; run the interval if it is
// complete. `lk` owns g->mu on entry and on return.
void submitted(mi355_agroup *g, std::unique_lock<std::mutex> &lk, int member, uint64_t *ticket, void *dst, const void *src, size_t bytes) {
  if (dst && bytes > (size_t)65536) {
    g->copying++;
    lk.unlock();
    std::memcpy(dst, src, bytes);
    lk.lock();
    g->copying--;
    g->cv.notify_all();
  } else if (dst && bytes) {
    std::memcpy(dst, src, bytes);   // (a 10 ms audio buffer is a few KB: cheaper than giving the lock away and taking it again)
  }
  g->sub[member].have = true;
  g->sub[member].interval = g->interval;
  if (ticket) *ticket = ticket_of(g, g->interval, member);
  if (everybody_here(g)) run_interval(g);
}

int check_member(mi355_agroup *g, int kind, int member) {
  if (g->kind != kind) return afail(g, MI355_ERR_INVALID_ARG, "agroup: this group batches another element kind");
  if (member < 0 || member >= g->n_members) return afail(g, MI355_ERR_INVALID_ARG, "agroup: no such member");
  if (!g->attached[member]) return afail(g, MI355_ERR_INVALID_ARG, "agroup: this member has been detached");
  if (g->sub[member].have) return afail(g, MI355_ERR_INVALID_ARG, "agroup: this member's previous buffer has not been waited for");
  return MI355_OK;
}

}  // namespace

extern "C" {

mi355_agroup *mi355_agroup_create_echo(int device, int n_members, size_t ring_len, int *status) {
  mi355_agroup *g = agroup_new(device, KIND_ECHO, n_members, status);
  if (!g) return nullptr;
  g->ring_len = ring_len;
  const size_t cells = (size_t)n_members * (ring_len ? ring_len : 1);
  int rc = ahip(g, hipMalloc((void **)&g->d_rings, cells * 8), "hipMalloc(agroup echo rings)");
  if (!rc) rc = ahip(g, hipMemset(g->d_rings, 0, cells * 8), "hipMemset(agroup echo rings)");
  if (!rc) rc = ahip(g, hipMalloc((void **)&g->d_jobs, (size_t)n_members * sizeof(EchoJob)), "hipMalloc(agroup echo jobs)");
  if (!rc) rc = ahip(g, hipHostMalloc((void **)&g->h_jobs, (size_t)n_members * sizeof(EchoJob), hipHostMallocDefault), "hipHostMalloc(agroup echo jobs)");
  if (!rc) rc = ahip(g, hipEventCreateWithFlags(&g->jobs_ev, hipEventDisableTiming), "hipEventCreate(agroup)");
  g->pos.assign((size_t)n_members, 0);
  if (rc) { if (status) *status = rc; mi355_agroup_destroy(g); return nullptr; }
  if (status) *status = MI355_OK;
  return g;
}

mi355_agroup *mi355_agroup_create_ebur128(int device, int n_members, unsigned channels, unsigned rate, unsigned mode, const int *channel_class, int *status) {
  mi355_agroup *g = agroup_new(device, KIND_EBUR128, n_members, status);
  if (!g) return nullptr;
  g->channels = channels;
  const int rc = ebur128_setup_batch(g->ctx, (unsigned)n_members, channels, rate, mode, channel_class);
  if (rc) { if (status) *status = rc; mi355_agroup_destroy(g); return nullptr; }
  if (status) *status = MI355_OK;
  return g;
}

mi355_agroup *mi355_agroup_create_loudnorm(int device, int n_members, unsigned channels, double loudness_target, double loudness_range_target,
                                           double max_true_peak, double offset, int *status) {
  mi355_agroup *g = agroup_new(device, KIND_LOUDNORM, n_members, status);
  if (!g) return nullptr;
  g->channels = channels;
  int rc = loudnorm_setup_batch(g->ctx, (unsigned)n_members, channels, loudness_target, loudness_range_target, max_true_peak, offset);
  if (!rc) {
    // the slots for the largest frames State::process sees (the 3 s first frame in, 3 s out at drain) now, not inside the first interval
    std::unique_lock<std::mutex> lk(g->mu);
    if ((size_t)n_members * 576000 * channels * 8 <= ((size_t)2 << 30))   // (thousands of members: sized when the frames come)
      rc = ensure_staging(g, lk, (size_t)576000 * channels * 8, (size_t)30 * 19200 * channels * 8);
  }
  if (rc) { if (status) *status = rc; mi355_agroup_destroy(g); return nullptr; }
  if (status) *status = MI355_OK;
  return g;
}

void mi355_agroup_destroy(mi355_agroup *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  {
    // members still blocked in wait are woken with a failure (destroying a group others still use is a caller bug; nothing hangs)
    std::lock_guard<std::mutex> lk(g->mu);
    for (int m = 0; m < g->n_members; m++) g->attached[m] = 0;
    g->cv.notify_all();
  }
  if (g->ctx) (void)hipStreamSynchronize(g->ctx->stream);
  if (g->h_in) (void)hipHostFree(g->h_in);
  if (g->d_in) (void)hipFree(g->d_in);
  if (g->h_out) (void)hipHostFree(g->h_out);
  if (g->d_out) (void)hipFree(g->d_out);
  if (g->d_rings) (void)hipFree(g->d_rings);
  if (g->d_w) (void)hipFree(g->d_w);
  if (g->d_jobs) (void)hipFree(g->d_jobs);
  if (g->h_jobs) (void)hipHostFree(g->h_jobs);
  if (g->jobs_ev) (void)hipEventDestroy(g->jobs_ev);
  if (g->ctx) mi355_ctx_destroy(g->ctx);   // releases the ebur128 / loudnorm batch engines with it
  delete g;
}

const char *mi355_agroup_last_error(mi355_agroup *g) { return g ? g->last_error.c_str() : "null agroup"; }

int mi355_agroup_set_linger(mi355_agroup *g, unsigned linger_us, unsigned timeout_ms) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  g->linger_us = linger_us;
  g->timeout_ms = timeout_ms;
  return MI355_OK;
}

int mi355_agroup_detach(mi355_agroup *g, int member) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  if (member < 0 || member >= g->n_members) return afail(g, MI355_ERR_INVALID_ARG, "agroup: no such member");
  g->attached[member] = 0;
  g->sub[member].have = false;
  // the others may have been waiting for this one
  if (everybody_here(g)) run_interval(g);
  g->cv.notify_all();
  return MI355_OK;
}

int mi355_agroup_submit_echo(mi355_agroup *g, int member, void *data, size_t n, int is_f64, size_t delay_samples, double intensity, double feedback,
                             int device_data, uint64_t *ticket) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  int rc = check_member(g, KIND_ECHO, member);
  if (rc) return rc;
  // RingBufferIter::new: assert!(size >= delay); assert_ne!(size, 0) (ring_buffer.rs:41-42)
  if (g->ring_len == 0) return afail(g, MI355_ERR_INVALID_ARG, "rsaudioecho: ring buffer size is 0");
  if (delay_samples > g->ring_len) return afail(g, MI355_ERR_INVALID_ARG, "rsaudioecho: delay exceeds ring buffer size");
  if (n && !data) return afail(g, MI355_ERR_INVALID_ARG, "agroup: null buffer");
  (void)hipSetDevice(g->device);
  Sub &s = g->sub[member];
  s.device = device_data != 0; s.data = data; s.n = n; s.fmt = is_f64 ? 1 : 0; s.delay = delay_samples; s.intensity = intensity; s.feedback = feedback;
  void *dst = nullptr;
  size_t bytes = 0;
  if (!s.device && n) {
    bytes = n * (is_f64 ? 8 : 4);
    if ((rc = ensure_staging(g, lk, bytes, 0))) return rc;
    dst = g->h_in + (size_t)member * g->cap_bytes;
  }
  submitted(g, lk, member, ticket, dst, data, bytes);
  return MI355_OK;
}

int mi355_agroup_submit_ebur128(mi355_agroup *g, int member, const void *data, size_t frames, int sample_format, int device_data, uint64_t *ticket) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  int rc = check_member(g, KIND_EBUR128, member);
  if (rc) return rc;
  if (sample_format < 0 || sample_format > 3) return afail(g, MI355_ERR_INVALID_ARG, "ebur128: bad sample format");
  if (frames == 0 || !data) return afail(g, MI355_ERR_INVALID_ARG, "agroup: empty buffer");
  for (int m = 0; m < g->n_members; m++)
    if (m != member && g->sub[m].have && g->sub[m].fmt != sample_format)
      return afail(g, MI355_ERR_INVALID_ARG, "agroup: the members of an ebur128level group submit one sample format");
  (void)hipSetDevice(g->device);
  static const size_t esz[4] = {2, 4, 4, 8};
  const size_t bytes = frames * g->channels * esz[sample_format];
  if ((rc = ensure_staging(g, lk, bytes, 0))) return rc;
  Sub &s = g->sub[member];
  s.device = device_data != 0; s.data = (void *)data; s.n = frames; s.fmt = sample_format;
  submitted(g, lk, member, ticket, s.device ? nullptr : g->h_in + (size_t)member * g->cap_bytes, data, bytes);
  return MI355_OK;
}

// the `reset` action of one ebur128level instance (imp.rs:124-139, :320-333): this member's meter back to the state of a new one -
// history, histograms, peaks and its 100 ms phase; the other members are not touched. Not while this member has a buffer pending.
int mi355_agroup_ebur128_reset(mi355_agroup *g, int member) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  int rc = check_member(g, KIND_EBUR128, member);
  if (rc) return rc;
  (void)hipSetDevice(g->device);
  if ((rc = ebur128_reset_stream(g->ctx, (unsigned)member))) { g->last_error = g->ctx->last_error; return rc; }
  for (int k = 0; k < 5; k++) g->query_interval[k] = ~(uint64_t)0;   // cached answers are stale
  g->peak_interval[0] = g->peak_interval[1] = ~(uint64_t)0;
  return MI355_OK;
}

int mi355_agroup_submit_loudnorm(mi355_agroup *g, int member, const double *data, size_t frames, double *out, size_t out_capacity_frames, int final_frame,
                                 int device_data, uint64_t *ticket) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  int rc = check_member(g, KIND_LOUDNORM, member);
  if (rc) return rc;
  if (frames && !data) return afail(g, MI355_ERR_INVALID_ARG, "agroup: null buffer");
  (void)hipSetDevice(g->device);
  const size_t ch = g->channels, bytes = frames * ch * 8;
  {
    // the output capacity is checked BEFORE anything changes (as mi355_loudnorm_process_batch does): a first or inner frame answers
    // 100 ms, the final frame what is still inside (imp.rs:270-310), a stream that ends inside its first 3 s its own length
    const size_t cur = loudnorm_member_frame_size(g->ctx, (unsigned)member);
    const size_t need = final_frame ? (cur == 19200 ? (size_t)30 * 19200 - (19200 - (frames < 19200 ? frames : 19200)) : frames) : (size_t)19200;
    if (!final_frame && frames != cur) return afail(g, MI355_ERR_INVALID_ARG, "audioloudnorm: a member hands over whole frames (mi355_agroup_loudnorm_frame_size)");
    if (final_frame && frames >= cur && frames != 0) return afail(g, MI355_ERR_INVALID_ARG, "audioloudnorm: the final frame is shorter than a full one");
    if (need > out_capacity_frames || (need && !out)) return afail(g, MI355_ERR_INVALID_ARG, "audioloudnorm: output buffer too small");
  }
  // what State::process can hand back for this frame: the first frame answers 100 ms, the final one up to 3 s (imp.rs:226-310)
  const size_t worst = final_frame ? (size_t)30 * 19200 : (frames > 19200 ? frames : (size_t)19200);
  if ((rc = ensure_staging(g, lk, bytes ? bytes : 8, worst * ch * 8))) return rc;
  Sub &s = g->sub[member];
  s.device = device_data != 0; s.data = (void *)data; s.out = out; s.n = frames; s.out_cap = out_capacity_frames; s.final_frame = final_frame ? 1 : 0;
  submitted(g, lk, member, ticket, s.device ? nullptr : g->h_in + (size_t)member * g->cap_bytes, data, bytes);
  return MI355_OK;
}

// the frame member `member` hands over next: its first 3 s (576,000 frames at 192 kHz), then 100 ms (19,200)
size_t mi355_agroup_loudnorm_frame_size(mi355_agroup *g, int member) {
  if (!g || g->kind != KIND_LOUDNORM || member < 0 || member >= g->n_members) return 0;
  std::lock_guard<std::mutex> lk(g->mu);
  return loudnorm_member_frame_size(g->ctx, (unsigned)member);
}

// Waits until the member's interval has run; copies a host member's result back to its buffer. *out_frames (optional): frames
// produced (audioloudnorm), samples processed (rsaudioecho) or frames metered (ebur128level).
int mi355_agroup_wait(mi355_agroup *g, uint64_t ticket, size_t *out_frames) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  if (ticket == 0) return afail(g, MI355_ERR_INVALID_ARG, "agroup: unknown ticket");
  const uint64_t interval = (ticket - 1) / (uint64_t)g->n_members;
  const int member = (int)((ticket - 1) % (uint64_t)g->n_members);
  if (interval == 0 || interval > g->interval) return afail(g, MI355_ERR_INVALID_ARG, "agroup: unknown ticket");
  const auto t0 = std::chrono::steady_clock::now();
  while (g->interval <= interval) {
    if (!g->attached[member]) return afail(g, MI355_ERR_INVALID_ARG, "agroup: destroyed or detached while waiting");
    if (everybody_here(g)) { run_interval(g); continue; }
    // independent members: linger for the others, then launch whoever is there
    if (g->linger_us == 0 || g->cv.wait_until(lk, t0 + std::chrono::microseconds(g->linger_us)) == std::cv_status::timeout) {
      if (g->interval <= interval) run_interval(g);
    }
  }
  if (g->res_interval[member] != interval) return afail(g, MI355_ERR_INVALID_ARG, "agroup: this ticket has been waited for already");
  const int rc = g->res_status[member];
  if (rc) return rc;   // last_error is the batch's
  const Sub &s = g->sub[member];
  const size_t frames = g->res_frames[member];
  if (out_frames) *out_frames = frames;
  // copy-out happens under the lock only for its pointer arithmetic: the member's slot is not written again before this member
  // submits again, and the slab is not reallocated while a copy is counted (a member that grows the slots waits in ensure_staging:
  // round 6's stress run caught a result read from a slab another member's larger buffer had just replaced)
  if (g->kind == KIND_ECHO) {
    if (!s.device && s.n) {
      const char *src = g->h_in + (size_t)member * g->cap_bytes;
      void *dst = s.data;
      const size_t bytes = s.n * (s.fmt ? 8 : 4);
      if (bytes <= (size_t)65536) {
        std::memcpy(dst, src, bytes);   // (a 10 ms audio buffer is a few KB: cheaper than giving the lock away and taking it again)
      } else {
        g->copying++;   // (the slab must not be reallocated under this copy: ensure_staging waits for copies in either direction)
        lk.unlock();
        std::memcpy(dst, src, bytes);
        lk.lock();
        g->copying--;
        g->cv.notify_all();
      }
    }
    g->res_pending[(size_t)member] = 0;
  } else if (g->kind == KIND_LOUDNORM) {
    if (frames > s.out_cap) return afail(g, MI355_ERR_INVALID_ARG, "audioloudnorm: output buffer too small");
    if (!s.device && frames && s.out) {
      const size_t ch = g->channels, cap_frames = g->out_cap_bytes / (ch * 8);
      const char *src = g->h_out + (size_t)member * cap_frames * ch * 8;
      void *dst = s.out;
      g->copying++;
      lk.unlock();
      std::memcpy(dst, src, frames * ch * 8);
      lk.lock();
      g->copying--;
      g->cv.notify_all();
    }
    g->res_pending[(size_t)member] = 0;
  }
  return MI355_OK;
}

// ebur128level's queries for one member (ebur128level/imp.rs:378-452). The engine answers for every member at once; the answers
// of an interval are computed once and served to the members that ask (what: 0 momentary, 1 short-term, 2 global, 3 relative
// threshold, 4 loudness range).
int mi355_agroup_ebur128_loudness(mi355_agroup *g, int member, int what, double *out) {
  if (!g || !out) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->kind != KIND_EBUR128) return afail(g, MI355_ERR_INVALID_ARG, "agroup: not an ebur128level group");
  if (member < 0 || member >= g->n_members || what < 0 || what > 4) return afail(g, MI355_ERR_INVALID_ARG, "agroup: bad member / query");
  if (g->query_interval[what] != g->interval) {
    (void)hipSetDevice(g->device);
    g->query_cache[what].assign((size_t)g->n_members, 0.0);
    const int rc = ebur128_query_batch(g->ctx, what, g->query_cache[what].data());
    if (rc) { g->last_error = g->ctx->last_error; return rc; }
    g->query_interval[what] = g->interval;
  }
  *out = g->query_cache[what][(size_t)member];
  return MI355_OK;
}

int mi355_agroup_ebur128_peak(mi355_agroup *g, int member, int true_peak, unsigned channel, double *out) {
  if (!g || !out) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->kind != KIND_EBUR128) return afail(g, MI355_ERR_INVALID_ARG, "agroup: not an ebur128level group");
  if (member < 0 || member >= g->n_members || channel >= g->channels) return afail(g, MI355_ERR_INVALID_ARG, "agroup: bad member / channel");
  const int k = true_peak ? 1 : 0;
  if (g->peak_interval[k] != g->interval) {
    g->peak_cache[k].assign((size_t)g->n_members * g->channels, 0.0);
    const int rc = ebur128_peak_batch(g->ctx, k, g->peak_cache[k].data());
    if (rc) { g->last_error = g->ctx->last_error; return rc; }
    g->peak_interval[k] = g->interval;
  }
  *out = g->peak_cache[k][(size_t)member * g->channels + channel];
  return MI355_OK;
}

int mi355_agroup_echo_get_state(mi355_agroup *g, int member, double *ring_out, size_t ring_len, size_t *pos_out) {
  if (!g) return MI355_ERR_INVALID_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->kind != KIND_ECHO || member < 0 || member >= g->n_members) return afail(g, MI355_ERR_INVALID_ARG, "agroup: bad member");
  if (ring_out && ring_len != g->ring_len) return afail(g, MI355_ERR_INVALID_ARG, "agroup: ring length mismatch");
  (void)hipSetDevice(g->device);
  if (ring_out && g->ring_len) {
    int rc = ahip(g, hipMemcpy(ring_out, g->d_rings + (size_t)member * g->ring_len, g->ring_len * 8, hipMemcpyDeviceToHost), "agroup echo: ring D2H");
    if (rc) return rc;
  }
  if (pos_out) *pos_out = g->pos[(size_t)member];
  return MI355_OK;
}

int mi355_agroup_stats(mi355_agroup *g, uint64_t stats[3]) {
  if (!g || !stats) return MI355_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  stats[0] = g->n_buffers;
  stats[1] = g->n_batches;
  stats[2] = g->n_largest;
  return MI355_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- process-wide groups
// Elements of independent pipelines cannot hand a group to each other; what they share is the process. mi355_agroup_shared_*
// returns THE group of this configuration (kind, device, member count, parameters), creating it at first use, and the next free
// member index; a group whose members have all been handed out is not offered again (the next element of that configuration
// starts a new one). mi355_agroup_release = detach; the last member out destroys the group.
namespace {
std::mutex g_shared_mu;
std::vector<mi355_agroup *> g_shared;

template <typename Make>
mi355_agroup *shared_get(const std::string &key, int n_members, int *member, int *status, Make make) {
  std::lock_guard<std::mutex> lk(g_shared_mu);
  for (mi355_agroup *g : g_shared)
    if (g->shared_key == key && g->handed_out < g->n_members) {
      if (member) *member = g->handed_out;
      g->handed_out++;
      if (status) *status = MI355_OK;
      return g;
    }
  mi355_agroup *g = make();
  if (!g) return nullptr;
  g->shared_key = key;
  g->handed_out = 1;
  if (member) *member = 0;
  g_shared.push_back(g);
  (void)n_members;
  return g;
}

std::string key_of(const char *kind, int device, int n_members, const double *v, int n) {
  std::string k = std::string(kind) + ":" + std::to_string(device) + ":" + std::to_string(n_members);
  for (int i = 0; i < n; i++) { char b[40]; std::snprintf(b, sizeof b, ":%.17g", v[i]); k += b; }
  return k;
}
}  // namespace

extern "C" {

mi355_agroup *mi355_agroup_shared_echo(int device, int n_members, size_t ring_len, int *member, int *status) {
  const double v[1] = {(double)ring_len};
  return shared_get(key_of("echo", device, n_members, v, 1), n_members, member, status, [&] { return mi355_agroup_create_echo(device, n_members, ring_len, status); });
}

mi355_agroup *mi355_agroup_shared_ebur128(int device, int n_members, unsigned channels, unsigned rate, unsigned mode, const int *channel_class, int *member,
                                          int *status) {
  std::vector<double> v = {(double)channels, (double)rate, (double)mode};
  for (unsigned c = 0; c < channels && c < 64; c++) v.push_back(channel_class ? (double)channel_class[c] : -1.0);
  return shared_get(key_of("ebur128", device, n_members, v.data(), (int)v.size()), n_members, member, status,
                    [&] { return mi355_agroup_create_ebur128(device, n_members, channels, rate, mode, channel_class, status); });
}

mi355_agroup *mi355_agroup_shared_loudnorm(int device, int n_members, unsigned channels, double loudness_target, double loudness_range_target,
                                           double max_true_peak, double offset, int *member, int *status) {
  const double v[5] = {(double)channels, loudness_target, loudness_range_target, max_true_peak, offset};
  return shared_get(key_of("loudnorm", device, n_members, v, 5), n_members, member, status,
                    [&] { return mi355_agroup_create_loudnorm(device, n_members, channels, loudness_target, loudness_range_target, max_true_peak, offset, status); });
}

// audioloudnorm's sink_chain / drain for a member (audioloudnorm/imp.rs:1545-1586 -> drain_full_frames :226-268, drain :270-310) with
// the adapter on this side: what mi355_loudnorm_push / _drain are for a single-instance context. push appends the buffer and
// hands every whole frame over in lock step with the other members (blocking like wait); drain hands over the rest as the final frame.
int mi355_agroup_loudnorm_push(mi355_agroup *g, int member, const double *data, size_t frames, double *out, size_t out_capacity_frames, size_t *out_frames) {
  if (!g || !out_frames) return MI355_ERR_INVALID_ARG;
  *out_frames = 0;
  std::vector<double> *ad = nullptr;
  size_t ch = 0;
  {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->kind != KIND_LOUDNORM || member < 0 || member >= g->n_members) return afail(g, MI355_ERR_INVALID_ARG, "agroup: not an audioloudnorm member");
    if (frames && !data) return afail(g, MI355_ERR_INVALID_ARG, "agroup: null buffer");
    if (g->adapter.empty()) g->adapter.resize((size_t)g->n_members);
    ad = &g->adapter[(size_t)member];   // (only this member's thread touches its adapter)
    ch = g->channels;
  }
  ad->insert(ad->end(), data, data + frames * ch);
  for (;;) {
    const size_t fs = mi355_agroup_loudnorm_frame_size(g, member);
    if (fs == 0 || ad->size() / ch < fs) break;
    uint64_t t = 0;
    size_t n = 0;
    int rc = mi355_agroup_submit_loudnorm(g, member, ad->data(), fs, out + *out_frames * ch, out_capacity_frames - *out_frames, 0, 0, &t);
    if (!rc) rc = mi355_agroup_wait(g, t, &n);
    if (rc) return rc;
    ad->erase(ad->begin(), ad->begin() + (std::ptrdiff_t)(fs * ch));
    *out_frames += n;
  }
  return MI355_OK;
}

int mi355_agroup_loudnorm_drain(mi355_agroup *g, int member, double *out, size_t out_capacity_frames, size_t *out_frames, int *eos) {
  if (!g || !out_frames) return MI355_ERR_INVALID_ARG;
  *out_frames = 0;
  if (eos) *eos = 0;
  std::vector<double> *ad = nullptr;
  size_t ch = 0;
  {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->kind != KIND_LOUDNORM || member < 0 || member >= g->n_members) return afail(g, MI355_ERR_INVALID_ARG, "agroup: not an audioloudnorm member");
    if (g->adapter.empty()) g->adapter.resize((size_t)g->n_members);
    ad = &g->adapter[(size_t)member];
    ch = g->channels;
  }
  const size_t rest = ad->size() / ch;
  uint64_t t = 0;
  int rc = mi355_agroup_submit_loudnorm(g, member, ad->data(), rest, out, out_capacity_frames, 1, 0, &t);
  if (!rc) rc = mi355_agroup_wait(g, t, out_frames);
  if (rc) return rc;
  ad->clear();
  if (eos && rest == 0 && *out_frames == 0) *eos = 1;   // nothing at all to drain: FlowError::Eos (imp.rs:289-293)
  return MI355_OK;
}

void mi355_agroup_release(mi355_agroup *g, int member) {
  if (!g) return;
  (void)mi355_agroup_detach(g, member);
  bool last = false;
  {
    std::lock_guard<std::mutex> lk(g_shared_mu);
    g->released++;
    // members never handed out count as gone once everybody who came has left
    last = g->released >= g->handed_out;
    if (last)
      for (size_t i = 0; i < g_shared.size(); i++)
        if (g_shared[i] == g) { g_shared.erase(g_shared.begin() + (std::ptrdiff_t)i); break; }
  }
  if (last) mi355_agroup_destroy(g);
}

}  // extern "C"
