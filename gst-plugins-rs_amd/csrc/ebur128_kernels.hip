// ebur128_kernels.hip — EBU R128 / ITU-R BS.1770 loudness measurement for `ebur128level`
// (and the r128_in/r128_out meters of `audioloudnorm`).
//
// Reference side: the element only slices frames and posts messages
// (audio/audiofx/src/ebur128level/imp.rs:296-486, :682-745); all arithmetic is inside the third-party crate
// `ebur128 = 0.1.10` (Cargo.lock:3685-3686, a Rust port of libebur128), which is not vendored. This file
// implements the published algorithm in libebur128's formulation; parity is checked against
// oracle/ebur128_oracle.c (serial f64) with a tolerance, and against the EBU Tech 3341/3342 readings.
//
// GPU mapping
//   * K-weighting (4th-order IIR, direct form II, f64): one lane per channel runs the literal recurrence
//     (bit-identical to a serial CPU run); parallelism is across channels (and across streams/contexts).
//   * Gating-block / short-term energies: per-channel sum of squares over the ring window, wave shuffle
//     (__shfl_down, 64 lanes) + LDS reduction, channel weights applied at the end.
//   * Sample peak / true peak: max|x| reductions; the true-peak interpolator is the 49-tap Hann-windowed
//     sinc polyphase FIR (x4 below 96 kHz, x2 below 192 kHz), one output sample per lane-iteration,
//     accumulated in the reference's tap order.
//   * Histograms, gating and the loudness formulas are O(1000) per query and stay on the host.
#include "internal.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace mi355 {

enum { EB_M = 1, EB_S = 2, EB_I = 4, EB_LRA = 8, EB_SAMPLE_PEAK = 16, EB_TRUE_PEAK = 32 };
constexpr int kHistBins = 1000;
constexpr int kEbNT = 256;
constexpr double kPi = 3.14159265358979323846;

struct EbFilterK {
  double b[5], a[5];
};

// Streams of a batch are independent meters: each has its own 100 ms phase (a member of an audio group resets, joins late or
// submits another buffer size). One launch covers one ROUND - the next segment of every stream - and reads what that is for its
// stream from a table in device memory: a stream with nothing to do in a round has n == 0 (segment) or slot < 0 (energy).
struct EbSeg {
  unsigned long long src0, n, ring0;   // first source frame, frames, first ring frame of this stream's segment
};
struct EbEv {
  unsigned long long end_frame;        // the window ends before this ring frame
  long long slot;                      // energy slot of this stream the result goes to, < 0: nothing for this stream
};

template <typename T>
__device__ __forceinline__ double eb_to_double(T v);
template <> __device__ __forceinline__ double eb_to_double<int16_t>(int16_t v) { return (double)v / 32768.0; }
template <> __device__ __forceinline__ double eb_to_double<int32_t>(int32_t v) { return (double)v / 2147483648.0; }
template <> __device__ __forceinline__ double eb_to_double<float>(float v) { return (double)v; }
template <> __device__ __forceinline__ double eb_to_double<double>(double v) { return v; }

// K-weighting + sample peak. The 4th-order recurrence is serial per channel by nature (SURVEY.md section 7
// item 6); a chunked parallel scan was implemented and rejected: superposing zero-state and homogeneous
// responses through A^L loses ~7 digits because the direct-form-II state of the high-pass stage is ~1e4 times
// larger than its output (1e-9 LU deviations). The literal recurrence is kept, but only its loop-carried part
// runs serially: per chunk of kEbChunk frames
//   (1) all lanes convert the chunk to f64 into LDS (coalesced loads) and track the sample peak,
//   (2) lane c walks   v0 = x - a1*v1 - a2*v2 - a3*v3 - a4*v4   for channel c (8 f64 ops per sample, the only
//       dependent chain) and leaves the v sequence in LDS,
//   (3) all lanes evaluate   y = b0*v0 + b1*v1 + b2*v2 + b3*v3 + b4*v4   (same expression, same order) and store the
//       ring coalesced.
// Results are bit-identical to the one-lane-does-everything form; the serial lane executes ~1/3 of the instructions.
// src element (i,c) at src[i*stride_f + c*stride_c].
constexpr int kEbChunk = 256;
template <typename T>
__global__ __launch_bounds__(256) void eb_filter_kernel(const T *__restrict__ src, const EbSeg *__restrict__ segs, size_t stride_f, size_t stride_c,
                                                        double *__restrict__ ring, unsigned channels,
                                                        const int *__restrict__ channel_class, double *__restrict__ vstate,
                                                        unsigned long long *__restrict__ peak, EbFilterK k, size_t src_ss, size_t ring_ss) {
  // batch of independent streams: block y works on stream y (src_ss / ring_ss = elements between consecutive streams)
  const EbSeg seg = segs[blockIdx.y];
  const size_t n = (size_t)seg.n, ring_frame0 = (size_t)seg.ring0;
  if (n == 0) return;   // (block-uniform)
  src += (size_t)blockIdx.y * src_ss + (size_t)seg.src0 * stride_f;
  ring += (size_t)blockIdx.y * ring_ss;
  vstate += (size_t)blockIdx.y * channels * 4;
  if (peak) peak += (size_t)blockIdx.y * 2 * channels;
  extern __shared__ double eb_sm[];          // [channels][kEbChunk + 4] : 4 carried v values, then x -> v in place
  __shared__ unsigned long long s_peak[64];
  const unsigned tid = threadIdx.x;
  const unsigned row = kEbChunk + 4;
  if (tid < channels) {
    s_peak[tid] = 0ull;
    // carried state v4, v3, v2, v1 in front of the chunk (oldest first)
    eb_sm[tid * row + 0] = vstate[tid * 4 + 3];
    eb_sm[tid * row + 1] = vstate[tid * 4 + 2];
    eb_sm[tid * row + 2] = vstate[tid * 4 + 1];
    eb_sm[tid * row + 3] = vstate[tid * 4 + 0];
  }
  __syncthreads();
  for (size_t i0 = 0; i0 < n; i0 += kEbChunk) {
    const unsigned len = (unsigned)(n - i0 < (size_t)kEbChunk ? n - i0 : (size_t)kEbChunk);
    // (1) stage + peak
    for (unsigned e = tid; e < len * channels; e += 256) {
      unsigned f, c;
      if (stride_c == 1) { f = e / channels; c = e - f * channels; }   // interleaved: consecutive lanes = consecutive elements
      else { c = e / len; f = e - c * len; }                            // planar
      const double x = eb_to_double<T>(src[(i0 + f) * stride_f + (size_t)c * stride_c]);
      eb_sm[c * row + 4 + f] = x;
      if (peak) {
        const double ax = x < 0.0 ? -x : x;
        atomicMax(&s_peak[c], (unsigned long long)__double_as_longlong(ax));  // non-negative doubles order like their bits
      }
    }
    __syncthreads();
    // (2) the recurrence, one lane per used channel
    if (tid < channels && channel_class[tid] != 0) {
      double *r = eb_sm + tid * row;
      double v4 = r[0], v3 = r[1], v2 = r[2], v1 = r[3];
      const double a1 = k.a[1], a2 = k.a[2], a3 = k.a[3], a4 = k.a[4];
      unsigned f = 0;
      // 8 samples per trip: the x values are read ahead of the dependent chain and the v values written after it, so
      // the LDS latency is off the loop-carried path (mul + 4 sub per sample)
      for (; f + 8 <= len; f += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = r[4 + f + u];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const double v0 = x[u] - a1 * v1 - a2 * v2 - a3 * v3 - a4 * v4;
          x[u] = v0;
          v4 = v3; v3 = v2; v2 = v1; v1 = v0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) r[4 + f + u] = x[u];
      }
      for (; f < len; f++) {
        const double v0 = r[4 + f] - a1 * v1 - a2 * v2 - a3 * v3 - a4 * v4;
        r[4 + f] = v0;
        v4 = v3; v3 = v2; v2 = v1; v1 = v0;
      }
    }
    __syncthreads();
    // (3) outputs
    for (unsigned e = tid; e < len * channels; e += 256) {
      const unsigned f = e / channels, c = e - f * channels;
      if (channel_class[c] == 0) continue;
      const double *r = eb_sm + c * row + 4 + f;  // r[0] = v0 of this frame, r[-1] = v1, ...
      ring[(ring_frame0 + i0 + f) * channels + c] = k.b[0] * r[0] + k.b[1] * r[-1] + k.b[2] * r[-2] + k.b[3] * r[-3] + k.b[4] * r[-4];
    }
    __syncthreads();
    // carry the last four v values to the front for the next chunk
    if (tid < channels && channel_class[tid] != 0) {
      double *r = eb_sm + tid * row;
      const double a0 = r[len + 0], a1 = r[len + 1], a2 = r[len + 2], a3 = r[len + 3];  // safe for len >= 1: indices len..len+3 hold v(len-4..len-1)
      r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
    }
    __syncthreads();
  }
  if (tid < channels) {
    if (channel_class[tid] != 0) {  // libebur128 flushes denormal state at the end of every filtered segment
      const double *r = eb_sm + tid * row;
      const double v1 = r[3], v2 = r[2], v3 = r[1], v4 = r[0];
      vstate[tid * 4 + 0] = fabs(v1) < DBL_MIN ? 0.0 : v1;
      vstate[tid * 4 + 1] = fabs(v2) < DBL_MIN ? 0.0 : v2;
      vstate[tid * 4 + 2] = fabs(v3) < DBL_MIN ? 0.0 : v3;
      vstate[tid * 4 + 3] = fabs(v4) < DBL_MIN ? 0.0 : v4;
    }
    if (peak && s_peak[tid] > peak[tid]) peak[tid] = s_peak[tid];
  }
}

// Weighted mean square of the last `frames` frames before ring frame `end_frame` (wrapping), all channels.
// eb_energy_partial_kernel: grid of kEbEnergyBlocks blocks, block b sums its slice per channel -> partial[slot][b][c];
// eb_energy_final_kernel: fixed-order sum over the blocks, channel weights, division -> out[slot] (deterministic).
constexpr unsigned kEbEnergyBlocks = 32;
__global__ __launch_bounds__(kEbNT) void eb_energy_partial_kernel(const double *__restrict__ ring, size_t ring_frames, const EbEv *__restrict__ evs,
                                                                  size_t frames, unsigned channels, double *__restrict__ partial,
                                                                  size_t ring_ss, unsigned slots_per_stream) {
  const EbEv ev = evs[blockIdx.y];
  if (ev.slot < 0) return;   // (block-uniform)
  const size_t end_frame = (size_t)ev.end_frame;
  const unsigned slot = (unsigned)ev.slot;
  ring += (size_t)blockIdx.y * ring_ss;
  partial += (size_t)blockIdx.y * slots_per_stream * gridDim.x * channels;
  __shared__ double wave_sum[kEbNT / 64];
  const size_t per = (frames + gridDim.x - 1) / gridDim.x;
  const size_t i0 = (size_t)blockIdx.x * per, i1 = i0 + per < frames ? i0 + per : frames;
  for (unsigned c = 0; c < channels; c++) {
    double s = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += kEbNT) {
      size_t f = end_frame + ring_frames - frames + i;  // end_frame - frames + i, modulo the ring
      if (f >= ring_frames) f -= ring_frames;
      const double x = ring[f * channels + c];
      s += x * x;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double cs = 0.0;
      for (int w = 0; w < kEbNT / 64; w++) cs += wave_sum[w];
      partial[((size_t)slot * gridDim.x + blockIdx.x) * channels + c] = cs;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void eb_energy_final_kernel(const double *__restrict__ partial, unsigned blocks, size_t frames, unsigned channels,
                                                             const int *__restrict__ channel_class, double *__restrict__ out, const EbEv *__restrict__ evs,
                                                             unsigned slots_per_stream) {
  if (evs[blockIdx.x].slot < 0) return;
  const unsigned slot = (unsigned)evs[blockIdx.x].slot;
  partial += (size_t)blockIdx.x * slots_per_stream * blocks * channels;
  out += (size_t)blockIdx.x * slots_per_stream;
  if (threadIdx.x != 0) return;
  double total = 0.0;
  for (unsigned c = 0; c < channels; c++) {
    const int cls = channel_class[c];
    if (cls == 0) continue;
    double cs = 0.0;
    for (unsigned b = 0; b < blocks; b++) cs += partial[((size_t)slot * blocks + b) * channels + c];
    if (cls == 2) cs *= 1.41; else if (cls == 3) cs *= 2.0;
    total += cs;
  }
  out[slot] = total / (double)frames;
}

// True-peak interpolator: polyphase FIR with per-phase tap lists (coeff/index tables, `delay` = taps per phase).
// Input sample i of channel c is xin(i) = i >= 0 ? (float)src : tail[c][delay + i]; output = max |(float)acc|.
struct EbInterpK {
  unsigned factor, delay;
  unsigned count[4];
  unsigned index[4][25];   // 49 taps: <= 13 per phase at factor 4, <= 25 at factor 2
  double coeff[4][25];
};

template <typename T>
__global__ __launch_bounds__(kEbNT) void eb_truepeak_kernel(const T *__restrict__ src, const EbSeg *__restrict__ segs, size_t stride_f, size_t stride_c,
                                                            float *__restrict__ tail, unsigned long long *__restrict__ peak,
                                                            EbInterpK ik, size_t src_ss, unsigned channels) {
  const unsigned c = blockIdx.x;
  const EbSeg seg = segs[blockIdx.y];
  const size_t n = (size_t)seg.n;
  if (n == 0) return;   // (block-uniform)
  src += (size_t)blockIdx.y * src_ss + (size_t)seg.src0 * stride_f;
  tail += (size_t)blockIdx.y * channels * ik.delay;
  peak += (size_t)blockIdx.y * 2 * channels;
  const T *sp = src + (size_t)c * stride_c;
  float *tl = tail + (size_t)c * ik.delay;
  __shared__ double wave_max[kEbNT / 64];
  double mx = 0.0;
  for (size_t i = threadIdx.x; i < n; i += kEbNT) {
    for (unsigned f = 0; f < ik.factor; f++) {
      double acc = 0.0;
      for (unsigned t = 0; t < ik.count[f]; t++) {
        const long j = (long)i - (long)ik.index[f][t];
        const float z = j >= 0 ? (float)eb_to_double<T>(sp[(size_t)j * stride_f]) : tl[(long)ik.delay + j];
        acc += (double)z * ik.coeff[f][t];
      }
      double o = (double)(float)acc;
      if (o < 0.0) o = -o;
      if (o > mx) mx = o;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(mx, off, 64);
    if (o > mx) mx = o;
  }
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = wave_max[0];
    for (int w = 1; w < kEbNT / 64; w++) if (wave_max[w] > m) m = wave_max[w];
    atomicMax(&peak[c], (unsigned long long)__double_as_longlong(m));
  }
  __syncthreads();
  // new history: the last `delay` inputs (older history shifts down when n < delay)
  float z = 0.0f;
  if (threadIdx.x < ik.delay) {
    const long j = (long)n - (long)ik.delay + (long)threadIdx.x;
    z = j >= 0 ? (float)eb_to_double<T>(sp[(size_t)j * stride_f]) : tl[(long)ik.delay + j];
  }
  __syncthreads();
  if (threadIdx.x < ik.delay) tl[threadIdx.x] = z;
}

// ---------------------------------------------------------------- host side state machine

struct Ebur128State {
  unsigned n_streams = 1;  // independent streams of identical configuration (batch entry points): fed together or one by one, each with its own phase
  unsigned channels = 0, rate = 0, mode = 0;
  std::vector<int> channel_class;
  EbFilterK fk{};
  EbInterpK ik{};
  bool have_interp = false;
  size_t samples_in_100ms = 0, ring_frames = 0;
  std::vector<size_t> index_frames, needed_frames, st_counter;   // [n_streams]: every stream has its own 100 ms phase
  std::vector<unsigned long> block_hist, st_hist;  // [n_streams][kHistBins]
  std::vector<double> sample_peak, true_peak;      // [n_streams][channels]
  // device
  double *d_ring = nullptr, *d_vstate = nullptr, *d_energy = nullptr, *d_partial = nullptr;
  int *d_class = nullptr;
  unsigned long long *d_peak = nullptr;  // [channels] sample peak bits, [channels] true peak bits
  float *d_tail = nullptr;
  void *d_in = nullptr;
  size_t d_in_bytes = 0;
  size_t energy_cap = 0;
  // the rounds of one call: [rounds][n_streams] segments, then [rounds][2][n_streams] energy events (pinned + device copy)
  char *h_tab = nullptr, *d_tab = nullptr;
  size_t tab_bytes = 0;
};

static double g_hist_energy[kHistBins], g_hist_bound[kHistBins + 1];
static std::once_flag g_hist_once;
static void hist_tables() {
  std::call_once(g_hist_once, [] {
    for (int i = 0; i < kHistBins; i++) g_hist_energy[i] = std::pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
    for (int i = 0; i <= kHistBins; i++) g_hist_bound[i] = std::pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
  });
}
static size_t hist_index(double e) {
  size_t lo = 0, hi = kHistBins;
  while (hi - lo != 1) {
    const size_t mid = (lo + hi) / 2;
    if (e >= g_hist_bound[mid]) lo = mid; else hi = mid;
  }
  return lo;
}
static double to_loudness(double e) { return 10.0 * (std::log(e) / std::log(10.0)) - 0.691; }

static void k_weighting(unsigned rate, EbFilterK *fk) {
  // BS.1770 pre-filter (high shelf) and RLB (high pass), bilinear transform at `rate`, convolved
  double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
  double K = std::tan(kPi * f0 / (double)rate);
  const double Vh = std::pow(10.0, G / 20.0), Vb = std::pow(Vh, 0.4996667741545416);
  double pb[3], pa[3] = {1.0, 0.0, 0.0};
  const double rb[3] = {1.0, -2.0, 1.0};
  double ra[3] = {1.0, 0.0, 0.0};
  const double a0 = 1.0 + K / Q + K * K;
  pb[0] = (Vh + Vb * K / Q + K * K) / a0;
  pb[1] = 2.0 * (K * K - Vh) / a0;
  pb[2] = (Vh - Vb * K / Q + K * K) / a0;
  pa[1] = 2.0 * (K * K - 1.0) / a0;
  pa[2] = (1.0 - K / Q + K * K) / a0;
  f0 = 38.13547087602444; Q = 0.5003270373238773;
  K = std::tan(kPi * f0 / (double)rate);
  ra[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
  ra[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
  fk->b[0] = pb[0] * rb[0];
  fk->b[1] = pb[0] * rb[1] + pb[1] * rb[0];
  fk->b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
  fk->b[3] = pb[1] * rb[2] + pb[2] * rb[1];
  fk->b[4] = pb[2] * rb[2];
  fk->a[0] = pa[0] * ra[0];
  fk->a[1] = pa[0] * ra[1] + pa[1] * ra[0];
  fk->a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
  fk->a[3] = pa[1] * ra[2] + pa[2] * ra[1];
  fk->a[4] = pa[2] * ra[2];
}

static void interp_tables(unsigned taps, unsigned factor, EbInterpK *ik) {
  std::memset(ik, 0, sizeof *ik);
  ik->factor = factor;
  ik->delay = (taps + factor - 1) / factor;
  for (unsigned j = 0; j < taps; j++) {
    const double m = (double)j - (double)(taps - 1) / 2.0;
    double c = 1.0;
    if (std::fabs(m) > 0.000001) c = std::sin(m * kPi / factor) / (m * kPi / factor);
    c *= 0.5 * (1.0 - std::cos(2.0 * kPi * j / (taps - 1)));
    if (std::fabs(c) > 0.000001) {
      const unsigned f = j % factor, t = ik->count[f]++;
      ik->coeff[f][t] = c;
      ik->index[f][t] = j / factor;
    }
  }
}

void ebur128_release(mi355_ctx *ctx) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return;
  if (st->d_ring) (void)hipFree(st->d_ring);
  if (st->d_vstate) (void)hipFree(st->d_vstate);
  if (st->d_partial) (void)hipFree(st->d_partial);
  if (st->d_energy) (void)hipFree(st->d_energy);
  if (st->d_class) (void)hipFree(st->d_class);
  if (st->d_peak) (void)hipFree(st->d_peak);
  if (st->d_tail) (void)hipFree(st->d_tail);
  if (st->d_in) (void)hipFree(st->d_in);
  if (st->h_tab) (void)hipHostFree(st->h_tab);
  if (st->d_tab) (void)hipFree(st->d_tab);
  delete st;
  ctx->ebur128 = nullptr;
}

// streams [s0, s0 + S) back to the state of a new meter
static int eb_reset_device(mi355_ctx *ctx, Ebur128State *st, size_t s0, size_t S) {
  const size_t C = st->channels;
  int rc = check_hip(ctx, hipMemsetAsync(st->d_ring + s0 * st->ring_frames * C, 0, S * st->ring_frames * C * sizeof(double), ctx->stream), "hipMemset(ebur128 ring)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMemsetAsync(st->d_vstate + s0 * C * 4, 0, S * C * 4 * sizeof(double), ctx->stream), "hipMemset(ebur128 state)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMemsetAsync(st->d_peak + s0 * 2 * C, 0, S * 2 * C * sizeof(unsigned long long), ctx->stream), "hipMemset(ebur128 peaks)");
  if (rc) return rc;
  if (st->d_tail) rc = check_hip(ctx, hipMemsetAsync(st->d_tail + s0 * C * st->ik.delay, 0, S * C * st->ik.delay * sizeof(float), ctx->stream), "hipMemset(ebur128 tail)");
  return rc;
}
static void eb_reset_host(Ebur128State *st, size_t s0, size_t S) {
  const size_t C = st->channels;
  std::fill(st->block_hist.begin() + s0 * kHistBins, st->block_hist.begin() + (s0 + S) * kHistBins, 0ul);
  std::fill(st->st_hist.begin() + s0 * kHistBins, st->st_hist.begin() + (s0 + S) * kHistBins, 0ul);
  std::fill(st->sample_peak.begin() + s0 * C, st->sample_peak.begin() + (s0 + S) * C, 0.0);
  std::fill(st->true_peak.begin() + s0 * C, st->true_peak.begin() + (s0 + S) * C, 0.0);
  for (size_t s = s0; s < s0 + S; s++) { st->index_frames[s] = 0; st->needed_frames[s] = st->samples_in_100ms * 4; st->st_counter[s] = 0; }
}

// n_streams independent meters of one configuration (ebur128_setup = one stream): mi355_ebur128_add_frames_batch feeds them the same
// number of frames each, ebur128_add_frames_streams (the audio groups) any number per stream
int ebur128_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, unsigned rate, unsigned mode, const int *channel_class) {
  hist_tables();
  ebur128_release(ctx);
  if (n_streams == 0 || n_streams > 65535) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: bad stream count");
  if (channels == 0 || channels > 64 || rate < 16 || rate > 2822400) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: bad channels/rate");
  // cumulative mode bits as in libebur128 (TRUE_PEAK -> SAMPLE_PEAK, LRA -> S, S/I -> M)
  if (mode & EB_TRUE_PEAK) mode |= EB_SAMPLE_PEAK;
  if (mode & EB_LRA) mode |= EB_S;
  if (mode & (EB_S | EB_I)) mode |= EB_M;
  Ebur128State *st = new Ebur128State();
  ctx->ebur128 = st;
  st->n_streams = n_streams;
  st->channels = channels; st->rate = rate; st->mode = mode;
  st->channel_class.assign(channels, 1);
  for (unsigned c = 0; c < channels; c++) {
    if (channel_class) st->channel_class[c] = channel_class[c];
    else st->channel_class[c] = (c == 3) ? 0 : ((c == 4 || c == 5) ? 2 : 1);  // libebur128 default map
    if (st->channel_class[c] < 0 || st->channel_class[c] > 3) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: bad channel class");
  }
  k_weighting(rate, &st->fk);
  st->samples_in_100ms = (rate + 5) / 10;
  const size_t window = (mode & EB_S) ? 3000 : 400;
  st->ring_frames = (size_t)rate * window / 1000;
  if (st->ring_frames % st->samples_in_100ms) st->ring_frames += st->samples_in_100ms - st->ring_frames % st->samples_in_100ms;
  const size_t S = n_streams;
  st->index_frames.assign(S, 0);
  st->needed_frames.assign(S, st->samples_in_100ms * 4);
  st->st_counter.assign(S, 0);
  st->block_hist.assign(S * kHistBins, 0ul);
  st->st_hist.assign(S * kHistBins, 0ul);
  st->sample_peak.assign(S * channels, 0.0);
  st->true_peak.assign(S * channels, 0.0);
  if ((mode & EB_TRUE_PEAK) && rate < 192000) {
    interp_tables(49, rate < 96000 ? 4 : 2, &st->ik);
    st->have_interp = true;
  }
  int rc = check_hip(ctx, hipMalloc((void **)&st->d_ring, S * st->ring_frames * channels * sizeof(double)), "hipMalloc(ebur128 ring)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMalloc((void **)&st->d_vstate, S * channels * 4 * sizeof(double)), "hipMalloc(ebur128 state)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMalloc((void **)&st->d_class, channels * sizeof(int)), "hipMalloc(ebur128 classes)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMalloc((void **)&st->d_peak, S * 2 * channels * sizeof(unsigned long long)), "hipMalloc(ebur128 peaks)");
  if (rc) return rc;
  if (st->have_interp) {
    rc = check_hip(ctx, hipMalloc((void **)&st->d_tail, S * channels * st->ik.delay * sizeof(float)), "hipMalloc(ebur128 tail)");
    if (rc) return rc;
  }
  rc = check_hip(ctx, hipMemcpyAsync(st->d_class, st->channel_class.data(), channels * sizeof(int), hipMemcpyHostToDevice, ctx->stream),
                 "hipMemcpy(ebur128 classes)");
  if (rc) return rc;
  rc = eb_reset_device(ctx, st, 0, S);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "ebur128: stream synchronize");
}

int ebur128_setup(mi355_ctx *ctx, unsigned channels, unsigned rate, unsigned mode, const int *channel_class) {
  return ebur128_setup_batch(ctx, 1, channels, rate, mode, channel_class);
}

int ebur128_reset(mi355_ctx *ctx) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  eb_reset_host(st, 0, st->n_streams);
  int rc = eb_reset_device(ctx, st, 0, st->n_streams);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "ebur128: stream synchronize");
}

// one stream of a batch back to the state of a new meter (the `reset` action of one ebur128level instance, imp.rs:320-333); the
// other streams keep their history and their phase
int ebur128_reset_stream(mi355_ctx *ctx, unsigned stream) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (stream >= st->n_streams) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: no such stream");
  eb_reset_host(st, stream, 1);
  int rc = eb_reset_device(ctx, st, stream, 1);
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "ebur128: stream synchronize");
}

// room for `bytes` of round tables (pinned + device); the previous call's tables are dead: every call ends with a stream synchronize
static int eb_table_capacity(mi355_ctx *ctx, Ebur128State *st, size_t bytes) {
  if (st->tab_bytes >= bytes) return MI355_OK;
  if (st->h_tab) (void)hipHostFree(st->h_tab);
  if (st->d_tab) (void)hipFree(st->d_tab);
  st->h_tab = nullptr; st->d_tab = nullptr; st->tab_bytes = 0;
  size_t cap = 4096;
  while (cap < bytes) cap *= 2;
  int rc = check_hip(ctx, hipHostMalloc((void **)&st->h_tab, cap, hipHostMallocDefault), "hipHostMalloc(ebur128 round tables)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMalloc((void **)&st->d_tab, cap), "hipMalloc(ebur128 round tables)");
  if (rc) return rc;
  st->tab_bytes = cap;
  return MI355_OK;
}

// energy of the `frames` ring frames before evs[stream].end_frame of every stream that has an event -> d_energy[stream][slot]
static void eb_launch_energy(mi355_ctx *ctx, Ebur128State *st, size_t frames, const EbEv *d_evs) {
  const size_t ring_ss = st->ring_frames * st->channels;
  hipLaunchKernelGGL(eb_energy_partial_kernel, dim3(kEbEnergyBlocks, st->n_streams), dim3(kEbNT), 0, ctx->stream, (const double *)st->d_ring,
                     st->ring_frames, d_evs, frames, st->channels, st->d_partial, ring_ss, (unsigned)st->energy_cap);
  hipLaunchKernelGGL(eb_energy_final_kernel, dim3(st->n_streams), dim3(64), 0, ctx->stream, (const double *)st->d_partial, kEbEnergyBlocks, frames,
                     st->channels, (const int *)st->d_class, st->d_energy, d_evs, (unsigned)st->energy_cap);
}

// one round: the segment d_segs[stream] of every stream. src_ss: elements between the buffers of consecutive streams
template <typename T>
static void eb_launch_segment(mi355_ctx *ctx, Ebur128State *st, const T *d_src, const EbSeg *d_segs, size_t stride_f, size_t stride_c, size_t src_ss) {
  unsigned long long *speak = (st->mode & EB_SAMPLE_PEAK) ? st->d_peak : nullptr;
  if (st->have_interp)
    hipLaunchKernelGGL((eb_truepeak_kernel<T>), dim3(st->channels, st->n_streams), dim3(kEbNT), 0, ctx->stream, d_src, d_segs, stride_f, stride_c, st->d_tail,
                       st->d_peak + st->channels, st->ik, src_ss, st->channels);
  const size_t filter_lds = (size_t)st->channels * (kEbChunk + 4) * sizeof(double);
  (void)hipFuncSetAttribute((const void *)eb_filter_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)filter_lds);
  hipLaunchKernelGGL((eb_filter_kernel<T>), dim3(1, st->n_streams), dim3(256), filter_lds, ctx->stream, d_src, d_segs, stride_f, stride_c, st->d_ring,
                     st->channels, (const int *)st->d_class, st->d_vstate, speak, st->fk, src_ss, st->ring_frames * st->channels);
}

static int eb_energy_capacity(mi355_ctx *ctx, Ebur128State *st, size_t events) {
  if (st->energy_cap >= events) return MI355_OK;
  if (st->d_energy) (void)hipFree(st->d_energy);
  if (st->d_partial) (void)hipFree(st->d_partial);
  st->d_energy = nullptr; st->d_partial = nullptr; st->energy_cap = 0;
  int rc = check_hip(ctx, hipMalloc((void **)&st->d_energy, st->n_streams * events * sizeof(double)), "hipMalloc(ebur128 energies)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMalloc((void **)&st->d_partial, st->n_streams * events * kEbEnergyBlocks * st->channels * sizeof(double)), "hipMalloc(ebur128 partial energies)");
  if (rc) return rc;
  st->energy_cap = events;
  return MI355_OK;
}

// fmt: 0 s16, 1 s32, 2 f32, 3 f64. `planes`: nullptr for interleaved `data`, else `channels` plane pointers (one stream only).
// Batch: `data` holds n_streams slots of `slot_elems` samples each; stream s takes frames_per[s] frames (x channels interleaved
// samples) from the start of its slot - none is allowed.
template <typename T>
static int eb_add_frames_t(mi355_ctx *ctx, Ebur128State *st, const T *data, const T *const *planes, const size_t *frames_per, size_t slot_elems, bool device_data) {
  const unsigned C = st->channels;
  const size_t S = st->n_streams;
  size_t frames = 0;   // the longest buffer of this call
  for (size_t s = 0; s < S; s++) frames = frames_per[s] > frames ? frames_per[s] : frames;
  if (frames == 0) return MI355_OK;
  const size_t bytes = S * slot_elems * sizeof(T);
  const T *d_src = data;
  size_t stride_f, stride_c;
  if (device_data) {
    stride_f = C; stride_c = 1;
  } else {
    if (st->d_in_bytes < bytes) {
      if (st->d_in) (void)hipFree(st->d_in);
      st->d_in = nullptr; st->d_in_bytes = 0;
      int rc = check_hip(ctx, hipMalloc(&st->d_in, bytes), "hipMalloc(ebur128 input)");
      if (rc) return rc;
      st->d_in_bytes = bytes;
    }
    if (planes) {
      for (unsigned c = 0; c < C; c++) {
        int rc = check_hip(ctx, hipMemcpyAsync((T *)st->d_in + (size_t)c * frames, planes[c], frames * sizeof(T), hipMemcpyHostToDevice, ctx->stream),
                           "hipMemcpyAsync(ebur128 plane)");
        if (rc) return rc;
      }
      stride_f = 1; stride_c = frames;
    } else {
      int rc = check_hip(ctx, hipMemcpyAsync(st->d_in, data, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(ebur128 input)");
      if (rc) return rc;
      stride_f = C; stride_c = 1;
    }
    d_src = (const T *)st->d_in;
  }
  const size_t src_ss = slot_elems;
  // worst case one gating block + one short-term block per 100 ms
  int rc = eb_energy_capacity(ctx, st, 2 * (frames / st->samples_in_100ms + 2));
  if (rc) return rc;
  // ---- the add_frames loop of libebur128 (filter up to the next 100 ms boundary, then gate), walked per stream: round r of a
  // stream is its r-th segment and the blocks that segment completes
  struct Round { EbSeg seg; EbEv ev[2]; };
  std::vector<std::vector<Round>> rounds(S);
  std::vector<std::vector<int>> event_kind(S);  // per stream: 0 = gating block (I), 1 = short-term block (LRA), in stream order
  size_t R = 0;
  for (size_t s = 0; s < S; s++) {
    size_t src_index = 0, left = frames_per[s];
    size_t &index = st->index_frames[s], &needed = st->needed_frames[s], &stc = st->st_counter[s];
    while (left > 0) {
      Round rd;
      rd.ev[0].slot = rd.ev[1].slot = -1;
      rd.ev[0].end_frame = rd.ev[1].end_frame = 0;
      if (left >= needed) {
        rd.seg = EbSeg{src_index, needed, index};
        src_index += needed;
        left -= needed;
        index += needed;
        if (st->mode & EB_I) {
          rd.ev[0] = EbEv{index, (long long)event_kind[s].size()};
          event_kind[s].push_back(0);
        }
        if (st->mode & EB_LRA) {
          stc += needed;
          if (stc == st->samples_in_100ms * 30) {
            rd.ev[1] = EbEv{index, (long long)event_kind[s].size()};
            event_kind[s].push_back(1);
            stc = st->samples_in_100ms * 20;
          }
        }
        needed = st->samples_in_100ms;
        if (index == st->ring_frames) index = 0;
      } else {
        rd.seg = EbSeg{src_index, left, index};
        index += left;
        if (st->mode & EB_LRA) stc += left;
        needed -= left;
        left = 0;
      }
      rounds[s].push_back(rd);
    }
    R = rounds[s].size() > R ? rounds[s].size() : R;
  }
  const size_t seg_bytes = R * S * sizeof(EbSeg), ev_bytes = R * 2 * S * sizeof(EbEv);
  if ((rc = eb_table_capacity(ctx, st, seg_bytes + ev_bytes))) return rc;
  EbSeg *h_segs = (EbSeg *)st->h_tab;
  EbEv *h_evs = (EbEv *)(st->h_tab + seg_bytes);
  std::vector<char> any_ev(R * 2, 0);
  for (size_t r = 0; r < R; r++)
    for (size_t s = 0; s < S; s++) {
      const bool have = r < rounds[s].size();
      h_segs[r * S + s] = have ? rounds[s][r].seg : EbSeg{0, 0, 0};
      for (int k = 0; k < 2; k++) {
        h_evs[(r * 2 + k) * S + s] = have ? rounds[s][r].ev[k] : EbEv{0, -1};
        if (have && rounds[s][r].ev[k].slot >= 0) any_ev[r * 2 + k] = 1;
      }
    }
  rc = check_hip(ctx, hipMemcpyAsync(st->d_tab, st->h_tab, seg_bytes + ev_bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(ebur128 round tables)");
  if (rc) return rc;
  const EbSeg *d_segs = (const EbSeg *)st->d_tab;
  const EbEv *d_evs = (const EbEv *)(st->d_tab + seg_bytes);
  bool any_event = false;
  for (size_t r = 0; r < R; r++) {
    eb_launch_segment<T>(ctx, st, d_src, d_segs + r * S, stride_f, stride_c, src_ss);
    if (any_ev[r * 2 + 0]) { eb_launch_energy(ctx, st, st->samples_in_100ms * 4, d_evs + (r * 2 + 0) * S); any_event = true; }
    if (any_ev[r * 2 + 1]) { eb_launch_energy(ctx, st, st->samples_in_100ms * 30, d_evs + (r * 2 + 1) * S); any_event = true; }
  }
  rc = check_hip(ctx, hipGetLastError(), "ebur128 kernel launch");
  if (rc) return rc;
  std::vector<double> energies(S * st->energy_cap);
  std::vector<unsigned long long> peaks(S * 2 * C);
  if (any_event) {
    rc = check_hip(ctx, hipMemcpyAsync(energies.data(), st->d_energy, energies.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream),
                   "hipMemcpyAsync(ebur128 energies)");
    if (rc) return rc;
  }
  rc = check_hip(ctx, hipMemcpyAsync(peaks.data(), st->d_peak, peaks.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream),
                 "hipMemcpyAsync(ebur128 peaks)");
  if (rc) return rc;
  rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "ebur128: stream synchronize");
  if (rc) return rc;
  for (size_t s = 0; s < S; s++) {
    for (size_t k = 0; k < event_kind[s].size(); k++) {
      const double e = energies[s * st->energy_cap + k];
      if (e >= g_hist_bound[0]) {
        if (event_kind[s][k] == 0) st->block_hist[s * kHistBins + hist_index(e)]++;
        else st->st_hist[s * kHistBins + hist_index(e)]++;
      }
    }
    for (unsigned c = 0; c < C; c++) {  // device peaks are running maxima since the last reset
      double sp, tp;
      std::memcpy(&sp, &peaks[s * 2 * C + c], 8);
      std::memcpy(&tp, &peaks[s * 2 * C + C + c], 8);
      if (sp > st->sample_peak[s * C + c]) st->sample_peak[s * C + c] = sp;
      if (tp > st->true_peak[s * C + c]) st->true_peak[s * C + c] = tp;
    }
  }
  return MI355_OK;
}

static int eb_add_dispatch(mi355_ctx *ctx, Ebur128State *st, const void *data, const void *const *planes, const size_t *frames_per, size_t slot_elems, int fmt,
                           bool device_data) {
  switch (fmt) {
    case 0: return eb_add_frames_t<int16_t>(ctx, st, (const int16_t *)data, (const int16_t *const *)planes, frames_per, slot_elems, device_data);
    case 1: return eb_add_frames_t<int32_t>(ctx, st, (const int32_t *)data, (const int32_t *const *)planes, frames_per, slot_elems, device_data);
    case 2: return eb_add_frames_t<float>(ctx, st, (const float *)data, (const float *const *)planes, frames_per, slot_elems, device_data);
    case 3: return eb_add_frames_t<double>(ctx, st, (const double *)data, (const double *const *)planes, frames_per, slot_elems, device_data);
    default: return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: bad sample format");
  }
}

int ebur128_add_frames(mi355_ctx *ctx, const void *data, const void *const *planes, size_t frames, int fmt) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (st->n_streams != 1) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: this meter is a batch, use the batch entry points");
  if (frames && !data && !planes) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: null data");
  return eb_add_dispatch(ctx, st, data, planes, &frames, frames * st->channels, fmt, false);
}

int ebur128_add_frames_batch(mi355_ctx *ctx, const void *data, size_t frames, int fmt, int device_data) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (frames && !data) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: null data");
  const std::vector<size_t> per(st->n_streams, frames);
  return eb_add_dispatch(ctx, st, data, nullptr, per.data(), frames * st->channels, fmt, device_data != 0);
}

// the streams of a batch fed independently: stream s takes frames_per[s] frames (none is allowed) from slot s of `data`, the slots
// `slot_elems` samples apart - every stream keeps its own 100 ms phase, so buffer sizes may differ between streams and calls
int ebur128_add_frames_streams(mi355_ctx *ctx, const void *data, size_t slot_elems, const size_t *frames_per, int fmt, int device_data) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (!data || !frames_per) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: null data");
  for (size_t s = 0; s < st->n_streams; s++)
    if (frames_per[s] * st->channels > slot_elems) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: a stream's buffer is larger than its slot");
  return eb_add_dispatch(ctx, st, data, nullptr, frames_per, slot_elems, fmt, device_data != 0);
}

// window energy of every stream -> out[n_streams]
static int eb_window_energy(mi355_ctx *ctx, Ebur128State *st, size_t frames, std::vector<double> &out) {
  if (frames > st->ring_frames) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: interval larger than the history window");
  int rc = eb_energy_capacity(ctx, st, 16);
  if (rc) return rc;
  if ((rc = eb_table_capacity(ctx, st, st->n_streams * sizeof(EbEv)))) return rc;
  EbEv *h_evs = (EbEv *)st->h_tab;
  for (size_t s = 0; s < st->n_streams; s++) h_evs[s] = EbEv{st->index_frames[s], 0};   // every stream's window ends at its own write position
  rc = check_hip(ctx, hipMemcpyAsync(st->d_tab, st->h_tab, st->n_streams * sizeof(EbEv), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(ebur128 query table)");
  if (rc) return rc;
  eb_launch_energy(ctx, st, frames, (const EbEv *)st->d_tab);
  std::vector<double> all(st->n_streams * st->energy_cap);
  rc = check_hip(ctx, hipMemcpyAsync(all.data(), st->d_energy, all.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(ebur128 energy)");
  if (rc) return rc;
  rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "ebur128: stream synchronize");
  if (rc) return rc;
  out.resize(st->n_streams);
  for (size_t s = 0; s < st->n_streams; s++) out[s] = all[s * st->energy_cap];
  return MI355_OK;
}

// histogram-derived values of stream s. what: 2 global, 3 relative threshold, 4 loudness range
static double eb_hist_query(const Ebur128State *st, size_t s, int what) {
  const unsigned long *bh = &st->block_hist[s * kHistBins], *sh = &st->st_hist[s * kHistBins];
  if (what == 2 || what == 3) {
    double thr = 0.0; size_t n = 0;
    for (int j = 0; j < kHistBins; j++) { thr += (double)bh[j] * g_hist_energy[j]; n += bh[j]; }
    if (what == 3) return n ? to_loudness(thr / (double)n * 0.1) : -70.0;
    if (!n) return -HUGE_VAL;
    thr = thr / (double)n * 0.1;  // relative gate: -10 LU
    size_t start;
    if (thr < g_hist_bound[0]) start = 0;
    else { start = hist_index(thr); if (thr > g_hist_energy[start]) ++start; }
    double g = 0.0; n = 0;
    for (size_t j = start; j < (size_t)kHistBins; j++) { g += (double)bh[j] * g_hist_energy[j]; n += bh[j]; }
    return n ? to_loudness(g / (double)n) : -HUGE_VAL;
  }
  size_t size = 0; double power = 0.0;
  for (int j = 0; j < kHistBins; j++) { size += sh[j]; power += (double)sh[j] * g_hist_energy[j]; }
  if (!size) return 0.0;
  const double integrated = 0.01 * (power / (double)size);  // -20 LU
  size_t index;
  if (integrated < g_hist_bound[0]) index = 0;
  else { index = hist_index(integrated); if (integrated > g_hist_energy[index]) ++index; }
  size = 0;
  for (size_t j = index; j < (size_t)kHistBins; j++) size += sh[j];
  if (!size) return 0.0;
  const size_t plow = (size_t)((double)(size - 1) * 0.1 + 0.5), phigh = (size_t)((double)(size - 1) * 0.95 + 0.5);
  size_t acc = 0, j = index;
  while (acc <= plow) acc += sh[j++];
  const double l_en = g_hist_energy[j - 1];
  while (acc <= phigh) acc += sh[j++];
  const double h_en = g_hist_energy[j - 1];
  return to_loudness(h_en) - to_loudness(l_en);
}

// what: 0 momentary, 1 short-term, 2 global, 3 relative threshold, 4 loudness range; out[n_streams]
int ebur128_query_batch(mi355_ctx *ctx, int what, double *out) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (!out) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: null output");
  if (what == 0 || what == 1) {
    if (what == 1 && !(st->mode & EB_S)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: short-term mode not enabled");
    std::vector<double> e;
    int rc = eb_window_energy(ctx, st, st->samples_in_100ms * (what == 0 ? 4 : 30), e);
    if (rc) return rc;
    for (size_t s = 0; s < st->n_streams; s++) out[s] = e[s] <= 0.0 ? -HUGE_VAL : to_loudness(e[s]);
    return MI355_OK;
  }
  if (what == 2 || what == 3) {
    if (!(st->mode & EB_I)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: global mode not enabled");
  } else if (what == 4) {
    if (!(st->mode & EB_LRA)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: loudness-range mode not enabled");
  } else {
    return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: unknown query");
  }
  for (size_t s = 0; s < st->n_streams; s++) out[s] = eb_hist_query(st, s, what);
  return MI355_OK;
}

int ebur128_query(mi355_ctx *ctx, int what, double *out) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (!out) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: null output");
  std::vector<double> v(st->n_streams);
  int rc = ebur128_query_batch(ctx, what, v.data());
  if (rc) return rc;
  *out = v[0];
  return MI355_OK;
}

// peaks of every stream: out[n_streams][channels]
int ebur128_peak_batch(mi355_ctx *ctx, int true_peak, double *out) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (!out) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: null output");
  if (true_peak && !(st->mode & EB_TRUE_PEAK)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: true-peak mode not enabled");
  if (!true_peak && !(st->mode & EB_SAMPLE_PEAK)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: sample-peak mode not enabled");
  for (size_t i = 0; i < (size_t)st->n_streams * st->channels; i++)
    out[i] = true_peak ? (st->true_peak[i] > st->sample_peak[i] ? st->true_peak[i] : st->sample_peak[i]) : st->sample_peak[i];
  return MI355_OK;
}

int ebur128_peak(mi355_ctx *ctx, int true_peak, unsigned channel, double *out) {
  Ebur128State *st = (Ebur128State *)ctx->ebur128;
  if (!st) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "ebur128: Have no state yet");
  if (channel >= st->channels || !out) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: bad channel");
  if (true_peak) {
    if (!(st->mode & EB_TRUE_PEAK)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: true-peak mode not enabled");
    *out = st->true_peak[channel] > st->sample_peak[channel] ? st->true_peak[channel] : st->sample_peak[channel];
  } else {
    if (!(st->mode & EB_SAMPLE_PEAK)) return set_error(ctx, MI355_ERR_INVALID_ARG, "ebur128: sample-peak mode not enabled");
    *out = st->sample_peak[channel];
  }
  return MI355_OK;
}

}  // namespace mi355
