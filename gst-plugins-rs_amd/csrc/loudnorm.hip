// loudnorm.hip — audioloudnorm (audio/audiofx/src/audioloudnorm/imp.rs) on the device.
//
// The element's State machine stays a host-side state machine (it is one: gain history, limiter state, frame
// type), but every per-sample loop of the reference runs as a kernel over device-resident rings, and the serial
// peak search of the limiter becomes a parallel first-match reduction:
//   reference loop (imp.rs)                                   device form
//   process_first_frame  :368-442  limiter_buf = buf*gain     ln_scale_kernel
//   process_fill_inner_frame :444-524 (+ final :654-697)      ln_fill_kernel   (per-sample interpolated gain, ring -> ring,
//                                                                               new input written 210 ms behind)
//   detect_peak :1438-1524  serial scan for the first local   ln_detect_kernel (every position tested independently, block
//     maximum above the ceiling, 10 followers checked                           min-reduction; the carried prev_smp only
//                                                                               matters for n == 0, which can never match)
//   attack / sustain / release envelopes :905-1330            ln_envelope_kernel (each sample derives its envelope value from
//                                                                               its own position, same f64 expression)
//   output copy + hard clamp :1404-1429                       ln_output_kernel
// State transitions (:845-1330) are transcribed literally on the host between those launches. r128_in / r128_out are
// two instances of the device loudness meter of ebur128_kernels.hip (modes I|S|LRA|SAMPLE_PEAK, :131-150).
// f64 throughout, unfused: samples are bit-identical to the CPU restatement (oracle/loudnorm_oracle.c).
#include "internal.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace mi355 {

namespace {
constexpr size_t GAIN_LOOKAHEAD = 3 * 192000, FRAME_SIZE = 19200, LIMITER_ATTACK_WINDOW = 1920, LIMITER_RELEASE_WINDOW = 19200,
                 LIMITER_LOOKAHEAD = 1920;
enum { FT_FIRST, FT_INNER, FT_FINAL, FT_LINEAR };
enum { LS_OUT, LS_ATTACK, LS_SUSTAIN, LS_RELEASE };
enum { ENV_CONST = 0, ENV_ATTACK = 1, ENV_RELEASE = 2 };
}  // namespace

struct LoudNormState {
  size_t channels = 0;
  size_t current_samples_per_frame = GAIN_LOOKAHEAD;
  double offset = 1, target_i = 0, target_lra = 0, target_tp = 0;
  double *d_buf = nullptr; size_t buf_len = 0, buf_index = 0, prev_buf_index = 0;
  double weights[21], delta[30]; size_t index = 1; double prev_delta = 0;
  double gain_reduction[2] = {0, 0};
  double *d_limiter = nullptr; size_t limiter_len = 0, limiter_buf_index = 0;
  int limiter_state = LS_OUT; size_t env_cnt = 0; bool have_sustain = false; size_t sustain_cnt = 0;
  int frame_type = FT_FIRST; bool above_threshold = false;
  void *r128_in = nullptr, *r128_out = nullptr;  // Ebur128State
  double *d_src = nullptr, *d_dst = nullptr;      // staging: up to 3 s in, up to 3 s out
  unsigned long long *d_peak = nullptr;           // {first matching n or ~0, bits of the peak value}
  std::vector<double> adapter;                    // UniqueAdapter (host side, f64 samples)
};

static LoudNormState *ln_of(mi355_ctx *ctx) { return (LoudNormState *)ctx->loudnorm; }

// ------------------------------------------------------------------ kernels

__global__ __launch_bounds__(256) void ln_scale_kernel(double *__restrict__ limiter, const double *__restrict__ buf, size_t n, double prev_delta,
                                                       double offset) {
  const size_t gs = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) limiter[i] = buf[i] * prev_delta * offset;
}

// frames [n0, n1): limiter[lidx + (n-n0)] = buf[bidx + (n-n0)] * gain(n); optionally buf[pidx + (n-n0)] = src[n-n0]
__global__ __launch_bounds__(256) void ln_fill_kernel(double *__restrict__ limiter, size_t llen, size_t lidx, double *__restrict__ buf, size_t blen,
                                                      size_t bidx, size_t pidx, const double *__restrict__ src, size_t ch, size_t n0, size_t n1,
                                                      double denom, double gain, double gain_next, double offset) {
  const size_t total = (n1 - n0) * ch, gs = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gs) {
    const size_t n = n0 + i / ch;
    const double current_gain = (gain + (((double)n / denom) * (gain_next - gain))) * offset;
    size_t r = bidx + i; if (r >= blen) r -= blen;
    size_t l = lidx + i; if (l >= llen) l -= llen;
    const double v = buf[r];
    if (src) { size_t w = pidx + i; if (w >= blen) w -= blen; buf[w] = src[i]; }
    limiter[l] = v * current_gain;
  }
}

// first n in [1, samples) at which some channel is a local maximum above the ceiling with no higher sample among the
// 10 followers (detect_peak). One block; result[0] = n (or ~0), result[1] = bits of max_c |this[c]|.
__global__ __launch_bounds__(1024) void ln_detect_kernel(const double *__restrict__ limiter, size_t llen, size_t base /* index of n == 0 */,
                                                         size_t ch, size_t samples, double target_tp, unsigned long long *__restrict__ result) {
  __shared__ unsigned long long s_min;
  if (threadIdx.x == 0) s_min = ~0ull;
  __syncthreads();
  auto at = [&](size_t n, size_t c) -> double {
    size_t i = base + n * ch + c;
    i %= llen;
    return fabs(limiter[i]);
  };
  for (size_t n0 = 1; n0 < samples; n0 += 1024) {
    const size_t n = n0 + threadIdx.x;
    if (n < samples) {
      bool hit = false;
      for (size_t c = 0; c < ch && !hit; c++) {
        const double th = at(n, c);
        if (at(n - 1, c) <= th && th >= at(n + 1, c) && th > target_tp) {
          bool ok = true;
          for (size_t i = 2; i < 12; i++)
            if (at(n + i, c) > th) { ok = false; break; }
          hit = ok;
        }
      }
      if (hit) atomicMin(&s_min, (unsigned long long)n);
    }
    __syncthreads();
    if (s_min != ~0ull) break;  // block-uniform after the barrier
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    result[0] = s_min;
    double mx = 0.0;
    if (s_min != ~0ull)
      for (size_t c = 0; c < ch; c++) { const double v = at((size_t)s_min, c); if (c == 0 || v > mx) mx = v; }
    result[1] = (unsigned long long)__double_as_longlong(mx);
  }
}

// limiter[lidx + (smp + i)] *= env(i), i in [0, count): constant g1, attack ramp or release ramp
__global__ __launch_bounds__(256) void ln_envelope_kernel(double *__restrict__ limiter, size_t llen, size_t start /* element index */, size_t ch,
                                                          size_t count, int mode, double g0, double g1, size_t env_cnt0) {
  const size_t total = count * ch, gs = (size_t)gridDim.x * 256;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += gs) {
    const size_t i = e / ch;
    double env = g1;
    if (mode == ENV_ATTACK) env = g0 - ((double)(env_cnt0 + i) / ((double)LIMITER_ATTACK_WINDOW - 1.0) * (g0 - g1));
    else if (mode == ENV_RELEASE) env = g0 - ((double)(env_cnt0 + i) / ((double)LIMITER_RELEASE_WINDOW - 1.0) * (g1 - g0));
    size_t l = start + e;
    l %= llen;
    limiter[l] *= env;
  }
}

__global__ __launch_bounds__(256) void ln_output_kernel(double *__restrict__ dst, const double *__restrict__ limiter, size_t llen, size_t lidx, size_t n,
                                                        double target_tp) {
  const size_t gs = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
    size_t l = lidx + i; if (l >= llen) l -= llen;
    double o = limiter[l];
    if (fabs(o) > target_tp) o = target_tp * (signbit(o) ? -1.0 : 1.0);
    dst[i] = o;
  }
}

__global__ __launch_bounds__(256) void ln_linear_kernel(double *__restrict__ dst, const double *__restrict__ src, size_t n, double offset) {
  const size_t gs = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) dst[i] = src[i] * offset;
}

static unsigned ln_blocks(size_t n, int n_cu) {
  size_t b = (n + 255) / 256;
  const size_t cap = (size_t)n_cu * 4;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------ meters (two instances of the ebur128 state)

struct MeterSwap {  // the meter functions address ctx->ebur128: point it at one of ours for the duration of a call
  mi355_ctx *ctx; void *saved;
  MeterSwap(mi355_ctx *c, void *m) : ctx(c), saved(c->ebur128) { c->ebur128 = m; }
  ~MeterSwap() { ctx->ebur128 = saved; }
};

static int meter_add(mi355_ctx *ctx, void *m, const double *host, size_t frames) {
  MeterSwap sw(ctx, m);
  return ebur128_add_frames(ctx, host, nullptr, frames, 3);
}
static int meter_query(mi355_ctx *ctx, void *m, int what, double *out) {
  MeterSwap sw(ctx, m);
  return ebur128_query(ctx, what, out);
}

// ------------------------------------------------------------------ limiter (host state machine over device data)

static int ln_detect(mi355_ctx *ctx, LoudNormState *s, size_t offset, size_t samples, bool *found, size_t *peak_delta, double *peak_value) {
  *found = false;
  if (samples < 2) return MI355_OK;
  size_t base = s->limiter_buf_index + (offset + LIMITER_LOOKAHEAD) * s->channels;
  if (base >= s->limiter_len) base -= s->limiter_len;
  hipLaunchKernelGGL(ln_detect_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)s->d_limiter, s->limiter_len, base, s->channels, samples,
                     s->target_tp, s->d_peak);
  unsigned long long r[2];
  int rc = check_hip(ctx, hipMemcpyAsync(r, s->d_peak, 16, hipMemcpyDeviceToHost, ctx->stream), "loudnorm: peak D2H");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "loudnorm: sync"))) return rc;
  if (r[0] != ~0ull) {
    *found = true;
    *peak_delta = (size_t)r[0];
    std::memcpy(peak_value, &r[1], 8);
  }
  return MI355_OK;
}

static void ln_envelope(mi355_ctx *ctx, LoudNormState *s, size_t smp_cnt, size_t count, int mode, double g0, double g1, size_t env_cnt0) {
  if (!count) return;
  size_t start = s->limiter_buf_index + smp_cnt * s->channels;
  if (start >= s->limiter_len) start -= s->limiter_len;
  hipLaunchKernelGGL(ln_envelope_kernel, dim3(ln_blocks(count * s->channels, ctx->n_cu)), dim3(256), 0, ctx->stream, s->d_limiter, s->limiter_len, start,
                     s->channels, count, mode, g0, g1, env_cnt0);
}

static int limiter_out(mi355_ctx *ctx, LoudNormState *s, size_t *smp_cnt, size_t nb) {
  bool peak; size_t pd = 0; double pv = 0;
  int rc = ln_detect(ctx, s, *smp_cnt, nb - *smp_cnt, &peak, &pd, &pv);
  if (rc) return rc;
  if (peak) {
    s->limiter_state = LS_ATTACK;
    s->env_cnt = 0;
    s->have_sustain = false;
    s->gain_reduction[0] = 1.0;
    s->gain_reduction[1] = s->target_tp / pv;
    *smp_cnt += LIMITER_LOOKAHEAD + pd - LIMITER_ATTACK_WINDOW;
  } else {
    *smp_cnt = nb;
  }
  return MI355_OK;
}

static int limiter_attack(mi355_ctx *ctx, LoudNormState *s, size_t *smp_cnt_io, size_t nb) {
  size_t smp_cnt = *smp_cnt_io;
  bool peak; size_t pd = 0; double pv = 0;
  int rc = ln_detect(ctx, s, smp_cnt, nb - smp_cnt, &peak, &pd, &pv);
  if (rc) return rc;
  const size_t new_peak_smp_cnt = smp_cnt + pd;
  // while env_cnt < WINDOW && smp_cnt < nb (&& smp_cnt != new_peak_smp_cnt): ramp samples
  size_t ramp = 0;
  if (s->env_cnt < LIMITER_ATTACK_WINDOW && smp_cnt < nb) {
    ramp = LIMITER_ATTACK_WINDOW - s->env_cnt;
    if (ramp > nb - smp_cnt) ramp = nb - smp_cnt;
    if (peak && new_peak_smp_cnt >= smp_cnt && new_peak_smp_cnt - smp_cnt < ramp) ramp = new_peak_smp_cnt - smp_cnt;
  }
  ln_envelope(ctx, s, smp_cnt, ramp, ENV_ATTACK, s->gain_reduction[0], s->gain_reduction[1], s->env_cnt);
  smp_cnt += ramp;
  s->env_cnt += ramp;
  if (peak) {
    if (smp_cnt < new_peak_smp_cnt) {
      ln_envelope(ctx, s, smp_cnt, new_peak_smp_cnt - smp_cnt, ENV_CONST, 0.0, s->gain_reduction[1], 0);
      smp_cnt = new_peak_smp_cnt;
    }
    const double gain_reduction = s->target_tp / pv;
    if (gain_reduction < s->gain_reduction[1]) {
      const double current = s->gain_reduction[0] - ((double)s->env_cnt / ((double)LIMITER_ATTACK_WINDOW - 1.0) * (s->gain_reduction[0] - s->gain_reduction[1]));
      const double old_slope = -(s->gain_reduction[0] - s->gain_reduction[1]);
      const double new_slope = -(current - gain_reduction);
      if (new_slope <= old_slope) {
        s->limiter_state = LS_ATTACK;
        s->gain_reduction[0] = current;
        s->gain_reduction[1] = gain_reduction;
        s->env_cnt = 0;
        s->have_sustain = false;
      } else {
        double new_end = (gain_reduction - s->gain_reduction[0]) / old_slope;
        new_end = std::fmax(new_end, 1.0);
        const double new_start = new_end - 1.0;
        s->gain_reduction[0] = s->gain_reduction[0] + new_start * old_slope;
        s->gain_reduction[1] = gain_reduction;
        double cur_pos = (current - s->gain_reduction[0]) / old_slope;
        if (cur_pos < 0.0) cur_pos = 0.0; else if (cur_pos > 1.0) cur_pos = 1.0;  // f64::clamp
        const double pos = ((double)LIMITER_ATTACK_WINDOW - 1.0) * cur_pos;
        s->env_cnt = (pos != pos) ? 0 : (size_t)pos;                               // `as usize`
        s->have_sustain = true;
        s->sustain_cnt = s->env_cnt;
      }
      *smp_cnt_io = smp_cnt;
      return MI355_OK;
    } else if (s->env_cnt < LIMITER_ATTACK_WINDOW) {
      s->have_sustain = true;
      s->sustain_cnt = s->env_cnt;
    }
  }
  if (s->env_cnt == LIMITER_ATTACK_WINDOW && smp_cnt < nb) s->limiter_state = LS_SUSTAIN;
  *smp_cnt_io = smp_cnt;
  return MI355_OK;
}

static int limiter_sustain(mi355_ctx *ctx, LoudNormState *s, size_t *smp_cnt_io, size_t nb) {
  size_t smp_cnt = *smp_cnt_io;
  bool peak; size_t pd = 0; double pv = 0;
  int rc = ln_detect(ctx, s, smp_cnt, nb - smp_cnt, &peak, &pd, &pv);
  if (rc) return rc;
  if (peak || s->have_sustain) {
    const size_t sustain_cnt = peak ? pd : s->sustain_cnt;
    size_t k = sustain_cnt;
    if (k > nb - smp_cnt) k = nb - smp_cnt;
    ln_envelope(ctx, s, smp_cnt, k, ENV_CONST, 0.0, s->gain_reduction[1], 0);
    smp_cnt += k;
    if (peak) {
      const double gain_reduction = s->target_tp / pv;
      if (gain_reduction < s->gain_reduction[1]) {
        s->limiter_state = LS_ATTACK;
        s->env_cnt = 0;
        s->have_sustain = false;
        s->gain_reduction[0] = s->gain_reduction[1];
        s->gain_reduction[1] = gain_reduction;
      } else {
        s->have_sustain = true;
        s->sustain_cnt = LIMITER_LOOKAHEAD;
      }
    } else {
      s->sustain_cnt -= k;
      if (s->sustain_cnt == 0) s->have_sustain = false;
    }
  } else {
    s->limiter_state = LS_RELEASE;
    s->gain_reduction[0] = s->gain_reduction[1];
    s->gain_reduction[1] = 1.0;
    s->env_cnt = 0;
  }
  *smp_cnt_io = smp_cnt;
  return MI355_OK;
}

static int limiter_release(mi355_ctx *ctx, LoudNormState *s, size_t *smp_cnt_io, size_t nb) {
  size_t smp_cnt = *smp_cnt_io;
  bool peak; size_t pd = 0; double pv = 0;
  int rc = ln_detect(ctx, s, smp_cnt, nb - smp_cnt, &peak, &pd, &pv);
  if (rc) return rc;
  if (peak) {
    const double gain_reduction = s->target_tp / pv;
    const double current = s->gain_reduction[0] - ((double)s->env_cnt / ((double)LIMITER_RELEASE_WINDOW - 1.0) * (s->gain_reduction[1] - s->gain_reduction[0]));
    if (gain_reduction < current) {
      ln_envelope(ctx, s, smp_cnt, pd, ENV_CONST, 0.0, s->gain_reduction[1], 0);  // sic: multiplies by gain_reduction[1] (imp.rs:1252-1263)
      smp_cnt += pd;
      s->limiter_state = LS_ATTACK;
      s->env_cnt = 0;
      s->have_sustain = false;
      s->gain_reduction[0] = current;
      s->gain_reduction[1] = gain_reduction;
    } else {
      s->gain_reduction[1] = current;
      s->limiter_state = LS_SUSTAIN;
    }
    *smp_cnt_io = smp_cnt;
    return MI355_OK;
  }
  size_t ramp = 0;
  if (s->env_cnt < LIMITER_RELEASE_WINDOW && smp_cnt < nb) {
    ramp = LIMITER_RELEASE_WINDOW - s->env_cnt;
    if (ramp > nb - smp_cnt) ramp = nb - smp_cnt;
  }
  ln_envelope(ctx, s, smp_cnt, ramp, ENV_RELEASE, s->gain_reduction[0], s->gain_reduction[1], s->env_cnt);
  smp_cnt += ramp;
  s->env_cnt += ramp;
  if (smp_cnt < nb) s->limiter_state = LS_OUT;
  *smp_cnt_io = smp_cnt;
  return MI355_OK;
}

static int limiter_first_frame(mi355_ctx *ctx, LoudNormState *s) {
  // sequential scan with the reference's quirk (`max` keeps the SIGNED sample, imp.rs:1339-1342): tiny, done on the host
  const size_t n = (LIMITER_LOOKAHEAD + 1) * s->channels;
  std::vector<double> head(n);
  int rc = check_hip(ctx, hipMemcpyAsync(head.data(), s->d_limiter, n * 8, hipMemcpyDeviceToHost, ctx->stream), "loudnorm: head D2H");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipStreamSynchronize(ctx->stream), "loudnorm: sync"))) return rc;
  double max = 0.0;
  for (size_t i = 0; i < n; i++)
    if (std::fabs(head[i]) > max) max = head[i];
  if (max > s->target_tp) {
    s->limiter_state = LS_SUSTAIN;
    s->have_sustain = true;
    s->sustain_cnt = LIMITER_LOOKAHEAD;
    s->gain_reduction[1] = s->target_tp / max;
  }
  return MI355_OK;
}

// true_peak_limiter (imp.rs:1374-1430): nb frames of output into d_dst + dst_off
static int true_peak_limiter(mi355_ctx *ctx, LoudNormState *s, double *d_dst, size_t nb) {
  int rc;
  if (s->frame_type == FT_FIRST && (rc = limiter_first_frame(ctx, s))) return rc;
  size_t smp_cnt = 0;
  while (smp_cnt < nb) {
    switch (s->limiter_state) {
      case LS_OUT: rc = limiter_out(ctx, s, &smp_cnt, nb); break;
      case LS_ATTACK: rc = limiter_attack(ctx, s, &smp_cnt, nb); break;
      case LS_SUSTAIN: rc = limiter_sustain(ctx, s, &smp_cnt, nb); break;
      default: rc = limiter_release(ctx, s, &smp_cnt, nb); break;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ln_output_kernel, dim3(ln_blocks(nb * s->channels, ctx->n_cu)), dim3(256), 0, ctx->stream, d_dst, (const double *)s->d_limiter,
                     s->limiter_len, s->limiter_buf_index, nb * s->channels, s->target_tp);
  return check_hip(ctx, hipGetLastError(), "loudnorm kernel launch");
}

// ------------------------------------------------------------------ gain path

static double gaussian_filter(const LoudNormState *s, size_t index) {
  double result = 0.0;
  index = index > 10 ? index - 10 : index + 20;
  for (size_t k = 0; k < 21; k++) {
    const size_t j = index + k < 30 ? index + k : index + k - 30;
    result += s->delta[j] * s->weights[k];
  }
  return result;
}

static void gains(const LoudNormState *s, double *gain, double *gain_next) {
  *gain = gaussian_filter(s, s->index + 10 < 30 ? s->index + 10 : s->index + 10 - 30);
  *gain_next = gaussian_filter(s, s->index + 11 < 30 ? s->index + 11 : s->index + 11 - 30);
}

static void advance(size_t *i, size_t by, size_t len) { *i += by; if (*i >= len) *i -= len; }

// process_fill_inner_frame: d_src holds `frames` new frames
static void fill_inner_frame(mi355_ctx *ctx, LoudNormState *s, size_t frames) {
  if (!frames) return;
  double gain, gain_next;
  gains(s, &gain, &gain_next);
  const size_t ch = s->channels;
  hipLaunchKernelGGL(ln_fill_kernel, dim3(ln_blocks(frames * ch, ctx->n_cu)), dim3(256), 0, ctx->stream, s->d_limiter, s->limiter_len, s->limiter_buf_index,
                     s->d_buf, s->buf_len, s->buf_index, s->prev_buf_index, (const double *)s->d_src, ch, (size_t)0, frames, (double)FRAME_SIZE, gain, gain_next,
                     s->offset);
  advance(&s->limiter_buf_index, frames * ch, s->limiter_len);
  advance(&s->prev_buf_index, frames * ch, s->buf_len);
  advance(&s->buf_index, frames * ch, s->buf_len);
}

static void fill_final_frame(mi355_ctx *ctx, LoudNormState *s, size_t idx, size_t num_samples) {
  if (idx >= num_samples) return;
  double gain, gain_next;
  gains(s, &gain, &gain_next);
  const size_t ch = s->channels, frames = num_samples - idx;
  hipLaunchKernelGGL(ln_fill_kernel, dim3(ln_blocks(frames * ch, ctx->n_cu)), dim3(256), 0, ctx->stream, s->d_limiter, s->limiter_len, s->limiter_buf_index,
                     s->d_buf, s->buf_len, s->buf_index, (size_t)0, (const double *)nullptr, ch, idx, num_samples, (double)num_samples, gain, gain_next, s->offset);
  advance(&s->limiter_buf_index, frames * ch, s->limiter_len);
  advance(&s->buf_index, frames * ch, s->buf_len);
}

static int update_gain_inner_frame(mi355_ctx *ctx, LoudNormState *s) {
  double global = 0, shortterm = 0, relative_threshold = 0;
  int rc;
  if ((rc = meter_query(ctx, s->r128_in, 2, &global))) return rc;
  if ((rc = meter_query(ctx, s->r128_in, 1, &shortterm))) return rc;
  if ((rc = meter_query(ctx, s->r128_in, 3, &relative_threshold))) return rc;
  if (!s->above_threshold) {
    if (shortterm > -70.0) s->prev_delta *= 1.0058;
    double shortterm_out = 0;
    if ((rc = meter_query(ctx, s->r128_out, 1, &shortterm_out))) return rc;
    if (shortterm_out >= s->target_i) s->above_threshold = true;
  }
  if (shortterm < relative_threshold || shortterm <= -70.0 || !s->above_threshold) {
    s->delta[s->index] = s->prev_delta;
  } else {
    double env_global;
    if (std::fabs(shortterm - global) < (s->target_lra / 2.0)) env_global = shortterm - global;
    else if ((s->target_lra / 2.0) * (shortterm - global) < 0.0) env_global = -1.0;
    else env_global = 1.0;
    const double env_shortterm = s->target_i - shortterm;
    s->delta[s->index] = std::pow(10.0, (env_global + env_shortterm) / 20.0);
  }
  s->prev_delta = s->delta[s->index];
  s->index += 1;
  if (s->index >= 30) s->index -= 30;
  return MI355_OK;
}

static int download(mi355_ctx *ctx, double *host, const double *dev, size_t n) {
  int rc = check_hip(ctx, hipMemcpyAsync(host, dev, n * 8, hipMemcpyDeviceToHost, ctx->stream), "loudnorm: D2H");
  if (rc) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "loudnorm: sync");
}

// State::process (imp.rs:800-828). src: host, `frames` frames. dst: host, receives *out_frames frames.
static int ln_process(mi355_ctx *ctx, LoudNormState *s, const double *src, size_t frames, double *dst, size_t dst_cap_frames, size_t *out_frames) {
  const size_t ch = s->channels;
  int rc;
  *out_frames = 0;
  if ((rc = meter_add(ctx, s->r128_in, src, frames))) return rc;
  if (s->frame_type == FT_FIRST && frames < s->current_samples_per_frame) {  // process_first_frame_is_last
    double global = 0;
    if ((rc = meter_query(ctx, s->r128_in, 2, &global))) return rc;
    double true_peak = 0.0;
    for (size_t c = 0; c < ch; c++) {
      double peak = 0;
      { MeterSwap sw(ctx, s->r128_in); if ((rc = ebur128_peak(ctx, 0, (unsigned)c, &peak))) return rc; }
      if (c == 0 || peak > true_peak) true_peak = peak;
    }
    const double offset = std::pow(10.0, (s->target_i - global) / 20.0);
    const double offset_tp = true_peak * offset;
    s->offset = offset_tp < s->target_tp ? offset : s->target_tp / true_peak;
    s->frame_type = FT_LINEAR;
  }
  const size_t need = s->frame_type == FT_FINAL ? 30 * FRAME_SIZE - (FRAME_SIZE - frames)
                      : (s->frame_type == FT_LINEAR ? frames : (s->frame_type == FT_FIRST ? FRAME_SIZE : s->current_samples_per_frame));
  if (need > dst_cap_frames) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: output buffer too small");
  if (frames && (rc = check_hip(ctx, hipMemcpyAsync(s->d_src, src, frames * ch * 8, hipMemcpyHostToDevice, ctx->stream), "loudnorm: H2D"))) return rc;
  switch (s->frame_type) {
    case FT_FIRST: {
      if ((rc = check_hip(ctx, hipMemcpyAsync(s->d_buf, s->d_src, s->buf_len * 8, hipMemcpyDeviceToDevice, ctx->stream), "loudnorm: buf fill"))) return rc;
      double shortterm = 0;
      if ((rc = meter_query(ctx, s->r128_in, 1, &shortterm))) return rc;
      double env_shortterm;
      if (shortterm < -70.0) { s->above_threshold = false; env_shortterm = 0.0; }
      else { s->above_threshold = true; env_shortterm = s->target_i - shortterm; }
      for (int i = 0; i < 30; i++) s->delta[i] = std::pow(10.0, env_shortterm / 20.0);
      s->prev_delta = s->delta[s->index];
      hipLaunchKernelGGL(ln_scale_kernel, dim3(ln_blocks(s->limiter_len, ctx->n_cu)), dim3(256), 0, ctx->stream, s->d_limiter, (const double *)s->d_buf,
                         s->limiter_len, s->prev_delta, s->offset);
      s->buf_index = s->limiter_len;
      s->limiter_buf_index = 0;
      if ((rc = true_peak_limiter(ctx, s, s->d_dst, FRAME_SIZE))) return rc;
      if ((rc = download(ctx, dst, s->d_dst, FRAME_SIZE * ch))) return rc;
      if ((rc = meter_add(ctx, s->r128_out, dst, FRAME_SIZE))) return rc;
      s->current_samples_per_frame = FRAME_SIZE;
      s->frame_type = FT_INNER;
      *out_frames = FRAME_SIZE;
      return MI355_OK;
    }
    case FT_INNER: {
      fill_inner_frame(ctx, s, frames);
      if ((rc = true_peak_limiter(ctx, s, s->d_dst, s->current_samples_per_frame))) return rc;
      if ((rc = download(ctx, dst, s->d_dst, s->current_samples_per_frame * ch))) return rc;
      if ((rc = meter_add(ctx, s->r128_out, dst, s->current_samples_per_frame))) return rc;
      if ((rc = update_gain_inner_frame(ctx, s))) return rc;
      *out_frames = s->current_samples_per_frame;
      return MI355_OK;
    }
    case FT_FINAL: {
      const size_t num_samples = frames;
      fill_inner_frame(ctx, s, frames);
      if (num_samples != FRAME_SIZE) fill_final_frame(ctx, s, num_samples, FRAME_SIZE);
      const size_t out_num_samples = need;
      size_t smp_cnt = 0;
      while (smp_cnt < out_num_samples) {
        const size_t frame_size = out_num_samples - smp_cnt < FRAME_SIZE ? out_num_samples - smp_cnt : FRAME_SIZE;
        double *d = dst + smp_cnt * ch;
        if ((rc = true_peak_limiter(ctx, s, s->d_dst, frame_size))) return rc;
        if ((rc = download(ctx, d, s->d_dst, frame_size * ch))) return rc;
        smp_cnt += frame_size;
        if (smp_cnt == out_num_samples) break;
        if ((rc = meter_add(ctx, s->r128_out, d, frame_size))) return rc;
        if ((rc = update_gain_inner_frame(ctx, s))) return rc;
        const size_t next_frame_size = out_num_samples - smp_cnt < FRAME_SIZE ? out_num_samples - smp_cnt : FRAME_SIZE;
        fill_final_frame(ctx, s, 0, next_frame_size);
        if (next_frame_size < FRAME_SIZE) advance(&s->limiter_buf_index, FRAME_SIZE - next_frame_size, s->limiter_len);  // sic (imp.rs:763)
      }
      *out_frames = out_num_samples;
      return MI355_OK;
    }
    default: {
      if (frames) {
        hipLaunchKernelGGL(ln_linear_kernel, dim3(ln_blocks(frames * ch, ctx->n_cu)), dim3(256), 0, ctx->stream, s->d_dst, (const double *)s->d_src, frames * ch,
                           s->offset);
        if ((rc = download(ctx, dst, s->d_dst, frames * ch))) return rc;
        if ((rc = meter_add(ctx, s->r128_out, dst, frames))) return rc;
      }
      *out_frames = frames;
      return MI355_OK;
    }
  }
}

// ------------------------------------------------------------------ entry points used by the C ABI

void loudnorm_release(mi355_ctx *ctx) {
  LoudNormState *s = ln_of(ctx);
  if (!s) return;
  for (void *m : {s->r128_in, s->r128_out}) {
    if (!m) continue;
    MeterSwap sw(ctx, m);
    ebur128_release(ctx);  // frees ctx->ebur128 (== m) and nulls it; the swap restores the caller's meter
  }
  double **bufs[] = {&s->d_buf, &s->d_limiter, &s->d_src, &s->d_dst};
  for (auto b : bufs) if (*b) (void)hipFree(*b);
  if (s->d_peak) (void)hipFree(s->d_peak);
  delete s;
  ctx->loudnorm = nullptr;
}

static int make_meter(mi355_ctx *ctx, unsigned channels, void **out) {
  void *saved = ctx->ebur128;
  ctx->ebur128 = nullptr;
  const int rc = ebur128_setup(ctx, channels, 192000, 4 | 2 | 8 | 16, nullptr);  // I | S | LRA | SAMPLE_PEAK
  *out = ctx->ebur128;
  ctx->ebur128 = saved;
  return rc;
}

// State::new (imp.rs:130-205)
int loudnorm_setup(mi355_ctx *ctx, unsigned channels, double loudness_target, double loudness_range_target, double max_true_peak, double offset_db) {
  loudnorm_release(ctx);
  if (channels < 1 || channels > 64) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: channels must be 1..64");
  LoudNormState *s = new LoudNormState();
  ctx->loudnorm = s;
  s->channels = channels;
  int rc;
  if ((rc = make_meter(ctx, channels, &s->r128_in)) || (rc = make_meter(ctx, channels, &s->r128_out))) { loudnorm_release(ctx); return rc; }
  s->buf_len = GAIN_LOOKAHEAD * channels;
  s->limiter_len = (2 * FRAME_SIZE + LIMITER_LOOKAHEAD) * channels;
  struct { double **p; size_t n; } bufs[] = {{&s->d_buf, s->buf_len}, {&s->d_limiter, s->limiter_len}, {&s->d_src, s->buf_len}, {&s->d_dst, s->buf_len}};
  for (auto &b : bufs) {
    if ((rc = check_hip(ctx, hipMalloc((void **)b.p, b.n * 8), "hipMalloc(loudnorm)")) || (rc = check_hip(ctx, hipMemset(*b.p, 0, b.n * 8), "hipMemset(loudnorm)"))) {
      loudnorm_release(ctx);
      return rc;
    }
  }
  if ((rc = check_hip(ctx, hipMalloc((void **)&s->d_peak, 16), "hipMalloc(loudnorm peak)"))) { loudnorm_release(ctx); return rc; }
  s->offset = std::pow(10.0, offset_db / 20.0);
  s->target_tp = std::pow(10.0, max_true_peak / 20.0);
  s->target_i = loudness_target;
  s->target_lra = loudness_range_target;
  // init_gaussian_filter (imp.rs:1893-1914)
  double total = 0.0;
  const double sigma = 3.5, c1 = 1.0 / (sigma * std::sqrt(2.0 * M_PI)), c2 = 2.0 * std::pow(sigma, 2.0);
  for (int i = 0; i < 21; i++) {
    const double x = (double)i - (double)(21 / 2);
    s->weights[i] = c1 * std::exp(-(std::pow(x, 2.0) / c2));
    total += s->weights[i];
  }
  const double adjust = 1.0 / total;
  for (int i = 0; i < 21; i++) s->weights[i] *= adjust;
  for (int i = 0; i < 30; i++) s->delta[i] = 0.0;
  return MI355_OK;
}

// sink_chain -> drain_full_frames (imp.rs:226-268)
int loudnorm_push(mi355_ctx *ctx, const double *data, size_t frames, double *out, size_t out_cap_frames, size_t *out_frames) {
  LoudNormState *s = ln_of(ctx);
  if (!s) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "audioloudnorm: not negotiated (setup not called)");
  *out_frames = 0;
  const size_t ch = s->channels;
  s->adapter.insert(s->adapter.end(), data, data + frames * ch);
  size_t used = 0;
  while (s->adapter.size() / ch - used >= s->current_samples_per_frame) {
    const size_t take = s->current_samples_per_frame;
    size_t n = 0;
    int rc = ln_process(ctx, s, s->adapter.data() + used * ch, take, out + *out_frames * ch, out_cap_frames - *out_frames, &n);
    if (rc) return rc;
    *out_frames += n;
    used += take;
  }
  s->adapter.erase(s->adapter.begin(), s->adapter.begin() + (std::ptrdiff_t)(used * ch));
  return MI355_OK;
}

// drain (imp.rs:270-310). *eos = 1 for "nothing to drain at all" (FlowError::Eos)
int loudnorm_drain(mi355_ctx *ctx, double *out, size_t out_cap_frames, size_t *out_frames, int *eos) {
  LoudNormState *s = ln_of(ctx);
  if (!s) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "audioloudnorm: not negotiated (setup not called)");
  *out_frames = 0;
  *eos = 0;
  const size_t avail = s->adapter.size() / s->channels;
  if (s->current_samples_per_frame == FRAME_SIZE) s->frame_type = FT_FINAL;
  else if (avail == 0) { *eos = 1; return MI355_OK; }
  int rc = ln_process(ctx, s, s->adapter.data(), avail, out, out_cap_frames, out_frames);
  s->adapter.clear();
  return rc;
}


// ================================================================== batches of streams (round 3)
// n_streams instances of the element, advanced TOGETHER by one call: the streams a call advances (all of them, or the members
// loudnorm_process_members names) get the same number of frames of the same frame type, so the launch shapes are shared; ring
// positions, gain indices and values are per stream (round 6: a stream that started later, or sat calls out, keeps its own). What kept round 2 from batching - the
// limiter's state machines of different streams take different transitions - is solved by running the state machine ON THE
// DEVICE: one 1024-lane block per stream executes true_peak_limiter (imp.rs:1374-1430) for the whole frame: every lane
// carries an identical copy of the state and takes the transitions together (they are scalar f64 expressions, the host
// code above transcribed once more), detect_peak is the block-wide first-match search of ln_detect_kernel, the envelopes
// are applied by all lanes, and the output copy with its hard clamp closes the kernel. A 100 ms frame of 256 stereo streams
// is then: one meter pass over the input, one fill launch, ONE limiter launch, one meter pass over the output, and the gain
// bookkeeping of update_gain_inner_frame on the host (four meter queries for all streams: two syncs) - instead of 256
// host-driven state machines with a device round trip per limiter transition.
// Samples are bit-identical to n separate single-stream contexts (tests/test_gpu_loudnorm.py).

struct LnbLimiter {  // per stream, device
  int state, have_sustain;
  unsigned long long env_cnt, sustain_cnt;
  double gr0, gr1;
};
// per stream and launch: the fill's gains and where the stream's rings stand (the values of a launch are snapshotted right before it)
struct LnbGain {
  double gain, gain_next, offset;
  unsigned long long lidx, bidx, pidx;   // limiter ring index, buf read index, buf write index
  int active, pad;                        // not a member of this call: the stream's blocks return at once
};

struct LoudNormBatch {
  size_t S = 0, channels = 0;
  std::vector<size_t> current_samples_per_frame;   // [S]: GAIN_LOOKAHEAD until the first frame has been taken, FRAME_SIZE afterwards
  double target_i = 0, target_lra = 0, target_tp = 0;
  double weights[21];
  std::vector<size_t> index;                       // [S]: advanced once per inner frame of the stream
  std::vector<double> delta, prev_delta, offset;  // [S][30], [S], [S]
  std::vector<char> above_threshold;              // [S]
  size_t buf_len = 0, limiter_len = 0;
  std::vector<size_t> buf_index, prev_buf_index, limiter_buf_index;   // [S]
  std::vector<int> frame_type;                     // [S]
  std::vector<char> active;                        // [S]: the streams the call in progress advances
  size_t adv = 0;                                  // how far the members' limiter rings have moved since the last upload of the table
  double *d_buf = nullptr, *d_limiter = nullptr, *d_src = nullptr, *d_dst = nullptr;  // [S][...]
  LnbLimiter *d_lim = nullptr;
  LnbGain *d_gain = nullptr, *h_gain = nullptr;
  hipEvent_t gain_ev = nullptr;
  void *r128_in = nullptr, *r128_out = nullptr;
};

__global__ __launch_bounds__(256) void lnb_scale_kernel(double *__restrict__ limiter, size_t llen, const double *__restrict__ buf, size_t blen,
                                                        const LnbGain *__restrict__ g) {
  const size_t s = blockIdx.y, gs = (size_t)gridDim.x * 256;
  if (!g[s].active) return;
  const double pd = g[s].gain, off = g[s].offset;  // gain = prev_delta here
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < llen; i += gs) limiter[s * llen + i] = buf[s * blen + i] * pd * off;
}

__global__ __launch_bounds__(256) void lnb_fill_kernel(double *__restrict__ limiter, size_t llen, double *__restrict__ buf, size_t blen,
                                                       const double *__restrict__ src, size_t src_stride, size_t ch, size_t n0, size_t n1,
                                                       double denom, const LnbGain *__restrict__ g) {
  const size_t s = blockIdx.y, total = (n1 - n0) * ch, gs = (size_t)gridDim.x * 256;
  if (!g[s].active) return;
  const size_t lidx = (size_t)g[s].lidx, bidx = (size_t)g[s].bidx, pidx = (size_t)g[s].pidx;
  const double gain = g[s].gain, gain_next = g[s].gain_next, offset = g[s].offset;
  double *lim = limiter + s * llen, *b = buf + s * blen;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += gs) {
    const size_t n = n0 + i / ch;
    const double current_gain = (gain + (((double)n / denom) * (gain_next - gain))) * offset;  // ln_fill_kernel's expression
    size_t r = bidx + i; if (r >= blen) r -= blen;
    size_t l = lidx + i; if (l >= llen) l -= llen;
    const double v = b[r];
    if (src) { size_t w = pidx + i; if (w >= blen) w -= blen; b[w] = src[s * src_stride + i]; }
    lim[l] = v * current_gain;
  }
}

__global__ __launch_bounds__(256) void lnb_linear_kernel(double *__restrict__ dst, size_t dst_stride, const double *__restrict__ src, size_t src_stride, size_t n,
                                                         const LnbGain *__restrict__ g) {
  const size_t s = blockIdx.y, gs = (size_t)gridDim.x * 256;
  if (!g[s].active) return;
  const double off = g[s].offset;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) dst[s * dst_stride + i] = src[s * src_stride + i] * off;
}

// detect_peak for one stream by its block (the search of ln_detect_kernel): n in [1, samples) relative to `base`
__device__ __forceinline__ bool lnb_detect(const double *lim, size_t llen, size_t base, size_t ch, size_t samples, double target_tp,
                                           unsigned long long *s_min, double *s_val, size_t *pd, double *pv) {
  if (samples < 2) return false;
  if (threadIdx.x == 0) *s_min = ~0ull;
  __syncthreads();
  auto at = [&](size_t n, size_t c) -> double { return fabs(lim[(base + n * ch + c) % llen]); };
  for (size_t n0 = 1; n0 < samples; n0 += 1024) {
    const size_t n = n0 + threadIdx.x;
    if (n < samples) {
      bool hit = false;
      for (size_t c = 0; c < ch && !hit; c++) {
        const double th = at(n, c);
        if (at(n - 1, c) <= th && th >= at(n + 1, c) && th > target_tp) {
          bool ok = true;
          for (size_t i = 2; i < 12; i++)
            if (at(n + i, c) > th) { ok = false; break; }
          hit = ok;
        }
      }
      if (hit) atomicMin(s_min, (unsigned long long)n);
    }
    __syncthreads();
    if (*s_min != ~0ull) break;
    __syncthreads();
  }
  const unsigned long long m = *s_min;
  if (m == ~0ull) { __syncthreads(); return false; }
  if (threadIdx.x == 0) {
    double mx = 0.0;
    for (size_t c = 0; c < ch; c++) { const double v = at((size_t)m, c); if (c == 0 || v > mx) mx = v; }
    *s_val = mx;
  }
  __syncthreads();
  *pd = (size_t)m;
  *pv = *s_val;
  __syncthreads();
  return true;
}

__device__ __forceinline__ void lnb_envelope(double *lim, size_t llen, size_t lidx, size_t ch, size_t smp_cnt, size_t count, int mode, double g0, double g1,
                                             size_t env_cnt0) {
  if (count) {
    size_t start = lidx + smp_cnt * ch;
    if (start >= llen) start -= llen;
    const size_t total = count * ch;
    for (size_t e = threadIdx.x; e < total; e += 1024) {
      const size_t i = e / ch;
      double env = g1;  // ln_envelope_kernel's expressions
      if (mode == ENV_ATTACK) env = g0 - ((double)(env_cnt0 + i) / ((double)LIMITER_ATTACK_WINDOW - 1.0) * (g0 - g1));
      else if (mode == ENV_RELEASE) env = g0 - ((double)(env_cnt0 + i) / ((double)LIMITER_RELEASE_WINDOW - 1.0) * (g1 - g0));
      lim[(start + e) % llen] *= env;
    }
  }
  __syncthreads();  // later searches read what the envelope wrote
}

// true_peak_limiter (imp.rs:1374-1430) of stream blockIdx.x over `nb` frames, then the output copy with the hard clamp.
// Every lane keeps the same copy of the limiter state; the transitions below are limiter_out / _attack / _sustain /
// _release of the host path, line for line.
__global__ __launch_bounds__(1024) void lnb_limiter_kernel(double *__restrict__ limiter, size_t llen, const LnbGain *__restrict__ g, size_t lidx_add, size_t ch, size_t nb,
                                                           double target_tp, int first_frame, LnbLimiter *__restrict__ state, double *__restrict__ dst, size_t dst_stride) {
  __shared__ unsigned long long s_min;
  __shared__ double s_val;
  const size_t s = blockIdx.x;
  if (!g[s].active) return;   // (block-uniform)
  // the table is the snapshot of the call's last upload; the members' rings have all moved on by lidx_add since
  const size_t lidx = ((size_t)g[s].lidx + lidx_add) % llen;
  double *lim = limiter + s * llen;
  LnbLimiter L = state[s];
  if (first_frame) {  // limiter_first_frame: the serial scan with the reference's quirk (`max` keeps the SIGNED sample, imp.rs:1339-1342)
    if (threadIdx.x == 0) {
      const size_t n = (LIMITER_LOOKAHEAD + 1) * ch;
      double mx = 0.0;
      for (size_t i = 0; i < n; i++)
        if (fabs(lim[i]) > mx) mx = lim[i];
      s_val = mx;
    }
    __syncthreads();
    const double mx = s_val;
    __syncthreads();
    if (mx > target_tp) { L.state = LS_SUSTAIN; L.have_sustain = 1; L.sustain_cnt = LIMITER_LOOKAHEAD; L.gr1 = target_tp / mx; }
  }
  size_t smp_cnt = 0;
  while (smp_cnt < nb) {
    size_t pd = 0;
    double pv = 0.0;
    size_t base = lidx + (smp_cnt + LIMITER_LOOKAHEAD) * ch;
    if (base >= llen) base -= llen;
    const bool peak = lnb_detect(lim, llen, base, ch, nb - smp_cnt, target_tp, &s_min, &s_val, &pd, &pv);
    if (L.state == LS_OUT) {
      if (peak) {
        L.state = LS_ATTACK; L.env_cnt = 0; L.have_sustain = 0; L.gr0 = 1.0; L.gr1 = target_tp / pv;
        smp_cnt += LIMITER_LOOKAHEAD + pd - LIMITER_ATTACK_WINDOW;
      } else {
        smp_cnt = nb;
      }
    } else if (L.state == LS_ATTACK) {
      const size_t new_peak_smp_cnt = smp_cnt + pd;
      size_t ramp = 0;
      if (L.env_cnt < LIMITER_ATTACK_WINDOW && smp_cnt < nb) {
        ramp = LIMITER_ATTACK_WINDOW - (size_t)L.env_cnt;
        if (ramp > nb - smp_cnt) ramp = nb - smp_cnt;
        if (peak && new_peak_smp_cnt >= smp_cnt && new_peak_smp_cnt - smp_cnt < ramp) ramp = new_peak_smp_cnt - smp_cnt;
      }
      lnb_envelope(lim, llen, lidx, ch, smp_cnt, ramp, ENV_ATTACK, L.gr0, L.gr1, (size_t)L.env_cnt);
      smp_cnt += ramp;
      L.env_cnt += ramp;
      bool done = false;
      if (peak) {
        if (smp_cnt < new_peak_smp_cnt) {
          lnb_envelope(lim, llen, lidx, ch, smp_cnt, new_peak_smp_cnt - smp_cnt, ENV_CONST, 0.0, L.gr1, 0);
          smp_cnt = new_peak_smp_cnt;
        }
        const double gain_reduction = target_tp / pv;
        if (gain_reduction < L.gr1) {
          const double current = L.gr0 - ((double)L.env_cnt / ((double)LIMITER_ATTACK_WINDOW - 1.0) * (L.gr0 - L.gr1));
          const double old_slope = -(L.gr0 - L.gr1);
          const double new_slope = -(current - gain_reduction);
          if (new_slope <= old_slope) {
            L.state = LS_ATTACK; L.gr0 = current; L.gr1 = gain_reduction; L.env_cnt = 0; L.have_sustain = 0;
          } else {
            double new_end = (gain_reduction - L.gr0) / old_slope;
            new_end = fmax(new_end, 1.0);
            const double new_start = new_end - 1.0;
            L.gr0 = L.gr0 + new_start * old_slope;
            L.gr1 = gain_reduction;
            double cur_pos = (current - L.gr0) / old_slope;
            if (cur_pos < 0.0) cur_pos = 0.0; else if (cur_pos > 1.0) cur_pos = 1.0;  // f64::clamp
            const double pos = ((double)LIMITER_ATTACK_WINDOW - 1.0) * cur_pos;
            L.env_cnt = (pos != pos) ? 0ull : (unsigned long long)pos;                 // `as usize`
            L.have_sustain = 1;
            L.sustain_cnt = L.env_cnt;
          }
          done = true;
        } else if (L.env_cnt < LIMITER_ATTACK_WINDOW) {
          L.have_sustain = 1;
          L.sustain_cnt = L.env_cnt;
        }
      }
      if (!done && L.env_cnt == LIMITER_ATTACK_WINDOW && smp_cnt < nb) L.state = LS_SUSTAIN;
    } else if (L.state == LS_SUSTAIN) {
      if (peak || L.have_sustain) {
        const size_t sustain_cnt = peak ? pd : (size_t)L.sustain_cnt;
        size_t k = sustain_cnt;
        if (k > nb - smp_cnt) k = nb - smp_cnt;
        lnb_envelope(lim, llen, lidx, ch, smp_cnt, k, ENV_CONST, 0.0, L.gr1, 0);
        smp_cnt += k;
        if (peak) {
          const double gain_reduction = target_tp / pv;
          if (gain_reduction < L.gr1) { L.state = LS_ATTACK; L.env_cnt = 0; L.have_sustain = 0; L.gr0 = L.gr1; L.gr1 = gain_reduction; }
          else { L.have_sustain = 1; L.sustain_cnt = LIMITER_LOOKAHEAD; }
        } else {
          L.sustain_cnt -= k;
          if (L.sustain_cnt == 0) L.have_sustain = 0;
        }
      } else {
        L.state = LS_RELEASE; L.gr0 = L.gr1; L.gr1 = 1.0; L.env_cnt = 0;
      }
    } else {  // LS_RELEASE
      if (peak) {
        const double gain_reduction = target_tp / pv;
        const double current = L.gr0 - ((double)L.env_cnt / ((double)LIMITER_RELEASE_WINDOW - 1.0) * (L.gr1 - L.gr0));
        if (gain_reduction < current) {
          lnb_envelope(lim, llen, lidx, ch, smp_cnt, pd, ENV_CONST, 0.0, L.gr1, 0);  // sic: multiplies by gain_reduction[1] (imp.rs:1252-1263)
          smp_cnt += pd;
          L.state = LS_ATTACK; L.env_cnt = 0; L.have_sustain = 0; L.gr0 = current; L.gr1 = gain_reduction;
        } else {
          L.gr1 = current;
          L.state = LS_SUSTAIN;
        }
      } else {
        size_t ramp = 0;
        if (L.env_cnt < LIMITER_RELEASE_WINDOW && smp_cnt < nb) {
          ramp = LIMITER_RELEASE_WINDOW - (size_t)L.env_cnt;
          if (ramp > nb - smp_cnt) ramp = nb - smp_cnt;
        }
        lnb_envelope(lim, llen, lidx, ch, smp_cnt, ramp, ENV_RELEASE, L.gr0, L.gr1, (size_t)L.env_cnt);
        smp_cnt += ramp;
        L.env_cnt += ramp;
        if (smp_cnt < nb) L.state = LS_OUT;
      }
    }
  }
  __syncthreads();
  const size_t total = nb * ch;
  for (size_t i = threadIdx.x; i < total; i += 1024) {  // ln_output_kernel
    size_t l = lidx + i; if (l >= llen) l -= llen;
    double o = lim[l];
    if (fabs(o) > target_tp) o = target_tp * (signbit(o) ? -1.0 : 1.0);
    dst[s * dst_stride + i] = o;
  }
  if (threadIdx.x == 0) state[s] = L;
}

static LoudNormBatch *lnb_of(mi355_ctx *ctx) { return (LoudNormBatch *)ctx->loudnorm_batch; }

void loudnorm_batch_release(mi355_ctx *ctx) {
  LoudNormBatch *b = lnb_of(ctx);
  if (!b) return;
  for (void *m : {b->r128_in, b->r128_out}) {
    if (!m) continue;
    MeterSwap sw(ctx, m);
    ebur128_release(ctx);
  }
  for (double **p : {&b->d_buf, &b->d_limiter, &b->d_src, &b->d_dst}) if (*p) (void)hipFree(*p);
  if (b->d_lim) (void)hipFree(b->d_lim);
  if (b->d_gain) (void)hipFree(b->d_gain);
  if (b->h_gain) (void)hipHostFree(b->h_gain);
  if (b->gain_ev) (void)hipEventDestroy(b->gain_ev);
  delete b;
  ctx->loudnorm_batch = nullptr;
}

static int lnb_make_meter(mi355_ctx *ctx, unsigned S, unsigned channels, void **out) {
  void *saved = ctx->ebur128;
  ctx->ebur128 = nullptr;
  const int rc = ebur128_setup_batch(ctx, S, channels, 192000, 4 | 2 | 8 | 16, nullptr);  // I | S | LRA | SAMPLE_PEAK (imp.rs:131-150)
  *out = ctx->ebur128;
  ctx->ebur128 = saved;
  return rc;
}

int loudnorm_setup_batch(mi355_ctx *ctx, unsigned n_streams, unsigned channels, double loudness_target, double loudness_range_target, double max_true_peak,
                         double offset_db) {
  loudnorm_batch_release(ctx);
  if (n_streams < 1 || n_streams > 4096) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: 1..4096 streams per batch");
  if (channels < 1 || channels > 64) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: channels must be 1..64");
  LoudNormBatch *b = new LoudNormBatch();
  ctx->loudnorm_batch = b;
  const size_t S = n_streams;
  b->S = S; b->channels = channels;
  int rc;
  if ((rc = lnb_make_meter(ctx, n_streams, channels, &b->r128_in)) || (rc = lnb_make_meter(ctx, n_streams, channels, &b->r128_out))) { loudnorm_batch_release(ctx); return rc; }
  b->buf_len = GAIN_LOOKAHEAD * channels;
  b->limiter_len = (2 * FRAME_SIZE + LIMITER_LOOKAHEAD) * channels;
  struct { double **p; size_t n; } bufs[] = {{&b->d_buf, S * b->buf_len}, {&b->d_limiter, S * b->limiter_len}, {&b->d_src, S * b->buf_len}, {&b->d_dst, S * b->buf_len}};
  for (auto &x : bufs)
    if ((rc = check_hip(ctx, hipMalloc((void **)x.p, x.n * 8), "hipMalloc(loudnorm batch)")) || (rc = check_hip(ctx, hipMemset(*x.p, 0, x.n * 8), "hipMemset(loudnorm batch)"))) {
      loudnorm_batch_release(ctx);
      return rc;
    }
  if ((rc = check_hip(ctx, hipMalloc((void **)&b->d_lim, S * sizeof(LnbLimiter)), "hipMalloc(loudnorm limiter states)")) ||
      (rc = check_hip(ctx, hipMalloc((void **)&b->d_gain, S * sizeof(LnbGain)), "hipMalloc(loudnorm gains)")) ||
      (rc = check_hip(ctx, hipHostMalloc((void **)&b->h_gain, S * sizeof(LnbGain), hipHostMallocDefault), "hipHostMalloc(loudnorm gains)")) ||
      (rc = check_hip(ctx, hipEventCreateWithFlags(&b->gain_ev, hipEventDisableTiming), "hipEventCreate(loudnorm)"))) { loudnorm_batch_release(ctx); return rc; }
  std::vector<LnbLimiter> lim(S, LnbLimiter{LS_OUT, 0, 0ull, 0ull, 0.0, 0.0});
  if ((rc = check_hip(ctx, hipMemcpy(b->d_lim, lim.data(), S * sizeof(LnbLimiter), hipMemcpyHostToDevice), "loudnorm: limiter states"))) { loudnorm_batch_release(ctx); return rc; }
  b->target_tp = std::pow(10.0, max_true_peak / 20.0);
  b->target_i = loudness_target;
  b->target_lra = loudness_range_target;
  b->offset.assign(S, std::pow(10.0, offset_db / 20.0));
  b->delta.assign(S * 30, 0.0);
  b->prev_delta.assign(S, 0.0);
  b->above_threshold.assign(S, 0);
  b->current_samples_per_frame.assign(S, (size_t)GAIN_LOOKAHEAD);
  b->index.assign(S, 1);
  b->buf_index.assign(S, 0);
  b->prev_buf_index.assign(S, 0);
  b->limiter_buf_index.assign(S, 0);
  b->frame_type.assign(S, FT_FIRST);
  b->active.assign(S, 1);
  {  // init_gaussian_filter (imp.rs:1893-1914)
    double total = 0.0;
    const double sigma = 3.5, c1 = 1.0 / (sigma * std::sqrt(2.0 * M_PI)), c2 = 2.0 * std::pow(sigma, 2.0);
    for (int i = 0; i < 21; i++) { const double x = (double)i - (double)(21 / 2); b->weights[i] = c1 * std::exp(-(std::pow(x, 2.0) / c2)); total += b->weights[i]; }
    const double adjust = 1.0 / total;
    for (int i = 0; i < 21; i++) b->weights[i] *= adjust;
  }
  return MI355_OK;
}

static double lnb_gaussian(const LoudNormBatch *b, size_t s, size_t index) {
  double result = 0.0;
  index = index > 10 ? index - 10 : index + 20;
  for (size_t k = 0; k < 21; k++) {
    const size_t j = index + k < 30 ? index + k : index + k - 30;
    result += b->delta[s * 30 + j] * b->weights[k];
  }
  return result;
}

// per-stream {gains, ring positions, member or not} of the next launch -> device, in stream order: a snapshot taken right before
// every launch that reads it (the positions move between the launches of a call)
static int lnb_upload_gains(mi355_ctx *ctx, LoudNormBatch *b, bool scale_first) {
  int rc = check_hip(ctx, hipEventSynchronize(b->gain_ev), "hipEventSynchronize(loudnorm gains)");  // the previous upload has been consumed
  if (rc) return rc;
  for (size_t s = 0; s < b->S; s++) {
    LnbGain &g = b->h_gain[s];
    const size_t ix = b->index[s];
    if (scale_first) { g.gain = b->prev_delta[s]; g.gain_next = 0.0; }
    else { g.gain = lnb_gaussian(b, s, ix + 10 < 30 ? ix + 10 : ix + 10 - 30); g.gain_next = lnb_gaussian(b, s, ix + 11 < 30 ? ix + 11 : ix + 11 - 30); }
    g.offset = b->offset[s];
    g.lidx = b->limiter_buf_index[s]; g.bidx = b->buf_index[s]; g.pidx = b->prev_buf_index[s];
    g.active = b->active[s] ? 1 : 0; g.pad = 0;
  }
  b->adv = 0;
  if ((rc = check_hip(ctx, hipMemcpyAsync(b->d_gain, b->h_gain, b->S * sizeof(LnbGain), hipMemcpyHostToDevice, ctx->stream), "loudnorm: gains H2D"))) return rc;
  return check_hip(ctx, hipEventRecord(b->gain_ev, ctx->stream), "hipEventRecord(loudnorm gains)");
}

// `frames` frames of every member of the call (none of the others) from the packed device buffer d_data [S][frames * channels]
static int lnb_meter_add(mi355_ctx *ctx, LoudNormBatch *b, void *m, const double *d_data, size_t frames) {
  MeterSwap sw(ctx, m);
  std::vector<size_t> per(b->S);
  for (size_t s = 0; s < b->S; s++) per[s] = b->active[s] ? frames : 0;
  return ebur128_add_frames_streams(ctx, d_data, frames * b->channels, per.data(), 3, 1);
}
static int lnb_meter_query(mi355_ctx *ctx, void *m, int what, std::vector<double> &out) {
  MeterSwap sw(ctx, m);
  return ebur128_query_batch(ctx, what, out.data());
}

// f(first, count) for every run of consecutive members of the call
template <typename F>
static int lnb_for_runs(const LoudNormBatch *b, F f) {
  for (size_t s = 0; s < b->S;) {
    if (!b->active[s]) { s++; continue; }
    size_t e = s;
    while (e + 1 < b->S && b->active[e + 1]) e++;
    const int rc = f(s, e - s + 1);
    if (rc) return rc;
    s = e + 1;
  }
  return MI355_OK;
}

static int lnb_fill(mi355_ctx *ctx, LoudNormBatch *b, const double *d_src, size_t n0, size_t n1, double denom) {
  if (n0 >= n1) return MI355_OK;
  int rc = lnb_upload_gains(ctx, b, false);
  if (rc) return rc;
  const size_t ch = b->channels, frames = n1 - n0;
  hipLaunchKernelGGL(lnb_fill_kernel, dim3(ln_blocks(frames * ch, ctx->n_cu / 4 + 1), (unsigned)b->S), dim3(256), 0, ctx->stream, b->d_limiter, b->limiter_len,
                     b->d_buf, b->buf_len, d_src, frames * ch, ch, n0, n1, denom, (const LnbGain *)b->d_gain);
  for (size_t s = 0; s < b->S; s++) {
    if (!b->active[s]) continue;
    advance(&b->limiter_buf_index[s], frames * ch, b->limiter_len);
    if (d_src) advance(&b->prev_buf_index[s], frames * ch, b->buf_len);
    advance(&b->buf_index[s], frames * ch, b->buf_len);
  }
  b->adv += frames * ch;
  return MI355_OK;
}

// (every call uploads the table - its members, where their rings stand - before its first limiter launch: the fill's or the first
// frame's upload; the limiter only needs to know how far the rings have moved since)
static int lnb_limit(mi355_ctx *ctx, LoudNormBatch *b, double *d_dst, size_t dst_stride, size_t nb, bool first_frame) {
  hipLaunchKernelGGL(lnb_limiter_kernel, dim3((unsigned)b->S), dim3(1024), 0, ctx->stream, b->d_limiter, b->limiter_len, (const LnbGain *)b->d_gain, b->adv % b->limiter_len,
                     b->channels, nb, b->target_tp, first_frame ? 1 : 0, b->d_lim, d_dst, dst_stride);
  return check_hip(ctx, hipGetLastError(), "loudnorm batch kernel launch");
}

// process_update_gain_inner_frame (imp.rs:526-608) for every member of the call
static int lnb_update_gain(mi355_ctx *ctx, LoudNormBatch *b) {
  const size_t S = b->S;
  std::vector<double> global(S), shortterm(S), rel(S), shortterm_out(S);
  int rc;
  if ((rc = lnb_meter_query(ctx, b->r128_in, 2, global)) || (rc = lnb_meter_query(ctx, b->r128_in, 1, shortterm)) || (rc = lnb_meter_query(ctx, b->r128_in, 3, rel))) return rc;
  bool need_out = false;
  for (size_t s = 0; s < S; s++) need_out |= b->active[s] && !b->above_threshold[s];
  if (need_out && (rc = lnb_meter_query(ctx, b->r128_out, 1, shortterm_out))) return rc;
  for (size_t s = 0; s < S; s++) {
    if (!b->active[s]) continue;
    if (!b->above_threshold[s]) {
      if (shortterm[s] > -70.0) b->prev_delta[s] *= 1.0058;
      if (shortterm_out[s] >= b->target_i) b->above_threshold[s] = 1;
    }
    double &d = b->delta[s * 30 + b->index[s]];
    if (shortterm[s] < rel[s] || shortterm[s] <= -70.0 || !b->above_threshold[s]) {
      d = b->prev_delta[s];
    } else {
      double env_global;
      if (std::fabs(shortterm[s] - global[s]) < (b->target_lra / 2.0)) env_global = shortterm[s] - global[s];
      else if ((b->target_lra / 2.0) * (shortterm[s] - global[s]) < 0.0) env_global = -1.0;
      else env_global = 1.0;
      const double env_shortterm = b->target_i - shortterm[s];
      d = std::pow(10.0, (env_global + env_shortterm) / 20.0);
    }
    b->prev_delta[s] = d;
    b->index[s] += 1;
    if (b->index[s] >= 30) b->index[s] -= 30;
  }
  return MI355_OK;
}

// the frame the next call must hand over for stream `stream` (its first 3 s, then 100 ms); streams of a uniform batch agree
size_t loudnorm_member_frame_size(mi355_ctx *ctx, unsigned stream) {
  LoudNormBatch *b = lnb_of(ctx);
  return b && stream < b->S ? b->current_samples_per_frame[stream] : 0;
}
size_t loudnorm_batch_frame_size(mi355_ctx *ctx) { return loudnorm_member_frame_size(ctx, 0); }
// where stream `stream` stands: its frame type (FT_*) - members of one call must agree in it and in their frame size
int loudnorm_member_frame_type(mi355_ctx *ctx, unsigned stream) {
  LoudNormBatch *b = lnb_of(ctx);
  return b && stream < b->S ? b->frame_type[stream] : -1;
}

static int lnb_process(mi355_ctx *ctx, LoudNormBatch *b, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stride, size_t out_cap_frames,
                       size_t *out_frames, int device_data, int final_frame);

// State::process (imp.rs:800-828) for the batch. `data`: stream s at data + s * stream_stride, `frames` frames each; full frames
// (frames == current_samples_per_frame) or, with `final_frame`, the shorter tail at drain. Output likewise.
int loudnorm_process_batch(mi355_ctx *ctx, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stride, size_t out_cap_frames,
                           size_t *out_frames, int device_data, int final_frame) {
  LoudNormBatch *b = lnb_of(ctx);
  if (!b) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "audioloudnorm: not negotiated (setup_batch not called)");
  std::fill(b->active.begin(), b->active.end(), (char)1);
  return lnb_process(ctx, b, data, stream_stride, frames, out, out_stride, out_cap_frames, out_frames, device_data, final_frame);
}

// The same for the streams with members[s] != 0 only - they must stand at the same frame type and frame size (streams that started
// together do; the audio groups call this once per such class) -; the other streams do not move and their rows of `data` / `out`
// are neither read nor written.
int loudnorm_process_members(mi355_ctx *ctx, const unsigned char *members, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stride,
                             size_t out_cap_frames, size_t *out_frames, int device_data, int final_frame) {
  LoudNormBatch *b = lnb_of(ctx);
  if (!b) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "audioloudnorm: not negotiated (setup_batch not called)");
  if (!members) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: null member list");
  bool any = false;
  for (size_t s = 0; s < b->S; s++) { b->active[s] = members[s] ? 1 : 0; any |= members[s] != 0; }
  if (!any) { *out_frames = 0; return MI355_OK; }
  return lnb_process(ctx, b, data, stream_stride, frames, out, out_stride, out_cap_frames, out_frames, device_data, final_frame);
}

static int lnb_process(mi355_ctx *ctx, LoudNormBatch *b, const double *data, size_t stream_stride, size_t frames, double *out, size_t out_stride, size_t out_cap_frames,
                       size_t *out_frames, int device_data, int final_frame) {
  *out_frames = 0;
  const size_t ch = b->channels, S = b->S;
  // the members of the call stand at ONE frame type and frame size (the class): taken from the first, checked for the rest
  size_t first = S;
  for (size_t s = 0; s < S; s++) if (b->active[s]) { first = s; break; }
  if (first == S) return MI355_OK;
  const size_t cspf = b->current_samples_per_frame[first];
  int ft = b->frame_type[first];
  for (size_t s = first; s < S; s++)
    if (b->active[s] && (b->current_samples_per_frame[s] != cspf || b->frame_type[s] != ft))
      return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: the streams of one call stand at one frame type and size");
  if (!final_frame && frames != cspf) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: a batch takes whole frames (mi355_loudnorm_batch_frame_size)");
  if (final_frame && frames >= cspf && !(frames == 0)) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: the final frame is shorter than a full one");
  if (frames && (!data || stream_stride < frames * ch)) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: bad input / stream stride");
  if (final_frame && cspf != FRAME_SIZE && frames == 0) return MI355_OK;  // nothing at all: the element answers FlowError::Eos (imp.rs:289-293)
  {
    // The output capacity is checked BEFORE anything changes: the meter has not seen the frame, frame_type has not moved. A
    // caller told "output buffer too small" can come back with a larger one and gets what State::process gives.
    int f2 = ft;
    if (final_frame && cspf == FRAME_SIZE) f2 = FT_FINAL;
    if (f2 == FT_FIRST && frames < cspf) f2 = FT_LINEAR;  // process_first_frame_is_last
    const size_t need = f2 == FT_FINAL ? 30 * FRAME_SIZE - (FRAME_SIZE - frames) : (f2 == FT_LINEAR ? frames : (f2 == FT_FIRST ? FRAME_SIZE : cspf));
    if (need > out_cap_frames || (need && (!out || out_stride < need * ch))) return set_error(ctx, MI355_ERR_INVALID_ARG, "audioloudnorm: output buffer too small");
  }
  auto set_type = [&](int t) { for (size_t s = 0; s < S; s++) if (b->active[s]) b->frame_type[s] = t; };
  if (final_frame && cspf == FRAME_SIZE) { ft = FT_FINAL; set_type(ft); }
  int rc;
  // the frame on the device, packed [S][frames * ch] (the members' rows only)
  const double *d_in = b->d_src;
  if (frames) {
    if (device_data && stream_stride == frames * ch) d_in = data;
    else if ((rc = lnb_for_runs(b, [&](size_t s0, size_t n) {
               return check_hip(ctx, hipMemcpy2DAsync(b->d_src + s0 * frames * ch, frames * ch * 8, data + s0 * stream_stride, stream_stride * 8, frames * ch * 8, n,
                                                      device_data ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream), "loudnorm: input copy");
             }))) return rc;
    if ((rc = lnb_meter_add(ctx, b, b->r128_in, d_in, frames))) return rc;
  }
  if (ft == FT_FIRST && frames < cspf) {  // process_first_frame_is_last (imp.rs:334-366)
    std::vector<double> global(S), peaks(S * ch);
    if ((rc = lnb_meter_query(ctx, b->r128_in, 2, global))) return rc;
    { MeterSwap sw(ctx, b->r128_in); if ((rc = ebur128_peak_batch(ctx, 0, peaks.data()))) return rc; }
    for (size_t s = 0; s < S; s++) {
      if (!b->active[s]) continue;
      double true_peak = 0.0;
      for (size_t c = 0; c < ch; c++) if (c == 0 || peaks[s * ch + c] > true_peak) true_peak = peaks[s * ch + c];
      const double offset = std::pow(10.0, (b->target_i - global[s]) / 20.0);
      const double offset_tp = true_peak * offset;
      b->offset[s] = offset_tp < b->target_tp ? offset : b->target_tp / true_peak;
    }
    ft = FT_LINEAR; set_type(ft);
  }
  const size_t need = ft == FT_FINAL ? 30 * FRAME_SIZE - (FRAME_SIZE - frames) : (ft == FT_LINEAR ? frames : (ft == FT_FIRST ? FRAME_SIZE : cspf));
  // a sub-frame of `n` frames from d_dst (packed [S][n * ch]) to the caller's buffer at frame offset `at`: the members' rows
  auto deliver = [&](size_t at, size_t n) -> int {
    return lnb_for_runs(b, [&](size_t s0, size_t cnt) {
      return check_hip(ctx, hipMemcpy2DAsync(out + s0 * out_stride + at * ch, out_stride * 8, b->d_dst + s0 * n * ch, n * ch * 8, n * ch * 8, cnt,
                                             device_data ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream), "loudnorm: output copy");
    });
  };
  switch (ft) {
    case FT_FIRST: {
      if ((rc = lnb_for_runs(b, [&](size_t s0, size_t n) {
             return check_hip(ctx, hipMemcpyAsync(b->d_buf + s0 * b->buf_len, d_in + s0 * b->buf_len, n * b->buf_len * 8, hipMemcpyDeviceToDevice, ctx->stream), "loudnorm: buf fill");
           }))) return rc;
      std::vector<double> shortterm(S);
      if ((rc = lnb_meter_query(ctx, b->r128_in, 1, shortterm))) return rc;
      for (size_t s = 0; s < S; s++) {
        if (!b->active[s]) continue;
        double env_shortterm;
        if (shortterm[s] < -70.0) { b->above_threshold[s] = 0; env_shortterm = 0.0; }
        else { b->above_threshold[s] = 1; env_shortterm = b->target_i - shortterm[s]; }
        for (int i = 0; i < 30; i++) b->delta[s * 30 + i] = std::pow(10.0, env_shortterm / 20.0);
        b->prev_delta[s] = b->delta[s * 30 + b->index[s]];
      }
      if ((rc = lnb_upload_gains(ctx, b, true))) return rc;
      hipLaunchKernelGGL(lnb_scale_kernel, dim3(ln_blocks(b->limiter_len, ctx->n_cu / 4 + 1), (unsigned)S), dim3(256), 0, ctx->stream, b->d_limiter, b->limiter_len,
                         (const double *)b->d_buf, b->buf_len, (const LnbGain *)b->d_gain);
      for (size_t s = 0; s < S; s++)
        if (b->active[s]) { b->buf_index[s] = b->limiter_len; b->limiter_buf_index[s] = 0; }   // (the snapshot just uploaded says 0 as well: a first frame's rings have not moved)
      if ((rc = lnb_limit(ctx, b, b->d_dst, FRAME_SIZE * ch, FRAME_SIZE, true))) return rc;
      if ((rc = lnb_meter_add(ctx, b, b->r128_out, b->d_dst, FRAME_SIZE))) return rc;
      if ((rc = deliver(0, FRAME_SIZE))) return rc;
      for (size_t s = 0; s < S; s++)
        if (b->active[s]) b->current_samples_per_frame[s] = FRAME_SIZE;
      set_type(FT_INNER);
      *out_frames = FRAME_SIZE;
      break;
    }
    case FT_INNER: {
      if ((rc = lnb_fill(ctx, b, d_in, 0, frames, (double)FRAME_SIZE))) return rc;
      if ((rc = lnb_limit(ctx, b, b->d_dst, frames * ch, frames, false))) return rc;
      if ((rc = lnb_meter_add(ctx, b, b->r128_out, b->d_dst, frames))) return rc;
      if ((rc = deliver(0, frames))) return rc;
      if ((rc = lnb_update_gain(ctx, b))) return rc;
      *out_frames = frames;
      break;
    }
    case FT_FINAL: {
      const size_t num_samples = frames;
      if ((rc = lnb_fill(ctx, b, d_in, 0, frames, (double)FRAME_SIZE))) return rc;
      if (num_samples != FRAME_SIZE && (rc = lnb_fill(ctx, b, nullptr, num_samples, FRAME_SIZE, (double)FRAME_SIZE))) return rc;  // process_fill_final_frame(num_samples, FRAME_SIZE)
      size_t smp_cnt = 0;
      while (smp_cnt < need) {
        const size_t frame_size = need - smp_cnt < FRAME_SIZE ? need - smp_cnt : FRAME_SIZE;
        if ((rc = lnb_limit(ctx, b, b->d_dst, frame_size * ch, frame_size, false))) return rc;
        if ((rc = deliver(smp_cnt, frame_size))) return rc;
        smp_cnt += frame_size;
        if (smp_cnt == need) break;
        if ((rc = lnb_meter_add(ctx, b, b->r128_out, b->d_dst, frame_size))) return rc;
        if ((rc = lnb_update_gain(ctx, b))) return rc;
        const size_t next_frame_size = need - smp_cnt < FRAME_SIZE ? need - smp_cnt : FRAME_SIZE;
        if ((rc = lnb_fill(ctx, b, nullptr, 0, next_frame_size, (double)next_frame_size))) return rc;
        if (next_frame_size < FRAME_SIZE)
          for (size_t s = 0; s < S; s++)
            if (b->active[s]) advance(&b->limiter_buf_index[s], FRAME_SIZE - next_frame_size, b->limiter_len);  // sic (imp.rs:763)
        if (next_frame_size < FRAME_SIZE) b->adv += FRAME_SIZE - next_frame_size;
      }
      *out_frames = need;
      break;
    }
    default: {
      if (frames) {
        if ((rc = lnb_upload_gains(ctx, b, true))) return rc;
        hipLaunchKernelGGL(lnb_linear_kernel, dim3(ln_blocks(frames * ch, ctx->n_cu / 4 + 1), (unsigned)S), dim3(256), 0, ctx->stream, b->d_dst, frames * ch, d_in, frames * ch,
                           frames * ch, (const LnbGain *)b->d_gain);
        if ((rc = lnb_meter_add(ctx, b, b->r128_out, b->d_dst, frames))) return rc;
        if ((rc = deliver(0, frames))) return rc;
      }
      *out_frames = frames;
      break;
    }
  }
  // host buffers are the caller's again when this returns; device callers stay asynchronous
  return device_data ? check_hip(ctx, hipGetLastError(), "loudnorm batch") : check_hip(ctx, hipStreamSynchronize(ctx->stream), "loudnorm: sync");
}

}  // namespace mi355
