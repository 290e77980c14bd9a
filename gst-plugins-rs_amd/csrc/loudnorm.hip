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

}  // namespace mi355
