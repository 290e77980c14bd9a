/* oracle/loudnorm_oracle.c — CPU restatement of audioloudnorm's State machine. TEST INFRASTRUCTURE ONLY.
 *
 * Follows audio/audiofx/src/audioloudnorm/imp.rs (itself a port of ffmpeg's af_loudnorm.c) function by function:
 *   State::new                          :130-205      process                       :800-828
 *   process_first_frame_is_last         :312-366      true_peak_limiter_out         :845-888
 *   process_first_frame                 :368-442      true_peak_limiter_attack      :890-1094
 *   process_fill_inner_frame            :444-524      true_peak_limiter_sustain     :1096-1216
 *   process_update_gain_inner_frame     :526-608      true_peak_limiter_release     :1218-1330
 *   process_inner_frame                 :610-652      true_peak_limiter_first_frame :1332-1372
 *   process_fill_final_frame            :654-697      true_peak_limiter             :1374-1430
 *   process_final_frame                 :699-779      detect_peak                   :1438-1524
 *   process_linear_frame                :781-816      gaussian_filter               :1526-1541
 *   init_gaussian_filter                :1893-1914
 * The two loudness meters (r128_in / r128_out, crate ebur128 0.1.10, modes HISTOGRAM|I|S|LRA|SAMPLE_PEAK, :131-150) are
 * the restatement in ebur128_oracle.c: PARITY UNPINNED for the crate part, exact for everything in this file.
 * Input/output: interleaved f64 at 192 kHz (the only caps the element accepts, :1848-1851).
 * The adapter logic around it (drain_full_frames / drain, :226-310) is restated by loudnorm_push / loudnorm_drain. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ebur128_oracle.c */
typedef struct oracle_ebur128 oracle_ebur128;
oracle_ebur128 *oracle_ebur128_new(unsigned channels, unsigned long rate, unsigned mode);
void oracle_ebur128_free(oracle_ebur128 *st);
void oracle_ebur128_add_frames(oracle_ebur128 *st, const double *src, size_t frames, size_t stride_f, size_t stride_c);
int oracle_ebur128_loudness_shortterm(const oracle_ebur128 *st, double *out);
int oracle_ebur128_loudness_global(const oracle_ebur128 *st, double *out);
int oracle_ebur128_relative_threshold(const oracle_ebur128 *st, double *out);
double oracle_ebur128_sample_peak(const oracle_ebur128 *st, unsigned c);

enum { GAIN_LOOKAHEAD = 3 * 192000, FRAME_SIZE = 19200, LIMITER_ATTACK_WINDOW = 1920, LIMITER_RELEASE_WINDOW = 19200, LIMITER_LOOKAHEAD = 1920 };
enum { FT_FIRST, FT_INNER, FT_FINAL, FT_LINEAR };
enum { LS_OUT, LS_ATTACK, LS_SUSTAIN, LS_RELEASE };

typedef struct {
  size_t channels;
  size_t current_samples_per_frame;
  double offset, target_i, target_lra, target_tp;
  double *buf; size_t buf_len, buf_index, prev_buf_index;
  double weights[21], delta[30]; size_t index; double prev_delta;
  double gain_reduction[2];
  double *limiter_buf; size_t limiter_len, limiter_buf_index;
  double *prev_smp;
  int limiter_state; size_t env_cnt; int have_sustain; size_t sustain_cnt;
  int frame_type, above_threshold;
  oracle_ebur128 *r128_in, *r128_out;
  /* adapter */
  double *adapter; size_t adapter_len, adapter_cap;
} loudnorm;

static void init_gaussian_filter(double w[21]) {
  double total = 0.0;
  const double sigma = 3.5;
  const int offset = 21 / 2;
  const double c1 = 1.0 / (sigma * sqrt(2.0 * M_PI));
  const double c2 = 2.0 * pow(sigma, 2.0);
  for (int i = 0; i < 21; i++) {
    const double x = (double)i - (double)offset;
    w[i] = c1 * exp(-(pow(x, 2.0) / c2));
    total += w[i];
  }
  const double adjust = 1.0 / total;
  for (int i = 0; i < 21; i++) w[i] *= adjust;
}

loudnorm *oracle_loudnorm_new(unsigned channels, double loudness_target, double loudness_range_target, double max_true_peak, double offset_db) {
  loudnorm *s = (loudnorm *)calloc(1, sizeof(*s));
  const unsigned mode = 4 | 2 | 8 | 16; /* I | S | LRA | SAMPLE_PEAK (histogram is how the restated meter always works) */
  s->channels = channels;
  s->r128_in = oracle_ebur128_new(channels, 192000, mode);
  s->r128_out = oracle_ebur128_new(channels, 192000, mode);
  s->buf_len = (size_t)GAIN_LOOKAHEAD * channels;
  s->buf = (double *)calloc(s->buf_len, sizeof(double));
  s->limiter_len = (size_t)(2 * FRAME_SIZE + LIMITER_LOOKAHEAD) * channels;
  s->limiter_buf = (double *)calloc(s->limiter_len, sizeof(double));
  s->prev_smp = (double *)calloc(channels, sizeof(double));
  s->current_samples_per_frame = GAIN_LOOKAHEAD;
  s->index = 1;
  s->limiter_state = LS_OUT;
  s->offset = pow(10.0, offset_db / 20.0);
  s->target_tp = pow(10.0, max_true_peak / 20.0);
  s->target_i = loudness_target;
  s->target_lra = loudness_range_target;
  init_gaussian_filter(s->weights);
  s->frame_type = FT_FIRST;
  return s;
}

void oracle_loudnorm_free(loudnorm *s) {
  if (!s) return;
  oracle_ebur128_free(s->r128_in); oracle_ebur128_free(s->r128_out);
  free(s->buf); free(s->limiter_buf); free(s->prev_smp); free(s->adapter); free(s);
}

static double gaussian_filter(const loudnorm *s, size_t index) {
  double result = 0.0;
  index = index > 10 ? index - 10 : index + 20;
  for (size_t k = 0; k < 21; k++) {  /* delta[index..] chained with delta[..] */
    const size_t j = index + k < 30 ? index + k : index + k - 30;
    result += s->delta[j] * s->weights[k];
  }
  return result;
}

static int detect_peak(loudnorm *s, size_t offset, size_t samples, size_t *peak_delta, double *peak_value) {
  const size_t ch = s->channels, len = s->limiter_len;
  size_t index = s->limiter_buf_index + (offset + LIMITER_LOOKAHEAD) * ch;
  if (index >= len) index -= len;
  for (size_t n = 0; n < samples; n++) {
    size_t next_index = index + ch;
    if (next_index >= len) next_index -= len;
    const double *this_ = s->limiter_buf + index, *next_ = s->limiter_buf + next_index;
    int detected = 0;
    for (size_t c = 0; c < ch; c++) {
      const double th = fabs(this_[c]), nx = fabs(next_[c]);
      detected = 0;
      if (s->prev_smp[c] <= th && th >= nx && th > s->target_tp && n > 0) {
        detected = 1;
        for (size_t i = 2; i < 12; i++) {
          size_t ni = index + c + i * ch;
          if (ni >= len) ni -= len;
          if (fabs(s->limiter_buf[ni]) > th) { detected = 0; break; }
        }
        if (detected) break;
      }
      s->prev_smp[c] = th;
    }
    if (detected) {
      double max_peak = 0.0;
      for (size_t c = 0; c < ch; c++) {
        if (c == 0 || fabs(this_[c]) > max_peak) max_peak = fabs(this_[c]);
        s->prev_smp[c] = fabs(this_[c]);
      }
      *peak_delta = n; *peak_value = max_peak;
      return 1;
    }
    index = next_index;
  }
  return 0;
}

static size_t limiter_out(loudnorm *s, size_t smp_cnt, size_t nb) {
  size_t pd; double pv;
  if (detect_peak(s, smp_cnt, nb - smp_cnt, &pd, &pv)) {
    s->limiter_state = LS_ATTACK;
    s->env_cnt = 0;
    s->have_sustain = 0;
    s->gain_reduction[0] = 1.0;
    s->gain_reduction[1] = s->target_tp / pv;
    smp_cnt += LIMITER_LOOKAHEAD + pd - LIMITER_ATTACK_WINDOW;
  } else {
    smp_cnt = nb;
  }
  return smp_cnt;
}

static void mul_frame(loudnorm *s, size_t *index, double g) {
  for (size_t c = 0; c < s->channels; c++) s->limiter_buf[*index + c] *= g;
  *index += s->channels;
  if (*index >= s->limiter_len) *index -= s->limiter_len;
}

static size_t limiter_attack(loudnorm *s, size_t smp_cnt, size_t nb) {
  const size_t ch = s->channels;
  size_t pd = 0; double pv = 0.0;
  const int peak = detect_peak(s, smp_cnt, nb - smp_cnt, &pd, &pv);
  const int have_new = peak;
  const size_t new_peak_smp_cnt = smp_cnt + pd;
  size_t index = s->limiter_buf_index + smp_cnt * ch;
  if (index >= s->limiter_len) index -= s->limiter_len;
  while (s->env_cnt < LIMITER_ATTACK_WINDOW && smp_cnt < nb) {
    if (have_new && smp_cnt == new_peak_smp_cnt) break;
    const double env = s->gain_reduction[0] - ((double)s->env_cnt / ((double)LIMITER_ATTACK_WINDOW - 1.0) * (s->gain_reduction[0] - s->gain_reduction[1]));
    mul_frame(s, &index, env);
    smp_cnt += 1;
    s->env_cnt += 1;
  }
  if (have_new) {
    if (smp_cnt < new_peak_smp_cnt) {
      for (size_t k = smp_cnt; k < new_peak_smp_cnt; k++) mul_frame(s, &index, s->gain_reduction[1]);
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
        s->have_sustain = 0;
      } else {
        double new_end = (gain_reduction - s->gain_reduction[0]) / old_slope;
        new_end = fmax(new_end, 1.0);
        const double new_start = new_end - 1.0;
        s->gain_reduction[0] = s->gain_reduction[0] + new_start * old_slope;
        s->gain_reduction[1] = gain_reduction;
        double cur_pos = (current - s->gain_reduction[0]) / old_slope;
        /* f64::clamp(cur_pos, 0.0, 1.0): NaN stays NaN, and `NaN as usize` is 0 */
        if (cur_pos < 0.0) cur_pos = 0.0; else if (cur_pos > 1.0) cur_pos = 1.0;
        const double pos = ((double)LIMITER_ATTACK_WINDOW - 1.0) * cur_pos;
        s->env_cnt = (pos != pos) ? 0 : (size_t)pos;
        s->have_sustain = 1;
        s->sustain_cnt = s->env_cnt;
      }
      return smp_cnt;
    } else if (s->env_cnt < LIMITER_ATTACK_WINDOW) {
      s->have_sustain = 1;
      s->sustain_cnt = s->env_cnt;
    }
  }
  if (s->env_cnt == LIMITER_ATTACK_WINDOW && smp_cnt < nb) s->limiter_state = LS_SUSTAIN;
  return smp_cnt;
}

static size_t limiter_sustain(loudnorm *s, size_t smp_cnt, size_t nb) {
  const size_t ch = s->channels;
  size_t pd = 0; double pv = 0.0;
  const int peak = detect_peak(s, smp_cnt, nb - smp_cnt, &pd, &pv);
  if (peak || s->have_sustain) {
    const size_t sustain_cnt = peak ? pd : s->sustain_cnt;
    size_t index = s->limiter_buf_index + smp_cnt * ch;
    if (index >= s->limiter_len) index -= s->limiter_len;
    size_t k = 0;
    while (k < sustain_cnt && smp_cnt < nb) {
      mul_frame(s, &index, s->gain_reduction[1]);
      smp_cnt += 1;
      k += 1;
    }
    if (peak) {
      const double gain_reduction = s->target_tp / pv;
      if (gain_reduction < s->gain_reduction[1]) {
        s->limiter_state = LS_ATTACK;
        s->env_cnt = 0;
        s->have_sustain = 0;
        s->gain_reduction[0] = s->gain_reduction[1];
        s->gain_reduction[1] = gain_reduction;
      } else {
        s->have_sustain = 1;
        s->sustain_cnt = LIMITER_LOOKAHEAD;
      }
    } else {
      s->sustain_cnt -= k;
      if (s->sustain_cnt == 0) s->have_sustain = 0;
    }
  } else {
    s->limiter_state = LS_RELEASE;
    s->gain_reduction[0] = s->gain_reduction[1];
    s->gain_reduction[1] = 1.0;
    s->env_cnt = 0;
  }
  return smp_cnt;
}

static size_t limiter_release(loudnorm *s, size_t smp_cnt, size_t nb) {
  const size_t ch = s->channels;
  size_t index = s->limiter_buf_index + smp_cnt * ch;
  if (index >= s->limiter_len) index -= s->limiter_len;
  size_t pd = 0; double pv = 0.0;
  if (detect_peak(s, smp_cnt, nb - smp_cnt, &pd, &pv)) {
    const double gain_reduction = s->target_tp / pv;
    const double current = s->gain_reduction[0] - ((double)s->env_cnt / ((double)LIMITER_RELEASE_WINDOW - 1.0) * (s->gain_reduction[1] - s->gain_reduction[0]));
    if (gain_reduction < current) {
      for (size_t k = 0; k < pd; k++) { mul_frame(s, &index, s->gain_reduction[1]); smp_cnt += 1; }
      s->limiter_state = LS_ATTACK;
      s->env_cnt = 0;
      s->have_sustain = 0;
      s->gain_reduction[0] = current;
      s->gain_reduction[1] = gain_reduction;
    } else {
      s->gain_reduction[1] = current;
      s->limiter_state = LS_SUSTAIN;
    }
    return smp_cnt;
  }
  while (s->env_cnt < LIMITER_RELEASE_WINDOW && smp_cnt < nb) {
    const double env = s->gain_reduction[0] - ((double)s->env_cnt / ((double)LIMITER_RELEASE_WINDOW - 1.0) * (s->gain_reduction[1] - s->gain_reduction[0]));
    mul_frame(s, &index, env);
    smp_cnt += 1;
    s->env_cnt += 1;
  }
  if (smp_cnt < nb) s->limiter_state = LS_OUT;
  return smp_cnt;
}

static void limiter_first_frame(loudnorm *s) {
  const size_t ch = s->channels;
  double max = 0.0;
  for (size_t i = 0; i < (LIMITER_LOOKAHEAD + 1) * ch; i++)
    if (fabs(s->limiter_buf[i]) > max) max = s->limiter_buf[i];  /* sic: the signed value is kept (imp.rs:1339-1342) */
  for (size_t c = 0; c < ch; c++) s->prev_smp[c] = fabs(s->limiter_buf[LIMITER_LOOKAHEAD * ch + c]);
  if (max > s->target_tp) {
    s->limiter_state = LS_SUSTAIN;
    s->have_sustain = 1;
    s->sustain_cnt = LIMITER_LOOKAHEAD;
    s->gain_reduction[1] = s->target_tp / max;
  }
}

static void true_peak_limiter(loudnorm *s, double *dst, size_t nb) {
  const size_t ch = s->channels;
  if (s->frame_type == FT_FIRST) limiter_first_frame(s);
  size_t smp_cnt = 0;
  while (smp_cnt < nb) {
    switch (s->limiter_state) {
      case LS_OUT: smp_cnt = limiter_out(s, smp_cnt, nb); break;
      case LS_ATTACK: smp_cnt = limiter_attack(s, smp_cnt, nb); break;
      case LS_SUSTAIN: smp_cnt = limiter_sustain(s, smp_cnt, nb); break;
      default: smp_cnt = limiter_release(s, smp_cnt, nb); break;
    }
  }
  size_t index = s->limiter_buf_index;
  for (size_t n = 0; n < nb; n++) {
    for (size_t c = 0; c < ch; c++) {
      double o = s->limiter_buf[index + c];
      if (fabs(o) > s->target_tp) o = s->target_tp * (signbit(o) ? -1.0 : 1.0);  /* f64::signum: +-1 (NaN never passes the test) */
      dst[n * ch + c] = o;
    }
    index += ch;
    if (index >= s->limiter_len) index -= s->limiter_len;
  }
}

static void first_frame_is_last(loudnorm *s) {
  double global = 0.0;
  oracle_ebur128_loudness_global(s->r128_in, &global);
  double true_peak = 0.0;
  for (size_t c = 0; c < s->channels; c++) {
    const double peak = oracle_ebur128_sample_peak(s->r128_in, (unsigned)c);
    if (c == 0 || peak > true_peak) true_peak = peak;
  }
  const double offset = pow(10.0, (s->target_i - global) / 20.0);
  const double offset_tp = true_peak * offset;
  s->offset = offset_tp < s->target_tp ? offset : s->target_tp / true_peak;
  s->frame_type = FT_LINEAR;
}

static void fill_inner_frame(loudnorm *s, const double *src, size_t frames) {
  const size_t ch = s->channels;
  const double gain = gaussian_filter(s, s->index + 10 < 30 ? s->index + 10 : s->index + 10 - 30);
  const double gain_next = gaussian_filter(s, s->index + 11 < 30 ? s->index + 11 : s->index + 11 - 30);
  for (size_t n = 0; n < frames; n++) {
    /* write the new input 210 ms ... behind the read position, read the oldest buffered frame */
    const double current_gain = (gain + (((double)n / (double)FRAME_SIZE) * (gain_next - gain))) * s->offset;
    double tmp[64];
    for (size_t c = 0; c < ch; c++) tmp[c] = s->buf[s->buf_index + c];
    for (size_t c = 0; c < ch; c++) s->buf[s->prev_buf_index + c] = src[n * ch + c];
    /* buf_read and buf_write never alias (imp.rs:470-490): prev_buf_index != buf_index by construction; when they
     * are equal (only before the first inner frame completes a lap) the read happened first above */
    for (size_t c = 0; c < ch; c++) s->limiter_buf[s->limiter_buf_index + c] = tmp[c] * current_gain;
    s->limiter_buf_index += ch; if (s->limiter_buf_index >= s->limiter_len) s->limiter_buf_index -= s->limiter_len;
    s->prev_buf_index += ch; if (s->prev_buf_index >= s->buf_len) s->prev_buf_index -= s->buf_len;
    s->buf_index += ch; if (s->buf_index >= s->buf_len) s->buf_index -= s->buf_len;
  }
}

static void fill_final_frame(loudnorm *s, size_t idx, size_t num_samples) {
  const size_t ch = s->channels;
  const double gain = gaussian_filter(s, s->index + 10 < 30 ? s->index + 10 : s->index + 10 - 30);
  const double gain_next = gaussian_filter(s, s->index + 11 < 30 ? s->index + 11 : s->index + 11 - 30);
  for (size_t n = idx; n < num_samples; n++) {
    const double current_gain = (gain + (((double)n / (double)num_samples) * (gain_next - gain))) * s->offset;
    for (size_t c = 0; c < ch; c++) s->limiter_buf[s->limiter_buf_index + c] = s->buf[s->buf_index + c] * current_gain;
    s->limiter_buf_index += ch; if (s->limiter_buf_index >= s->limiter_len) s->limiter_buf_index -= s->limiter_len;
    s->buf_index += ch; if (s->buf_index >= s->buf_len) s->buf_index -= s->buf_len;
  }
}

static void update_gain_inner_frame(loudnorm *s) {
  double global = 0, shortterm = 0, relative_threshold = 0;
  oracle_ebur128_loudness_global(s->r128_in, &global);
  oracle_ebur128_loudness_shortterm(s->r128_in, &shortterm);
  oracle_ebur128_relative_threshold(s->r128_in, &relative_threshold);
  if (!s->above_threshold) {
    if (shortterm > -70.0) s->prev_delta *= 1.0058;
    double shortterm_out = 0;
    oracle_ebur128_loudness_shortterm(s->r128_out, &shortterm_out);
    if (shortterm_out >= s->target_i) s->above_threshold = 1;
  }
  if (shortterm < relative_threshold || shortterm <= -70.0 || !s->above_threshold) {
    s->delta[s->index] = s->prev_delta;
  } else {
    double env_global;
    if (fabs(shortterm - global) < (s->target_lra / 2.0)) env_global = shortterm - global;
    else if ((s->target_lra / 2.0) * (shortterm - global) < 0.0) env_global = -1.0;
    else env_global = 1.0;
    const double env_shortterm = s->target_i - shortterm;
    s->delta[s->index] = pow(10.0, (env_global + env_shortterm) / 20.0);
  }
  s->prev_delta = s->delta[s->index];
  s->index += 1;
  if (s->index >= 30) s->index -= 30;
}

/* One call of State::process (imp.rs:800-828). `src`: frames x channels; `dst` must hold the output of the frame type:
 * first/inner: 19200 frames; final: 30*19200 - (19200 - frames); linear: `frames`. Returns the frames written. */
size_t oracle_loudnorm_process(loudnorm *s, const double *src, size_t frames, double *dst) {
  const size_t ch = s->channels;
  oracle_ebur128_add_frames(s->r128_in, src, frames, ch, 1);
  if (s->frame_type == FT_FIRST && frames < s->current_samples_per_frame) first_frame_is_last(s);
  switch (s->frame_type) {
    case FT_FIRST: {
      memcpy(s->buf, src, sizeof(double) * s->buf_len);
      double shortterm = 0;
      oracle_ebur128_loudness_shortterm(s->r128_in, &shortterm);
      double env_shortterm;
      if (shortterm < -70.0) { s->above_threshold = 0; env_shortterm = 0.0; }
      else { s->above_threshold = 1; env_shortterm = s->target_i - shortterm; }
      for (int i = 0; i < 30; i++) s->delta[i] = pow(10.0, env_shortterm / 20.0);
      s->prev_delta = s->delta[s->index];
      for (size_t i = 0; i < s->limiter_len; i++) s->limiter_buf[i] = s->buf[i] * s->prev_delta * s->offset;
      s->buf_index = s->limiter_len;
      s->limiter_buf_index = 0;
      true_peak_limiter(s, dst, FRAME_SIZE);
      oracle_ebur128_add_frames(s->r128_out, dst, FRAME_SIZE, ch, 1);
      s->current_samples_per_frame = FRAME_SIZE;
      s->frame_type = FT_INNER;
      return FRAME_SIZE;
    }
    case FT_INNER: {
      fill_inner_frame(s, src, frames);
      true_peak_limiter(s, dst, s->current_samples_per_frame);
      oracle_ebur128_add_frames(s->r128_out, dst, s->current_samples_per_frame, ch, 1);
      update_gain_inner_frame(s);
      return s->current_samples_per_frame;
    }
    case FT_FINAL: {
      const size_t num_samples = frames;
      fill_inner_frame(s, src, frames);
      if (num_samples != FRAME_SIZE) fill_final_frame(s, num_samples, FRAME_SIZE);
      const size_t out_num_samples = 30 * (size_t)FRAME_SIZE - (FRAME_SIZE - num_samples);
      size_t smp_cnt = 0;
      while (smp_cnt < out_num_samples) {
        const size_t frame_size = out_num_samples - smp_cnt < FRAME_SIZE ? out_num_samples - smp_cnt : FRAME_SIZE;
        double *d = dst + smp_cnt * ch;
        true_peak_limiter(s, d, frame_size);
        smp_cnt += frame_size;
        if (smp_cnt == out_num_samples) break;
        oracle_ebur128_add_frames(s->r128_out, d, frame_size, ch, 1);
        update_gain_inner_frame(s);
        const size_t next_frame_size = out_num_samples - smp_cnt < FRAME_SIZE ? out_num_samples - smp_cnt : FRAME_SIZE;
        fill_final_frame(s, 0, next_frame_size);
        if (next_frame_size < FRAME_SIZE) {
          s->limiter_buf_index += (FRAME_SIZE - next_frame_size);  /* sic: samples, not samples*channels (imp.rs:763) */
          if (s->limiter_buf_index >= s->limiter_len) s->limiter_buf_index -= s->limiter_len;
        }
      }
      return out_num_samples;
    }
    default: {
      for (size_t i = 0; i < frames * ch; i++) dst[i] = src[i] * s->offset;
      oracle_ebur128_add_frames(s->r128_out, dst, frames, ch, 1);
      return frames;
    }
  }
}

/* drain_full_frames (imp.rs:226-268): push interleaved input, process every full frame; returns frames written to dst
 * (capacity checked by the caller: at most (frames + pending) rounded to 19200). */
size_t oracle_loudnorm_push(loudnorm *s, const double *src, size_t frames, double *dst) {
  const size_t ch = s->channels;
  if ((s->adapter_len + frames) * ch > s->adapter_cap) {
    s->adapter_cap = (s->adapter_len + frames) * ch * 2 + 1024;
    s->adapter = (double *)realloc(s->adapter, s->adapter_cap * sizeof(double));
  }
  memcpy(s->adapter + s->adapter_len * ch, src, frames * ch * sizeof(double));
  s->adapter_len += frames;
  size_t written = 0, used = 0;
  while (s->adapter_len - used >= s->current_samples_per_frame) {
    const size_t take = s->current_samples_per_frame;
    written += oracle_loudnorm_process(s, s->adapter + used * ch, take, dst + written * ch);
    used += take;
  }
  memmove(s->adapter, s->adapter + used * ch, (s->adapter_len - used) * ch * sizeof(double));
  s->adapter_len -= used;
  return written;
}

/* drain (imp.rs:270-310): returns frames written, or (size_t)-1 for "nothing to drain" (FlowError::Eos) */
size_t oracle_loudnorm_drain(loudnorm *s, double *dst) {
  const size_t avail = s->adapter_len;
  if (s->current_samples_per_frame == FRAME_SIZE) s->frame_type = FT_FINAL;
  else if (avail == 0) return (size_t)-1;
  const size_t n = oracle_loudnorm_process(s, s->adapter, avail, dst);
  s->adapter_len = 0;
  return n;
}

int oracle_loudnorm_frame_type(const loudnorm *s) { return s->frame_type; }
int oracle_loudnorm_limiter_state(const loudnorm *s) { return s->limiter_state; }
double oracle_loudnorm_offset(const loudnorm *s) { return s->offset; }
