/* oracle/ebur128_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the loudness measurement behind `ebur128level` (and `audioloudnorm`):
 * the arithmetic lives in the third-party crate `ebur128 = 0.1.10` (Cargo.lock:3685-3686), a Rust port
 * of libebur128, which is NOT vendored under /root/reference — **parity unpinned** against the crate.
 * What is restated here is the published algorithm (ITU-R BS.1770-4, EBU R 128 / Tech 3341 / Tech 3342)
 * in libebur128's formulation, serial and in f64, anchored on:
 *   - the reference call sites: EbuR128::new(channels, rate, mode) with Mode::HISTOGRAM always set
 *     (audio/audiofx/src/ebur128level/imp.rs:53-78,518), add_frames_{i16,i32,f32,f64}[_planar]
 *     (:690-739), loudness_momentary/shortterm/global, relative_threshold, loudness_range,
 *     sample_peak, true_peak (:378-452), reset() (:329), set_channel_map (:522-595);
 *   - the EBU Tech 3341 / 3342 minimum-requirement test signals (synthetic sines with published
 *     expected readings, +-0.1 LU / +-1 LU), replayed by tests/test_oracle_ebur128.py.
 *
 * Algorithm summary
 *   K-weighting: high-shelf (f0 1681.97 Hz, +4 dB) and high-pass (f0 38.135 Hz) biquads, bilinear
 *   transformed for the sample rate and convolved into one 4th-order section, direct form II, per channel.
 *   Every 100 ms: energy of the last 400 ms (channel-weighted mean square) -> block histogram if above the
 *   -70 LUFS absolute gate. Loudness = -0.691 + 10 log10(energy). Integrated loudness: mean of the
 *   histogram above the relative gate (-10 LU below the mean of all absolute-gated blocks).
 *   LRA: 3 s short-term energies every 1 s, relative gate -20 LU, 10th..95th percentile spread.
 *   Sample peak: max |x|. True peak: 49-tap Hann-windowed sinc polyphase interpolator, x4 below 96 kHz,
 *   x2 below 192 kHz.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { EB_MODE_M = 1, EB_MODE_S = 2, EB_MODE_I = 4, EB_MODE_LRA = 8, EB_MODE_SAMPLE_PEAK = 16, EB_MODE_TRUE_PEAK = 32 };
/* channel classes: 0 unused, 1 normal weight 1.0, 2 surround weight 1.41, 3 dual mono weight 2.0 */
enum { EB_CH_UNUSED = 0, EB_CH_NORMAL = 1, EB_CH_SURROUND = 2, EB_CH_DUAL_MONO = 3 };

#define EB_PI 3.14159265358979323846
#define HIST_BINS 1000

typedef struct {
  unsigned count;
  unsigned *index;
  double *coeff;
} interp_filter;

typedef struct oracle_ebur128 {
  unsigned channels, mode;
  unsigned long rate;
  int *channel_class;
  double b[5], a[5];
  double (*v)[5];
  size_t samples_in_100ms, audio_data_frames, audio_data_index, needed_frames, short_term_frame_counter;
  double *audio_data;
  unsigned long *block_hist, *st_hist;
  double *sample_peak, *prev_sample_peak, *true_peak, *prev_true_peak;
  /* true peak interpolator */
  unsigned interp_factor, interp_taps, interp_delay, interp_zi;
  interp_filter *interp_filters;
  float **interp_z;
} oracle_ebur128;

static double hist_energies[HIST_BINS];
static double hist_boundaries[HIST_BINS + 1];
static int hist_ready = 0;

static void hist_init(void) {
  if (hist_ready) return;
  hist_boundaries[0] = pow(10.0, (-70.0 + 0.691) / 10.0);
  for (int i = 0; i < HIST_BINS; i++) hist_energies[i] = pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
  for (int i = 1; i <= HIST_BINS; i++) hist_boundaries[i] = pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
  hist_ready = 1;
}

static size_t find_hist_index(double energy) {
  size_t lo = 0, hi = HIST_BINS;
  do {
    size_t mid = (lo + hi) / 2;
    if (energy >= hist_boundaries[mid]) lo = mid; else hi = mid;
  } while (hi - lo != 1);
  return lo;
}

static double energy_to_loudness(double e) { return 10.0 * (log(e) / log(10.0)) - 0.691; }

static void init_filter(oracle_ebur128 *st) {
  double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
  double K = tan(EB_PI * f0 / (double)st->rate);
  double Vh = pow(10.0, G / 20.0), Vb = pow(Vh, 0.4996667741545416);
  double pb[3] = {0, 0, 0}, pa[3] = {1, 0, 0}, rb[3] = {1, -2, 1}, ra[3] = {1, 0, 0};
  double a0 = 1.0 + K / Q + K * K;
  pb[0] = (Vh + Vb * K / Q + K * K) / a0;
  pb[1] = 2.0 * (K * K - Vh) / a0;
  pb[2] = (Vh - Vb * K / Q + K * K) / a0;
  pa[1] = 2.0 * (K * K - 1.0) / a0;
  pa[2] = (1.0 - K / Q + K * K) / a0;
  f0 = 38.13547087602444; Q = 0.5003270373238773;
  K = tan(EB_PI * f0 / (double)st->rate);
  ra[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
  ra[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
  st->b[0] = pb[0] * rb[0];
  st->b[1] = pb[0] * rb[1] + pb[1] * rb[0];
  st->b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
  st->b[3] = pb[1] * rb[2] + pb[2] * rb[1];
  st->b[4] = pb[2] * rb[2];
  st->a[0] = pa[0] * ra[0];
  st->a[1] = pa[0] * ra[1] + pa[1] * ra[0];
  st->a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
  st->a[3] = pa[1] * ra[2] + pa[2] * ra[1];
  st->a[4] = pa[2] * ra[2];
}

static void interp_create(oracle_ebur128 *st, unsigned taps, unsigned factor) {
  st->interp_taps = taps;
  st->interp_factor = factor;
  st->interp_delay = (taps + factor - 1) / factor;
  st->interp_filters = (interp_filter *)calloc(factor, sizeof(interp_filter));
  for (unsigned j = 0; j < factor; j++) {
    st->interp_filters[j].index = (unsigned *)calloc(st->interp_delay, sizeof(unsigned));
    st->interp_filters[j].coeff = (double *)calloc(st->interp_delay, sizeof(double));
  }
  st->interp_z = (float **)calloc(st->channels, sizeof(float *));
  for (unsigned c = 0; c < st->channels; c++) st->interp_z[c] = (float *)calloc(st->interp_delay, sizeof(float));
  for (unsigned j = 0; j < taps; j++) {
    double m = (double)j - (double)(taps - 1) / 2.0;
    double c = 1.0;
    if (fabs(m) > 0.000001) c = sin(m * EB_PI / factor) / (m * EB_PI / factor);
    c *= 0.5 * (1.0 - cos(2.0 * EB_PI * j / (taps - 1))); /* Hann window */
    if (fabs(c) > 0.000001) {
      unsigned f = j % factor, t = st->interp_filters[f].count++;
      st->interp_filters[f].coeff[t] = c;
      st->interp_filters[f].index[t] = j / factor;
    }
  }
  st->interp_zi = 0;
}

oracle_ebur128 *oracle_ebur128_new(unsigned channels, unsigned long rate, unsigned mode) {
  hist_init();
  if (channels == 0 || rate < 16) return NULL;
  oracle_ebur128 *st = (oracle_ebur128 *)calloc(1, sizeof *st);
  /* libebur128 mode bits are cumulative: TRUE_PEAK implies SAMPLE_PEAK, LRA implies S, S and I imply M */
  if (mode & EB_MODE_TRUE_PEAK) mode |= EB_MODE_SAMPLE_PEAK;
  if (mode & EB_MODE_LRA) mode |= EB_MODE_S;
  if (mode & (EB_MODE_S | EB_MODE_I)) mode |= EB_MODE_M;
  st->channels = channels; st->rate = rate; st->mode = mode;
  st->channel_class = (int *)calloc(channels, sizeof(int));
  /* libebur128 default map: L, R, C, unused (LFE), Ls, Rs */
  for (unsigned c = 0; c < channels; c++) st->channel_class[c] = (c == 3) ? EB_CH_UNUSED : ((c == 4 || c == 5) ? EB_CH_SURROUND : EB_CH_NORMAL);
  st->v = (double(*)[5])calloc(channels, sizeof(double[5]));
  st->sample_peak = (double *)calloc(channels, sizeof(double));
  st->prev_sample_peak = (double *)calloc(channels, sizeof(double));
  st->true_peak = (double *)calloc(channels, sizeof(double));
  st->prev_true_peak = (double *)calloc(channels, sizeof(double));
  st->samples_in_100ms = (rate + 5) / 10;
  unsigned long window = (mode & (EB_MODE_S | EB_MODE_LRA)) ? 3000 : 400; /* LRA implies S in libebur128's mode bits */
  st->audio_data_frames = rate * window / 1000;
  if (st->audio_data_frames % st->samples_in_100ms) st->audio_data_frames += st->samples_in_100ms - (st->audio_data_frames % st->samples_in_100ms);
  st->audio_data = (double *)calloc(st->audio_data_frames * channels, sizeof(double));
  st->needed_frames = st->samples_in_100ms * 4;
  st->block_hist = (unsigned long *)calloc(HIST_BINS, sizeof(unsigned long));
  st->st_hist = (unsigned long *)calloc(HIST_BINS, sizeof(unsigned long));
  init_filter(st);
  if (mode & EB_MODE_TRUE_PEAK) {
    if (rate < 96000) interp_create(st, 49, 4);
    else if (rate < 192000) interp_create(st, 49, 2);
  }
  return st;
}

void oracle_ebur128_free(oracle_ebur128 *st) {
  if (!st) return;
  if (st->interp_filters) {
    for (unsigned j = 0; j < st->interp_factor; j++) { free(st->interp_filters[j].index); free(st->interp_filters[j].coeff); }
    free(st->interp_filters);
    for (unsigned c = 0; c < st->channels; c++) free(st->interp_z[c]);
    free(st->interp_z);
  }
  free(st->channel_class); free(st->v); free(st->sample_peak); free(st->prev_sample_peak); free(st->true_peak);
  free(st->prev_true_peak); free(st->audio_data); free(st->block_hist); free(st->st_hist); free(st);
}

void oracle_ebur128_set_channel_class(oracle_ebur128 *st, unsigned c, int cls) { if (c < st->channels) st->channel_class[c] = cls; }

/* EbuR128::reset(): clears measurement state, keeps configuration */
void oracle_ebur128_reset(oracle_ebur128 *st) {
  memset(st->audio_data, 0, st->audio_data_frames * st->channels * sizeof(double));
  memset(st->v, 0, st->channels * sizeof(double[5]));
  memset(st->block_hist, 0, HIST_BINS * sizeof(unsigned long));
  memset(st->st_hist, 0, HIST_BINS * sizeof(unsigned long));
  for (unsigned c = 0; c < st->channels; c++) st->sample_peak[c] = st->prev_sample_peak[c] = st->true_peak[c] = st->prev_true_peak[c] = 0.0;
  st->audio_data_index = 0; st->needed_frames = st->samples_in_100ms * 4; st->short_term_frame_counter = 0;
  if (st->interp_z) { for (unsigned c = 0; c < st->channels; c++) memset(st->interp_z[c], 0, st->interp_delay * sizeof(float)); st->interp_zi = 0; }
}

/* filter `frames` frames given as doubles in [-1,1] scale, element (i,c) at src[i*stride_f + c*stride_c] */
static void filter_frames(oracle_ebur128 *st, const double *src, size_t frames, size_t stride_f, size_t stride_c) {
  if (st->mode & EB_MODE_SAMPLE_PEAK) {
    for (unsigned c = 0; c < st->channels; c++) {
      double max = 0.0;
      for (size_t i = 0; i < frames; i++) {
        double x = src[i * stride_f + c * stride_c];
        if (x > max) max = x; else if (-x > max) max = -x;
      }
      if (max > st->prev_sample_peak[c]) st->prev_sample_peak[c] = max;
    }
  }
  if ((st->mode & EB_MODE_TRUE_PEAK) && st->interp_filters) {
    for (size_t i = 0; i < frames; i++) {
      for (unsigned c = 0; c < st->channels; c++) {
        st->interp_z[c][st->interp_zi] = (float)src[i * stride_f + c * stride_c];
        for (unsigned f = 0; f < st->interp_factor; f++) {
          double acc = 0.0;
          for (unsigned t = 0; t < st->interp_filters[f].count; t++) {
            int k = (int)st->interp_zi - (int)st->interp_filters[f].index[t];
            if (k < 0) k += (int)st->interp_delay;
            acc += (double)st->interp_z[c][k] * st->interp_filters[f].coeff[t];
          }
          double o = (double)(float)acc; /* resampler output buffer is f32 */
          if (o > st->prev_true_peak[c]) st->prev_true_peak[c] = o; else if (-o > st->prev_true_peak[c]) st->prev_true_peak[c] = -o;
        }
      }
      if (++st->interp_zi == st->interp_delay) st->interp_zi = 0;
    }
  }
  for (unsigned c = 0; c < st->channels; c++) {
    if (st->channel_class[c] == EB_CH_UNUSED) continue;
    double *v = st->v[c];
    double *dst = st->audio_data + st->audio_data_index + c;
    for (size_t i = 0; i < frames; i++) {
      v[0] = src[i * stride_f + c * stride_c] - st->a[1] * v[1] - st->a[2] * v[2] - st->a[3] * v[3] - st->a[4] * v[4];
      dst[i * st->channels] = st->b[0] * v[0] + st->b[1] * v[1] + st->b[2] * v[2] + st->b[3] * v[3] + st->b[4] * v[4];
      v[4] = v[3]; v[3] = v[2]; v[2] = v[1]; v[1] = v[0];
    }
    for (int k = 1; k <= 4; k++) if (fabs(v[k]) < DBL_MIN) v[k] = 0.0;
  }
}

static double gating_block_energy(const oracle_ebur128 *st, size_t frames_per_block) {
  double sum = 0.0;
  const size_t ch = st->channels;
  for (unsigned c = 0; c < ch; c++) {
    if (st->channel_class[c] == EB_CH_UNUSED) continue;
    double cs = 0.0;
    if (st->audio_data_index < frames_per_block * ch) {
      for (size_t i = 0; i < st->audio_data_index / ch; i++) cs += st->audio_data[i * ch + c] * st->audio_data[i * ch + c];
      for (size_t i = st->audio_data_frames - (frames_per_block - st->audio_data_index / ch); i < st->audio_data_frames; i++)
        cs += st->audio_data[i * ch + c] * st->audio_data[i * ch + c];
    } else {
      for (size_t i = st->audio_data_index / ch - frames_per_block; i < st->audio_data_index / ch; i++)
        cs += st->audio_data[i * ch + c] * st->audio_data[i * ch + c];
    }
    if (st->channel_class[c] == EB_CH_SURROUND) cs *= 1.41; else if (st->channel_class[c] == EB_CH_DUAL_MONO) cs *= 2.0;
    sum += cs;
  }
  return sum / (double)frames_per_block;
}

/* add `frames` frames; src element (i,c) at src[i*stride_f + c*stride_c], already scaled to [-1,1] */
void oracle_ebur128_add_frames(oracle_ebur128 *st, const double *src, size_t frames, size_t stride_f, size_t stride_c) {
  size_t src_index = 0;
  for (unsigned c = 0; c < st->channels; c++) { st->prev_sample_peak[c] = 0.0; st->prev_true_peak[c] = 0.0; }
  while (frames > 0) {
    if (frames >= st->needed_frames) {
      filter_frames(st, src + src_index * stride_f, st->needed_frames, stride_f, stride_c);
      src_index += st->needed_frames;
      frames -= st->needed_frames;
      st->audio_data_index += st->needed_frames * st->channels;
      if (st->mode & EB_MODE_I) {
        double e = gating_block_energy(st, st->samples_in_100ms * 4);
        if (e >= hist_boundaries[0]) st->block_hist[find_hist_index(e)]++;
      }
      if (st->mode & EB_MODE_LRA) {
        st->short_term_frame_counter += st->needed_frames;
        if (st->short_term_frame_counter == st->samples_in_100ms * 30) {
          double e = gating_block_energy(st, st->samples_in_100ms * 30);
          if (e >= hist_boundaries[0]) st->st_hist[find_hist_index(e)]++;
          st->short_term_frame_counter = st->samples_in_100ms * 20;
        }
      }
      st->needed_frames = st->samples_in_100ms;
      if (st->audio_data_index == st->audio_data_frames * st->channels) st->audio_data_index = 0;
    } else {
      filter_frames(st, src + src_index * stride_f, frames, stride_f, stride_c);
      st->audio_data_index += frames * st->channels;
      if (st->mode & EB_MODE_LRA) st->short_term_frame_counter += frames;
      st->needed_frames -= frames;
      frames = 0;
    }
  }
  for (unsigned c = 0; c < st->channels; c++) {
    if (st->prev_sample_peak[c] > st->sample_peak[c]) st->sample_peak[c] = st->prev_sample_peak[c];
    if (st->prev_true_peak[c] > st->true_peak[c]) st->true_peak[c] = st->prev_true_peak[c];
  }
}

static int energy_in_interval(const oracle_ebur128 *st, size_t interval_frames, double *out) {
  if (interval_frames > st->audio_data_frames) return -1;
  *out = gating_block_energy(st, interval_frames);
  return 0;
}

int oracle_ebur128_loudness_momentary(const oracle_ebur128 *st, double *out) {
  double e;
  if (energy_in_interval(st, st->samples_in_100ms * 4, &e)) return -1;
  *out = e <= 0.0 ? -HUGE_VAL : energy_to_loudness(e);
  return 0;
}
int oracle_ebur128_loudness_shortterm(const oracle_ebur128 *st, double *out) {
  double e;
  if (energy_in_interval(st, st->samples_in_100ms * 30, &e)) return -1;
  *out = e <= 0.0 ? -HUGE_VAL : energy_to_loudness(e);
  return 0;
}

static int relative_threshold_energy(const oracle_ebur128 *st, double *thr, size_t *count) {
  double t = 0.0; size_t n = 0;
  for (int j = 0; j < HIST_BINS; j++) { t += (double)st->block_hist[j] * hist_energies[j]; n += st->block_hist[j]; }
  *thr = t; *count = n;
  return 0;
}

int oracle_ebur128_relative_threshold(const oracle_ebur128 *st, double *out) {
  double t; size_t n;
  relative_threshold_energy(st, &t, &n);
  if (!n) { *out = -70.0; return 0; }
  t /= (double)n; t *= 0.1; /* -10 LU */
  *out = energy_to_loudness(t);
  return 0;
}

int oracle_ebur128_loudness_global(const oracle_ebur128 *st, double *out) {
  double t; size_t n;
  relative_threshold_energy(st, &t, &n);
  if (!n) { *out = -HUGE_VAL; return 0; }
  t /= (double)n; t *= 0.1;
  size_t start;
  if (t < hist_boundaries[0]) start = 0;
  else { start = find_hist_index(t); if (t > hist_energies[start]) ++start; }
  double g = 0.0; n = 0;
  for (size_t j = start; j < HIST_BINS; j++) { g += (double)st->block_hist[j] * hist_energies[j]; n += st->block_hist[j]; }
  if (!n) { *out = -HUGE_VAL; return 0; }
  *out = energy_to_loudness(g / (double)n);
  return 0;
}

int oracle_ebur128_loudness_range(const oracle_ebur128 *st, double *out) {
  size_t size = 0; double power = 0.0;
  for (int j = 0; j < HIST_BINS; j++) { size += st->st_hist[j]; power += (double)st->st_hist[j] * hist_energies[j]; }
  if (!size) { *out = 0.0; return 0; }
  power /= (double)size;
  double integrated = 0.01 * power; /* -20 LU */
  size_t index;
  if (integrated < hist_boundaries[0]) index = 0;
  else { index = find_hist_index(integrated); if (integrated > hist_energies[index]) ++index; }
  size = 0;
  for (size_t j = index; j < HIST_BINS; j++) size += st->st_hist[j];
  if (!size) { *out = 0.0; return 0; }
  size_t plow = (size_t)((double)(size - 1) * 0.1 + 0.5), phigh = (size_t)((double)(size - 1) * 0.95 + 0.5);
  size_t acc = 0, j = index;
  while (acc <= plow) acc += st->st_hist[j++];
  double l_en = hist_energies[j - 1];
  while (acc <= phigh) acc += st->st_hist[j++];
  double h_en = hist_energies[j - 1];
  *out = energy_to_loudness(h_en) - energy_to_loudness(l_en);
  return 0;
}

double oracle_ebur128_sample_peak(const oracle_ebur128 *st, unsigned c) { return c < st->channels ? st->sample_peak[c] : -1.0; }
double oracle_ebur128_true_peak(const oracle_ebur128 *st, unsigned c) {
  if (c >= st->channels) return -1.0;
  /* libebur128 reports max(true peak, sample peak) */
  return st->true_peak[c] > st->sample_peak[c] ? st->true_peak[c] : st->sample_peak[c];
}
void oracle_ebur128_filter_coeffs(const oracle_ebur128 *st, double b[5], double a[5]) { memcpy(b, st->b, sizeof st->b); memcpy(a, st->a, sizeof st->a); }
