/* oracle/echo_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of rsaudioecho's per-buffer loop. Checker + "port" CPU baseline; never
 * linked into the product path.
 *
 * Follows:
 *   audio/audiofx/src/audioecho/ring_buffer.rs:9-29   RingBuffer::new            -> oracle_echo_new
 *   audio/audiofx/src/audioecho/ring_buffer.rs:37-82  RingBufferIter (+Drop)     -> loop indices below
 *   audio/audiofx/src/audioecho/imp.rs:69-85          AudioEcho::process         -> oracle_echo_process_f32/f64
 *   audio/audiofx/src/audioecho/imp.rs:205-227        transform_ip (delay clamp) -> oracle_echo_delay_samples
 *   audio/audiofx/src/audioecho/imp.rs:248-259        setup (ring sizing)        -> oracle_echo_ring_len
 *
 * Parity pinning: the reference has NO rsaudioecho test — "parity unpinned" by
 * reference vectors; pinned by source semantics only.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct oracle_echo {
  double *buffer;
  size_t size;
  size_t pos;
} oracle_echo;

/* imp.rs:250-251: size = (max_delay * rate).seconds(); buffer_size = size * channels */
size_t oracle_echo_ring_len(uint64_t max_delay_ns, uint32_t rate, uint32_t channels) {
  return (size_t)((max_delay_ns * (uint64_t)rate) / 1000000000ull) * (size_t)channels;
}
/* imp.rs:207 + :74-77: delay = min(max_delay, delay); delay_frames = (delay*channels*rate).seconds() */
size_t oracle_echo_delay_samples(uint64_t delay_ns, uint64_t max_delay_ns, uint32_t rate, uint32_t channels) {
  uint64_t d = delay_ns < max_delay_ns ? delay_ns : max_delay_ns;
  return (size_t)((d * (uint64_t)channels * (uint64_t)rate) / 1000000000ull);
}

oracle_echo *oracle_echo_new(size_t size) {
  oracle_echo *e = (oracle_echo *)calloc(1, sizeof *e);
  e->buffer = (double *)calloc(size ? size : 1, sizeof(double));
  e->size = size;
  return e;
}
void oracle_echo_free(oracle_echo *e) { if (e) { free(e->buffer); free(e); } }
const double *oracle_echo_ring(const oracle_echo *e) { return e->buffer; }
size_t oracle_echo_pos(const oracle_echo *e) { return e->pos; }

#define ECHO_BODY(T)                                                              \
  size_t size = e->size;                                                          \
  if (size == 0 || delay > size) return -1; /* assert!(size >= delay); assert_ne!(size, 0) */ \
  size_t read_pos = (size - delay + e->pos) % size;                               \
  size_t write_pos = e->pos % size;                                               \
  for (size_t i = 0; i < n; i++) {                                                \
    double ev = e->buffer[read_pos];                                              \
    double inp = (double)data[i];                                                 \
    double out = inp + intensity * ev;                                            \
    e->buffer[write_pos] = inp + feedback * ev;                                   \
    data[i] = (T)out;                                                             \
    write_pos = (write_pos + 1) % size;                                           \
    read_pos = (read_pos + 1) % size;                                             \
  }                                                                               \
  e->pos = write_pos;                                                             \
  return 0;

int oracle_echo_process_f32(oracle_echo *e, float *data, size_t n, size_t delay, double intensity, double feedback) { ECHO_BODY(float) }
int oracle_echo_process_f64(oracle_echo *e, double *data, size_t n, size_t delay, double intensity, double feedback) { ECHO_BODY(double) }
