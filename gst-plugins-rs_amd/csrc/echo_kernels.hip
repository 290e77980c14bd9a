// echo_kernels.hip — gfx950 kernels for rsaudioecho.
//
// Reference loop replaced: AudioEcho::process (audio/audiofx/src/audioecho/imp.rs:69-85) zipped
// with RingBufferIter (audio/audiofx/src/audioecho/ring_buffer.rs:37-82):
//     e = ring[read]; out = inp + intensity*e; ring[write] = inp + feedback*e; data = out as F
// with read = write - delay (mod size), all in f64, unfused.
//
// Parallel form. Let W[i] be the value sample i of this buffer writes into the ring and
// D = delay, or `size` when delay is 0 or size (read == write index: the slot read is the one
// written `size` samples earlier, ring_buffer.rs:44-45). Then e_i = W[i-D] for i >= D and
// e_i = ring[(pos + i + size - D) % size] (history) for i < D. The only dependency is i -> i-D:
//   feedback == 0 : W[i] = inp[i]; every sample is independent            (3 parallel launches)
//   feedback != 0 : comb filter; lane t owns the chain t, t+D, t+2D, ...   (D independent chains)
// W is staged in a scratch array and committed to the ring afterwards, so no lane ever reads a
// ring slot another lane has already overwritten (the write-after-read hazard of updating in place).
// Arithmetic is plain f64 mul/add (-ffp-contract=off), i.e. bit-identical to the reference.
#include "internal.hpp"

namespace mi355 {

template <typename T>
__global__ __launch_bounds__(256) void echo_widen_kernel(const T *__restrict__ data, double *__restrict__ w, size_t n) {
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) w[i] = (double)data[i];
}

// feedback == 0: w[] holds the widened input (== what the ring receives).
template <typename T>
__global__ __launch_bounds__(256) void echo_nofb_kernel(T *__restrict__ data, const double *__restrict__ w,
                                                        const double *__restrict__ ring, size_t n, size_t size,
                                                        size_t pos, size_t D, double intensity) {
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
    const double e = (i < D) ? ring[(pos + i + size - D) % size] : w[i - D];
    const double out = w[i] + intensity * e;
    data[i] = (T)out;
  }
}

// feedback != 0: one lane per residue class mod D.
template <typename T>
__global__ __launch_bounds__(256) void echo_chain_kernel(T *__restrict__ data, double *__restrict__ w,
                                                         const double *__restrict__ ring, size_t n, size_t size,
                                                         size_t pos, size_t D, double intensity, double feedback) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= D || t >= n) return;
  double e = ring[(pos + t + size - D) % size];
  for (size_t i = t; i < n; i += D) {
    const double inp = (double)data[i];
    const double out = inp + intensity * e;
    const double wv = inp + feedback * e;
    data[i] = (T)out;
    w[i] = wv;
    e = wv;
  }
}

// ring[(pos + i) % size] = W[i] for the last min(n,size) samples (earlier ones are overwritten anyway).
__global__ __launch_bounds__(256) void echo_commit_kernel(double *__restrict__ ring, const double *__restrict__ w,
                                                          size_t n, size_t size, size_t pos) {
  const size_t first = n > size ? n - size : 0;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs)
    ring[(pos + i) % size] = w[i];
}

static unsigned blocks_for(size_t n, int n_cu) {
  size_t b = (n + 255) / 256;
  const size_t cap = (size_t)n_cu * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T>
static int echo_run(mi355_ctx *ctx, T *d_data, size_t n, size_t delay, double intensity, double feedback) {
  EchoDevice &E = ctx->echo;
  const size_t size = E.ring_len;
  const size_t D = (delay == 0) ? size : delay;
  // scratch W[n] lives in staging slot 1
  const size_t need = n * sizeof(double);
  if (ctx->d_stage_bytes[1] < need) {
    if (ctx->d_stage[1]) (void)hipFree(ctx->d_stage[1]);
    ctx->d_stage[1] = nullptr;
    ctx->d_stage_bytes[1] = 0;
    int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[1], need), "hipMalloc(echo scratch)");
    if (rc) return rc;
    ctx->d_stage_bytes[1] = need;
  }
  double *w = (double *)ctx->d_stage[1];
  const unsigned gb = blocks_for(n, ctx->n_cu);
  if (feedback == 0.0) {
    hipLaunchKernelGGL((echo_widen_kernel<T>), dim3(gb), dim3(256), 0, ctx->stream, (const T *)d_data, w, n);
    hipLaunchKernelGGL((echo_nofb_kernel<T>), dim3(gb), dim3(256), 0, ctx->stream, d_data, (const double *)w,
                       (const double *)E.d_ring, n, size, E.pos, D, intensity);
  } else {
    const size_t chains = D < n ? D : n;
    const unsigned cb = (unsigned)((chains + 255) / 256);
    hipLaunchKernelGGL((echo_chain_kernel<T>), dim3(cb), dim3(256), 0, ctx->stream, d_data, w,
                       (const double *)E.d_ring, n, size, E.pos, D, intensity, feedback);
  }
  hipLaunchKernelGGL(echo_commit_kernel, dim3(gb), dim3(256), 0, ctx->stream, E.d_ring, (const double *)w, n, size, E.pos);
  int rc = check_hip(ctx, hipGetLastError(), "echo kernel launch");
  if (rc) return rc;
  E.pos = (E.pos + n) % size;  // RingBufferIter::drop (ring_buffer.rs:78-82)
  return MI355_OK;
}

int launch_echo(mi355_ctx *ctx, void *d_data, size_t n, int is_f64, size_t delay, double intensity, double feedback) {
  EchoDevice &E = ctx->echo;
  if (!E.configured) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "rsaudioecho: not negotiated (setup not called)");
  // RingBufferIter::new: assert!(size >= delay); assert_ne!(size, 0) (ring_buffer.rs:41-42)
  if (E.ring_len == 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: ring buffer size is 0");
  if (delay > E.ring_len) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: delay exceeds ring buffer size");
  if (n == 0) return MI355_OK;
  return is_f64 ? echo_run<double>(ctx, (double *)d_data, n, delay, intensity, feedback)
                : echo_run<float>(ctx, (float *)d_data, n, delay, intensity, feedback);
}

}  // namespace mi355
