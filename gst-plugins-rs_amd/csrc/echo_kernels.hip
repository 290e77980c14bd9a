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

#include <cstring>
#include <vector>

namespace mi355 {

// Batches: blockIdx.y = stream. Stream s has its own ring (ring + s * size), its own slice of the buffer (data + s *
// stream_stride elements) and of the scratch (w + s * n), and its own {D, intensity, feedback} (device array `par`): the
// streams of a batch are independent AudioEcho instances advanced by the same number of samples per call.
struct EchoPar { unsigned long long D; double intensity, feedback; };

// Where a launch finds the per-stream parameters: up to kEchoInline streams travel in the kernel arguments (no copy, no
// event: the single-stream element path), larger batches through the device array refreshed in stream order.
constexpr int kEchoInline = 4;
struct EchoParSrc {
  const EchoPar *dev;
  EchoPar inl[kEchoInline];
  __device__ __forceinline__ EchoPar at(unsigned s) const { return dev ? dev[s] : inl[s]; }
};

template <typename T>
__global__ __launch_bounds__(256) void echo_widen_kernel(const T *__restrict__ data, double *__restrict__ w, size_t n, size_t stream_stride,
                                                         EchoParSrc par) {
  const unsigned s = blockIdx.y;
  if (par.at(s).feedback != 0.0) return;  // the chain kernel widens as it goes
  data += (size_t)s * stream_stride;
  w += (size_t)s * n;
  const size_t gs = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) w[i] = (double)data[i];
}

// feedback == 0: w[] holds the widened input (== what the ring receives): every sample is independent.
// feedback != 0: one lane per residue class mod D.
template <typename T>
__global__ __launch_bounds__(256) void echo_main_kernel(T *__restrict__ data, double *__restrict__ w, const double *__restrict__ ring, size_t n,
                                                        size_t size, size_t pos, size_t stream_stride, EchoParSrc par) {
  const unsigned s = blockIdx.y;
  const EchoPar P = par.at(s);
  const size_t D = (size_t)P.D;
  data += (size_t)s * stream_stride;
  w += (size_t)s * n;
  ring += (size_t)s * size;
  if (P.feedback == 0.0) {
    const size_t gs = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
      const double e = (i < D) ? ring[(pos + i + size - D) % size] : w[i - D];
      const double out = w[i] + P.intensity * e;
      data[i] = (T)out;
    }
  } else {
    // chains t, t + D, t + 2D, ...: grid-stride over the residue classes
    const size_t gs = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < D && t < n; t += gs) {
      double e = ring[(pos + t + size - D) % size];
      for (size_t i = t; i < n; i += D) {
        const double inp = (double)data[i];
        const double out = inp + P.intensity * e;
        const double wv = inp + P.feedback * e;
        data[i] = (T)out;
        w[i] = wv;
        e = wv;
      }
    }
  }
}

// ring[(pos + i) % size] = W[i] for the last min(n,size) samples (earlier ones are overwritten anyway).
__global__ __launch_bounds__(256) void echo_commit_kernel(double *__restrict__ ring, const double *__restrict__ w,
                                                          size_t n, size_t size, size_t pos) {
  ring += (size_t)blockIdx.y * size;
  w += (size_t)blockIdx.y * n;
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

static int ensure_echo_scratch(mi355_ctx *ctx, size_t need) {
  if (ctx->d_stage_bytes[1] >= need) return MI355_OK;
  if (ctx->d_stage[1]) (void)hipFree(ctx->d_stage[1]);
  ctx->d_stage[1] = nullptr;
  ctx->d_stage_bytes[1] = 0;
  int rc = check_hip(ctx, hipMalloc(&ctx->d_stage[1], need), "hipMalloc(echo scratch)");
  if (rc) return rc;
  ctx->d_stage_bytes[1] = need;
  return MI355_OK;
}

// d_data: n_streams slices of n samples, stream_stride elements apart. par_host: n_streams entries (delay already mapped to D).
template <typename T>
static int echo_run(mi355_ctx *ctx, T *d_data, size_t n, size_t stream_stride, const EchoPar *par_host) {
  EchoDevice &E = ctx->echo;
  const size_t size = E.ring_len;
  const unsigned S = (unsigned)E.n_streams;
  // scratch W[S][n] lives in staging slot 1
  int rc = ensure_echo_scratch(ctx, (size_t)S * n * sizeof(double));
  if (rc) return rc;
  double *w = (double *)ctx->d_stage[1];
  // parameters: pinned host copy -> device, in stream order (a call may change delay / intensity / feedback)
  EchoParSrc par;
  par.dev = nullptr;
  if (S <= (unsigned)kEchoInline) {
    for (unsigned s = 0; s < (unsigned)kEchoInline; s++) par.inl[s] = par_host[s < S ? s : 0];
  } else {
    // the previous call's parameter copy must have left the pinned block (long done in practice)
    if ((rc = check_hip(ctx, hipEventSynchronize(E.par_ev), "hipEventSynchronize(echo)"))) return rc;
    std::memcpy(E.h_par, par_host, S * sizeof(EchoPar));
    if ((rc = check_hip(ctx, hipMemcpyAsync(E.d_par, E.h_par, S * sizeof(EchoPar), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(echo parameters)"))) return rc;
    // the pinned parameter block may be rewritten by the next call once THIS COPY has been consumed: the event sits right
    // behind the copy, not behind the kernels (behind them, the next call's hipEventSynchronize would wait for the whole
    // previous buffer and serialise host and device)
    if ((rc = check_hip(ctx, hipEventRecord(E.par_ev, ctx->stream), "hipEventRecord(echo)"))) return rc;
    par.dev = (const EchoPar *)E.d_par;
    for (int s = 0; s < kEchoInline; s++) par.inl[s] = par_host[0];
  }
  bool any_nofb = false;
  size_t most = 1;  // widest parallel extent of the main kernel over the streams
  for (unsigned s = 0; s < S; s++) {
    if (par_host[s].feedback == 0.0) { any_nofb = true; most = n > most ? n : most; }
    else { const size_t chains = par_host[s].D < n ? (size_t)par_host[s].D : n; most = chains > most ? chains : most; }
  }
  const unsigned gb = blocks_for(n, ctx->n_cu), mb = blocks_for(most, ctx->n_cu);
  if (any_nofb) hipLaunchKernelGGL((echo_widen_kernel<T>), dim3(gb, S), dim3(256), 0, ctx->stream, (const T *)d_data, w, n, stream_stride, par);
  hipLaunchKernelGGL((echo_main_kernel<T>), dim3(mb, S), dim3(256), 0, ctx->stream, d_data, w, (const double *)E.d_ring, n, size, E.pos, stream_stride, par);
  hipLaunchKernelGGL(echo_commit_kernel, dim3(gb, S), dim3(256), 0, ctx->stream, E.d_ring, (const double *)w, n, size, E.pos);
  rc = check_hip(ctx, hipGetLastError(), "echo kernel launch");
  if (rc) return rc;
  E.pos = (E.pos + n) % size;  // RingBufferIter::drop (ring_buffer.rs:78-82)
  return MI355_OK;
}

int echo_setup(mi355_ctx *ctx, int n_streams, size_t ring_len) {
  (void)hipStreamSynchronize(ctx->stream);
  echo_release(ctx);
  if (n_streams < 1 || n_streams > 65535) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: 1..65535 streams per batch");
  EchoDevice &E = ctx->echo;
  const size_t cells = (size_t)n_streams * (ring_len ? ring_len : 1);
  int rc = check_hip(ctx, hipMalloc((void **)&E.d_ring, cells * sizeof(double)), "hipMalloc(echo ring)");
  if (rc) return rc;
  if ((rc = check_hip(ctx, hipMemset(E.d_ring, 0, cells * sizeof(double)), "hipMemset(echo ring)"))) return rc;
  if ((rc = check_hip(ctx, hipMalloc(&E.d_par, (size_t)n_streams * sizeof(EchoPar)), "hipMalloc(echo parameters)"))) return rc;
  if ((rc = check_hip(ctx, hipHostMalloc(&E.h_par, (size_t)n_streams * sizeof(EchoPar), hipHostMallocDefault), "hipHostMalloc(echo parameters)"))) return rc;
  if ((rc = check_hip(ctx, hipEventCreateWithFlags(&E.par_ev, hipEventDisableTiming), "hipEventCreate(echo)"))) return rc;
  E.ring_len = ring_len;
  E.n_streams = n_streams;
  E.pos = 0;
  E.configured = true;
  return MI355_OK;
}

void echo_release(mi355_ctx *ctx) {
  EchoDevice &E = ctx->echo;
  if (E.d_ring) (void)hipFree(E.d_ring);
  if (E.d_par) (void)hipFree(E.d_par);
  if (E.h_par) (void)hipHostFree(E.h_par);
  if (E.par_ev) (void)hipEventDestroy(E.par_ev);
  E = EchoDevice{};
}

int launch_echo_batch(mi355_ctx *ctx, void *d_data, size_t stream_stride, size_t n, int is_f64, const size_t *delay, const double *intensity,
                      const double *feedback) {
  EchoDevice &E = ctx->echo;
  if (!E.configured) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "rsaudioecho: not negotiated (setup not called)");
  // RingBufferIter::new: assert!(size >= delay); assert_ne!(size, 0) (ring_buffer.rs:41-42)
  if (E.ring_len == 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: ring buffer size is 0");
  for (int s = 0; s < E.n_streams; s++)
    if (delay[s] > E.ring_len) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: delay exceeds ring buffer size");
  if (n == 0) return MI355_OK;
  if (E.n_streams > 1 && stream_stride < n) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: stream stride shorter than the buffer");
  std::vector<EchoPar> par((size_t)E.n_streams);
  for (int s = 0; s < E.n_streams; s++) par[s] = EchoPar{(unsigned long long)(delay[s] == 0 ? E.ring_len : delay[s]), intensity[s], feedback[s]};
  return is_f64 ? echo_run<double>(ctx, (double *)d_data, n, stream_stride, par.data()) : echo_run<float>(ctx, (float *)d_data, n, stream_stride, par.data());
}

int launch_echo(mi355_ctx *ctx, void *d_data, size_t n, int is_f64, size_t delay, double intensity, double feedback) {
  if (ctx->echo.configured && ctx->echo.n_streams != 1) return set_error(ctx, MI355_ERR_INVALID_ARG, "rsaudioecho: set up as a batch (use the _batch entry points)");
  return launch_echo_batch(ctx, d_data, n, n, is_f64, &delay, &intensity, &feedback);
}

}  // namespace mi355
