// sofa_kernels.hip — gfx950 kernels for `sofalizer`: uniformly partitioned FFT convolution of every input channel with
// its head-related impulse-response pair, mixed to stereo.
//
// Reference path replaced: the per-block loop of Sofalizer::process (audio/hrtf/src/sofa/imp.rs:234-300): per input block
// of `block-length` frames and per channel that is not dropped (LFE1 / LFE2 are ChannelProcessor::Drop, :808-821) the
// channel is de-interleaved (:250-256), run through the third-party crate sofar's `Renderer::process_block` (built with
// `partition-length`, :793-798; the block length must be a multiple of it, :775-781) and mixed
// `out[2i] += l * gain; out[2i+1] += r * gain` in channel order (:282-297). Which HRIR pair a channel uses is decided on
// the host (Sofar::filter lookup in the SOFA file, State::update_filters :129-160) and handed over with
// mi355_sofa_set_filter: reading SOFA/HDF5 files is not part of the per-buffer path.
// sofar's sources are not in the reference tree (Cargo dependency `sofar`, features "dsp"): PARITY UNPINNED. What a
// uniformly partitioned convolver computes is a streaming linear convolution, y = x * h per ear, with h changing at block
// boundaries; that is the contract here, checked against a time-domain oracle (oracle/oracle.py: SofaRenderer) within 2e-6 of
// full scale (f32 FFT round-off), not bit for bit.
//
// Algorithm (uniformly partitioned overlap-save, partition P, FFT size N = 2P, K = ceil(L / P) filter partitions):
//   per sub-block j of P input samples:  X_j = FFT([x_{j-1} | x_j]);  Y = sum_k X_{j-k} . H_k;  y_j = IFFT(Y)[P .. 2P)
// One workgroup per channel walks the B / P sub-blocks of a block: the 2P-point transforms run in LDS (radix-2, all lanes
// of the workgroup on N/2 butterflies per stage, twiddles from an LDS table), the frequency-domain delay line (K spectra
// per channel) lives in global memory (L2-resident: K * N * 8 B per channel), the two ears share X.
#include "internal.hpp"

#include <cmath>
#include <cstring>
#include <vector>

namespace mi355 {

struct SofaState {
  int channels = 0, filter_len = 0, P = 0, B = 0, K = 0, N = 0, logN = 0;
  float2 *d_H = nullptr;     // [C][2][K][N] filter partition spectra
  float2 *d_fdl = nullptr;   // [C][K][N] spectra of the last K input windows (ring, slot = sub-block counter mod K)
  float *d_prev = nullptr;   // [C][P] previous sub-block of every channel
  float *d_partial = nullptr;  // [C][B][2]
  float *d_in = nullptr, *d_out = nullptr;  // staging for the host entry point
  float *d_gain = nullptr;   // [C]
  float *d_taps = nullptr;   // [2][K*P] staging for set_filter
  int *d_drop = nullptr;     // [C]
  std::vector<int> drop;
  std::vector<unsigned char> have_filter;
  unsigned long long counter = 0;  // sub-blocks processed (FDL write slot = counter mod K)
};
static SofaState *sofa_of(mi355_ctx *ctx) { return (SofaState *)ctx->sofa; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// In-place radix-2 decimation-in-time FFT of buf[0..N) in LDS by all lanes of the workgroup. tw[k] = exp(-2 pi i k / N),
// k < N/2. INVERSE: conjugated twiddles, no scaling. The caller has stored the input in bit-reversed order.
template <bool INVERSE>
__device__ __forceinline__ void fft_lds(float2 *buf, const float2 *tw, int N, int logN) {
  for (int s = 1; s <= logN; s++) {
    const int half = 1 << (s - 1);
    __syncthreads();
    for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
      const int j = t & (half - 1), base = (t >> (s - 1)) << s;
      float2 w = tw[j << (logN - s)];
      if (INVERSE) w.y = -w.y;
      const float2 a = buf[base + j], b = cmul(buf[base + j + half], w);
      buf[base + j] = make_float2(a.x + b.x, a.y + b.y);
      buf[base + j + half] = make_float2(a.x - b.x, a.y - b.y);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

// spectra of one channel's filter partitions: grid (K, 2 ears), taps[ear][K*P] zero padded
__global__ __launch_bounds__(256) void sofa_filter_fft_kernel(const float *__restrict__ taps, float2 *__restrict__ H, int P, int K, int N, int logN) {
  extern __shared__ float2 sm[];  // [N] buffer + [N/2] twiddles
  float2 *buf = sm, *tw = sm + N;
  const int k = blockIdx.x, ear = blockIdx.y;
  for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {
    float s, c;
    sincospif(-2.0f * (float)i / (float)N, &s, &c);
    tw[i] = make_float2(c, s);
  }
  for (int i = threadIdx.x; i < N; i += blockDim.x) buf[bitrev(i, logN)] = make_float2(i < P ? taps[(size_t)ear * K * P + (size_t)k * P + i] : 0.0f, 0.0f);
  fft_lds<false>(buf, tw, N, logN);
  for (int i = threadIdx.x; i < N; i += blockDim.x) H[((size_t)ear * K + k) * N + i] = buf[i];
}

// one workgroup per channel: the B / P sub-blocks of one input block
__global__ __launch_bounds__(256) void sofa_convolve_kernel(const float *__restrict__ in, int C, const float2 *__restrict__ H, float2 *__restrict__ fdl,
                                                            float *__restrict__ prev, float *__restrict__ partial, const int *__restrict__ drop,
                                                            int P, int B, int K, int N, int logN, unsigned slot0) {
  extern __shared__ float2 sm[];  // [N] X, [N] Yl, [N] Yr, [N/2] twiddles
  float2 *X = sm, *Yl = sm + N, *Yr = sm + 2 * N, *tw = sm + 3 * N;
  const int c = blockIdx.x;
  float *out = partial + (size_t)c * B * 2;
  if (drop[c]) {  // ChannelProcessor::Drop contributes nothing (sofa/imp.rs:244-246)
    for (int i = threadIdx.x; i < 2 * B; i += blockDim.x) out[i] = 0.0f;
    return;
  }
  for (int i = threadIdx.x; i < N / 2; i += blockDim.x) {
    float s, co;
    sincospif(-2.0f * (float)i / (float)N, &s, &co);
    tw[i] = make_float2(co, s);
  }
  const float2 *Hl = H + (size_t)c * 2 * K * N, *Hr = Hl + (size_t)K * N;
  float2 *F = fdl + (size_t)c * K * N;
  float *pv = prev + (size_t)c * P;
  const float inv_n = 1.0f / (float)N;
  for (int j = 0; j < B / P; j++) {
    const unsigned slot = (slot0 + (unsigned)j) % (unsigned)K;
    __syncthreads();
    // window [previous sub-block | this sub-block], stored bit-reversed for the in-place transform
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      const float v = i < P ? pv[i] : in[(size_t)(j * P + i - P) * C + c];  // de-interleave (sofa/imp.rs:250-256)
      X[bitrev(i, logN)] = make_float2(v, 0.0f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) pv[i] = in[(size_t)(j * P + i) * C + c];
    fft_lds<false>(X, tw, N, logN);
    for (int i = threadIdx.x; i < N; i += blockDim.x) F[(size_t)slot * N + i] = X[i];
    __syncthreads();
    // Y = sum_k X_{j-k} . H_k, partitions in ascending k (fixed order: deterministic)
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      float2 al = make_float2(0.0f, 0.0f), ar = al;
      for (int k = 0; k < K; k++) {
        const unsigned s = (slot + (unsigned)K - (unsigned)k) % (unsigned)K;
        const float2 x = k == 0 ? X[i] : F[(size_t)s * N + i];
        const float2 pl = cmul(x, Hl[(size_t)k * N + i]), pr = cmul(x, Hr[(size_t)k * N + i]);
        al.x += pl.x; al.y += pl.y; ar.x += pr.x; ar.y += pr.y;
      }
      Yl[bitrev(i, logN)] = al;
      Yr[bitrev(i, logN)] = ar;
    }
    fft_lds<true>(Yl, tw, N, logN);
    fft_lds<true>(Yr, tw, N, logN);
    for (int i = threadIdx.x; i < P; i += blockDim.x) {  // overlap-save: the last P samples are the valid ones
      out[(size_t)(j * P + i) * 2 + 0] = Yl[P + i].x * inv_n;
      out[(size_t)(j * P + i) * 2 + 1] = Yr[P + i].x * inv_n;
    }
  }
}

// out[n] = ((0 + l_0 * g_0) + l_1 * g_1) + ...   (sofa/imp.rs:207 zero fill, :282-297 accumulation in channel order)
__global__ __launch_bounds__(256) void sofa_mix_kernel(const float *__restrict__ partial, const float *__restrict__ gain, const int *__restrict__ drop,
                                                       float *__restrict__ out, int C, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * B) return;
  float acc = 0.0f;
  for (int c = 0; c < C; c++) {
    if (drop[c]) continue;
    acc += partial[(size_t)c * B * 2 + i] * gain[c];
  }
  out[i] = acc;
}

// ------------------------------------------------------------------ host side

void sofa_release(mi355_ctx *ctx) {
  SofaState *S = sofa_of(ctx);
  if (!S) return;
  void *ptrs[] = {S->d_H, S->d_fdl, S->d_prev, S->d_partial, S->d_in, S->d_out, S->d_gain, S->d_taps, S->d_drop};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete S;
  ctx->sofa = nullptr;
}

int sofa_setup(mi355_ctx *ctx, int channels, int filter_len, int partition_len, int block_len) {
  sofa_release(ctx);
  if (channels < 1 || channels > 64) return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: bad channel count");
  if (filter_len < 1 || filter_len > (1 << 20)) return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: bad filter length");
  if (partition_len < 1 || partition_len > 65535 || block_len < 1 || block_len > 65535)  // property ranges, sofa/imp.rs:384-397
    return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: partition / block length out of range");
  if (block_len % partition_len != 0) return set_error(ctx, MI355_ERR_INVALID_ARG, "Block Length is not multiple of Partition Length");  // :775-781
  if ((partition_len & (partition_len - 1)) != 0 || partition_len < 8 || partition_len > 2048)
    return set_error(ctx, MI355_ERR_UNSUPPORTED, "sofalizer: partition length must be a power of two in 8..2048 (radix-2 transforms in LDS)");
  SofaState *S = new SofaState();
  ctx->sofa = S;
  S->channels = channels; S->filter_len = filter_len; S->P = partition_len; S->B = block_len;
  S->K = (filter_len + partition_len - 1) / partition_len;
  S->N = 2 * partition_len;
  S->logN = 0;
  while ((1 << S->logN) < S->N) S->logN++;
  S->drop.assign(channels, 0);
  S->have_filter.assign(channels, 0);
  const size_t C = channels, K = S->K, N = S->N;
  int rc;
#define MI355_SOFA_ALLOC(p, bytes, what) if ((rc = check_hip(ctx, hipMalloc((void **)&(p), (bytes)), what))) return rc; \
  if ((rc = check_hip(ctx, hipMemsetAsync((p), 0, (bytes), ctx->stream), what))) return rc;
  MI355_SOFA_ALLOC(S->d_H, C * 2 * K * N * sizeof(float2), "hipMalloc(sofalizer filter spectra)")
  MI355_SOFA_ALLOC(S->d_fdl, C * K * N * sizeof(float2), "hipMalloc(sofalizer delay line)")
  MI355_SOFA_ALLOC(S->d_prev, C * S->P * sizeof(float), "hipMalloc(sofalizer history)")
  MI355_SOFA_ALLOC(S->d_partial, C * S->B * 2 * sizeof(float), "hipMalloc(sofalizer partial outputs)")
  MI355_SOFA_ALLOC(S->d_in, C * S->B * sizeof(float), "hipMalloc(sofalizer input)")
  MI355_SOFA_ALLOC(S->d_out, (size_t)S->B * 2 * sizeof(float), "hipMalloc(sofalizer output)")
  MI355_SOFA_ALLOC(S->d_gain, C * sizeof(float), "hipMalloc(sofalizer gains)")
  MI355_SOFA_ALLOC(S->d_taps, 2 * K * S->P * sizeof(float), "hipMalloc(sofalizer taps)")
  MI355_SOFA_ALLOC(S->d_drop, C * sizeof(int), "hipMalloc(sofalizer drop flags)")
#undef MI355_SOFA_ALLOC
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "sofalizer: stream synchronize");
}

// Renderer::set_filter for one channel: FIR pair of filter_len taps and whole-sample onset delays (>= 0) that are folded
// into the taps; the pair takes effect with the next block. Taps beyond filter_len after the delay are cut.
int sofa_set_filter(mi355_ctx *ctx, int channel, const float *left, const float *right, int delay_left, int delay_right) {
  SofaState *S = sofa_of(ctx);
  if (!S) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "sofalizer: not configured");
  if (channel < 0 || channel >= S->channels || !left || !right || delay_left < 0 || delay_right < 0)
    return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: bad filter argument");
  const size_t KP = (size_t)S->K * S->P;
  std::vector<float> taps(2 * KP, 0.0f);
  for (int e = 0; e < 2; e++) {
    const float *h = e ? right : left;
    const int d = e ? delay_right : delay_left;
    for (int i = 0; i + d < S->filter_len; i++) taps[e * KP + (size_t)(i + d)] = h[i];
  }
  int rc = check_hip(ctx, hipMemcpyAsync(S->d_taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(sofalizer taps)");
  if (rc) return rc;
  const size_t lds = ((size_t)S->N + S->N / 2) * sizeof(float2);
  hipLaunchKernelGGL(sofa_filter_fft_kernel, dim3(S->K, 2), dim3(256), lds, ctx->stream, (const float *)S->d_taps,
                     S->d_H + (size_t)channel * 2 * S->K * S->N, S->P, S->K, S->N, S->logN);
  if ((rc = check_hip(ctx, hipGetLastError(), "sofalizer filter kernel launch"))) return rc;
  S->have_filter[channel] = 1;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "sofalizer: stream synchronize");  // `taps` is host memory of this call
}

int sofa_set_drop(mi355_ctx *ctx, int channel, int drop) {
  SofaState *S = sofa_of(ctx);
  if (!S) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "sofalizer: not configured");
  if (channel < 0 || channel >= S->channels) return set_error(ctx, MI355_ERR_INVALID_ARG, "sofalizer: bad channel");
  S->drop[channel] = drop ? 1 : 0;
  return check_hip(ctx, hipMemcpy(S->d_drop, S->drop.data(), S->drop.size() * sizeof(int), hipMemcpyHostToDevice), "hipMemcpy(sofalizer drop flags)");
}

// ChannelProcessor::reset (flush-stop, sofa/imp.rs:840-848): the input history goes, the filters stay until the element sets new ones
int sofa_reset(mi355_ctx *ctx) {
  SofaState *S = sofa_of(ctx);
  if (!S) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "sofalizer: not configured");
  int rc = check_hip(ctx, hipMemsetAsync(S->d_fdl, 0, (size_t)S->channels * S->K * S->N * sizeof(float2), ctx->stream), "hipMemset(sofalizer delay line)");
  if (rc) return rc;
  rc = check_hip(ctx, hipMemsetAsync(S->d_prev, 0, (size_t)S->channels * S->P * sizeof(float), ctx->stream), "hipMemset(sofalizer history)");
  S->counter = 0;
  return rc;
}

int sofa_process_block_device(mi355_ctx *ctx, const float *d_in, float *d_out, const float *gains) {
  SofaState *S = sofa_of(ctx);
  if (!S) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "sofalizer: not configured");
  for (int c = 0; c < S->channels; c++)
    if (!S->drop[c] && !S->have_filter[c]) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "sofalizer: a channel has no filter yet");
  int rc = check_hip(ctx, hipMemcpyAsync(S->d_gain, gains, (size_t)S->channels * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(sofalizer gains)");
  if (rc) return rc;
  const size_t lds = (3 * (size_t)S->N + S->N / 2) * sizeof(float2);
  (void)hipFuncSetAttribute((const void *)sofa_convolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(sofa_convolve_kernel, dim3(S->channels), dim3(256), lds, ctx->stream, d_in, S->channels, (const float2 *)S->d_H, S->d_fdl, S->d_prev,
                     S->d_partial, (const int *)S->d_drop, S->P, S->B, S->K, S->N, S->logN, (unsigned)(S->counter % (unsigned long long)S->K));
  hipLaunchKernelGGL(sofa_mix_kernel, dim3((2 * S->B + 255) / 256), dim3(256), 0, ctx->stream, (const float *)S->d_partial, (const float *)S->d_gain,
                     (const int *)S->d_drop, d_out, S->channels, S->B);
  S->counter += (unsigned long long)(S->B / S->P);
  if ((rc = check_hip(ctx, hipGetLastError(), "sofalizer kernel launch"))) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "sofalizer: stream synchronize");  // `gains` is host memory of this call
}

int sofa_process_block_host(mi355_ctx *ctx, const float *in, float *out, const float *gains) {
  SofaState *S = sofa_of(ctx);
  if (!S) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "sofalizer: not configured");
  int rc = check_hip(ctx, hipMemcpyAsync(S->d_in, in, (size_t)S->channels * S->B * sizeof(float), hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(sofalizer input)");
  if (rc) return rc;
  if ((rc = sofa_process_block_device(ctx, S->d_in, S->d_out, gains))) return rc;
  if ((rc = check_hip(ctx, hipMemcpyAsync(out, S->d_out, (size_t)S->B * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(sofalizer output)"))) return rc;
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "sofalizer: stream synchronize");
}

}  // namespace mi355
