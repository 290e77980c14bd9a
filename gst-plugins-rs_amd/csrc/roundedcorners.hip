// roundedcorners.hip — the device side of `roundedcorners` (video/videofx/src/border/imp.rs).
//
// The reference has no per-pixel work per buffer: generate_alpha_mask / draw_rounded_corners (imp.rs:57-180) render ONE A8
// plane per caps / radius change (cairo: four arcs, antialiased fill + 1 px stroke), and prepare_output_buffer (imp.rs:482-559)
// appends that one shared GstMemory to every buffer as plane 3 of A420 (stride[3] x round_up_2(height) bytes, imp.rs:469-470).
// What the bytes ARE is cairo's business and stays on the host (host/elements.cpp renders them through the system libcairo);
// what this file adds is what a device-resident I420 stream needs to get its alpha plane without a host trip:
//   * mi355_roundedcorners_set_mask: the host-rendered plane, uploaded once and kept in HBM - the device twin of the
//     reference's `alpha_mem` (a device GstMemory wraps mi355_roundedcorners_mask_device and is appended by reference, as the
//     reference appends its memory: no bytes move per buffer);
//   * mi355_roundedcorners_append_device: for consumers that want the A420 frame contiguous, plane 3 written behind the
//     I420 planes of every frame of a batch - one launch, 16-byte vectors where the addresses allow it.
// Integer bytes copied: bit-exact by construction; tests/test_gpu_roundedcorners.py compares with the cairo goldens.
#include "internal.hpp"

namespace mi355 {

struct RoundedMask {
  uint8_t *d = nullptr;
  size_t bytes = 0;
  int width = 0, height = 0, stride = 0;
};

namespace {

// every frame f gets the mask at dst + f * pitch: V bytes per lane and step (V = 16 / 4 / 1 by what the addresses allow)
template <typename T>
__global__ __launch_bounds__(256) void plane_append_kernel(const T *__restrict__ mask, uint8_t *__restrict__ dst, size_t pitch, size_t n_vec, int n_frames) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, total = n_vec * (size_t)n_frames;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t f = i / n_vec, k = i - f * n_vec;
    __builtin_nontemporal_store(mask[k], (T *)(dst + f * pitch) + k);
  }
}

}  // namespace

void roundedcorners_release(mi355_ctx *ctx) {
  auto *m = static_cast<RoundedMask *>(ctx->rounded);
  if (!m) return;
  if (m->d) (void)hipFree(m->d);
  delete m;
  ctx->rounded = nullptr;
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_roundedcorners_set_mask(mi355_ctx *ctx, const uint8_t *mask, int width, int height, int stride) {
  if (!ctx) return MI355_ERR_INVALID_ARG;
  int rc = check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  if (rc) return rc;
  if (!mask) {  // passthrough (I420 out): no plane
    // (the plane may still be read by launches on the stream: they were enqueued before this call, hipFree waits for them)
    roundedcorners_release(ctx);
    return MI355_OK;
  }
  if (width <= 0 || height <= 0 || stride < width) return set_error(ctx, MI355_ERR_INVALID_ARG, "roundedcorners: bad mask geometry");
  const size_t rows = ((size_t)height + 1) & ~(size_t)1;  // round_up_2 (imp.rs:469)
  const size_t bytes = (size_t)stride * rows;
  auto *m = static_cast<RoundedMask *>(ctx->rounded);
  if (!m) ctx->rounded = m = new RoundedMask();
  if (m->bytes != bytes) {
    if (m->d) (void)hipFree(m->d);
    m->d = nullptr;
    m->bytes = 0;
    if ((rc = check_hip(ctx, hipMalloc((void **)&m->d, (bytes + 15) & ~(size_t)15), "hipMalloc(roundedcorners mask)"))) return rc;
    m->bytes = bytes;
  }
  m->width = width; m->height = height; m->stride = stride;
  // in stream order behind the launches that still read the previous mask; the caller's buffer is read before this returns
  if ((rc = check_hip(ctx, hipMemcpyAsync(m->d, mask, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(roundedcorners mask)"))) return rc;
  __atomic_fetch_add(&ctx->n_h2d, 1ull, __ATOMIC_RELAXED);
  return check_hip(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int mi355_roundedcorners_mask_device(mi355_ctx *ctx, const uint8_t **d_mask, size_t *size, int *stride) {
  if (!ctx) return MI355_ERR_INVALID_ARG;
  auto *m = static_cast<RoundedMask *>(ctx->rounded);
  if (!m || !m->d) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "roundedcorners: no mask set");
  if (d_mask) *d_mask = m->d;
  if (size) *size = m->bytes;
  if (stride) *stride = m->stride;
  return MI355_OK;
}

int mi355_roundedcorners_append_device(mi355_ctx *ctx, uint8_t *d_frames, size_t frame_pitch, size_t alpha_offset, int n_frames) {
  if (!ctx) return MI355_ERR_INVALID_ARG;
  auto *m = static_cast<RoundedMask *>(ctx->rounded);
  if (!m || !m->d) return set_error(ctx, MI355_ERR_NOT_CONFIGURED, "roundedcorners: no mask set");
  if (n_frames <= 0) return MI355_OK;
  if (!d_frames || (n_frames > 1 && frame_pitch < alpha_offset + m->bytes)) return set_error(ctx, MI355_ERR_INVALID_ARG, "roundedcorners: frames overlap their alpha planes");
  int rc = check_hip(ctx, hipSetDevice(ctx->device), "hipSetDevice");
  if (rc) return rc;
  uint8_t *dst = d_frames + alpha_offset;
  const size_t bytes = m->bytes;
  const uintptr_t align = (uintptr_t)dst | (uintptr_t)frame_pitch | (uintptr_t)bytes;
  auto grid_for = [&](size_t n_vec) {
    size_t g = (n_vec * (size_t)n_frames + 255) / 256, cap = (size_t)ctx->n_cu * 32;
    return (unsigned)(g < cap ? (g ? g : 1) : cap);
  };
  if (align % 16 == 0) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    hipLaunchKernelGGL(plane_append_kernel<v4>, dim3(grid_for(bytes / 16)), dim3(256), 0, ctx->stream, (const v4 *)m->d, dst, frame_pitch, bytes / 16, n_frames);
  } else if (align % 4 == 0) {
    hipLaunchKernelGGL(plane_append_kernel<uint32_t>, dim3(grid_for(bytes / 4)), dim3(256), 0, ctx->stream, (const uint32_t *)m->d, dst, frame_pitch, bytes / 4, n_frames);
  } else {
    hipLaunchKernelGGL(plane_append_kernel<uint8_t>, dim3(grid_for(bytes)), dim3(256), 0, ctx->stream, (const uint8_t *)m->d, dst, frame_pitch, bytes, n_frames);
  }
  return check_hip(ctx, hipGetLastError(), "roundedcorners append launch");
}

}  // extern "C"
