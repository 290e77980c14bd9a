// lds_pair_probe.hip — is an 8-byte-aligned ds_write_b64 seen whole by ds_read_b64 / ds_read_b128 of OTHER waves on gfx950?
// colorlut_tagged_kernel (csrc/colorlut_window.hip) rests on it: a cache entry is the pair {colour, value} written with one
// ds_write_b64; a reader must never see the colour of one write with the value of another. Here half of each block's waves
// write pairs {x, f(x)} to pseudo-random 8-byte slots of a small LDS array (so that slots are rewritten constantly) while
// the other half read them with ds_read_b128 (both halves of a 16-byte set) or ds_read_b64 and count pairs that are not
// {x, f(x)}. Build: hipcc --offload-arch=gfx950 -O3 tools/lds_pair_probe.hip -o tools/lds_pair_probe ; prints the torn-pair count.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef volatile __attribute__((address_space(3))) u4_t lds_vu4;
typedef volatile __attribute__((address_space(3))) u2_t lds_vu2;

__device__ __forceinline__ uint32_t f(uint32_t x) { return (x * 2654435761u) ^ 0x5bd1e995u; }
__device__ __forceinline__ uint32_t rnd(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int READ128>
__global__ __launch_bounds__(1024) void probe(unsigned iters, unsigned sets, unsigned long long *out) {
  extern __shared__ unsigned char dyn[];
  const unsigned wave = threadIdx.x >> 6;
  for (unsigned i = threadIdx.x; i < sets * 2; i += 1024) {
    const u2_t z = {0u, f(0u)};
    *(lds_vu2 *)(lds_byte *)(uintptr_t)(8u * i) = z;
  }
  __syncthreads();
  uint32_t s = 0x9e3779b9u * (blockIdx.x * 1024u + threadIdx.x + 1u);
  unsigned long long torn = 0, seen = 0;
  if (wave & 1) {
    for (unsigned it = 0; it < iters; it++) {
      const uint32_t x = rnd(s), slot = rnd(s) % (sets * 2);
      const u2_t pr = {x, f(x)};
      *(lds_vu2 *)(lds_byte *)(uintptr_t)(8u * slot) = pr;
    }
  } else {
    for (unsigned it = 0; it < iters; it++) {
      if (READ128) {
        const uint32_t set = rnd(s) % sets;
        const u4_t e = *(lds_vu4 *)(lds_byte *)(uintptr_t)(16u * set);
        torn += (e.y != f(e.x)) + (e.w != f(e.z));
        seen += 2;
      } else {
        const uint32_t slot = rnd(s) % (sets * 2);
        const u2_t e = *(lds_vu2 *)(lds_byte *)(uintptr_t)(8u * slot);
        torn += e.y != f(e.x);
        seen += 1;
      }
    }
  }
  atomicAdd(out, torn);
  atomicAdd(out + 1, seen);
}

int main() {
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  for (int r128 = 0; r128 < 2; r128++)
    for (unsigned sets : {64u, 1024u, 8192u}) {
      hipMemset(d, 0, 16);
      if (r128) hipLaunchKernelGGL(probe<1>, dim3(512), dim3(1024), sets * 16, 0, 20000u, sets, d);
      else hipLaunchKernelGGL(probe<0>, dim3(512), dim3(1024), sets * 16, 0, 20000u, sets, d);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("ds_write_b64 vs %s, %5u sets: %llu pairs read, %llu torn\n", r128 ? "ds_read_b128" : "ds_read_b64 ", sets, h[1], h[0]);
    }
  return 0;
}
