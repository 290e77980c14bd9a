// tools/valu_bench.hip — VALU/LDS issue-rate microbenchmark for gfx950 (development tool, not product).
// Each kernel runs ITER iterations of 16 independent instructions of one kind per wave; we report
// lane-ops per clock per CU (64 lanes x instructions / cycles), using s_memtime-free wall timing via
// hipEvents and the nominal clock reported by the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 4096

#define DEF_KERNEL(NAME, BODY)                                                         \
  __global__ __launch_bounds__(256) void NAME(float *out, float a, float b) {          \
    float r0 = threadIdx.x * a, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f;            \
    float r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;                   \
    float s0 = r0 * b, s1 = r1 * b, s2 = r2 * b, s3 = r3 * b;                           \
    float s4 = r4 * b, s5 = r5 * b, s6 = r6 * b, s7 = r7 * b;                           \
    unsigned two = 2; (void)two; asm volatile("s_mov_b64 s[60:61], 0x5555aaaa" ::: "s60", "s61"); \
    for (int i = 0; i < ITERS; i++) {                                                  \
      BODY                                                                             \
    }                                                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7; \
  }

#define OP2(INS, D, A, B) asm volatile(INS " %0, %1, %2" : "=v"(D) : "v"(A), "v"(B));
#define OP2S(INS, D, A) asm volatile(INS " %0, %0, %1" : "+v"(D) : "v"(A));
#define X16(M) M(r0) M(r1) M(r2) M(r3) M(r4) M(r5) M(r6) M(r7) M(s0) M(s1) M(s2) M(s3) M(s4) M(s5) M(s6) M(s7)

#define M_MUL(R) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(R) : "v"(a));
#define M_ADD(R) asm volatile("v_add_f32 %0, %0, %1" : "+v"(R) : "v"(a));
#define M_FMA(R) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_CND(R) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(R) : "v"(a));
#define M_RCP(R) asm volatile("v_rcp_f32 %0, %0" : "+v"(R));
#define M_CVTU(R) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(R));
#define M_CVTB(R) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(R));
#define M_PERM(R) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_FRACT(R) asm volatile("v_fract_f32 %0, %0" : "+v"(R));
#define M_MAX(R) asm volatile("v_max_f32 %0, %0, %1" : "+v"(R) : "v"(a));
#define M_MED3(R) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_CMP(R) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(R), "v"(a) : "vcc");
#define M_LSHL(R) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(R));
#define M_MADU(R) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_SUB(R) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(R) : "v"(a));

#define M_CNDS(R) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[60:61]" : "+v"(R) : "v"(a));
#define M_FLOOR(R) asm volatile("v_floor_f32 %0, %0" : "+v"(R));
#define M_CVTPK(R) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(R) : "v"(a));
#define M_MIN3(R) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_BFE(R) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(R));
#define M_LSHLOR(R) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(R) : "v"(a));
#define M_ADDCL(R) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(R) : "v"(a));
#define M_FMAC(R) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_CMPS(R) asm volatile("v_cmp_lt_f32_e64 s[62:63], %0, %1" : : "v"(R), "v"(a) : "s62", "s63");
#define M_SDWASH(R) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(R) : "v"(two));
#define M_CMPSDWA(R) asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %0 src0_sel:BYTE_0 src1_sel:BYTE_1" : : "v"(R) : "vcc");
#define M_ANDOR(R) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(R) : "v"(a), "v"(b));
#define M_ASHR(R) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(R));
#define M_CVTF(R) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(R));
#define M_RNDNE(R) asm volatile("v_rndne_f32 %0, %0" : "+v"(R));
#define M_MULI(R) asm volatile("v_mul_f32 %0, 0x437f0000, %0" : "+v"(R));
DEF_KERNEL(k_mul, X16(M_MUL))
DEF_KERNEL(k_sub, X16(M_SUB))
DEF_KERNEL(k_floor, X16(M_FLOOR))
DEF_KERNEL(k_cvtpk, X16(M_CVTPK))
DEF_KERNEL(k_min3, X16(M_MIN3))
DEF_KERNEL(k_bfe, X16(M_BFE))
DEF_KERNEL(k_lshlor, X16(M_LSHLOR))
DEF_KERNEL(k_addcl, X16(M_ADDCL))
DEF_KERNEL(k_fmac, X16(M_FMAC))
DEF_KERNEL(k_andor, X16(M_ANDOR))
DEF_KERNEL(k_ashr, X16(M_ASHR))
DEF_KERNEL(k_cvtf, X16(M_CVTF))
DEF_KERNEL(k_rndne, X16(M_RNDNE))
DEF_KERNEL(k_muli, X16(M_MULI))
DEF_KERNEL(k_cmpsdwa, X16(M_CMPSDWA))
DEF_KERNEL(k_cnds, X16(M_CNDS))
DEF_KERNEL(k_cmps, X16(M_CMPS))
DEF_KERNEL(k_sdwash, X16(M_SDWASH))
DEF_KERNEL(k_add, X16(M_ADD))
DEF_KERNEL(k_fma, X16(M_FMA))
DEF_KERNEL(k_cnd, X16(M_CND))
DEF_KERNEL(k_rcp, X16(M_RCP))
DEF_KERNEL(k_cvtu, X16(M_CVTU))
DEF_KERNEL(k_cvtb, X16(M_CVTB))
DEF_KERNEL(k_perm, X16(M_PERM))
DEF_KERNEL(k_fract, X16(M_FRACT))
DEF_KERNEL(k_max, X16(M_MAX))
DEF_KERNEL(k_med3, X16(M_MED3))
DEF_KERNEL(k_cmp, X16(M_CMP))
DEF_KERNEL(k_lshl, X16(M_LSHL))
DEF_KERNEL(k_madu, X16(M_MADU))

// packed: 8 register pairs
#define DEF_PK(NAME, INS3)                                                                  \
  __global__ __launch_bounds__(256) void NAME(float *out, float a, float b) {               \
    typedef float f2 __attribute__((ext_vector_type(2)));                                   \
    f2 r0 = {threadIdx.x * a, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f;             \
    f2 r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;                           \
    f2 av = {a, b}, bv = {b, a};                                                            \
    for (int i = 0; i < ITERS; i++) {                                                       \
      INS3(r0) INS3(r1) INS3(r2) INS3(r3) INS3(r4) INS3(r5) INS3(r6) INS3(r7)               \
      INS3(r0) INS3(r1) INS3(r2) INS3(r3) INS3(r4) INS3(r5) INS3(r6) INS3(r7)               \
    }                                                                                       \
    f2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                 \
  }
#define P_MUL(R) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(R) : "v"(av));
#define P_ADD(R) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(R) : "v"(av));
#define P_FMA(R) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(R) : "v"(av), "v"(bv));
DEF_PK(k_pkmul, P_MUL)
DEF_PK(k_pkadd, P_ADD)
DEF_PK(k_pkfma, P_FMA)

// mixed dual-issue probe: alternate mul and cndmask / mul and pk_mul
#define M_MIX(R) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(R) : "v"(a), "v"(b));
DEF_KERNEL(k_muladd, X16(M_MIX))

// LDS read rates
__global__ __launch_bounds__(256) void k_ldsr32(float *out, float a, float b) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * a;
  __syncthreads();
  float acc = 0.f;
  const float *p = lds + threadIdx.x;
  for (int i = 0; i < ITERS; i++) {
    float v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n"
                 "ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"((unsigned)(size_t)p));
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + b;
}
__global__ __launch_bounds__(256) void k_ldsr2x32(float *out, float a, float b) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * a;
  __syncthreads();
  float acc = 0.f;
  const float *p = lds + threadIdx.x;
  for (int i = 0; i < ITERS; i++) {
    float2 v0, v1, v2, v3;
    asm volatile("ds_read2_b32 %0, %4 offset1:33\n ds_read2_b32 %1, %4 offset0:64 offset1:97\n ds_read2_b32 %2, %4 offset0:128 offset1:161\n ds_read2_b32 %3, %4 offset0:192 offset1:225\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"((unsigned)(size_t)p));
    acc += v0.x + v1.x + v2.x + v3.x + v0.y + v1.y + v2.y + v3.y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + b;
}

template <typename K>
static double run(const char *name, K kern, int insts_per_iter, double lanes_per_inst, int clock_khz, int n_cu, int waves_per_simd) {
  float *d;
  const int blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double total_inst = (double)blocks * 4 /*waves*/ * ITERS * insts_per_iter;
  const double cycles = ms * 1e-3 * clock_khz * 1e3;
  const double per_cu_per_clk = total_inst * lanes_per_inst / cycles / n_cu;
  printf("%-10s waves/SIMD=%d  %8.3f ms  wave-instr/clk/CU=%.3f  lane-results/clk/CU=%.1f (at nominal %d MHz)\n", name, waves_per_simd, ms,
         total_inst / cycles / n_cu, per_cu_per_clk, clock_khz / 1000);
  hipFree(d);
  return per_cu_per_clk;
}

__global__ void k_clock(unsigned long long *out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < 2000000; i++) asm volatile("v_mul_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(x));
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (unsigned long long)x; }
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int cu = p.multiProcessorCount, clk = p.clockRate;
  {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_clock, dim3(cu * 8), dim3(256), 0, 0, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    int wc = 0; hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    printf("clock probe: shader cycles=%llu wall ticks=%llu (wall rate %d kHz) event ms=%.3f => shader MHz (vs wall)=%.1f, (vs event)=%.1f\n", h[0], h[1], wc, ms,
           (double)h[0] / ((double)h[1] / wc) / 1e3, (double)h[0] / (ms * 1e3));
  }
  for (int w : {1, 2, 4}) {
    run("mul", k_mul, 16, 64, clk, cu, w);
    run("add", k_add, 16, 64, clk, cu, w);
    run("fma", k_fma, 16, 64, clk, cu, w);
    run("mul+add", k_muladd, 32, 64, clk, cu, w);
    run("cndmask", k_cnd, 16, 64, clk, cu, w);
    run("rcp", k_rcp, 16, 64, clk, cu, w);
    run("sub", k_sub, 16, 64, clk, cu, w);
    run("cnd_sgpr", k_cnds, 16, 64, clk, cu, w);
    run("cmp_sgpr", k_cmps, 16, 64, clk, cu, w);
    run("cmp_sdwa", k_cmpsdwa, 16, 64, clk, cu, w);
    run("sdwa_shl", k_sdwash, 16, 64, clk, cu, w);
    run("floor", k_floor, 16, 64, clk, cu, w);
    run("cvt_pk_u8", k_cvtpk, 16, 64, clk, cu, w);
    run("min3", k_min3, 16, 64, clk, cu, w);
    run("bfe", k_bfe, 16, 64, clk, cu, w);
    run("lshl_or", k_lshlor, 16, 64, clk, cu, w);
    run("add_clamp", k_addcl, 16, 64, clk, cu, w);
    run("fmac", k_fmac, 16, 64, clk, cu, w);
    run("and_or", k_andor, 16, 64, clk, cu, w);
    run("ashr", k_ashr, 16, 64, clk, cu, w);
    run("cvt_f32_u32", k_cvtf, 16, 64, clk, cu, w);
    run("rndne", k_rndne, 16, 64, clk, cu, w);
    run("mul_lit", k_muli, 16, 64, clk, cu, w);
    run("cvt_u32", k_cvtu, 16, 64, clk, cu, w);
    run("cvt_ubyte", k_cvtb, 16, 64, clk, cu, w);
    run("perm", k_perm, 16, 64, clk, cu, w);
    run("fract", k_fract, 16, 64, clk, cu, w);
    run("max", k_max, 16, 64, clk, cu, w);
    run("med3", k_med3, 16, 64, clk, cu, w);
    run("cmp", k_cmp, 16, 64, clk, cu, w);
    run("lshl", k_lshl, 16, 64, clk, cu, w);
    run("mad_u24", k_madu, 16, 64, clk, cu, w);
    run("pk_mul", k_pkmul, 16, 128, clk, cu, w);
    run("pk_add", k_pkadd, 16, 128, clk, cu, w);
    run("pk_fma", k_pkfma, 16, 128, clk, cu, w);
    run("ds_r_b32", k_ldsr32, 8, 64, clk, cu, w);
    run("ds_r2_b32", k_ldsr2x32, 4, 128, clk, cu, w);
  }
  return 0;
}
