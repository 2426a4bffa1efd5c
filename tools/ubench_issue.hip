// ubench_issue.hip -- issue-cost microbenchmarks on gfx950 that ground the kernel design choices
// (DESIGN.md section 5):  hipcc -O3 --offload-arch=gfx950 -o ubench_issue tools/ubench_issue.hip
//   * cycles per wave-instruction of v_pk_fma_f32 / v_fma_f32 / v_exp_f32 / v_rcp_f32 /
//     v_mfma_f32_16x16x4_f32 (independent and dependent chains), one wave per SIMD;
//   * the same with two waves per SIMD, and an MFMA-only wave beside a VALU-only wave on the same
//     SIMD: do f32 MFMA and f32 VALU overlap or add?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { K_PKFMA, K_FMA, K_EXP, K_RCP, K_MFMA_IND, K_MFMA_DEP, K_MAX_I32, K_MIX_SAME, K_MIX_FMA, K_MIX_MAX, K_MIX_EXP, K_MIX_FMA2, K_BF16_K16, K_BF16_K16_DEP, K_BF16_K32, K_BF16_K16_MIX4, K_BF16_K16_MIX2, K_BF16_K32_MIX4, K_BF16_K16_MIXPK2, K_BF16_K16_MIXEXP1, K_CVT, K_N };
static const char* kname[] = {"v_pk_fma_f32", "v_fma_f32", "v_exp_f32", "v_rcp_f32", "mfma16x16x4f32 indep",
                              "mfma16x16x4f32 dep", "v_max_i32", "mfma+4valu same wave"};

#define REP8(x) x x x x x x x x
#define REP4(x) x x x x

template <int KIND>
__device__ __forceinline__ void body(int iters, float seed, float* sink) {
  f2 a0 = {seed, seed}, a1 = a0, a2 = a0, a3 = a0, m = {1.0001f, 0.9999f}, c = {1e-9f, 1e-9f};
  float s0 = seed, s1 = seed, s2 = seed, s3 = seed, s4 = seed, s5 = seed, s6 = seed, s7 = seed;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, b0 = {seed, seed, seed, seed}, b1 = b0;
  for (int i = 0; i < iters; ++i) {
    if (KIND == K_PKFMA) {
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(c));)
    } else if (KIND == K_FMA) {
      REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MAX_I32) {
      REP8(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4"
                        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(m.x));)
    } else if (KIND == K_EXP) {
      REP4(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));)
    } else if (KIND == K_RCP) {
      REP4(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));)
    } else if (KIND == K_MFMA_IND) {
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                        "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MFMA_DEP) {
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n"
                        "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                        : "+v"(c0) : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MIX_SAME) {
      // per MFMA: 4 independent packed fmas in the same wave (32 MFMA + 128 VALU per iteration)
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                        : "v"(m.x), "v"(c.x), "v"(m), "v"(c));)
    } else if (KIND == K_BF16_K16) {
      REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x16_bf16 %1, %4, %5, %1\n"
                        "v_mfma_f32_16x16x16_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x16_bf16 %3, %4, %5, %3"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1));)
    } else if (KIND == K_BF16_K16_DEP) {
      REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n"
                        "v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0"
                        : "+v"(c0) : "v"(a0), "v"(a1));)
    } else if (KIND == K_BF16_K32) {
      REP8(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n"
                        "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b0), "v"(b1));)
    } else if (KIND == K_BF16_K16_MIX4) {
      REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %8, %9, %0\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %1, %8, %9, %1\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %2, %8, %9, %2\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %3, %8, %9, %3\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(a0), "v"(a1), "v"(m.x), "v"(c.x));)
    } else if (KIND == K_BF16_K16_MIX2) {
      REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %8, %9, %0\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %1, %8, %9, %1\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %2, %8, %9, %2\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %3, %8, %9, %3\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(a0), "v"(a1), "v"(m.x), "v"(c.x));)
    } else if (KIND == K_BF16_K32_MIX4) {
      REP8(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %8, %9, %0\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x32_bf16 %1, %8, %9, %1\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x32_bf16 %2, %8, %9, %2\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x32_bf16 %3, %8, %9, %3\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(b0), "v"(b1), "v"(m.x), "v"(c.x));)
    } else if (KIND == K_BF16_K16_MIXPK2) {
      REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %8, %9, %0\n v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %1, %8, %9, %1\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %2, %8, %9, %2\n v_pk_fma_f32 %4, %4, %10, %11\n v_pk_fma_f32 %5, %5, %10, %11\n"
                        "v_mfma_f32_16x16x16_bf16 %3, %8, %9, %3\n v_pk_fma_f32 %6, %6, %10, %11\n v_pk_fma_f32 %7, %7, %10, %11"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                        : "v"(b0.xy), "v"(b1.xy), "v"(m), "v"(c));)
    } else if (KIND == K_BF16_K16_MIXEXP1) {
      REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %8, %9, %0\n v_exp_f32 %4, %4\n"
                        "v_mfma_f32_16x16x16_bf16 %1, %8, %9, %1\n v_exp_f32 %5, %5\n"
                        "v_mfma_f32_16x16x16_bf16 %2, %8, %9, %2\n v_exp_f32 %6, %6\n"
                        "v_mfma_f32_16x16x16_bf16 %3, %8, %9, %3\n v_exp_f32 %7, %7"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(a0), "v"(a1));)
    } else if (KIND == K_CVT) {
      REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %4, %5\n v_cvt_pk_bf16_f32 %2, %4, %5\n v_cvt_pk_bf16_f32 %3, %4, %5"
                        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MIX_FMA) {
      // per MFMA: 4 independent NON-packed fmas in the same wave
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MIX_FMA2) {
      // per MFMA: 2 independent non-packed fmas
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                        "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                        "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MIX_MAX) {
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n"
                        "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n"
                        "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8\n"
                        "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(m.x), "v"(c.x));)
    } else if (KIND == K_MIX_EXP) {
      // per MFMA: 2 independent v_exp_f32
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                        "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                        "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                        : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                        : "v"(m.x), "v"(c.x));)
    }
  }
  float r = a0.x + a1.x + a2.x + a3.x + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + c0.x + c1.x + c2.x + c3.x;
  if (r == 12345.678f) *sink = r;
}

// kindA runs on waves 0..3 of the block (one per SIMD); kindB on waves 4..7 (the second wave of each SIMD)
template <int KA, int KB>
__global__ __launch_bounds__(512) void k(int iters, float seed, float* sink, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) body<KA>(iters, seed, sink);
  else body<KB>(iters, seed, sink);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int KA, int KB>
void run(const char* label, int threads, int instrA, int instrB) {
  float* sink;
  unsigned long long* cyc;
  hipMalloc(&sink, 4);
  hipMalloc(&cyc, 64);
  hipMemset(cyc, 0, 64);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KA, KB><<<256, threads>>>(iters, 1.0f, sink, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KA, KB><<<256, threads>>>(iters, 1.0f, sink, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-58s %8.3f ms | s_memtime ticks/instr: waveA %.2f", label, ms, (double)h[0] / ((double)iters * instrA));
  if (threads > 256) printf("  waveB %.2f", (double)h[4] / ((double)iters * instrB));
  printf("  | ns/instrA %.3f\n", ms * 1e6 / ((double)iters * instrA));
  hipFree(sink);
  hipFree(cyc);
}

int main() {
  // instructions per loop iteration: 32 for the VALU / MFMA kinds, 32 MFMA + 128 VALU for the mix
  printf("== one wave per SIMD (256 threads x 256 blocks)\n");
  run<K_PKFMA, K_PKFMA>(kname[K_PKFMA], 256, 32, 32);
  run<K_FMA, K_FMA>(kname[K_FMA], 256, 32, 32);
  run<K_MAX_I32, K_MAX_I32>(kname[K_MAX_I32], 256, 32, 32);
  run<K_EXP, K_EXP>(kname[K_EXP], 256, 32, 32);
  run<K_RCP, K_RCP>(kname[K_RCP], 256, 32, 32);
  run<K_MFMA_IND, K_MFMA_IND>(kname[K_MFMA_IND], 256, 32, 32);
  run<K_MFMA_DEP, K_MFMA_DEP>(kname[K_MFMA_DEP], 256, 32, 32);
  run<K_MIX_SAME, K_MIX_SAME>("1 mfma + 4 pk_fma interleaved (per mfma)", 256, 32, 32);
  run<K_MIX_FMA, K_MIX_FMA>("1 mfma + 4 v_fma_f32 (non-packed) interleaved (per mfma)", 256, 32, 32);
  run<K_MIX_FMA2, K_MIX_FMA2>("1 mfma + 2 v_fma_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_MIX_MAX, K_MIX_MAX>("1 mfma + 4 v_max_i32 interleaved (per mfma)", 256, 32, 32);
  run<K_MIX_EXP, K_MIX_EXP>("1 mfma + 2 v_exp_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_BF16_K16, K_BF16_K16>("v_mfma_f32_16x16x16_bf16 indep", 256, 32, 32);
  run<K_BF16_K16_DEP, K_BF16_K16_DEP>("v_mfma_f32_16x16x16_bf16 dep", 256, 32, 32);
  run<K_BF16_K32, K_BF16_K32>("v_mfma_f32_16x16x32_bf16 indep", 256, 32, 32);
  run<K_BF16_K16_MIX4, K_BF16_K16_MIX4>("1 mfma 16x16x16 bf16 + 4 v_fma_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_BF16_K16_MIX2, K_BF16_K16_MIX2>("1 mfma 16x16x16 bf16 + 2 v_fma_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_BF16_K32_MIX4, K_BF16_K32_MIX4>("1 mfma 16x16x32 bf16 + 4 v_fma_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_BF16_K16_MIXPK2, K_BF16_K16_MIXPK2>("1 mfma 16x16x16 bf16 + 2 v_pk_fma_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_BF16_K16_MIXEXP1, K_BF16_K16_MIXEXP1>("1 mfma 16x16x16 bf16 + 1 v_exp_f32 interleaved (per mfma)", 256, 32, 32);
  run<K_CVT, K_CVT>("v_cvt_pk_bf16_f32", 256, 32, 32);
  printf("== two waves per SIMD (512 threads x 256 blocks)\n");
  run<K_BF16_K16, K_PKFMA>("mfma 16x16x16 bf16 | v_pk_fma_f32", 512, 32, 32);
  run<K_BF16_K16, K_CVT>("mfma 16x16x16 bf16 | v_cvt_pk_bf16_f32", 512, 32, 32);
  run<K_BF16_K16, K_MAX_I32>("mfma 16x16x16 bf16 | v_max_i32", 512, 32, 32);
  run<K_BF16_K16, K_FMA>("mfma 16x16x16 bf16 | v_fma_f32", 512, 32, 32);
  run<K_BF16_K32, K_FMA>("mfma 16x16x32 bf16 | v_fma_f32", 512, 32, 32);
  run<K_BF16_K16, K_EXP>("mfma 16x16x16 bf16 | v_exp_f32", 512, 32, 32);
  run<K_MFMA_IND, K_FMA>("mfma | v_fma_f32 (non-packed)", 512, 32, 32);
  run<K_MFMA_IND, K_MAX_I32>("mfma | v_max_i32", 512, 32, 32);
  run<K_PKFMA, K_PKFMA>("pk_fma | pk_fma", 512, 32, 32);
  run<K_EXP, K_EXP>("exp | exp", 512, 32, 32);
  run<K_MFMA_IND, K_MFMA_IND>("mfma | mfma", 512, 32, 32);
  run<K_MFMA_IND, K_PKFMA>("mfma | pk_fma  (do f32 MFMA and f32 VALU overlap?)", 512, 32, 32);
  run<K_MFMA_IND, K_EXP>("mfma | exp", 512, 32, 32);
  run<K_MFMA_DEP, K_PKFMA>("mfma dep | pk_fma", 512, 32, 32);
  run<K_PKFMA, K_EXP>("pk_fma | exp", 512, 32, 32);
  return 0;
}
