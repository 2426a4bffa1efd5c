// ubench_trans.hip -- does a transcendental (v_exp_f32 / v_rcp_f32: 8-9 issue cycles per wave-instruction, profiles/r03_ubench_issue.txt)
// leave room for plain VALU work behind it?  The trajectory kernels' tails are ~46 % transcendentals by VALU time: if an independent
// v_fma_f32 / v_pk_fma_f32 issues in the shadow of a v_exp_f32, interleaving them in the instruction stream would shorten the tails.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/ubench_trans tools/ubench_trans.hip
// Prints shader cycles (s_memtime) per GROUP (one v_exp_f32 + n plain instructions), one wave per SIMD and two.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
template <int KIND>
__device__ __forceinline__ void body(float (&r)[16]) {
  if constexpr (KIND == 0) {          // 8 independent v_exp
    REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                      "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));)
  } else if constexpr (KIND == 1) {   // 8 independent v_fma
    REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                      "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]), "v"(r[9]));)
  } else if constexpr (KIND == 2) {   // (exp, fma) x 4: independent registers
    REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n"
                      "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %8, %9"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(r[8]), "v"(r[9]));)
  } else if constexpr (KIND == 3) {   // (exp, fma, fma) x 4
    REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %8, %8, %12, %13\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %12, %13\n v_fma_f32 %9, %9, %12, %13\n"
                      "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %10, %10, %12, %13\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %12, %13\n v_fma_f32 %11, %11, %12, %13"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[10]), "+v"(r[11]),
                        "+v"(r[12]), "+v"(r[13]) : "v"(r[8]), "v"(r[9]));)
  } else if constexpr (KIND == 4) {   // (exp, pk_fma) x 4
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {r[4], r[5]}, b = {r[6], r[7]}, c = {r[10], r[11]}, d = {r[12], r[13]}, m = {r[8], r[8]}, n = {r[9], r[9]};
    REP8(asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_pk_fma_f32 %5, %5, %8, %9\n"
                      "v_exp_f32 %2, %2\n v_pk_fma_f32 %6, %6, %8, %9\n v_exp_f32 %3, %3\n v_pk_fma_f32 %7, %7, %8, %9"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));)
    r[4] = a[0] + a[1]; r[6] = b[0] + b[1]; r[10] = c[0] + c[1]; r[12] = d[0] + d[1];
  } else if constexpr (KIND == 5) {   // dependent chain: exp -> fma -> exp -> fma (what a tanh / exp tail looks like for ONE element)
    REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n"
                      "v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %1, %2"
                      : "+v"(r[0]) : "v"(r[8]), "v"(r[9]));)
  } else if constexpr (KIND == 6) {   // four such chains interleaved
    REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                      "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(r[8]), "v"(r[9]));)
  } else if constexpr (KIND == 7) {   // the same four chains, exp of chain i next to fma of chain i - 1
    REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %3, %3, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %0, %0, %4, %5\n"
                      "v_exp_f32 %2, %2\n v_fma_f32 %1, %1, %4, %5\n v_exp_f32 %3, %3\n v_fma_f32 %2, %2, %4, %5"
                      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(r[8]), "v"(r[9]));)
  }
}
template <int KA, int KB>
__global__ __launch_bounds__(512) void k(int iters, float seed, float* sink, unsigned long long* cyc) {
  float r[16];
  for (int i = 0; i < 16; ++i) r[i] = seed * (i + 1) * 1e-3f + threadIdx.x * 1e-6f;
  r[8] = 0.999f; r[9] = 1e-3f;
  const bool second = (threadIdx.x >> 8) & 1;        // threads 256..511: the second wave of each SIMD
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (!second) body<KA>(r); else body<KB>(r);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += r[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}
template <int KA, int KB>
static void run(const char* name, int threads, int groups_a, int groups_b) {
  float* sink; unsigned long long* cyc;
  hipMalloc(&sink, sizeof(float) * 256 * 512); hipMalloc(&cyc, 16);
  hipMemset(cyc, 0, 16);
  const int iters = 2000;
  hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(threads), 0, 0, 10, 1.f, sink, cyc);
  hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(threads), 0, 0, iters, 1.f, sink, cyc);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-78s waveA %6.2f cycles/group", name, (double)h[0] / iters / groups_a);
  if (threads > 256) printf("   waveB %6.2f cycles/group", (double)h[1] / iters / groups_b);
  printf("\n");
  hipFree(sink); hipFree(cyc);
}
int main() {
  printf("== one wave per SIMD\n");
  run<0, 0>("v_exp_f32 (group = 1 instruction)", 256, 64, 64);
  run<1, 1>("v_fma_f32 (group = 1 instruction)", 256, 64, 64);
  run<2, 2>("1 v_exp_f32 + 1 independent v_fma_f32", 256, 32, 32);
  run<3, 3>("1 v_exp_f32 + 2 independent v_fma_f32", 256, 32, 32);
  run<4, 4>("1 v_exp_f32 + 1 independent v_pk_fma_f32", 256, 32, 32);
  run<5, 5>("DEPENDENT chain exp -> fma -> exp -> fma (group = exp + fma)", 256, 32, 32);
  run<6, 6>("four such chains, 4 exps then 4 fmas (group = exp + fma)", 256, 32, 32);
  run<7, 7>("four such chains, exp of chain i beside fma of chain i-1 (group = exp + fma)", 256, 32, 32);
  printf("== two waves per SIMD\n");
  run<0, 1>("v_exp_f32 | v_fma_f32", 512, 64, 64);
  run<0, 0>("v_exp_f32 | v_exp_f32", 512, 64, 64);
  run<1, 1>("v_fma_f32 | v_fma_f32", 512, 64, 64);
  run<6, 6>("4 exps then 4 fmas | same (group = exp + fma)", 512, 32, 32);
  return 0;
}
