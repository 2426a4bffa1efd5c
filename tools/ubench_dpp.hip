// ubench_dpp.hip -- issue cost of v_fmac_f32 with a DPP quad broadcast on its first source against the plain v_fmac_f32 and
// v_pk_fma_f32 (gfx950, one wave per SIMD, eight independent accumulators): does a weight held 4-per-VGPR (quad lane k) and
// broadcast inside the FMA cost more issue time than a wave-uniform VGPR operand?   (profiles/r06_lane_resident.txt)
//   hipcc -O3 --offload-arch=gfx950 -o ubench_dpp tools/ubench_dpp.hip && ./ubench_dpp
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(64) void k(int iters, float seed, float* sink, unsigned long long* ticks) {
  float a0 = seed, a1 = seed, a2 = seed, a3 = seed, a4 = seed, a5 = seed, a6 = seed, a7 = seed;
  f2 p0 = {seed, seed}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
  const float w = 1e-9f * (threadIdx.x & 3), x = 0.5f;
  const f2 w2 = {w, w};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                        "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(x));)
    } else if (KIND == 1) {
      REP8(asm volatile("v_fmac_f32_dpp %0, %8, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                        "v_fmac_f32_dpp %2, %8, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
                        "v_fmac_f32_dpp %4, %8, %9 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                        "v_fmac_f32_dpp %6, %8, %9 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(x));)
    } else {
      REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                        "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(w2), "v"(w2));)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

int main() {
  float* sink;
  unsigned long long* ticks;
  hipMalloc(&sink, 1024 * 64 * sizeof(float));
  hipMalloc(&ticks, 8);
  const int iters = 2000;
  const char* names[3] = {"v_fmac_f32 (VGPR weight)", "v_fmac_f32_dpp quad_perm broadcast", "v_pk_fma_f32"};
  for (int waves = 1; waves <= 2; ++waves)
    for (int kind = 0; kind < 3; ++kind) {
      unsigned long long t = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1024 * waves), dim3(64), 0, 0, iters, 1.f, sink, ticks);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1024 * waves), dim3(64), 0, 0, iters, 1.f, sink, ticks);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1024 * waves), dim3(64), 0, 0, iters, 1.f, sink, ticks);
        hipDeviceSynchronize();
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s) per SIMD  %-38s %.2f cycles per instruction\n", waves, names[kind], (double)t / (iters * 64.0));
    }
  return 0;
}
