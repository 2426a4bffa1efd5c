// Microbenchmark behind DESIGN.md section 8: ONE CHAIN PER LANE, the state x[50] in registers, the S/T/Q net's layer 1
// (50 -> 10) and heads (10 -> 50) as packed VALU FMAs whose weights are wave-uniform SGPR pairs
// (v_pk_fma_f32 v[..], s[w:w+1], v[x:x+1] op_sel_hi:[1,0,0]) -- no MFMA padding, no cross-wave exchange.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ub_lane tools/ubench_lane_per_chain.hip && tools/bin/ub_lane
// Measured on MI355X (2 waves per SIMD, 131 072 chains): 86 TFLOP/s of matrix work = 0.55 of the fp32 roof with naive
// scalar loads -- twice what the MFMA tiles sustain on the same net -- but a wave is 64 CHAINS: 4096 chains are 64 waves
// (6 % of the SIMDs).  The form pays from ~65 536 chains per GPU on; the MFMA tile exists to find parallelism INSIDE a
// chain when chains are scarce (the 4096 / 8192 chains per GPU of BASELINE.json).  (The compiler hoists the loop-invariant
// weight loads into ~1000 SGPRs it does not have and spills them through v_readlane unless a memory clobber keeps the
// loads inside the loop: 20 TFLOP/s without it.)
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
// one chain per lane: x[50] in registers; layer 1 (10 hidden) + heads (3 x 50) on the VALU with wave-uniform weights
template <int D, int H>
__global__ __launch_bounds__(64, 2) void lane_net(const float* __restrict__ W1, const float* __restrict__ Wh,
                                                  const float* __restrict__ xin, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x, n = blockIdx.x * 64 + lane;
  float x[D];
#pragma unroll
  for (int k = 0; k < D; ++k) x[k] = xin[(size_t)k * gridDim.x * 64 + n];
  float accum = 0.f;
  for (int it = 0; it < iters; ++it) {
    f2 h[H / 2];
#pragma unroll
    for (int j = 0; j < H / 2; ++j) h[j] = f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; ++k) {
      asm volatile("" ::: "memory");          // keep the (loop-invariant) weight loads inside: 1000 SGPRs do not exist
      const f2 xs = f2{x[k], x[k]};
#pragma unroll
      for (int j = 0; j < H / 2; ++j) {
        const f2 w = *reinterpret_cast<const f2*>(W1 + k * H + 2 * j);       // wave-uniform address
        h[j] = __builtin_elementwise_fma(w, xs, h[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < H / 2; ++j) { h[j].x = fmaxf(h[j].x, 0.f); h[j].y = fmaxf(h[j].y, 0.f); }
    // heads: out[k] = sum_j h[j] Wh[j][k], k in pairs
#pragma unroll
    for (int k = 0; k < D; k += 2) {
      asm volatile("" ::: "memory");
      f2 o = f2{0.f, 0.f};
#pragma unroll
      for (int j = 0; j < H; ++j) {
        const float hj = (j & 1) ? h[j / 2].y : h[j / 2].x;
        const f2 w = *reinterpret_cast<const f2*>(Wh + j * D + k);
        o = __builtin_elementwise_fma(w, f2{hj, hj}, o);
      }
      x[k] = x[k] * 0.999f + 1e-3f * o.x;
      if (k + 1 < D) x[k + 1] = x[k + 1] * 0.999f + 1e-3f * o.y;
    }
  }
#pragma unroll
  for (int k = 0; k < D; ++k) accum += x[k];
  out[n] = accum;
}
template __global__ void lane_net<50, 10>(const float*, const float*, const float*, float*, int);

#include <cstdio>
#include <vector>
int main() {
  const int D = 50, H = 10;
  const int blocks = 256 * 8, iters = 2000;
  float *W1, *Wh, *x, *out;
  hipMalloc(&W1, D * H * 4); hipMalloc(&Wh, H * D * 4); hipMalloc(&x, (size_t)D * blocks * 64 * 4); hipMalloc(&out, blocks * 64 * 4);
  std::vector<float> w(D * H, 0.01f), xs((size_t)D * blocks * 64, 0.5f);
  hipMemcpy(W1, w.data(), D * H * 4, hipMemcpyHostToDevice); hipMemcpy(Wh, w.data(), D * H * 4, hipMemcpyHostToDevice);
  hipMemcpy(x, xs.data(), xs.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((lane_net<50, 10>), dim3(blocks), dim3(64), 0, 0, W1, Wh, x, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * 64 * iters * (D * H + H * D);
    // waves per SIMD = blocks / 1024
    printf("lane_net<50,10>: %.3f ms, %.1f TFLOP/s of matrix work, %.0f SIMD-cycles (2.4 GHz) per wave-iteration\n", ms,
           2 * fma / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)blocks / 1024 * iters));
  }
  return 0;
}
