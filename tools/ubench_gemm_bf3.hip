// ubench_gemm_bf3.hip -- the split engine's NT GEMM on the f32-input MFMA vs the bf16x3 form (csrc/gemm_f32.hpp, BF3 = 1) vs the
// bf16x3 form on pre-split planes (csrc/gemm_xl.hpp; -DL2HMC_XL_TIMING adds the shader cycles of its k loop):
// time and accuracy on the decoder-sized products of config 5 (M = 8192 chains, 1024 x 1024 and 1024 x 784 weights).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I l2hmc_amd/csrc -I include -o tools/bin/ubench_gemm_bf3 tools/ubench_gemm_bf3.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gemm_f32.hpp"
#include "gemm_xl.hpp"
namespace l2hmc {   // the two symbols of l2hmc_abi.hip the header refers to
thread_local char g_err[512];
int fail(int code, const char* fmt, const char* s, long long a, long long b) { (void)fmt; (void)s; (void)a; (void)b; return code; }
}
using namespace l2hmc;

static double run(int epi_softplus, int M, int N, int K, int bf3, const float* A, const float* B, const float* bias, float* C,
                  float* C2, int reps) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.beta = 1.f; g.bias = bias;
  g.C2 = C2; g.ldc2 = N; g.bf3 = bf3;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { if (epi_softplus) launch_gemm<EPI_BIAS_SOFTPLUS>(g, 0); else launch_gemm<EPI_BIAS>(g, 0); }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) { if (epi_softplus) launch_gemm<EPI_BIAS_SOFTPLUS>(g, 0); else launch_gemm<EPI_BIAS>(g, 0); }
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

static double run_xlp(int epi_softplus, int M, int N, int K, const unsigned short* Ap, const unsigned short* Bp, const float* bias, float* C,
                      float* C2, int reps) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  const int ldp = ceil_to(K, 32);
  g.Ap = Ap; g.ap_plane = (long long)M * ldp; g.ldap = ldp; g.Bp = Bp; g.bp_plane = (long long)ceil_to(N, XLP_TN) * ldp; g.ldbp = ldp;
  g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.beta = 1.f; g.bias = bias; g.C2 = C2; g.ldc2 = N; g.bf3 = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { if (epi_softplus) launch_gemm_planes<EPI_BIAS_SOFTPLUS>(g, 0); else launch_gemm_planes<EPI_BIAS>(g, 0); }
  if (hipDeviceSynchronize() != hipSuccess) { printf("xlp launch failed: %s\n", hipGetErrorString(hipGetLastError())); return -1; }
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) { if (epi_softplus) launch_gemm_planes<EPI_BIAS_SOFTPLUS>(g, 0); else launch_gemm_planes<EPI_BIAS>(g, 0); }
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main() {
  const int M = 8192;
  const int shapes[3][2] = {{1024, 1024}, {784, 1024}, {1024, 784}};     // (N, K)
  for (int sidx = 0; sidx < 3; ++sidx) {
    const int N = shapes[sidx][0], K = shapes[sidx][1];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N);
    srand(1 + sidx);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 0.5f;          // softplus-like activations: mostly positive
    for (auto& v : hB) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *A, *B, *b, *C, *C2;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&b, N * 4);
    hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&C2, (size_t)M * N * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> out[2];
    double us[2];
    for (int bf3 = 0; bf3 < 2; ++bf3) {
      us[bf3] = run(0, M, N, K, bf3, A, B, b, C, nullptr, 20);
      out[bf3].resize((size_t)64 * N);
      hipMemcpy(out[bf3].data(), C + (size_t)4000 * N, out[bf3].size() * 4, hipMemcpyDeviceToHost);    // rows 4000 .. 4063
    }
    const double us_sp = run(1, M, N, K, 1, A, B, b, C, C2, 20), us_sp0 = run(1, M, N, K, 0, A, B, b, C, C2, 20);
    double e[2] = {0, 0}, scale = 0;
    for (int r = 0; r < 64; ++r)
      for (int n = 0; n < N; ++n) {
        double ref = hb[n];
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)(4000 + r) * K + k] * (double)hB[(size_t)n * K + k];
        scale = fmax(scale, fabs(ref));
        for (int v = 0; v < 2; ++v) e[v] = fmax(e[v], fabs(out[v][(size_t)r * N + n] - ref));
      }
    const double fl = 2.0 * M * N * K;
    printf("M=%d N=%d K=%d: f32 MFMA %.1f us = %.1f TFLOP/s | bf16x3 %.1f us = %.1f TFLOP/s (algorithmic; x%.2f) | "
           "softplus epilogue: f32 %.1f us, bf16x3 %.1f us | max |err| vs fp64: f32 %.2e  bf16x3 %.2e  (|C| <= %.2f)\n",
           M, N, K, us[0], fl / us[0] * 1e-6, us[1], fl / us[1] * 1e-6, us[0] / us[1], us_sp0, us_sp, e[0], e[1], scale);
    if (K % 8 == 0) {   // the same tile on pre-split bf16 planes (gemm_xlp_kernel): time, and bit-equality with the 128 x 128 bf16x3 kernel
      unsigned short *Ap, *Bp;
      const int ldp = ceil_to(K, 32), Np = ceil_to(N, XLP_TN);
      hipMalloc(&Ap, (size_t)M * ldp * 6); hipMalloc(&Bp, (size_t)Np * ldp * 6);
      to_planes(0, A, K, M, K, Ap, M, ldp);
      to_planes(0, B, K, N, K, Bp, Np, ldp);
      run(0, M, N, K, 1, A, B, b, C, nullptr, 1);
      std::vector<float> ref((size_t)M * N), got((size_t)M * N);
      hipMemcpy(ref.data(), C, ref.size() * 4, hipMemcpyDeviceToHost);
      hipMemset(C, 0, ref.size() * 4);
#ifdef L2HMC_XL_TIMING
      unsigned long long z[4] = {0, 0, 0, 0}, t[4];
      hipMemcpyToSymbol(HIP_SYMBOL(l2hmc::xl_ticks), z, sizeof(z));
#endif
      const double us_p = run_xlp(0, M, N, K, Ap, Bp, b, C, nullptr, 20);
#ifdef L2HMC_XL_TIMING
      hipMemcpyFromSymbol(t, HIP_SYMBOL(l2hmc::xl_ticks), sizeof(t));
      const double cyc = (double)t[0] / (double)t[1], wall_us = (double)t[2] / (double)t[1] / 100.0;
      printf("   planes: k loop of a workgroup: %.0f shader cycles = %.0f per k-tile of 32, %.1f us by the constant clock -> %.2f GHz\n",
             cyc, cyc / ((K + 31) / 32), wall_us, cyc / wall_us * 1e-3);
#endif
      hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost);
      size_t diff = 0;
      double worst = 0;
      for (size_t i = 0; i < ref.size(); ++i) { if (ref[i] != got[i]) ++diff; worst = fmax(worst, fabs((double)ref[i] - got[i])); }
      const double us_p_sp = run_xlp(1, M, N, K, Ap, Bp, b, C, C2, 20);
      printf("   256 x 128 tiles, 8 waves, on pre-split planes (gemm_xlp_kernel): %.1f us = %.1f TFLOP/s (x%.2f vs 128 x 128 bf16x3), softplus epilogue %.1f us | "
             "elements that differ from the 128 x 128 kernel: %zu (max %.2e)\n", us_p, fl / us_p * 1e-6, us[1] / us_p, us_p_sp, diff, worst);
      hipFree(Ap); hipFree(Bp);
    }
    hipFree(A); hipFree(B); hipFree(b); hipFree(C); hipFree(C2);
  }
  return 0;
}
