#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#define L2HMC_FAST_F16X2 1
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f2v __attribute__((ext_vector_type(2)));
struct WF16 { h8v a1; h8v a2; };
__device__ __forceinline__ h2v cvt_pk16(float a, float b) { return __builtin_convertvector(f2v{a, b}, h2v); }
// the product's split (l2hmc_amd/csrc/traj_fast.hpp): w a = (64 w)(a / 64); hd = f16(a / 64), lo = f16(a - 64 hd); w_hi = f16(w),
// w_lo = f16(64 (w - w_hi)) / 64; fragments [64 w_hi | w_hi], [64 w_lo | w_lo]
#define SPL 64.0f
__device__ __forceinline__ h2v lo_pair16(h2v h, float a0, float a1) {
  h2v r;
  const float ns = -SPL;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(ns), "v"(a0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "s"(ns), "v"(a1));
  return r;
}
__device__ __forceinline__ h8v split16(f4 a) {
  const f4 ad = a * (1.f / SPL);
  const h2v h01 = cvt_pk16(ad.x, ad.y), h23 = cvt_pk16(ad.z, ad.w);
  const h2v l01 = lo_pair16(h01, a.x, a.y), l23 = lo_pair16(h23, a.z, a.w);
  return h8v{h01.x, h01.y, h23.x, h23.y, l01.x, l01.y, l23.x, l23.y};
}
__device__ __forceinline__ WF16 wsplit16(f4 w) {
  const float sc = SPL;
  const h2v h01 = cvt_pk16(w.x, w.y), h23 = cvt_pk16(w.z, w.w);
  const f4 r = f4{w.x - (float)h01.x, w.y - (float)h01.y, w.z - (float)h23.x, w.w - (float)h23.y} * sc;
  const h2v L01 = cvt_pk16(r.x, r.y), L23 = cvt_pk16(r.z, r.w);
  const h2v H01 = cvt_pk16((float)h01.x * sc, (float)h01.y * sc), H23 = cvt_pk16((float)h23.x * sc, (float)h23.y * sc);
  const h2v l01 = cvt_pk16((float)L01.x / sc, (float)L01.y / sc), l23 = cvt_pk16((float)L23.x / sc, (float)L23.y / sc);
  WF16 f;
  f.a1 = h8v{H01.x, H01.y, H23.x, H23.y, h01.x, h01.y, h23.x, h23.y};
  f.a2 = h8v{L01.x, L01.y, L23.x, L23.y, l01.x, l01.y, l23.x, l23.y};
  return f;
}
__device__ __forceinline__ f4 mfma16x2(const WF16& W, h8v b, f4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.a1, b, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(W.a2, b, acc, 0, 0, 0);
}
__global__ void k(const f4* W, const f4* Z, f4* o32, f4* o16, f4* dbg) {
  const int l = threadIdx.x;
  f4 w = W[l], z = Z[l], a = {0, 0, 0, 0};
  for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], z[r], a, 0, 0, 0);
  o32[l] = a;
  f4 b = {0, 0, 0, 0};
  const WF16 wf = wsplit16(w);
  const h8v zs = split16(z);
  __builtin_amdgcn_sched_barrier(0);
  o16[l] = mfma16x2(wf, zs, b);
  __builtin_amdgcn_sched_barrier(0);
  h8v s = split16(z);
  dbg[l] = f4{(float)s[0] * SPL + (float)s[4] - z.x, (float)s[1] * SPL + (float)s[5] - z.y, (float)s[2] * SPL + (float)s[6] - z.z, (float)s[3] * SPL + (float)s[7] - z.w};
}
int main() {
  float hw[256], hz[256], r32[256], r16[256], d[256];
  f4 *W, *Z, *A, *B, *D;
  hipMalloc(&W, 1024); hipMalloc(&Z, 1024); hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&D, 1024);
  const float ws[7] = {1.f, 1.f, 0.01f, 0.01f, 1e-3f, 0.2f, 100.f}, zs[7] = {10.f, 0.05f, 10.f, 0.05f, 1e-3f, 300.f, 1e5f};
  for (int t = 0; t < 7; ++t) {
    srand(1 + t);
    for (int i = 0; i < 256; ++i) { hw[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f * ws[t]; hz[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f * zs[t]; }
    hipMemcpy(W, hw, 1024, hipMemcpyHostToDevice); hipMemcpy(Z, hz, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(W, Z, A, B, D);
    hipMemcpy(r32, A, 1024, hipMemcpyDeviceToHost); hipMemcpy(r16, B, 1024, hipMemcpyDeviceToHost); hipMemcpy(d, D, 1024, hipMemcpyDeviceToHost);
    // fp64 reference: D[row 4q+r][col c] = sum_{k-step s, kq} W_lane(i=row, kq)[s] * Z_lane(c, kq)[s]
    double e32 = 0, e16 = 0, ed = 0, mx = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      int c = l & 15, q = l >> 4, row = 4 * q + r; double ref = 0;
      for (int s = 0; s < 4; ++s) for (int kq = 0; kq < 4; ++kq) ref += (double)hw[(kq * 16 + row) * 4 + s] * (double)hz[(kq * 16 + c) * 4 + s];
      e32 = fmax(e32, fabs(r32[l * 4 + r] - ref)); e16 = fmax(e16, fabs(r16[l * 4 + r] - ref)); mx = fmax(mx, fabs(ref)); ed = fmax(ed, fabs(d[l * 4 + r]));
    }
    printf("|w| <= %.0e |z| <= %.0e: max|ref| %.3e  err f32-mfma %.3e  err f16x2 %.3e  (ratio %.2f)  split residual %.3e\n", ws[t], zs[t], mx, e32, e16, e16 / e32, ed);
  }
}
