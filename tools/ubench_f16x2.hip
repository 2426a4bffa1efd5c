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
__device__ __forceinline__ float sub_lo16(float a, h2v h) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a)); return r; }
__device__ __forceinline__ float sub_hi16(float a, h2v h) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a)); return r; }
__device__ __forceinline__ h8v split16(f4 a) {
  const h2v h01 = cvt_pk16(a.x, a.y), h23 = cvt_pk16(a.z, a.w);
  const h2v l01 = cvt_pk16(sub_lo16(a.x, h01), sub_hi16(a.y, h01)), l23 = cvt_pk16(sub_lo16(a.z, h23), sub_hi16(a.w, h23));
  return h8v{h01.x, h01.y, h23.x, h23.y, l01.x, l01.y, l23.x, l23.y};
}
__device__ __forceinline__ WF16 wsplit16(f4 w) {
  const h2v h01 = cvt_pk16(w.x, w.y), h23 = cvt_pk16(w.z, w.w);
  const h2v l01 = cvt_pk16(w.x - (float)h01.x, w.y - (float)h01.y), l23 = cvt_pk16(w.z - (float)h23.x, w.w - (float)h23.y);
  WF16 f; f.a1 = h8v{h01.x, h01.y, h23.x, h23.y, h01.x, h01.y, h23.x, h23.y}; f.a2 = h8v{l01.x, l01.y, l23.x, l23.y, l01.x, l01.y, l23.x, l23.y}; return f;
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
  dbg[l] = f4{(float)s[0] + (float)s[4] - z.x, (float)s[1] + (float)s[5] - z.y, (float)s[2] + (float)s[6] - z.z, (float)s[3] + (float)s[7] - z.w};
}
int main() {
  float hw[256], hz[256], r32[256], r16[256], d[256];
  srand(1);
  for (int i = 0; i < 256; ++i) { hw[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f; hz[i] = (rand() / (float)RAND_MAX - 0.5f) * 20.f; }
  f4 *W, *Z, *A, *B, *D;
  hipMalloc(&W, 1024); hipMalloc(&Z, 1024); hipMalloc(&A, 1024); hipMalloc(&B, 1024); hipMalloc(&D, 1024);
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
  printf("max|ref| %.3f  err f32-mfma %.3e  err f16x2 %.3e  split residual %.3e\n", mx, e32, e16, ed);
}
