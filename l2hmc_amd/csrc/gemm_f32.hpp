// gemm_f32.hpp -- hand-written fp32 MFMA GEMM with fused epilogues for the split engine (gfx950 / CDNA4).
//
//   C[M x N] = epilogue( A[M x K] . B[N x K]^T )          ("NT" form: BOTH operands K-contiguous)
//
// M = the chain batch (thousands), N / K = layer widths of the VAE decoder (mnist_vae.py:104-111:
// 50 -> 1024 -> 1024 -> 784), of the sampler's image branch (:134-140) and of the H = 200 S/T/Q nets
// (:142-167).  Forward layers contract with W^T (W is (in, out), layers.py:33), so the host side keeps a
// transposed copy of every weight for the forward products and uses W as stored for the input-gradient
// products (dA = dOut . W^T  ==  NT form with B = W).
//
// Tiling: a 256-thread workgroup owns a 128 x 128 tile of C; each of its 4 waves a 64 x 64 quadrant =
// 4 x 4 v_mfma_f32_16x16x4_f32 tiles (64 accumulator VGPRs).  The MFMA "A" operand carries the WEIGHT rows
// (n) and the "B" operand the activation rows (m), so that a lane ends up with 4 CONSECUTIVE columns
// n = n0 + 4q + r of one row m = m0 + c: bias / aux / sigmoid operands and the result move as dwordx4.
// K is consumed in tiles of GK = 16: both tiles are staged in LDS as [row][16 k + 4 pad] (80-byte rows: one
// conflict-free ds_read_b128 per lane fetches the operands of the 4 k-steps of a 16-wide tile), double
// buffered, the next tile's global loads issued before the 64 MFMAs of the current one.  (-DL2HMC_GEMM_GK=32 -- one
// barrier per 128 MFMAs, 74 KB of LDS per workgroup -- was measured: config 5 4.87 -> 5.34 ms per proposal; not used.)
//
// Epilogues (fused, so no activation makes an extra HBM round trip):
//   EPI_BIAS            C = acc + b
//   EPI_BIAS_SOFTPLUS   C = softplus(acc + b),  C2 = sigmoid(acc + b)   (C2 feeds the backward pass)
//   EPI_BIAS_RELU       C = relu(acc + b)
//   EPI_MUL             C = acc * E[m][n] (+ C if accum),  C2 = acc      (backward through softplus)
//   EPI_MASK            C = E[m][n] > 0 ? acc : 0                       (backward through relu; E = the activation)
//   EPI_TAN             C = E acc,  C2 = E (1 - E) acc E2               (tangent through softplus: E = sigmoid,
//                       E2 = the cotangent that multiplies the sigmoid in the reverse pass -- the two terms of
//                       the Hessian-vector product of the decoder energy, train_split.hpp)
//   EPI_ADD             C = acc + E[m][n]                               (d/dz of the prior term)
//   EPI_BCE             l = acc + b;  C = beta (sigmoid(l) - t),  rowsum[m][tile] = beta sum_n bce(l, t)
//                       (mnist_vae.py:122-126, TF's stable form max(l,0) - l t + log1p(e^{-|l|}))
//   EPI_NET1            C = relu(acc + tb[row(m)][n] + auxh[m][n])      (first hidden layer of an S/T/Q net:
//                       time/bias table row of the chain's schedule row, image-branch term)
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "l2hmc_kernels.hpp"

namespace l2hmc {

enum { EPI_BIAS = 0, EPI_BIAS_SOFTPLUS = 1, EPI_BIAS_RELU = 2, EPI_MUL = 3, EPI_ADD = 4, EPI_BCE = 5, EPI_NET1 = 6,
       EPI_MASK = 7, EPI_TAN = 8 };

struct GemmArgs {
  const float* A; int lda;       // (M, K)
  const float* B; int ldb;       // (N, K)
  float* C; int ldc;             // (M, N)
  int M, N, K;
  const float* bias;             // (N) or NULL
  const float* E; int lde;       // (M, N) second operand of the epilogue (sigmoid / aux / z / auxh) or NULL
  float* C2; int ldc2;           // second output (sigmoid / raw product / second tangent term) or NULL
  const float* E2; int lde2;     // EPI_TAN: third operand
  int accum;                     // EPI_MUL: C += instead of C =
  float* rowsum; int n_tiles;    // EPI_BCE: (M, bce_partials) partial sums, one per (column tile, wave column)
  float beta;                    // EPI_BCE scale
  // EPI_NET1
  const float* tb;               // (T, N) time/bias table of this net
  const unsigned char* dir; int dir_all, it, T;
  int bf3;                       // 1: the product runs on the bf16 MFMA with 3-way split operands (see gemm_nt_kernel)
  // pre-split operands (gemm_xl.hpp): three bf16 planes h | m | l of an fp32 matrix, plane p at P + p * plane, row stride ld
  // (elements).  Contract: ld >= ceil32(K) with the columns K .. ceil32(K) - 1 ZERO; Bp holds ceil128(N) rows, the rows beyond N
  // zero (the kernel reads whole tiles without predicates).  Cp: the epilogue ALSO (C != NULL) or ONLY (C == NULL) writes its
  // first output as planes -- every element is split once by its producer instead of once per consumer tile.
  const unsigned short* Ap; long long ap_plane; int ldap;
  const unsigned short* Bp; long long bp_plane; int ldbp;
  unsigned short* Cp; long long cp_plane; int ldcp;
  // plane arithmetic: 0 = bf16x3 (h | m | l, six products), 1 = f16x2 (three f16 planes X1 | X1 / 64 | 64 (x - X1), three
  // products: gemm_xl.hpp) -- what the planes Ap / Bp hold and what the epilogue writes to Cp
  int pm;
};

#ifndef L2HMC_GEMM_GK
#define L2HMC_GEMM_GK 16
#endif
constexpr int GK = L2HMC_GEMM_GK, GP = GK + 4;   // k-tile, padded LDS row (floats): 16 consecutive rows start in 16 distinct
                                               // bank quads for GP = 20 and 36 alike

// softplus(p) = max(p, 0) + log(1 + e^{-|p|}) and sigmoid(p) from ONE hardware exp2, one log2 and one rcp
// (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each).  With e = e^{-|p|} in (0, 1] the argument 1 + e lies in
// (1, 2]: log2 there has absolute error ~1e-7, which is also the error of dropping e below 6e-8 -- the same
// size as one rounding of the result (ocml's expf + log1pf + division cost ~4x the instructions for nothing
// the fp32 sums downstream could keep).
__device__ __forceinline__ float softplus_acc(float p, float& sig) {
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(p));
  const float u = 1.f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  sig = p >= 0.f ? r : e * r;
  return fmaxf(p, 0.f) + 0.6931471805599453f * __builtin_amdgcn_logf(u);
}
__device__ __forceinline__ f4 softplus4(f4 p, f4& sig) {
  float s0, s1, s2, s3;
  const f4 r = f4{softplus_acc(p.x, s0), softplus_acc(p.y, s1), softplus_acc(p.z, s2), softplus_acc(p.w, s3)};
  sig = f4{s0, s1, s2, s3};
  return r;
}

// one row quad (4 consecutive k of one row) of an operand tile; rows are 16-byte aligned when KV == 4
// (dwordx4), 8-byte aligned when KV == 2 (K even: the d = 50 latent rows), else guarded scalar loads
template <int KV>
__device__ __forceinline__ f4 load_kquad(const float* p, int k, int K) {
  f4 v = splat(0.f);
  if (KV == 4) {
    if (k < K) v = *reinterpret_cast<const f4*>(p);
  } else if (KV == 2) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    if (k < K) { const f2 a = *reinterpret_cast<const f2*>(p); v.x = a.x; v.y = a.y; }
    if (k + 2 < K) { const f2 b = *reinterpret_cast<const f2*>(p + 2); v.z = b.x; v.w = b.y; }
  } else {
    if (k + 0 < K) v.x = p[0];
    if (k + 1 < K) v.y = p[1];
    if (k + 2 < K) v.z = p[2];
    if (k + 3 < K) v.w = p[3];
  }
  return v;
}

// ---- error-compensated bf16 MFMA ("bf16x3") ---------------------------------------------------------------------------
// gfx950's bf16 MFMA runs at 16x the f32-input MFMA rate.  Every fp32 operand is split into three bf16 terms
// x = h + m + l (round-to-nearest at each level: 24+ significand bits in all, i.e. the split is EXACT for fp32 inputs), and
// of the nine cross products the six with weight >= 2^-16 are kept:  h h + (h m + m h) + (h l + m m + l h).  The dropped ones
// are <= 3 x 2^-24 |x y| -- the size of ONE fp32 rounding; products of bf16 terms are exact in the MFMA's fp32 datapath and
// the accumulation is fp32 as before.  Six v_mfma_f32_16x16x32_bf16 (K = 32, ~16 cycles each) replace eight
// v_mfma_f32_16x16x4_f32 (K = 4 each, 32 cycles each) per 16 x 16 x 32 block: 96 instead of 256 matrix-pipe cycles, and unlike
// the f32-input MFMA the bf16 one leaves the VALU free for the split (v_cvt_pk_bf16_f32 + shift / and + v_pk_add_f32:
// 4.5 VALU per operand element, done by the consuming wave right after its LDS read -- the staging path and the LDS layout
// are those of the fp32 kernel).  Same D layout as the 16x16x4 form, so the epilogues are shared.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
struct Split3 { u4v h, m, l; };
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ Split3 split3(f4 x0, f4 x1) {       // 8 consecutive k of one row
  const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  Split3 o;
#ifdef L2HMC_BF3_ABL_NOSPLIT      // timing ablation only (wrong numbers): no VALU split
  o.h = u4v{__float_as_uint(x0.x), __float_as_uint(x0.y), __float_as_uint(x0.z), __float_as_uint(x0.w)};
  o.m = u4v{__float_as_uint(x1.x), __float_as_uint(x1.y), __float_as_uint(x1.z), __float_as_uint(x1.w)};
  o.l = o.h;
  return o;
#endif
  // The residuals are PLAIN v_sub_f32 (inline asm): left to the compiler, each pair of subtractions becomes one v_pk_add_f32,
  // and the packed-f32 path shares the matrix pipe -- beside a wave streaming bf16 MFMAs a packed instruction issues once per
  // ~22 cycles, a plain one every ~6.5 (profiles/r03_ubench_issue.txt, finding 3): the eight packed subtractions of a fragment
  // cost more than its other 28 instructions together.  (L2HMC_BF3_PK_SUB restores the compiler's choice: the round-3 form.)
  auto sub = [](float x, float y) {
#ifdef L2HMC_BF3_PK_SUB
    return x - y;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = xs[2 * i], b = xs[2 * i + 1];
    const unsigned ph = pk_bf16(a, b);
    const float ra = sub(a, __uint_as_float(ph << 16)), rb = sub(b, __uint_as_float(ph & 0xffff0000u));
    const unsigned pm = pk_bf16(ra, rb);
    const float sa = sub(ra, __uint_as_float(pm << 16)), sb = sub(rb, __uint_as_float(pm & 0xffff0000u));
    o.h[i] = ph; o.m[i] = pm; o.l[i] = pk_bf16(sa, sb);
  }
  return o;
}
struct Split4 { unsigned h[2], m[2], l[2]; };            // four values as packed bf16 pairs (low half = the first)
__device__ __forceinline__ Split4 split4(f4 x) {
  Split4 o;
  auto sub = [](float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
  };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned ph = pk_bf16(a, b);
    const float ra = sub(a, __uint_as_float(ph << 16)), rb = sub(b, __uint_as_float(ph & 0xffff0000u));
    const unsigned pm = pk_bf16(ra, rb);
    const float sa = sub(ra, __uint_as_float(pm << 16)), sb = sub(rb, __uint_as_float(pm & 0xffff0000u));
    o.h[i] = ph; o.m[i] = pm; o.l[i] = pk_bf16(sa, sb);
  }
  return o;
}
// ---- f16x2 planes (GemmArgs.pm = 1) --------------------------------------------------------------------------------------
// x = X1 + X2 / 64 with X1 = f16(x), X2 = f16(64 (x - X1)) (both round-to-nearest, |x - X1 - X2 / 64| <= 2^-22 |x| -- two 11-bit terms -- while X2 is a
// normal f16: |x| >= 4e-3; below that its quantum is 2^-30 absolute; |x| < 65504).  Stored as TWO f16 planes X1 | X2 (4 bytes per
// element against bf16x3's 6); the consumer forms X1 / 64 on its fragment (exact), so that  x y = X1 Y1 + (X1 / 64) Y2 + X2 (Y1 / 64)  (+ X2 Y2 / 4096 <= 2^-22 |x y|, dropped) is three MFMAs on one accumulator
// with no rescaling: half the matrix-pipe work of the six bf16x3 products and two thirds of their plane traffic.  The f16 exponent range is what
// bounds it: the SAMPLER's decoder products (activations, logits, BCE gradients: O(1e-3 ... 1e2)) take it; the trainer's adjoint
// planes (entries scaled by 1 / chains) keep bf16x3.
typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ Split4 split4_f16(f4 x) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  Split4 o;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const hf2 h = __builtin_convertvector(f2{a, b}, hf2);                                        // v_cvt_pk_f16_f32
    const hf2 d = h * (_Float16)0.015625f;
    const hf2 l = __builtin_convertvector(f2{(a - (float)h.x) * 64.f, (b - (float)h.y) * 64.f}, hf2);
    o.h[i] = __builtin_bit_cast(unsigned, h); o.m[i] = __builtin_bit_cast(unsigned, d); o.l[i] = __builtin_bit_cast(unsigned, l);
  }
  return o;
}
__device__ __forceinline__ f4 mfma_f16(u4v a, u4v b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, a), __builtin_bit_cast(hf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma_bf16(u4v a, u4v b, f4 c) {
#ifdef L2HMC_BF3_ABL_NOMFMA       // timing ablation only: the split results are consumed by one VALU op instead
  c.x += __uint_as_float(a.x ^ b.y);
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// The shared epilogue: lane (c, q) of wave w holds acc[i][j] = C[m = m0 + wm + 16 j + c][n = n0 + wn + 16 i + 4 q + (0..3)].
template <int EPI, int WMB, int WNB, int WAVES_N>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f4 (&acc)[WNB][WMB], long long m0, int n0, int wm, int wn,
                                              int w, int c, int q) {
  // ---- epilogue: lane holds C[m = m0 + wm + 16 j + c][n = n0 + wn + 16 i + 4 q + (0..3)] -------------------
  float rs[WMB];                                                    // EPI_BCE: per m-block row partial
#pragma unroll
  for (int j = 0; j < WMB; ++j) rs[j] = 0.f;
#pragma unroll
  for (int j = 0; j < WMB; ++j) {
    const long long m = m0 + wm + 16 * j + c;
    const bool mok = m < g.M;
    int trow = 0;
    if (EPI == EPI_NET1 && mok) {
      const bool fwd = g.dir != nullptr ? g.dir[m] != 0 : (g.dir_all != 0);
      trow = fwd ? g.it : (g.T - 1 - g.it);
    }
#pragma unroll
    for (int i = 0; i < WNB; ++i) {
      const int n = n0 + wn + 16 * i + 4 * q;
      if (!mok || n >= g.N) continue;
      const bool full = n + 3 < g.N;
      f4 v = acc[i][j];
      f4 b = splat(0.f), e = splat(0.f);
      auto ld4 = [&](const float* p) {
        if (full && ((reinterpret_cast<size_t>(p) & 15) == 0)) return *reinterpret_cast<const f4*>(p);
        f4 r = splat(0.f);
        r.x = p[0];
        if (n + 1 < g.N) r.y = p[1];
        if (n + 2 < g.N) r.z = p[2];
        if (n + 3 < g.N) r.w = p[3];
        return r;
      };
      auto st4 = [&](float* p, f4 r) {
        if (full && ((reinterpret_cast<size_t>(p) & 15) == 0)) { *reinterpret_cast<f4*>(p) = r; return; }
        p[0] = r.x;
        if (n + 1 < g.N) p[1] = r.y;
        if (n + 2 < g.N) p[2] = r.z;
        if (n + 3 < g.N) p[3] = r.w;
      };
      if (g.bias != nullptr) b = ld4(g.bias + n);
      if (g.E != nullptr) e = ld4(g.E + m * g.lde + n);
      f4 out, out2 = splat(0.f);
      bool has2 = EPI == EPI_BIAS_SOFTPLUS;
      if (EPI == EPI_BIAS) {
        out = v + b;
      } else if (EPI == EPI_BIAS_SOFTPLUS) {
        const f4 p = v + b;
        out = softplus4(p, out2);
      } else if (EPI == EPI_BIAS_RELU) {
        const f4 p = v + b;
        out = f4{fmaxf(p.x, 0.f), fmaxf(p.y, 0.f), fmaxf(p.z, 0.f), fmaxf(p.w, 0.f)};
      } else if (EPI == EPI_MUL) {
        out = v * e;
        if (g.accum) out = out + ld4(g.C + m * g.ldc + n);
        out2 = v;
        has2 = true;
      } else if (EPI == EPI_MASK) {
        out = f4{e.x > 0.f ? v.x : 0.f, e.y > 0.f ? v.y : 0.f, e.z > 0.f ? v.z : 0.f, e.w > 0.f ? v.w : 0.f};
      } else if (EPI == EPI_TAN) {
        out = e * v;
        out2 = e * (1.f - e) * v * ld4(g.E2 + m * g.lde2 + n);
        has2 = true;
      } else if (EPI == EPI_ADD) {
        out = v + e;
      } else if (EPI == EPI_BCE) {
        const f4 l = v + b;
        f4 sg;
        const f4 sp = softplus4(l, sg);
        // bce = max(l, 0) - l t + log1p(e^{-|l|}) = softplus(l) - l t
        const f4 bce = sp - l * e;
        float s = bce.x;
        if (n + 1 < g.N) s += bce.y;
        if (n + 2 < g.N) s += bce.z;
        if (n + 3 < g.N) s += bce.w;
        rs[j] += s;
        out = g.beta * (sg - e);
      } else {  // EPI_NET1
        const f4 t = ld4(g.tb + (long long)trow * g.N + n);
        const f4 p = v + t + e;
        out = f4{fmaxf(p.x, 0.f), fmaxf(p.y, 0.f), fmaxf(p.z, 0.f), fmaxf(p.w, 0.f)};
      }
      if (g.C != nullptr) st4(g.C + m * g.ldc + n, out);
      if (has2 && g.C2 != nullptr) st4(g.C2 + m * g.ldc2 + n, out2);
      if (g.Cp != nullptr) {                     // the same four values as bf16 planes (the next product's A operand)
        const Split4 sp = g.pm ? split4_f16(out) : split4(out);
        unsigned short* pp = g.Cp + m * g.ldcp + n;
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        if (g.pm) {                              // f16x2: two planes, X1 | X2 (X1 / 64 is formed by the consumer)
          if (full) {
            *reinterpret_cast<u2v*>(pp) = u2v{sp.h[0], sp.h[1]};
            *reinterpret_cast<u2v*>(pp + g.cp_plane) = u2v{sp.l[0], sp.l[1]};
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (n + t < g.N) {
                pp[t] = (unsigned short)(sp.h[t >> 1] >> (16 * (t & 1)));
                pp[g.cp_plane + t] = (unsigned short)(sp.l[t >> 1] >> (16 * (t & 1)));
              }
          }
        } else if (full) {
          *reinterpret_cast<u2v*>(pp) = u2v{sp.h[0], sp.h[1]};
          *reinterpret_cast<u2v*>(pp + g.cp_plane) = u2v{sp.m[0], sp.m[1]};
          *reinterpret_cast<u2v*>(pp + 2 * g.cp_plane) = u2v{sp.l[0], sp.l[1]};
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (n + t < g.N) {
              pp[t] = (unsigned short)(sp.h[t >> 1] >> (16 * (t & 1)));
              pp[g.cp_plane + t] = (unsigned short)(sp.m[t >> 1] >> (16 * (t & 1)));
              pp[2 * g.cp_plane + t] = (unsigned short)(sp.l[t >> 1] >> (16 * (t & 1)));
            }
        }
      }
    }
  }
  if (EPI == EPI_BCE) {
    // per chain m: sum over this wave's 64 columns = over the 4 lanes (q) that share column c, fixed order
#pragma unroll
    for (int j = 0; j < WMB; ++j) {
      float s = rs[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const long long m = m0 + wm + 16 * j + c;
      if (q == 0 && m < g.M) g.rowsum[m * (WAVES_N * g.n_tiles) + WAVES_N * (n0 / (16 * WNB * WAVES_N)) + (w % WAVES_N)] = g.beta * s;
    }
  }
}

// WMB x WNB: 16 x 16 MFMA tiles per wave along m and n; the 2 x 2 waves of a workgroup cover a
// (32 WMB) x (32 WNB) tile of C.  4 x 4 (128 x 128) for the big decoder products, 2 x 2 (64 x 64) for the
// H = 200 net layers (fills the chip at M = 8192), 1 x 2 (32 x 64) for the N = d = 50 latent gradient.
// WAVES_N: how the 4 waves tile the workgroup's block -- 2 x 2 (default) or 4 x 1 (each wave spans the whole width
// 16 WNB: the 112-wide tiles that divide the decoder's 784 logits exactly)
// BF3 = 1: the bf16x3 inner product above on k-tiles of 32 (one barrier per 96 bf16 MFMAs of a 64 x 64 wave block).
template <int EPI, int KV, int WMB, int WNB, int WAVES_N = 2, int BF3 = 0>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs g) {
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int TM = 16 * WMB * WAVES_M, TN = 16 * WNB * WAVES_N;
  constexpr int GK = BF3 ? 32 : l2hmc::GK, GP = GK + 4;          // (shadow the file-level k-tile for this kernel)
  __shared__ __attribute__((aligned(16))) float sA[2][TM * GP];   // activations  [m][k]
  __shared__ __attribute__((aligned(16))) float sB[2][TN * GP];   // weights      [n][k]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int wm = (w / WAVES_N) * 16 * WMB, wn = (w % WAVES_N) * 16 * WNB;    // this wave's block
  // Tile of this workgroup.  Workgroups are dispatched round-robin over the 8 XCDs in linear order (x fastest), each XCD
  // with its own L2: with the plain (n-tile, m-tile) = (blockIdx.x, blockIdx.y) mapping and 8 n-tiles per row every XCD
  // reads ALL activation rows.  XCD-aware form (m-tile count a multiple of 8): XCD j owns the m-tiles j, j + 8, ... and
  // walks their n-tiles -- an activation tile is fetched into one L2 only.
  int bx = blockIdx.x, by = blockIdx.y;
#ifndef L2HMC_GEMM_NO_XCD_MAP
  if ((gridDim.y & 7) == 0 && gridDim.y >= 16) {
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, k = lin >> 3;
    by = (int)((k / gridDim.x) * 8 + xcd);
    bx = (int)(k % gridDim.x);
  }
#endif
  const long long m0 = (long long)by * TM;
  const int n0 = bx * TN;

  // global -> register staging: a row of the k-tile is GK / 4 quads; thread loads quad (tid % QPR) of rows
  // (tid / QPR) + RPP i
  constexpr int QPR = GK / 4, RPP = 256 / QPR;
  const int lr = tid / QPR, lk = (tid % QPR) * 4;
  constexpr int NA = (TM + RPP - 1) / RPP, NB = (TN + RPP - 1) / RPP;
  f4 ra[NA], rb[NB];
  // interior tiles (every row of both operand tiles exists, K a multiple of the k-tile, 16-byte rows): plain dwordx4 loads
  // without the per-load bounds branches -- wave-uniform, decided once (bf16x3 kernel: -10 %; the f32 kernel is 3 % slower with it)
  const bool interior = BF3 != 0 && KV == 4 && TM % RPP == 0 && TN % RPP == 0 && m0 + TM <= g.M && n0 + TN <= g.N && g.K % GK == 0;
  auto gload = [&](int k0) {
    const int k = k0 + lk;
    if (interior) {
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f4*>(g.A + (m0 + lr + RPP * i) * g.lda + k);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f4*>(g.B + (long long)(n0 + lr + RPP * i) * g.ldb + k);
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const long long m = m0 + lr + RPP * i;
      ra[i] = (lr + RPP * i < TM && m < g.M) ? load_kquad<KV>(g.A + m * g.lda + k, k, g.K) : splat(0.f);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + lr + RPP * i;
      rb[i] = (lr + RPP * i < TN && n < g.N) ? load_kquad<KV>(g.B + (long long)n * g.ldb + k, k, g.K) : splat(0.f);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (lr + RPP * i < TM) *reinterpret_cast<f4*>(&sA[buf][(lr + RPP * i) * GP + lk]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (lr + RPP * i < TN) *reinterpret_cast<f4*>(&sB[buf][(lr + RPP * i) * GP + lk]) = rb[i];
  };

  f4 acc[WNB][WMB];                                                // [n block][m block]
#pragma unroll
  for (int i = 0; i < WNB; ++i)
#pragma unroll
    for (int j = 0; j < WMB; ++j) acc[i][j] = splat(0.f);

  const int nk = (g.K + GK - 1) / GK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GK);                         // in flight under the MFMAs below
    if constexpr (BF3 != 0) {
      // lane (c, q) takes k = 8 q .. 8 q + 7 of its row for BOTH operands (any k permutation common to A and B is a valid
      // contraction order): two ds_read_b128 per fragment, split in registers
      // Software pipeline inside the k-tile: the fp32 fragments are all requested up front; block i's 24 MFMAs (six
      // products x WMB independent accumulators, product-major so consecutive MFMAs never wait on each other) issue while
      // the VALU splits weight fragment i + 1 -- sched_group_barrier pins that interleaving (2 MFMA : 3 VALU).
      f4 ra_[WMB][2], rw_[WNB][2];
#pragma unroll
      for (int j = 0; j < WMB; ++j) {
        const float* pa = &sA[buf][(wm + 16 * j + c) * GP + 8 * q];
        ra_[j][0] = *reinterpret_cast<const f4*>(pa); ra_[j][1] = *reinterpret_cast<const f4*>(pa + 4);
      }
#pragma unroll
      for (int i = 0; i < WNB; ++i) {
        const float* pw = &sB[buf][(wn + 16 * i + c) * GP + 8 * q];
        rw_[i][0] = *reinterpret_cast<const f4*>(pw); rw_[i][1] = *reinterpret_cast<const f4*>(pw + 4);
      }
      Split3 sa[WMB];
#pragma unroll
      for (int j = 0; j < WMB; ++j) sa[j] = split3(ra_[j][0], ra_[j][1]);
      Split3 sw = split3(rw_[0][0], rw_[0][1]);
#pragma unroll
      for (int i = 0; i < WNB; ++i) {
        Split3 nx = sw;
        if (i + 1 < WNB) nx = split3(rw_[i + 1][0], rw_[i + 1][1]);
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = mfma_bf16(sw.l, sa[j].h, acc[i][j]);
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = mfma_bf16(sw.h, sa[j].l, acc[i][j]);
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = mfma_bf16(sw.m, sa[j].m, acc[i][j]);
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = mfma_bf16(sw.m, sa[j].h, acc[i][j]);
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = mfma_bf16(sw.h, sa[j].m, acc[i][j]);
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = mfma_bf16(sw.h, sa[j].h, acc[i][j]);
#ifndef L2HMC_BF3_NO_SGB
        if (i + 1 < WNB) {
#pragma unroll
          for (int rep = 0; rep < 3 * WMB; ++rep) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // 3 VALU
          }
        }
#endif
        sw = nx;
      }
      if (kt + 1 < nk) sstore(buf ^ 1);
      __syncthreads();
      continue;
    }
    // all fragments of the k-tile are requested up front: the second 16-wide half arrives under the MFMAs of the first
    f4 fw[GK / 16][WNB], fa[GK / 16][WMB];
#pragma unroll
    for (int h = 0; h < GK / 16; ++h) {
#pragma unroll
      for (int i = 0; i < WNB; ++i) fw[h][i] = *reinterpret_cast<const f4*>(&sB[buf][(wn + 16 * i + c) * GP + 16 * h + 4 * q]);
#pragma unroll
      for (int j = 0; j < WMB; ++j) fa[h][j] = *reinterpret_cast<const f4*>(&sA[buf][(wm + 16 * j + c) * GP + 16 * h + 4 * q]);
    }
#pragma unroll
    for (int h = 0; h < GK / 16; ++h)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < WNB; ++i)
#pragma unroll
          for (int j = 0; j < WMB; ++j) acc[i][j] = MFMA16(fw[h][i][s], fa[h][j][s], acc[i][j]);
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<EPI, WMB, WNB, WAVES_N>(g, acc, m0, n0, wm, wn, w, c, q);
}

// (A warp-specialised form -- four producer waves split every element once per workgroup into three bf16 LDS planes, four
//  consumer waves issue nothing but the MFMAs; 120 KB of double-buffered planes, one 512-thread workgroup per CU -- was built
//  and measured in round 3: identical results, 142.7 us against 123.4 us for the split in the consumer on 8192 x 1024 x 1024
//  (profiles/r03_gemm_bf16x3.txt): 144 KB of plane traffic per k-tile through the 128 B/clk LDS pipe is as long as the k-tile's
//  matrix-pipe time.  Not kept.)
// ------------------------------------------------------------------------------------------------------------
// One S/T/Q net evaluation in ONE launch (the H = 200 nets of mnist_vae.py:142-167 and any H > 15):
//   out3 = relu(relu([a | b] [W1; W2] + time/bias row + aux_h) W4 + b4) [Ws | Wt | Wq]
// A workgroup owns a tile of 16 chains (one MFMA column block); the input tile and both hidden activations stay in
// LDS ([row][K] with a row stride whose quarter is odd: conflict-free ds_read_b128 operand fetches), the weights
// (K-contiguous transposed copies, rows and K zero-padded to multiples of 16 so the loops carry no guards;
// L2-resident: 80 + 160 + 120 KB) stream straight from global memory as the MFMA A operand -- each lane fetches the
// float4 of its own (row, k-quad) for a whole output block before the MFMAs that consume them; several workgroups
// per CU hide that latency.  Three dependent GEMM launches + two HBM round trips of the activations become one
// launch with two workgroup barriers.
struct NetEvalArgs {
  const float* AB; int ldab;        // (M, 2 d): the two first-layer inputs side by side
  const float* W12t;                // (Hp, K1p)   Hp = ceil16(H), K1p = ceil16(2 d), zero padded
  const float* W4t;                 // (Hp, Hp)
  const float* Wht;                 // (N3p, Hp)   N3p = ceil16(3 d)
  const float* b4;                  // (H)
  const float* tb;                  // (T, H) time/bias table of this net
  const float* auxh;                // (M, H) image-branch term or NULL
  const unsigned char* dir; int dir_all, it, T;
  float* out3;                      // (M, 3 d)  (upd.mode == 0)
  int M, d, H;
  float *keep_h1, *keep_h2;         // (M, H) or NULL: both hidden activations also go to HBM (the trainer's reverse sweep needs them)
  float* keep_out3;                 // (M, 3 d) or NULL: with a fused update (upd.mode != 0), the raw head products ALSO go to HBM
  // The leapfrog half-update that consumes this evaluation, fused behind the heads (upd.mode != 0: the head products stay in
  // LDS and out3 is not written).  Same formulas as the stand-alone update kernels of split.hip.
  struct Update {
    int mode;                       // 0 none, 1 momentum half-update (k_v_half), 2 masked position update (k_x_half)
    const float *bs, *bt, *bq, *lam_s, *lam_q;   // head biases and log-scales of this net (d each)
    const float* alpha; float eps_host;
    float* ld;                      // (M) log-det, accumulated
    // mode 1: v_out = v_half(v_in; g);  optionally xin = k1 x for the X-net evaluation that follows
    const float* vin; int ldvi; const float* g; int ldg; float* vout; int ldvo;
    const float* x; int ldx; float* xin; int ldxi;
    // mode 2: z_out = x_half(z_in; vh), second = 0 / 1;  optionally xin_next = (1 - kept) z_out
    const float* zin; int ldzi; const float* vh; int ldvh; float* zout; int ldzo; float* xin_next; int ldxn;
    const float* masks; int second;
  } upd;
};
__host__ __device__ inline int ceil16(int k) { return (k + 15) / 16 * 16; }
__host__ __device__ inline int odd_quarter_stride(int k) {   // smallest multiple of 4 >= k whose quarter is odd
  int p = (k + 3) / 4 * 4;
  if (((p / 4) & 1) == 0) p += 4;
  return p;
}
constexpr int NE_MAXKT = 16;       // k-tiles per layer the kernel is compiled for: K <= 256
// CB = chain blocks (of 16) per workgroup: every weight fragment fetched from L2 feeds CB MFMAs
// input tile, two hidden activations, head products -- the head products take the place of the FIRST hidden activation
// (dead once layer 2 is done) whenever their rows fit: 3 d <= the hidden row stride
inline bool net_eval_out_aliases_h1(int d, int H) { return ceil16(3 * d) <= odd_quarter_stride(ceil16(H)); }
inline size_t net_eval_lds_bytes(int d, int H, int CB = 1) {
  return sizeof(float) * 16 * CB * (size_t)(odd_quarter_stride(ceil16(2 * d)) + 2 * odd_quarter_stride(ceil16(H)) +
                                           (net_eval_out_aliases_h1(d, H) ? 0 : ceil16(3 * d)));
}

// NWV = waves per workgroup (the waves share out the 16-wide output blocks of every layer): <1, 4> = 16 chains on 4 waves,
// two workgroups per CU; <2, 8> = 32 chains on 8 waves, one workgroup per CU -- the same 8 waves per CU, but every weight
// fragment streamed from L2 now feeds two MFMAs (half the L2 traffic of a net evaluation).
//
// NK1 / NKH > 0 (round 5): the k-tile counts of the first layer and of the hidden layers are COMPILE-TIME (config 5: K1p = 112,
// Hp = 208 -> <7, 13>), a wave owns at most two output blocks per layer (NWV >= half the block count), and the weight
// fragments of the NEXT block -- of the next LAYER across the barrier: they depend on no activation -- are requested before
// the MFMAs of the current one (two register sets, `sched_barrier` pins the order: left alone, the scheduler sinks every load
// to its first use).  With runtime k-tile counts (the generic form, NK1 = 0) every fragment load sits behind a uniform branch
// and the compiler waits for vmcnt(0) at each join: one L2 round trip per block with nothing to run beside it.
template <int CB, int NWV = 4, int NK1 = 0, int NKH = 0>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void net_eval_kernel(const NetEvalArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  lds_poison(sm);
  constexpr int NE_MT = 16 * CB, NTHR = 64 * NWV;
  constexpr bool PIPE = NK1 > 0;
  constexpr int NKMAX = PIPE ? (NKH > NK1 ? NKH : NK1) : 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  f4 wset[2][NKMAX];                                         // (PIPE) the two fragment sets
  auto wload = [&](auto nkc, const float* Wt, int nb, f4* wf) {
    constexpr int NK = decltype(nkc)::value;
    const float* wrow = Wt + (long long)(nb * 16 + c) * (16 * NK) + 4 * q;
#pragma unroll
    for (int j = 0; j < NK; ++j) wf[j] = *reinterpret_cast<const f4*>(wrow + 16 * j);
  };
  // NCB = CB: the whole block; NCB = 1: its chain block `cb0` only (a block shared out between two waves, see the stage plan)
  auto wcomp = [&](auto nkc, auto ncbc, int cb0, const float* As, int ldA, const f4* wf, int nb, auto&& epi) {
    constexpr int NK = decltype(nkc)::value, NCB = decltype(ncbc)::value;
    f4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = splat(0.f);
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      f4 af[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) af[cb] = *reinterpret_cast<const f4*>(As + (16 * (cb0 + cb) + c) * ldA + j * 16 + 4 * q);
#pragma unroll
      for (int s = 0; s < 4; ++s)                              // (the chain blocks' accumulators alternate; measured equal to block-major)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = MFMA16(wf[j][s], af[cb][s], acc[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) epi(nb, cb0 + cb, acc[cb]);
  };
  typedef std::integral_constant<int, CB> ICB;
  typedef std::integral_constant<int, 1> IC1;
  // second slot of wave w in a layer of NBL output blocks (the first is block w): with rem = NBL - NWV blocks left, an odd
  // rem (or rem == 2) leaves the SIMD that holds waves 0 and NWV / 2 a whole block more than the others (13 blocks on 8 waves:
  // 4 / 3 / 3 / 3) -- so the last block (both, for rem == 2) is shared out by chain block between two waves of different
  // SIMDs: 3.5 / 3.5 / 3 / 3.  Returns the block (or -1) and which chain blocks (-1: all).
  auto slot1 = [&](int NBL, int& cbh) {
    const int rem = NBL - NWV, S = (rem & 1) ? 1 : (rem == 2 ? 2 : 0), F = rem - S;
    cbh = -1;
    if (rem <= 0) return -1;
    if (CB != 2 || F + 2 * S > NWV) return w < rem ? NWV + w : -1;
    if (w < F) return NWV + w;
    if (w < F + 2 * S) { cbh = (w - F) & 1; return NWV + F + ((w - F) >> 1); }
    return -1;
  };
  typedef std::integral_constant<int, NK1> IK1;
  typedef std::integral_constant<int, NKH> IKH;
#ifdef L2HMC_NE_TIMERS          // phase stamps of two workgroups, printed by lane 0 of every wave (tools/experiments: variant build only)
  unsigned long long ne_t[8];
  int ne_n = 0;
#define NE_STAMP() do { if (ne_n < 8) ne_t[ne_n++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NE_STAMP() do {} while (0)
#endif
  NE_STAMP();
  const int K1 = 2 * g.d, H = g.H, K1p = ceil16(K1), Hp = ceil16(H);
  const int ld1 = odd_quarter_stride(K1p), ldh = odd_quarter_stride(Hp);
  float* sIn = sm;
  float* sH1 = sIn + NE_MT * ld1;
  float* sH2 = sH1 + NE_MT * ldh;
  const int ldo = ceil16(3 * g.d);
  float* sOut = ldo <= ldh ? sH1 : sH2 + NE_MT * ldh;    // (NE_MT, ceil16(3 d)) head products of the fused update
  const long long m0 = (long long)blockIdx.x * NE_MT;

  // (PIPE) the input tile's loads go out FIRST -- loads return in order, whatever is requested ahead of them delays the first
  // barrier -- and are parked in registers while everything else this wave will need is requested behind them
  constexpr int IN_Q = PIPE ? 4 * NK1 : 1, IN_U = PIPE ? (NE_MT * IN_Q + NTHR - 1) / NTHR : 1;       // quads per row; per thread
  f4 vin_[IN_U];
  if constexpr (PIPE) {
#pragma unroll
    for (int u = 0; u < IN_U; ++u) {
      const int i = tid + u * NTHR, r = i / IN_Q, kq4 = (i % IN_Q) * 4;
      const bool valid = i < NE_MT * IN_Q && m0 + r < g.M && kq4 < K1;
      vin_[u] = *reinterpret_cast<const f4*>(g.AB + (valid ? (m0 + r) * g.ldab + kq4 : 0));      // (clamped: no branch around the load)
    }
    __builtin_amdgcn_sched_barrier(0);
    wload(IK1{}, g.W12t, w, wset[0]);                          // stage 0
  }
  // schedule rows of this lane's chains: requested early -- the epilogue of layer 1 hangs on them
  bool mok[CB], fwdc[CB];
  int trow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    const long long m = m0 + 16 * cb + c;
    mok[cb] = m < g.M;
    trow[cb] = 0;
    fwdc[cb] = true;
    if (mok[cb]) {
      fwdc[cb] = g.dir != nullptr ? g.dir[m] != 0 : (g.dir_all != 0);
      trow[cb] = fwdc[cb] ? g.it : (g.T - 1 - g.it);
    }
  }
  // (PIPE) the epilogue operands of this wave's two blocks of layers 1 and 2 -- the image-branch term (an HBM round trip per
  // block when it is requested where it is used: the phase stamps of round 5 showed layer 1 at twice its matrix-pipe time),
  // BOTH candidate rows of the time/bias table (which one depends on the chain's direction, itself still in flight), b4 --
  // are requested here, under the input tile's round trip
  f4 pe[2][CB], ptF[2], ptB[2], pb[2];
  int hH = -1, nbH = -1;
  if constexpr (PIPE) {
    nbH = slot1(NKH, hH);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int nb = sl == 0 ? w : nbH;
      const int n = nb * 16 + 4 * q;
      const bool nin = nb >= 0 && n < g.H;
      ptF[sl] = nin ? *reinterpret_cast<const f4*>(g.tb + (long long)g.it * g.H + n) : splat(0.f);
      ptB[sl] = nin ? *reinterpret_cast<const f4*>(g.tb + (long long)(g.T - 1 - g.it) * g.H + n) : splat(0.f);
      pb[sl] = nin ? *reinterpret_cast<const f4*>(g.b4 + n) : splat(0.f);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
        pe[sl][cb] = (nin && mok[cb] && g.auxh != nullptr) ? *reinterpret_cast<const f4*>(g.auxh + (m0 + 16 * cb + c) * g.H + n) : splat(0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const int N3 = 3 * g.d;
  const int mode = g.upd.mode;
  if constexpr (PIPE) {
#pragma unroll
    for (int u = 0; u < IN_U; ++u) {
      const int i = tid + u * NTHR, r = i / IN_Q, kq4 = (i % IN_Q) * 4;
      if (i < NE_MT * IN_Q) *reinterpret_cast<f4*>(sIn + r * ld1 + kq4) = (m0 + r < g.M && kq4 < K1) ? vin_[u] : splat(0.f);
    }
  } else {
    for (int i = tid; i < NE_MT * (K1p / 4); i += NTHR) {        // input tile, zero padded to K1p (K1 % 4 == 0)
      const int r = i / (K1p / 4), kq4 = (i % (K1p / 4)) * 4;
      f4 v = splat(0.f);
      if (m0 + r < g.M && kq4 < K1) v = *reinterpret_cast<const f4*>(g.AB + (m0 + r) * g.ldab + kq4);
      *reinterpret_cast<f4*>(sIn + r * ld1 + kq4) = v;
    }
  }
  for (int i = tid; i < NE_MT * (ldh - H) ; i += NTHR) {       // pad columns of the hidden activations
    const int r = i / (ldh - H), k = H + i % (ldh - H);
    sH1[r * ldh + k] = 0.f;
    sH2[r * ldh + k] = 0.f;
  }
  __syncthreads();
  NE_STAMP();

  // one layer: C[m][n] = sum_k As[m][k] Wt[n][k] over the padded Kp; epi(nb, cb, acc): lane holds
  // C[m = 16 cb + c][n = 16 nb + 4 q + r]
  auto layer = [&](const float* As, int ldA, int Kp, const float* Wt, int Np, auto&& epi) {
    const int nk = Kp >> 4;
    // (requesting block nb + NWV's weight fragments before block nb's MFMAs IN THIS FORM -- two register stages behind the
    //  uniform `j < nk` branches -- was measured in round 3: 3.84 vs 3.79 ms per config-5 proposal.  With compile-time k-tile
    //  counts and pinned load order it is what the PIPE form above gains from.)
    for (int nb = w; nb * 16 < Np; nb += NWV) {
      const float* wrow = Wt + (long long)(nb * 16 + c) * Kp + 4 * q;
      f4 wf[NE_MAXKT];
#pragma unroll
      for (int j = 0; j < NE_MAXKT; ++j)
        if (j < nk) wf[j] = *reinterpret_cast<const f4*>(wrow + j * 16);
      f4 acc[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = splat(0.f);
#pragma unroll
      for (int j = 0; j < NE_MAXKT; ++j) {
        if (j < nk) {
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            const f4 af = *reinterpret_cast<const f4*>(As + (16 * cb + c) * ldA + j * 16 + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[cb] = MFMA16(wf[j][s], af[s], acc[cb]);
          }
        }
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) epi(nb, cb, acc[cb]);
    }
  };
  auto relu4f = [](f4 p) { return f4{fmaxf(p.x, 0.f), fmaxf(p.y, 0.f), fmaxf(p.z, 0.f), fmaxf(p.w, 0.f)}; };
  auto epi1p = [&](auto sc, int nb, int cb, f4 v) {           // sc: the slot (PIPE: which preloaded operands)
    constexpr int SL = decltype(sc)::value;
    const int n = nb * 16 + 4 * q;
    if (n >= H) return;                                        // (H % 4 == 0)
    f4 t = splat(0.f), e = splat(0.f);
    if constexpr (PIPE) {
      if (mok[cb]) t = fwdc[cb] ? ptF[SL] : ptB[SL];
      e = pe[SL][cb];
    } else if (mok[cb]) {
      t = *reinterpret_cast<const f4*>(g.tb + (long long)trow[cb] * H + n);
      if (g.auxh != nullptr) e = *reinterpret_cast<const f4*>(g.auxh + (m0 + 16 * cb + c) * H + n);
    }
    const f4 h = relu4f(v + t + e);
    *reinterpret_cast<f4*>(sH1 + (16 * cb + c) * ldh + n) = h;
    if (g.keep_h1 != nullptr && mok[cb]) *reinterpret_cast<f4*>(g.keep_h1 + (m0 + 16 * cb + c) * H + n) = h;
  };
  auto epi2p = [&](auto sc, int nb, int cb, f4 v) {
    constexpr int SL = decltype(sc)::value;
    const int n = nb * 16 + 4 * q;
    if (n >= H) return;
    f4 b;
    if constexpr (PIPE) b = pb[SL];
    else b = *reinterpret_cast<const f4*>(g.b4 + n);
    const f4 h = relu4f(v + b);
    *reinterpret_cast<f4*>(sH2 + (16 * cb + c) * ldh + n) = h;
    if (g.keep_h2 != nullptr && mok[cb]) *reinterpret_cast<f4*>(g.keep_h2 + (m0 + 16 * cb + c) * H + n) = h;
  };
  typedef std::integral_constant<int, 0> IS0;
  auto epi1 = [&](int nb, int cb, f4 v) { epi1p(IS0{}, nb, cb, v); };
  auto epi1b = [&](int nb, int cb, f4 v) { epi1p(IC1{}, nb, cb, v); };
  auto epi2 = [&](int nb, int cb, f4 v) { epi2p(IS0{}, nb, cb, v); };
  auto epi2b = [&](int nb, int cb, f4 v) { epi2p(IC1{}, nb, cb, v); };
  auto epi3 = [&](int nb, int cb, f4 v) {
    const int n = nb * 16 + 4 * q;
    if (mode != 0) {                                         // (n + 3 < ceil16(3 d) = ldo always)
      *reinterpret_cast<f4*>(sOut + (16 * cb + c) * ldo + n) = v;
      if (g.keep_out3 != nullptr && mok[cb]) {
        float* o = g.keep_out3 + (m0 + 16 * cb + c) * N3 + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N3) o[r] = v[r];
      }
      return;
    }
    if (!mok[cb]) return;
    float* o = g.out3 + (m0 + 16 * cb + c) * N3 + n;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < N3) o[r] = v[r];
  };
  if constexpr (PIPE) {
    // stages (layer, slot): a wave's blocks are w and `slot1` of each layer; stage k's fragments live in set k % 2 and are
    // requested one stage ahead.  (K1p == 16 NK1, Hp == 16 NKH: the host picks the instantiation.)
    const int nb0 = w, NB3 = ceil16(N3) >> 4;
    int h3;
    const int nb3 = slot1(NB3, h3);                            // second block in the head layer (hidden layers: nbH, hH above)
    const bool one3 = nb0 < NB3;
    auto second = [&](auto nkc, const float* As, int ldA, const f4* wf, int nb, int cbh, auto&& epi) {
      if (nb < 0) return;
      if (cbh < 0) wcomp(nkc, ICB{}, 0, As, ldA, wf, nb, epi);
      else wcomp(nkc, IC1{}, cbh, As, ldA, wf, nb, epi);
    };
    if (nbH >= 0) wload(IK1{}, g.W12t, nbH, wset[1]);
    __builtin_amdgcn_sched_barrier(0);
    wcomp(IK1{}, ICB{}, 0, sIn, ld1, wset[0], nb0, epi1);
    __builtin_amdgcn_sched_barrier(0);
    wload(IKH{}, g.W4t, nb0, wset[0]);
    __builtin_amdgcn_sched_barrier(0);
    second(IK1{}, sIn, ld1, wset[1], nbH, hH, epi1b);
    NE_STAMP();
    __syncthreads();
    NE_STAMP();
    if (nbH >= 0) wload(IKH{}, g.W4t, nbH, wset[1]);
    __builtin_amdgcn_sched_barrier(0);
    wcomp(IKH{}, ICB{}, 0, sH1, ldh, wset[0], nb0, epi2);
    __builtin_amdgcn_sched_barrier(0);
    if (one3) wload(IKH{}, g.Wht, nb0, wset[0]);
    __builtin_amdgcn_sched_barrier(0);
    second(IKH{}, sH1, ldh, wset[1], nbH, hH, epi2b);
    NE_STAMP();
    __syncthreads();
    NE_STAMP();
    if (nb3 >= 0) wload(IKH{}, g.Wht, nb3, wset[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (one3) wcomp(IKH{}, ICB{}, 0, sH2, ldh, wset[0], nb0, epi3);
    second(IKH{}, sH2, ldh, wset[1], nb3, h3, epi3);
  } else {
    layer(sIn, ld1, K1p, g.W12t, Hp, epi1);
    __syncthreads();
    layer(sH1, ldh, Hp, g.W4t, Hp, epi2);
    __syncthreads();
    layer(sH2, ldh, Hp, g.Wht, ceil16(N3), epi3);
  }
  if (mode == 0) return;
  NE_STAMP();
  __syncthreads();
  NE_STAMP();
  // ---- fused half-update: TPC = 16 / CB threads per chain, dimensions strided by TPC; the chain's log-det share is
  // reduced over its TPC lanes in a fixed order
  const NetEvalArgs::Update& U = g.upd;
  constexpr int TPC = NTHR / NE_MT;
  const int r = tid / TPC, j = tid % TPC;
  const long long n = m0 + r;
  const bool ok = n < g.M;
  const int d = g.d;
  bool fwd = true;
  int srow = 0;
  if (ok) {
    fwd = g.dir != nullptr ? g.dir[n] != 0 : (g.dir_all != 0);
    srow = fwd ? g.it : (g.T - 1 - g.it);
  }
  const float eps = U.alpha != nullptr ? expf(*U.alpha) : U.eps_host, sgn = fwd ? 1.f : -1.f;
  const float* o3 = sOut + r * ldo;
  float acc = 0.f;
  const float heps = 0.5f * eps;
  auto upd_v = [&](int k, float vi, float gk, float xk) {      // momentum half-update of dimension k (k_v_half of split.hip)
    const float S = expf(U.lam_s[k]) * tanhf(o3[k] + U.bs[k]);
    const float T_ = o3[d + k] + U.bt[k];
    const float Q = expf(U.lam_q[k]) * tanhf(o3[2 * d + k] + U.bq[k]);
    const float sv = sgn * heps * S, ES = expf(sv), EQ = expf(eps * Q);
    const float cc = heps * (T_ - EQ * gk);
    U.vout[n * U.ldvo + k] = fwd ? vi * ES + cc : (vi - cc) * ES;
    acc += sv;
    if (U.xin != nullptr) {
      const float mk = U.masks[srow * d + k];
      U.xin[n * U.ldxi + k] = (fwd ? mk : 1.f - mk) * xk;
    }
  };
  auto upd_x = [&](int k, float zi, float vhk) {               // masked position update of dimension k (k_x_half)
    const float mk = U.masks[srow * d + k];
    const float k1 = fwd ? mk : 1.f - mk;
    const float kp = U.second ? 1.f - k1 : k1, up = 1.f - kp;
    const float S = expf(U.lam_s[k]) * tanhf(o3[k] + U.bs[k]);
    const float T_ = o3[d + k] + U.bt[k];
    const float Q = expf(U.lam_q[k]) * tanhf(o3[2 * d + k] + U.bq[k]);
    const float sx = sgn * eps * S, ES = expf(sx), EQ = expf(eps * Q);
    const float tr = eps * (EQ * vhk + T_);
    const float nw = fwd ? zi * ES + tr : ES * (zi - tr);
    const float zo = kp * zi + up * nw;
    U.zout[n * U.ldzo + k] = zo;
    if (U.xin_next != nullptr) U.xin_next[n * U.ldxn + k] = up * zo;
    acc += up * sx;
  };
  // (Requesting these operands at the top of the kernel, or right behind its first barrier, was measured in round 5: the update
  //  phase shrinks by 1200 cycles and the phase that issues the twelve extra loads per thread grows by as much.)
  if (ok) {
    if (mode == 1) {
      for (int k = j; k < d; k += TPC) upd_v(k, U.vin[n * U.ldvi + k], U.g[n * U.ldg + k], U.xin != nullptr ? U.x[n * U.ldx + k] : 0.f);
    } else {
      for (int k = j; k < d; k += TPC) upd_x(k, U.zin[n * U.ldzi + k], U.vh[n * U.ldvh + k]);
    }
  }
#pragma unroll
  for (int off = TPC / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (ok && j == 0) U.ld[n] += acc;
#ifdef L2HMC_NE_TIMERS
  {
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    if (PIPE && lane == 0 && (blockIdx.x == 3 || blockIdx.x == 131) && g.it == 2)
      printf("NE wg %d w %d mode %d : in %llu  L1 %llu bar %llu  L2 %llu bar %llu  L3 %llu bar %llu  upd %llu  total %llu\n", (int)blockIdx.x, w, mode,
             ne_t[1] - ne_t[0], ne_t[2] - ne_t[1], ne_t[3] - ne_t[2], ne_t[4] - ne_t[3], ne_t[5] - ne_t[4], ne_t[6] - ne_t[5],
             ne_t[7] - ne_t[6], te - ne_t[7], te - ne_t[0]);
  }
#endif
}

// which instantiation runs: 32 chains on 8 waves from `cb` == 2; the software-pipelined form for the shapes it is compiled for
inline bool net_eval_piped(int cb, int d, int H) { return cb == 2 && ceil16(2 * d) == 16 * 7 && ceil16(H) == 16 * 13; }
inline const void* net_eval_fn(int cb, int d, int H) {
  if (net_eval_piped(cb, d, H)) return reinterpret_cast<const void*>(net_eval_kernel<2, 8, 7, 13>);
  return cb == 2 ? reinterpret_cast<const void*>(net_eval_kernel<2, 8>) : reinterpret_cast<const void*>(net_eval_kernel<1, 4>);
}
inline void launch_net_eval(int cb, unsigned blocks, size_t lds, hipStream_t s, const NetEvalArgs& na) {
  if (net_eval_piped(cb, na.d, na.H)) hipLaunchKernelGGL((net_eval_kernel<2, 8, 7, 13>), dim3(blocks), dim3(512), lds, s, na);
  else if (cb == 2) hipLaunchKernelGGL((net_eval_kernel<2, 8>), dim3(blocks), dim3(512), lds, s, na);
  else hipLaunchKernelGGL((net_eval_kernel<1, 4>), dim3(blocks), dim3(256), lds, s, na);
}

// ------------------------------------------------------------------------------------------------------------
// The reverse of one S/T/Q net evaluation in ONE launch (the trainer's data path, train_split.hpp `net_bwd`):
//   d a2 = (d out3 [Ws | Wt | Wq]^T) [h2 > 0],   d a1 = (d a2 W4^T) [h1 > 0],   d [a | b] = d a1 [W1; W2]^T
// -- net_eval_kernel run backwards: the same 16 (x CB) chains per workgroup, the cotangent tile and both hidden
// cotangents resident in LDS, the weights (as stored = K-contiguous for these products; zero-padded copies, rows and K to
// multiples of 16) streamed from L2 as the MFMA A operand.  d a2 and d a1 also go to HBM: they are the B operands of the
// weight-gradient contractions over all (evaluation, chain) rows that follow the sweep.
struct NetBwdArgs {
  const float* dO3; int ldo;        // (M, 3 d) cotangents of the raw head products (d zs | d zt | d zq)
  const float* Whc;                 // (Hp, K3p)   [Ws | Wt | Wq] side by side, K3p = ceil16(3 d)
  const float* W4;                  // (Hp, Hp)    W4 as stored (in, out)
  const float* W12;                 // (K1p, Hp)   [W1; W2] stacked, K1p = ceil16(2 d)
  const float *h2, *h1;             // (M, H) the forward activations (relu masks)
  float *da2, *da1;                 // (M, H)
  float* dAB; int ldab;             // (M, 2 d)
  int M, d, H;
};
inline size_t net_bwd_lds_bytes(int d, int H, int CB = 1) {
  return sizeof(float) * 16 * CB * (size_t)(odd_quarter_stride(ceil16(3 * d)) + 2 * odd_quarter_stride(ceil16(H)));
}

// NK3 / NKH > 0: the software-pipelined form of net_eval_kernel<CB, NWV, NK1, NKH> (compile-time k-tile counts, fragments a stage
// ahead, the odd block shared out between two SIMDs, the relu masks of both blocks requested ahead of the first barrier) for
// ceil16(3 d) == 16 NK3 and ceil16(H) == 16 NKH.
template <int CB, int NWV = 4, int NK3 = 0, int NKH = 0>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void net_bwd_kernel(const NetBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  lds_poison(sm);
  constexpr int NE_MT = 16 * CB, NTHR = 64 * NWV;
  constexpr bool PIPE = NK3 > 0;
  constexpr int NKMAX = PIPE ? (NKH > NK3 ? NKH : NK3) : 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int N3 = 3 * g.d, K1 = 2 * g.d, H = g.H, K3p = ceil16(N3), K1p = ceil16(K1), Hp = ceil16(H);
  const int ld3 = odd_quarter_stride(K3p), ldh = odd_quarter_stride(Hp);
  float* sIn = sm;
  float* sD2 = sIn + NE_MT * ld3;
  float* sD1 = sD2 + NE_MT * ldh;
  const long long m0 = (long long)blockIdx.x * NE_MT;
  bool mok[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) mok[cb] = m0 + 16 * cb + c < g.M;
  auto masked = [](f4 v, f4 h) { return f4{h.x > 0.f ? v.x : 0.f, h.y > 0.f ? v.y : 0.f, h.z > 0.f ? v.z : 0.f, h.w > 0.f ? v.w : 0.f}; };

  if constexpr (PIPE) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef std::integral_constant<int, NK3> IK3;
    typedef std::integral_constant<int, NKH> IKH;
    typedef std::integral_constant<int, CB> ICB;
    typedef std::integral_constant<int, 1> IC1;
    typedef std::integral_constant<int, 0> IS0;
    f4 wset[2][NKMAX];
    auto wload = [&](auto nkc, const float* Wt, int nb, f4* wf) {
      constexpr int NK = decltype(nkc)::value;
      const float* wrow = Wt + (long long)(nb * 16 + c) * (16 * NK) + 4 * q;
#pragma unroll
      for (int j = 0; j < NK; ++j) wf[j] = *reinterpret_cast<const f4*>(wrow + 16 * j);
    };
    auto wcomp = [&](auto nkc, auto ncbc, int cb0, const float* As, int ldA, const f4* wf, int nb, auto&& epi) {
      constexpr int NK = decltype(nkc)::value, NCB = decltype(ncbc)::value;
      f4 acc[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = splat(0.f);
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        f4 af[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) af[cb] = *reinterpret_cast<const f4*>(As + (16 * (cb0 + cb) + c) * ldA + j * 16 + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb] = MFMA16(wf[j][s], af[cb][s], acc[cb]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) epi(nb, cb0 + cb, acc[cb]);
    };
    auto slot1 = [&](int NBL, int& cbh) {                      // as in net_eval_kernel
      const int rem = NBL - NWV, S = (rem & 1) ? 1 : (rem == 2 ? 2 : 0), F = rem - S;
      cbh = -1;
      if (rem <= 0) return -1;
      if (CB != 2 || F + 2 * S > NWV) return w < rem ? NWV + w : -1;
      if (w < F) return NWV + w;
      if (w < F + 2 * S) { cbh = (w - F) & 1; return NWV + F + ((w - F) >> 1); }
      return -1;
    };
    auto second = [&](auto nkc, const float* As, int ldA, const f4* wf, int nb, int cbh, auto&& epi) {
      if (nb < 0) return;
      if (cbh < 0) wcomp(nkc, ICB{}, 0, As, ldA, wf, nb, epi);
      else wcomp(nkc, IC1{}, cbh, As, ldA, wf, nb, epi);
    };
    // the cotangent tile's loads first (3 d even: 8-byte row segments), parked in registers; then stage 0 and the masks
    constexpr int IN_P = 8 * NK3, IN_U = (NE_MT * IN_P + NTHR - 1) / NTHR;       // pairs per row; per thread
    f2 vin_[IN_U];
#pragma unroll
    for (int u = 0; u < IN_U; ++u) {
      const int i = tid + u * NTHR, r = i / IN_P, k2 = (i % IN_P) * 2;
      const bool valid = i < NE_MT * IN_P && m0 + r < g.M && k2 < N3;
      vin_[u] = *reinterpret_cast<const f2*>(g.dO3 + (valid ? (m0 + r) * g.ldo + k2 : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    const int nb0 = w;
    int hH, hC;
    const int nbH = slot1(NKH, hH), NBC = K1p >> 4, nbC = slot1(NBC, hC);
    wload(IK3{}, g.Whc, nb0, wset[0]);
    f4 pm2[2][CB], pm1[2][CB];                                 // relu masks (the forward activations) of this wave's two blocks
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      const int nb = sl == 0 ? nb0 : nbH;
      const int n = nb * 16 + 4 * q;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const bool v = nb >= 0 && n < H && mok[cb];
        const long long at = v ? (m0 + 16 * cb + c) * H + n : 0;
        pm2[sl][cb] = *reinterpret_cast<const f4*>(g.h2 + at);
        pm1[sl][cb] = *reinterpret_cast<const f4*>(g.h1 + at);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < IN_U; ++u) {
      const int i = tid + u * NTHR, r = i / IN_P, k2 = (i % IN_P) * 2;
      if (i < NE_MT * IN_P)
        *reinterpret_cast<f2*>(sIn + r * ld3 + k2) = (m0 + r < g.M && k2 < N3) ? vin_[u] : f2{0.f, 0.f};
    }
    for (int i = tid; i < NE_MT * (ldh - H); i += NTHR) {      // pad columns of the hidden cotangents
      const int r = i / (ldh - H), k = H + i % (ldh - H);
      sD2[r * ldh + k] = 0.f;
      sD1[r * ldh + k] = 0.f;
    }
    __syncthreads();
    auto epiA = [&](auto sc, int nb, int cb, f4 v) {
      constexpr int SL = decltype(sc)::value;
      const int n = nb * 16 + 4 * q;
      if (n >= H) return;
      f4 o = splat(0.f);
      if (mok[cb]) {
        o = masked(v, pm2[SL][cb]);
        *reinterpret_cast<f4*>(g.da2 + (m0 + 16 * cb + c) * H + n) = o;
      }
      *reinterpret_cast<f4*>(sD2 + (16 * cb + c) * ldh + n) = o;
    };
    auto epiB = [&](auto sc, int nb, int cb, f4 v) {
      constexpr int SL = decltype(sc)::value;
      const int n = nb * 16 + 4 * q;
      if (n >= H) return;
      f4 o = splat(0.f);
      if (mok[cb]) {
        o = masked(v, pm1[SL][cb]);
        *reinterpret_cast<f4*>(g.da1 + (m0 + 16 * cb + c) * H + n) = o;
      }
      *reinterpret_cast<f4*>(sD1 + (16 * cb + c) * ldh + n) = o;
    };
    auto epiC = [&](int nb, int cb, f4 v) {
      const int n = nb * 16 + 4 * q;
      if (!mok[cb] || n >= K1) return;
      *reinterpret_cast<f4*>(g.dAB + (m0 + 16 * cb + c) * g.ldab + n) = v;
    };
    auto epiA0 = [&](int nb, int cb, f4 v) { epiA(IS0{}, nb, cb, v); };
    auto epiA1 = [&](int nb, int cb, f4 v) { epiA(IC1{}, nb, cb, v); };
    auto epiB0 = [&](int nb, int cb, f4 v) { epiB(IS0{}, nb, cb, v); };
    auto epiB1 = [&](int nb, int cb, f4 v) { epiB(IC1{}, nb, cb, v); };
    const bool oneC = nb0 < NBC;
    if (nbH >= 0) wload(IK3{}, g.Whc, nbH, wset[1]);
    __builtin_amdgcn_sched_barrier(0);
    wcomp(IK3{}, ICB{}, 0, sIn, ld3, wset[0], nb0, epiA0);
    __builtin_amdgcn_sched_barrier(0);
    wload(IKH{}, g.W4, nb0, wset[0]);
    __builtin_amdgcn_sched_barrier(0);
    second(IK3{}, sIn, ld3, wset[1], nbH, hH, epiA1);
    __syncthreads();
    if (nbH >= 0) wload(IKH{}, g.W4, nbH, wset[1]);
    __builtin_amdgcn_sched_barrier(0);
    wcomp(IKH{}, ICB{}, 0, sD2, ldh, wset[0], nb0, epiB0);
    __builtin_amdgcn_sched_barrier(0);
    if (oneC) wload(IKH{}, g.W12, nb0, wset[0]);
    __builtin_amdgcn_sched_barrier(0);
    second(IKH{}, sD2, ldh, wset[1], nbH, hH, epiB1);
    __syncthreads();
    if (nbC >= 0) wload(IKH{}, g.W12, nbC, wset[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (oneC) wcomp(IKH{}, ICB{}, 0, sD1, ldh, wset[0], nb0, epiC);
    second(IKH{}, sD1, ldh, wset[1], nbC, hC, epiC);
    return;
  }

  for (int i = tid; i < NE_MT * K3p; i += NTHR) {            // cotangent tile, zero padded to K3p (3 d need not be a multiple of 4)
    const int r = i / K3p, k = i % K3p;
    sIn[r * ld3 + k] = (m0 + r < g.M && k < N3) ? g.dO3[(m0 + r) * g.ldo + k] : 0.f;
  }
  for (int i = tid; i < NE_MT * (ldh - H); i += NTHR) {      // pad columns of the hidden cotangents
    const int r = i / (ldh - H), k = H + i % (ldh - H);
    sD2[r * ldh + k] = 0.f;
    sD1[r * ldh + k] = 0.f;
  }
  __syncthreads();

  auto layer = [&](const float* As, int ldA, int Kp, const float* Wt, int Np, auto&& epi) {      // as in net_eval_kernel
    const int nk = Kp >> 4;
    for (int nb = w; nb * 16 < Np; nb += NWV) {
      const float* wrow = Wt + (long long)(nb * 16 + c) * Kp + 4 * q;
      f4 wf[NE_MAXKT];
#pragma unroll
      for (int j = 0; j < NE_MAXKT; ++j)
        if (j < nk) wf[j] = *reinterpret_cast<const f4*>(wrow + j * 16);
      f4 acc[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) acc[cb] = splat(0.f);
#pragma unroll
      for (int j = 0; j < NE_MAXKT; ++j) {
        if (j < nk) {
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            const f4 af = *reinterpret_cast<const f4*>(As + (16 * cb + c) * ldA + j * 16 + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[cb] = MFMA16(wf[j][s], af[s], acc[cb]);
          }
        }
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) epi(nb, cb, acc[cb]);
    }
  };

  layer(sIn, ld3, K3p, g.Whc, Hp, [&](int nb, int cb, f4 v) {
    const int n = nb * 16 + 4 * q;
    if (n >= H) return;                                        // (H % 4 == 0)
    f4 o = splat(0.f);
    if (mok[cb]) {
      const long long at = (m0 + 16 * cb + c) * H + n;
      o = masked(v, *reinterpret_cast<const f4*>(g.h2 + at));
      *reinterpret_cast<f4*>(g.da2 + at) = o;
    }
    *reinterpret_cast<f4*>(sD2 + (16 * cb + c) * ldh + n) = o;
  });
  __syncthreads();
  layer(sD2, ldh, Hp, g.W4, Hp, [&](int nb, int cb, f4 v) {
    const int n = nb * 16 + 4 * q;
    if (n >= H) return;
    f4 o = splat(0.f);
    if (mok[cb]) {
      const long long at = (m0 + 16 * cb + c) * H + n;
      o = masked(v, *reinterpret_cast<const f4*>(g.h1 + at));
      *reinterpret_cast<f4*>(g.da1 + at) = o;
    }
    *reinterpret_cast<f4*>(sD1 + (16 * cb + c) * ldh + n) = o;
  });
  __syncthreads();
  layer(sD1, ldh, Hp, g.W12, K1p, [&](int nb, int cb, f4 v) {
    const int n = nb * 16 + 4 * q;
    if (!mok[cb] || n >= K1) return;                           // (2 d % 4 == 0)
    *reinterpret_cast<f4*>(g.dAB + (m0 + 16 * cb + c) * g.ldab + n) = v;
  });
}
inline bool net_bwd_piped(int cb, int d, int H) { return cb == 2 && d % 2 == 0 && ceil16(3 * d) == 16 * 10 && ceil16(H) == 16 * 13; }
inline const void* net_bwd_fn(int cb, int d, int H) {
  if (net_bwd_piped(cb, d, H)) return reinterpret_cast<const void*>(net_bwd_kernel<2, 8, 10, 13>);
  return cb == 2 ? reinterpret_cast<const void*>(net_bwd_kernel<2, 8>) : reinterpret_cast<const void*>(net_bwd_kernel<1, 4>);
}
inline void launch_net_bwd(int cb, unsigned blocks, size_t lds, hipStream_t s, const NetBwdArgs& nb) {
  if (net_bwd_piped(cb, nb.d, nb.H)) hipLaunchKernelGGL((net_bwd_kernel<2, 8, 10, 13>), dim3(blocks), dim3(512), lds, s, nb);
  else if (cb == 2) hipLaunchKernelGGL((net_bwd_kernel<2, 8>), dim3(blocks), dim3(512), lds, s, nb);
  else hipLaunchKernelGGL((net_bwd_kernel<1, 4>), dim3(blocks), dim3(256), lds, s, nb);
}


// ------------------------------------------------------------------------------------------------------------
// Weight-gradient products of the GEMM-engine trainer (train_split.hpp): contraction over the ROWS of two
// row-major activations ("TN" form),
//   part[z][i][j] = sum_{r in row chunk z} A[r][i] B[r][j],        A (R x I), B (R x J), R = chains x evaluations.
// A 256-thread workgroup owns a 64 x 64 tile of one chunk; each wave a 32 x 32 quadrant (2 x 2 MFMA tiles).  Row
// tiles of 16 are staged in LDS exactly as they lie in memory ([r][i], row stride 68: the 4 row segments a wave
// reads per ds_read_b32 fall into distinct banks), double buffered.  The chunks are added in chunk order by
// tn_reduce_kernel: no atomics, the gradient is bitwise reproducible.
struct GemmTnArgs {
  const float* A; int lda;
  const float* B; int ldb;
  long long R; int I, J;
  float* part;                   // (n_chunks, I, J)
  long long rows_per_chunk;      // multiple of 16
};
constexpr int TN_P = 68;

// VA / VB: what a row quad of A / B loads as -- 4: one dwordx4 (16-byte rows, column count a multiple of 4), 2: two dwordx2
// (8-byte rows, even column count: the 3 d = 150 head cotangents), 1: guarded scalars.  Both >= 2: the branchless two-deep form.
template <int VA, int VB>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTnArgs g) {
  constexpr bool V4 = VA == 4 && VB == 4;                      // (the guarded loop's dwordx4 fast path)
  constexpr bool DEEP = VA >= 2 && VB >= 2;
  __shared__ __attribute__((aligned(16))) float sA[2][16 * TN_P];
  __shared__ __attribute__((aligned(16))) float sB[2][16 * TN_P];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const long long r_begin = (long long)blockIdx.z * g.rows_per_chunk;
  long long r_end = r_begin + g.rows_per_chunk;
  if (r_end > g.R) r_end = g.R;
  const int lr = tid >> 4, lc = (tid & 15) * 4;              // this thread's (row, column quad) of a 16 x 64 tile
  f4 ra, rb;
  auto ldq = [&](const float* base, int ld, long long r, int col, int ncol) {
    f4 v = splat(0.f);
    if (r < r_end) {
      const float* p = base + r * ld + col;
      if (V4) {
        if (col + 3 < ncol) return *reinterpret_cast<const f4*>(p);
      }
      if (col + 0 < ncol) v.x = p[0];
      if (col + 1 < ncol) v.y = p[1];
      if (col + 2 < ncol) v.z = p[2];
      if (col + 3 < ncol) v.w = p[3];
    }
    return v;
  };
  auto gload = [&](long long r0) {
    ra = ldq(g.A, g.lda, r0 + lr, i0 + lc, g.I);
    rb = ldq(g.B, g.ldb, r0 + lr, j0 + lc, g.J);
  };
  auto sstore = [&](int buf) {
    *reinterpret_cast<f4*>(&sA[buf][lr * TN_P + lc]) = ra;
    *reinterpret_cast<f4*>(&sB[buf][lr * TN_P + lc]) = rb;
  };
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
  f4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = splat(0.f);
  const long long nt = (r_end - r_begin + 15) / 16;
  // 16 x 16 blocks of this wave that lie wholly beyond the matrix (I, J = 200 on 64-wide tiles: 87 of 256 blocks) are skipped:
  // wave-uniform, decided once
  bool live[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) live[a][b] = i0 + wi + 16 * a < g.I && j0 + wj + 16 * b < g.J;
  auto compute = [&](int buf) {
    float fa[2][4], fb[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a][s] = sA[buf][(4 * q + s) * TN_P + wi + 16 * a + c];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b][s] = sB[buf][(4 * q + s) * TN_P + wj + 16 * b + c];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (live[a][b]) acc[a][b] = MFMA16(fa[a][s], fb[b][s], acc[a][b]);
  };
  if constexpr (DEEP) {
    // (round 5) no branch around a load -- rows / quads (pairs) beyond the matrix are clamped to element 0 and zeroed when they
    // are stored -- and TWO row tiles in flight in registers (tile k in slot k & 1): with one, a 16-row tile's MFMAs (512 cycles
    // per wave) are a fraction of the round trip that feeds the next.
    typedef float f2 __attribute__((ext_vector_type(2)));
    f4 pa[2], pb[2];
    bool oka[2][2], okb[2][2];                                 // [slot][low / high pair of the quad]
    auto ld1 = [&](auto vc, const float* base, long long ld, long long r, int col, int ncol, f4& v, bool (&ok)[2]) {
      constexpr int V = decltype(vc)::value;
      ok[0] = r < r_end && col < ncol;
      ok[1] = r < r_end && col + 2 < ncol;
      if constexpr (V == 4) {
        v = *reinterpret_cast<const f4*>(base + (ok[0] ? r * ld + col : 0));
      } else {
        const f2 lo = *reinterpret_cast<const f2*>(base + (ok[0] ? r * ld + col : 0));
        const f2 hi = *reinterpret_cast<const f2*>(base + (ok[1] ? r * ld + col + 2 : 0));
        v = f4{lo.x, lo.y, hi.x, hi.y};
      }
    };
    auto gl = [&](auto sc, long long r0) {
      constexpr int S = decltype(sc)::value;
      ld1(std::integral_constant<int, VA>{}, g.A, g.lda, r0 + lr, i0 + lc, g.I, pa[S], oka[S]);
      ld1(std::integral_constant<int, VB>{}, g.B, g.ldb, r0 + lr, j0 + lc, g.J, pb[S], okb[S]);
    };
    auto st = [&](auto sc, int buf) {
      constexpr int S = decltype(sc)::value;
      const f4 a = pa[S], b = pb[S];
      *reinterpret_cast<f4*>(&sA[buf][lr * TN_P + lc]) = f4{oka[S][0] ? a.x : 0.f, oka[S][0] ? a.y : 0.f, oka[S][1] ? a.z : 0.f, oka[S][1] ? a.w : 0.f};
      *reinterpret_cast<f4*>(&sB[buf][lr * TN_P + lc]) = f4{okb[S][0] ? b.x : 0.f, okb[S][0] ? b.y : 0.f, okb[S][1] ? b.z : 0.f, okb[S][1] ? b.w : 0.f};
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    if (nt > 0) {
      gl(S0{}, r_begin);
      st(S0{}, 0);
      if (nt > 1) gl(S1{}, r_begin + 16);
      if (nt > 2) gl(S0{}, r_begin + 32);
    }
    __syncthreads();
    for (long long t = 0; t < nt; t += 2) {
      compute(0);                                              // tile t (even) from buffer 0
      if (t + 1 < nt) st(S1{}, 1);                             // tile t + 1 waits in slot 1
      if (t + 3 < nt) gl(S1{}, r_begin + (t + 3) * 16);
      __syncthreads();
      if (t + 1 < nt) {
        compute(1);
        if (t + 2 < nt) st(S0{}, 0);
        if (t + 4 < nt) gl(S0{}, r_begin + (t + 4) * 16);
        __syncthreads();
      }
    }
  } else {
    if (nt > 0) {
      gload(r_begin);
      sstore(0);
    }
    __syncthreads();
    for (long long t = 0; t < nt; ++t) {
      const int buf = (int)(t & 1);
      if (t + 1 < nt) gload(r_begin + (t + 1) * 16);
      compute(buf);
      if (t + 1 < nt) sstore(buf ^ 1);
      __syncthreads();
    }
  }
  // lane holds D[i = i0 + wi + 16 a + 4 q + r][j = j0 + wj + 16 b + c]
  float* out = g.part + (long long)blockIdx.z * g.I * g.J;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int j = j0 + wj + 16 * b + c;
      if (j >= g.J) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi + 16 * a + 4 * q + r;
        if (i < g.I) out[(long long)i * g.J + j] = acc[a][b][r];
      }
    }
}

// dst (+)= sum_z part[z][i][j], z ascending (fixed order).  The (I x J) result may be a stack of row blocks and / or
// column blocks that live apart in the destination (e.g. [W1; W2] of a net, or [Ws | Wt | Wq]):
//   element (i, j) -> dst[(i / iblock) * istride + (j / jblock) * jstride + (i % iblock) * ldd + (j % jblock)]
struct TnScatter { int iblock; long long istride; int jblock; long long jstride; };
// sum over z of p[z * stride] in chunk order -- the order every gradient's reproducibility rests on -- with the loads of 32
// chunks in flight at a time (a plain loop leaves the compiler ~4: 128 chunks were 25 us of pure L2 latency per reduction,
// eleven reductions per training step)
__device__ __forceinline__ float sum_chunks(const float* p, int n_chunks, long long stride) {
  float s = 0.f;
  int z = 0;
  for (; z + 32 <= n_chunks; z += 32) {
    float t[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = p[(long long)(z + u) * stride];
#pragma unroll
    for (int u = 0; u < 32; ++u) s += t[u];
  }
  if (z + 8 <= n_chunks) {
    for (; z + 8 <= n_chunks; z += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = p[(long long)(z + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
  }
  for (; z < n_chunks; ++z) s += p[(long long)z * stride];
  return s;
}
__global__ void tn_reduce_kernel(const float* part, int n_chunks, int I, int J, float* dst, int ldd, int accumulate,
                                 TnScatter sc) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)I * J) return;
  const float s = sum_chunks(part + e, n_chunks, (long long)I * J);
  const int i = (int)(e / J), j = (int)(e % J);
  float* o = dst + (long long)(i / sc.iblock) * sc.istride + (long long)(j / sc.jblock) * sc.jstride +
             (long long)(i % sc.iblock) * ldd + (j % sc.jblock);
  *o = accumulate ? *o + s : s;
}

// chunking of a TN product / column sum: chunks of >= 512 rows (multiple of 16), at most 64 of them and at most
// `part_cap` floats of partial results
inline void tn_chunks(long long R, long long IJ, long long part_cap, int& n_chunks, long long& rows_per_chunk) {
  long long n = (R + 511) / 512;
  if (n > 64) n = 64;
  if (n * IJ > part_cap) n = part_cap / IJ;
  if (n < 1) n = 1;
  rows_per_chunk = ((R + n - 1) / n + 15) / 16 * 16;
  n_chunks = (int)((R + rows_per_chunk - 1) / rows_per_chunk);
  if (n_chunks < 1) n_chunks = 1;
}

// dst (I x J, row stride ldd) (+)= A^T B; `part` holds at least part_cap floats (>= I * J)
inline void launch_gemm_tn(hipStream_t s, const float* A, int lda, const float* B, int ldb, long long R, int I, int J,
                           float* dst, int ldd, int accumulate, float* part, long long part_cap,
                           const TnScatter* scatter = nullptr) {
  if (I <= 0 || J <= 0) return;
  const TnScatter sc = scatter ? *scatter : TnScatter{I, 0, J, 0};
  GemmTnArgs g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.R = R; g.I = I; g.J = J; g.part = part;
  int nc;
  tn_chunks(R, (long long)I * J, part_cap, nc, g.rows_per_chunk);
  const dim3 grid((unsigned)((J + 63) / 64), (unsigned)((I + 63) / 64), (unsigned)nc);
  auto width = [](const float* p, int ld, int cols) {          // widest load a row quad of this operand can take
    if (ld % 4 == 0 && cols % 4 == 0 && (reinterpret_cast<size_t>(p) & 15) == 0) return 4;
    if (ld % 2 == 0 && cols % 2 == 0 && (reinterpret_cast<size_t>(p) & 7) == 0) return 2;
    return 1;
  };
  const int va = width(A, lda, I), vb = width(B, ldb, J);
  if (va == 4 && vb == 4) hipLaunchKernelGGL((gemm_tn_kernel<4, 4>), grid, dim3(256), 0, s, g);
  else if (va == 4 && vb == 2) hipLaunchKernelGGL((gemm_tn_kernel<4, 2>), grid, dim3(256), 0, s, g);
  else if (va >= 2 && vb >= 2) hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_tn_kernel<1, 1>), grid, dim3(256), 0, s, g);
  const long long n = (long long)I * J;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nc, I, J, dst, ldd, accumulate, sc);
}

// rowsum layout of EPI_BCE: (M, 2 * n_tiles) with n_tiles = ceil(N / (32 WNB)) of the shape the launcher picks
template <int EPI, int WMB, int WNB, int WAVES_N = 2>
int launch_gemm_shape(const GemmArgs& g, hipStream_t s) {
  constexpr int TM = 16 * WMB * (4 / WAVES_N), TN = 16 * WNB * WAVES_N;
  const dim3 grid((unsigned)((g.N + TN - 1) / TN), (unsigned)((g.M + TM - 1) / TM));
  const bool al16 = ((reinterpret_cast<size_t>(g.A) | reinterpret_cast<size_t>(g.B)) & 15) == 0;
  const bool al8 = ((reinterpret_cast<size_t>(g.A) | reinterpret_cast<size_t>(g.B)) & 7) == 0;
  constexpr bool bf3_shape = (WMB == 4 && WNB == 4) || (WMB == 2 && WNB == 7);      // the decoder-sized products only
  constexpr bool bf3_epi = EPI == EPI_BIAS_SOFTPLUS || EPI == EPI_BCE || EPI == EPI_MUL || EPI == EPI_TAN || EPI == EPI_BIAS;
  if (g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && al16) {
    if constexpr (bf3_shape && bf3_epi) {
      if (g.bf3) {
        hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, WMB, WNB, WAVES_N, 1>), grid, dim3(256), 0, s, g);
        return L2HMC_OK;
      }
    }
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, WMB, WNB, WAVES_N>), grid, dim3(256), 0, s, g);
  } else if (g.K % 2 == 0 && g.lda % 2 == 0 && g.ldb % 2 == 0 && al8)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, WMB, WNB, WAVES_N>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, 1, WMB, WNB, WAVES_N>), grid, dim3(256), 0, s, g);
  return L2HMC_OK;
}

// 128 x 128, 64 x 64, 32 x 64 workgroup tiles; SHAPE_W112: 128 x 112 (4 x 1 waves of 2 x 7 MFMA tiles) for widths that
// are multiples of 112 but not of 128 -- the decoder's 784 = 7 x 112 logits, where 128-wide tiles waste 1/8 of the work
// ------------------------------------------------------------------------------------------------------------
// The N = d <= 64 products with a long K (the latent gradient d z = d a1 W1^T + z of the decoder posterior, K = 1024, and the
// same shape inside its Hessian-vector product):  C = A B^T + E.  On the 32 x 64 tiles of gemm_nt_kernel this is 256 workgroups
// of four waves walking 64 k-tiles with ONE tile of loads in flight each: 38.7 us for 33.5 MB of A at 8192 chains (0.87 TB/s)
// -- latency, not bandwidth or the matrix pipe.  Here a workgroup owns 16 rows of A and its four waves a QUARTER of K each
// (all NBLK column blocks): a wave requests its whole A fragment of a chunk of 16 k-tiles up front (16 dwordx4 per lane in
// flight, 8 waves per CU), streams the weight fragments from L2 one k-tile ahead, and the four partial sums meet in LDS in
// wave order (fixed order: reproducible).  No staging of A in LDS -- every element is used by exactly one wave.
// (Measured at 19.6 us per 8192 x 50 x 1024 product against a matrix-pipe floor of 8-9: weight fragments 1 / 3 / 7 k-tiles ahead,
// 16 rows on four waves or 32 on eight, the weight rows at a stride that is not a power of two, chunks of 2 k-tiles requested one
// chunk ahead instead of a wave's whole share up front -- all within 2 us of each other.  Phase stamps: a wave's k loop takes
// 7400-10 000 cycles, its FIRST k-tile arrives after 10 000-12 000 (waves 0-3) or 24 000-25 000 (waves 4-7, which issue second on
// their SIMD): 33.5 MB in 64-byte pieces, 16 rows x 4 KB apart per load, at 1.7 TB/s -- the access pattern, not the schedule.)
// Contract: K % 256 == 0 (chunks of 4 k-tiles per wave; of 16 when K % 1024 == 0), lda / ldb % 4 == 0, 16-byte aligned A and B,
// N <= 16 NBLK.
template <int NBLK, int MB, int NW, int CH>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_add_kernel(const GemmArgs g) {
  static_assert(NW == 4 || NW == 8, "K is split over 4 or 8 waves");
  static_assert(MB * NBLK <= NW, "one wave per output block in the epilogue");
  __shared__ __attribute__((aligned(16))) f4 red[4][MB * NBLK][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const long long m0 = (long long)blockIdx.x * (16 * MB);
  const int nkq = (g.K >> 4) / NW;                               // k-tiles of 16 per wave (K % (16 NW CH) == 0)
  const int jb = w * nkq, je = jb + nkq;
  // No predicates in the loop: rows beyond M and weight rows beyond N are CLAMPED to row 0 -- an output element depends on its
  // own row of A and its own row of B only, and the elements of those rows / columns are never stored.  (With a guard per load
  // the compiler branches around every one of them and waits for vmcnt(0) at each join.)
  const float* arow[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) arow[b] = g.A + (m0 + 16 * b + c < g.M ? m0 + 16 * b + c : 0) * g.lda + 4 * q;
  const float* wrow[NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i) wrow[i] = g.B + (long long)(16 * i + c < g.N ? 16 * i + c : 0) * g.ldb + 4 * q;
  f4 acc[MB][NBLK];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[b][i] = splat(0.f);
  constexpr int WD = CH < 4 ? CH : 4;                            // ring of weight fragments, WD - 1 k-tiles ahead of the MFMAs
  for (int j0 = jb; j0 < je; j0 += CH) {
    f4 af[MB][CH];
#pragma unroll
    for (int u = 0; u < CH; ++u)
#pragma unroll
      for (int b = 0; b < MB; ++b) af[b][u] = *reinterpret_cast<const f4*>(arow[b] + 16 * (j0 + u));
    f4 wf[WD][NBLK];
#pragma unroll
    for (int t = 0; t < WD - 1; ++t)
#pragma unroll
      for (int i = 0; i < NBLK; ++i) wf[t][i] = *reinterpret_cast<const f4*>(wrow[i] + 16 * (j0 + t));
    __builtin_amdgcn_sched_barrier(0);     // (the scheduler otherwise sinks every load to its first use: one round trip per k-tile)
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (u + WD - 1 < CH) {
#pragma unroll
        for (int i = 0; i < NBLK; ++i) wf[(u + WD - 1) % WD][i] = *reinterpret_cast<const f4*>(wrow[i] + 16 * (j0 + u + WD - 1));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NBLK; ++i)
#pragma unroll
          for (int b = 0; b < MB; ++b) acc[b][i] = MFMA16(wf[u % WD][i][s], af[b][u][s], acc[b][i]);
    }
  }
  // the NW partial sums in a fixed order: (w + 4) onto w first (eight waves), then 0 + 1 + 2 + 3 by the epilogue threads
  if (NW == 8) {
    if (w >= 4) {
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int i = 0; i < NBLK; ++i) red[w - 4][b * NBLK + i][lane] = acc[b][i];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int i = 0; i < NBLK; ++i) acc[b][i] = acc[b][i] + red[w][b * NBLK + i][lane];
    }
    __syncthreads();
  }
  if (w < 4) {
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int i = 0; i < NBLK; ++i) red[w][b * NBLK + i][lane] = acc[b][i];
  }
  __syncthreads();
  // thread (block = w, lane): lane holds C[m0 + 16 b + c][16 i + 4 q + (0..3)] of block (b, i) = (w / NBLK, w % NBLK)
  const long long m = m0 + 16 * (w / NBLK) + c;
  if (w < MB * NBLK && m < g.M) {
    f4 v = red[0][w][lane];
    v = v + red[1][w][lane];
    v = v + red[2][w][lane];
    v = v + red[3][w][lane];
    const int n = 16 * (w % NBLK) + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < g.N) g.C[m * g.ldc + n + r] = v[r] + (g.E != nullptr ? g.E[m * g.lde + n + r] : 0.f);
  }
}
inline bool gemm_skinny_ok(const GemmArgs& g) {
  return g.N <= 64 && g.K % 256 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && g.bias == nullptr && g.C != nullptr &&
         g.Cp == nullptr && ((reinterpret_cast<size_t>(g.A) | reinterpret_cast<size_t>(g.B)) & 15) == 0;
}

enum { SHAPE_BIG = 0, SHAPE_MID = 1, SHAPE_SKINNY = 2, SHAPE_AUTO = 3, SHAPE_W112 = 4 };
inline int gemm_tile_n(int shape) { return shape == SHAPE_BIG ? 128 : (shape == SHAPE_W112 ? 112 : 64); }
inline int gemm_waves_n(int shape) { return shape == SHAPE_W112 ? 1 : 2; }
// 128 x 128 tiles when they fill the 256 CUs, else 64 x 64 (a 512-chain batch -- the reference's training batch -- gives
// only 4 x 8 big tiles of a 1024-wide layer)
inline int gemm_auto_shape(long long M, int N) {
  if (((M + 127) / 128) * ((N + 127) / 128) < 192) return SHAPE_MID;
  return (N % 112 == 0 && N % 128 != 0) ? SHAPE_W112 : SHAPE_BIG;
}

template <int EPI>
int launch_gemm(const GemmArgs& g, hipStream_t s, int shape = SHAPE_AUTO) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return L2HMC_OK;
  if (shape == SHAPE_AUTO) shape = gemm_auto_shape(g.M, g.N);
  if (shape == SHAPE_W112) {
    if (EPI == EPI_BCE || EPI == EPI_MUL) return launch_gemm_shape<EPI == EPI_BCE ? EPI_BCE : EPI_MUL, 2, 7, 1>(g, s);
    shape = SHAPE_BIG;                      // (only the logits-shaped products are instantiated for it)
  }
  if (shape == SHAPE_BIG) return launch_gemm_shape<EPI, 4, 4>(g, s);
  if (shape == SHAPE_MID) return launch_gemm_shape<EPI, 2, 2>(g, s);
  if constexpr (EPI == EPI_ADD) {
#ifndef L2HMC_NO_SKINNY_SPLITK
    if (gemm_skinny_ok(g)) {
      // 32 rows on eight waves (every weight fragment from L2 feeds two MFMAs) once that still gives every CU a workgroup
      if (g.K % 1024 == 0 && g.M >= 32LL * 200) {
        const dim3 grid((unsigned)((g.M + 31) / 32));
        if (g.N <= 32) hipLaunchKernelGGL((gemm_skinny_add_kernel<2, 2, 8, 8>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_skinny_add_kernel<4, 2, 8, 8>), grid, dim3(512), 0, s, g);
      } else {
        const dim3 grid((unsigned)((g.M + 15) / 16));
        if (g.N <= 32) hipLaunchKernelGGL((gemm_skinny_add_kernel<2, 1, 4, 4>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_skinny_add_kernel<4, 1, 4, 4>), grid, dim3(256), 0, s, g);
      }
      return L2HMC_OK;
    }
#endif
  }
  return launch_gemm_shape<EPI, 1, 2>(g, s);
}

}  // namespace l2hmc
