// traj_lane.hpp -- ONE CHAIN PER LANE: the fused trajectory / sampler-loop kernel for the many-chain regime.
//
// Same algorithm and argument block as traj_kernel (reference: utils/dynamics.py:115-309, utils/sampler.py:28-55), a
// different decomposition.  On gfx950 the fp32 VECTOR peak equals the fp32 MATRIX peak (157 TFLOP/s: the f32 MFMA runs
// on the VALU's multipliers, which is also why MFMA and VALU time add on a SIMD).  For the H ~ 10 nets of
// the reference the MFMA tile therefore buys only one thing -- parallelism INSIDE a chain when chains are scarce -- and
// costs padding (11 -> 16 hidden rows, 50 -> 64 dimensions, d = 2 -> 16) plus three cross-wave sums per leapfrog step.
// When there are enough chains to give every SIMD a wave of 64 of them, this form wins:
//   * a lane owns a whole chain: x, v, grad U live in its registers (DP = padded dimension count, compile time);
//   * every weight, bias, mask entry and energy parameter is WAVE-UNIFORM: it arrives by scalar loads (s_load_dwordxN,
//     scalar cache / L2) and enters the packed FMA as an SGPR pair -- v_pk_fma_f32 v[acc], s[w:w+1], v[x] op_sel_hi --
//     two hidden units (or two output dimensions) per instruction, no LDS, no barrier, no exchange, no padding beyond
//     the even pair;
//   * the chain's direction only changes per-lane scalars (f = 1 / 0, the time encoding, which of the two mask rows).
// tools/ubench_lane_per_chain.hip: 86 TFLOP/s on the net's matrix work against ~41 for the tiles.  A wave is 64
// chains, so the dispatcher takes this kernel only when the chains fill the chip (l2hmc_abi.hip).
//
// Lane layout of a net's weights (floats; rows padded to RS = 4 ceil(H / 4) so a row starts 16-byte aligned; written
// by pack_lane_kernel behind the MFMA fragments of l2hmc_pack_nets):
//   L1A [d][RS]   W1[k][j]        L1B [d][RS]   W2[k][j]
//   TB  [3][RS]   W3[0][j], W3[1][j], b1 + b2 + b3
//   L2  [H][RS]   W4[j][i]        B4 [RS]
//   HD  [ceil(d / 2)][6 H + 12]   per dimension pair: Ws[j][k0..k1], Wt[j][k0..k1], Wq[j][k0..k1] (j major), then
//                                 bs, bt, bq, e^{lam_s}, e^{lam_q} pairs, 2 pad
#pragma once
#include "l2hmc_kernels.hpp"

namespace l2hmc {

typedef float f2 __attribute__((ext_vector_type(2)));

// the dimension counts the kernels are compiled for (DP): the layout carries DP zero-padded rows / pairs and HU = 10 or
// 16 zero-padded hidden units, so that NO loop of the kernel has a runtime bound
__host__ __device__ inline int lane_dp(int d) {
  return d <= 2 ? 2 : d <= 4 ? 4 : d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : d <= 50 ? 50 : 64;
}
__host__ __device__ inline int lane_hu(int H) { return H <= 10 ? 10 : 16; }
struct LaneLayout { int RS, HU, DP, l1a, l1b, tb, l2, b4, hd, HB, total; };
__host__ __device__ inline LaneLayout lane_layout(int d, int H) {
  LaneLayout L;
  L.HU = lane_hu(H);
  L.DP = lane_dp(d);
  L.RS = (L.HU + 3) / 4 * 4;
  L.l1a = 0;
  L.l1b = L.DP * L.RS;
  L.tb = 2 * L.DP * L.RS;
  L.l2 = L.tb + 3 * L.RS;
  L.b4 = L.l2 + L.HU * L.RS;
  L.hd = L.b4 + L.RS;
  L.HB = 6 * L.HU + 12;
  L.total = (L.hd + (L.DP / 2) * L.HB + 3) / 4 * 4;
  return L;
}

// the same offsets as compile-time constants of a kernel instance (DP dimensions, HPR hidden pairs): every weight address
// is then base + immediate, no SGPR holds layout arithmetic
template <int DP, int HPR>
struct LaneL {
  static constexpr int HU = 2 * HPR, RS = (HU + 3) / 4 * 4;
  static constexpr int l1a = 0, l1b = DP * RS, tb = 2 * DP * RS, l2 = tb + 3 * RS, b4 = l2 + HU * RS, hd = b4 + RS;
  static constexpr int HB = 6 * HU + 12;
};

__device__ __forceinline__ f2 splat2(float a) { return f2{a, a}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 ld2(const float* p) { return *reinterpret_cast<const f2*>(p); }
__device__ __forceinline__ f2 ex2_2(f2 a) { return f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ f2 rcp2(f2 a) { return f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
// relu on the integer view: one v_max_i32 (fmaxf needs a canonicalising second v_max_f32)
__device__ __forceinline__ f2 relu2(f2 a) {
  return f2{__int_as_float(max(__float_as_int(a.x), 0)), __int_as_float(max(__float_as_int(a.y), 0))};
}
// tanh(z) = 1 - 2 / (1 + 2^(2 z log2 e))
__device__ __forceinline__ f2 tanh2(f2 z) { return 1.f - 2.f * rcp2(1.f + ex2_2(z * 2.8853900817779268f)); }
// keeps the (loop-invariant) wave-uniform loads inside their loop: hoisted, they would need thousands of SGPRs and
// come back as v_readlane spills (tools/ubench_lane_per_chain.hip: 20 instead of 86 TFLOP/s)
#define LANE_NO_HOIST() asm volatile("" ::: "memory")
// fine-grained fences (per layer-1 row / hidden unit / head) for wide states; for d <= 4 one fence per net evaluation
// lets the scheduler overlap the scalar loads of a whole evaluation (a few hundred floats)
#define LANE_FENCE_FINE(DPv) do { if constexpr ((DPv) > LANE_COARSE_DP) LANE_NO_HOIST(); } while (0)
#ifndef LANE_COARSE_DP
#define LANE_COARSE_DP 4
#endif
// (a single fence per leapfrog step was tried for d = 2: the scheduler then hoists across evaluations and spills SGPRs
//  through v_readlane again -- 272 cross-lane moves per step)
#define LANE_EVAL_FENCE() LANE_NO_HOIST()

// RES (round 6): where the weights of the step loop live.  At one wave per SIMD -- 65 536 chains of a d <= 2 target = 1024 waves --
// nothing hides the scalar loads: the step loop of <GMM, 2, 5> holds 137 s_load_* and 49 s_waitcnt lgkmcnt(0) beside ~720 VALU
// instructions, ~4000 cycles of issue and ~4500 of waiting per leapfrog step (8500 measured; profiles/r06_lane_resident.txt).
// A wave-uniform value lives in SGPRs by the compiler's choice (100 of them: the whole budget); the kernel uses 84 of 256 VGPRs.
//   RES = 0  weights by scalar loads inside the loop (rounds 2-5; any DP, HPR)
//   RES = 1  XNet's layer-2 and head pairs (evaluated twice per step: a third of the weight traffic) passed through v_mov_b64
//            once per launch -- ordinary VGPR pairs, which v_pk_fma_f32 takes as it takes the SGPR pairs (160 registers)
//   RES = 2  EVERY weight of both nets in VGPRs, FOUR per register: lane l holds float 4 i + (l & 3) of the net's lane layout in
//            register i, and the FMA reads it through a DPP quad broadcast -- v_fmac_f32_dpp acc, w4, x quad_perm:[k,k,k,k] --
//            at the issue cost of a v_pk_fma_f32 (tools/ubench_dpp.hip: 5.13 cycles both).  Twice the FMA instructions (one
//            hidden unit each instead of two), no scalar load and no s_waitcnt left in a net evaluation; 2 x 72 registers at
//            d = 2, H = 10.  The FMAs run in the order of the packed form on the same operands: results are bit-identical.
template <int HPR, int NP, int RES>
struct LaneRes {};
template <int HPR, int NP>
struct LaneRes<HPR, NP, 1> {
  f2 l2[2 * HPR][HPR];
  f2 hd[NP][3][2 * HPR];
};
__device__ __forceinline__ f2 to_vgpr2(f2 sv) {
  f2 v;
  asm("v_mov_b64 %0, %1" : "=v"(v) : "s"(sv));
  return v;
}
template <int DP, int HPR, int RES>
__device__ __forceinline__ void lane_res_load(const float* __restrict__ W, LaneRes<HPR, DP / 2, RES>& R) {
  if constexpr (RES == 1) {
    using L = LaneL<DP, HPR>;
#pragma unroll
    for (int j = 0; j < 2 * HPR; ++j)
#pragma unroll
      for (int i = 0; i < HPR; ++i) R.l2[j][i] = to_vgpr2(ld2(W + L::l2 + j * L::RS + 2 * i));
#pragma unroll
    for (int p = 0; p < DP / 2; ++p)
#pragma unroll
      for (int hd = 0; hd < 3; ++hd)
#pragma unroll
        for (int j = 0; j < 2 * HPR; ++j) R.hd[p][hd][j] = to_vgpr2(ld2(W + L::hd + p * L::HB + hd * 4 * HPR + 2 * j));
  }
}

// ---- RES = 2: four weights per VGPR, broadcast inside the FMA -----------------------------------------------------------------
// (EXEC is full wherever these run: dead lanes shadow chain 0, the step loop has no divergent branch.  The packed registers are
//  written once per launch, so the DPP read-after-VALU-write wait states never apply inside the loop.)
#define L2HMC_DPP_Q(k) " quad_perm:[" #k "," #k "," #k "," #k "] row_mask:0xf bank_mask:0xf"
__device__ __forceinline__ void fmac_q(float& acc, float w4, int q, float x) {      // acc += w4[quad lane q] * x  (q folds to a constant)
  switch (q & 3) {
    case 0: asm("v_fmac_f32_dpp %0, %1, %2" L2HMC_DPP_Q(0) : "+v"(acc) : "v"(w4), "v"(x)); break;
    case 1: asm("v_fmac_f32_dpp %0, %1, %2" L2HMC_DPP_Q(1) : "+v"(acc) : "v"(w4), "v"(x)); break;
    case 2: asm("v_fmac_f32_dpp %0, %1, %2" L2HMC_DPP_Q(2) : "+v"(acc) : "v"(w4), "v"(x)); break;
    default: asm("v_fmac_f32_dpp %0, %1, %2" L2HMC_DPP_Q(3) : "+v"(acc) : "v"(w4), "v"(x)); break;
  }
}
__device__ __forceinline__ float bcast_q(float w4, int q) {
  float r;
  switch (q & 3) {
    case 0: asm("v_mov_b32_dpp %0, %1" L2HMC_DPP_Q(0) : "=v"(r) : "v"(w4)); break;
    case 1: asm("v_mov_b32_dpp %0, %1" L2HMC_DPP_Q(1) : "=v"(r) : "v"(w4)); break;
    case 2: asm("v_mov_b32_dpp %0, %1" L2HMC_DPP_Q(2) : "=v"(r) : "v"(w4)); break;
    default: asm("v_mov_b32_dpp %0, %1" L2HMC_DPP_Q(3) : "=v"(r) : "v"(w4)); break;
  }
  return r;
}
template <int DP, int HPR>
struct LaneDpp {
  static constexpr int NF = (LaneL<DP, HPR>::hd + (DP / 2) * LaneL<DP, HPR>::HB + 3) / 4 * 4, NR = NF / 4;
  float r[NR];
  __device__ __forceinline__ void load(const float* __restrict__ W, int lane) {
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = W[4 * i + (lane & 3)];
  }
  __device__ __forceinline__ void fmac(float& acc, int o, float x) const { fmac_q(acc, r[o >> 2], o, x); }   // float o of the layout
  __device__ __forceinline__ float get(int o) const { return bcast_q(r[o >> 2], o); }
};
// lane_hidden / lane_heads below on the packed registers: the same sums in the same order, one hidden unit per instruction
template <int DP, int HPR, class FA, class FB>
__device__ __forceinline__ void lane_hidden_dpp(const LaneDpp<DP, HPR>& W, FA&& a_of, FB&& b_of, float tc, float ts, f2 (&h2)[HPR]) {
  using L = LaneL<DP, HPR>;
  float acc[2 * HPR], o2[2 * HPR];
  // (one pass over the hidden units per term: consecutive instructions write different accumulators -- a lone wave pays the
  //  latency of every dependent pair; the order of the terms per accumulator is the packed form's)
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j) acc[j] = W.get(L::tb + 2 * L::RS + j);
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j) W.fmac(acc[j], L::tb + L::RS + j, ts);
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j) W.fmac(acc[j], L::tb + j, tc);
#pragma unroll
  for (int k = 0; k < DP; ++k) {
    const float ak = a_of(k), bk = b_of(k);
#pragma unroll
    for (int j = 0; j < 2 * HPR; ++j) W.fmac(acc[j], L::l1b + k * L::RS + j, bk);
#pragma unroll
    for (int j = 0; j < 2 * HPR; ++j) W.fmac(acc[j], L::l1a + k * L::RS + j, ak);
  }
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j) acc[j] = __int_as_float(max(__float_as_int(acc[j]), 0));
#pragma unroll
  for (int i = 0; i < 2 * HPR; ++i) o2[i] = W.get(L::b4 + i);
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j)
#pragma unroll
    for (int i = 0; i < 2 * HPR; ++i) W.fmac(o2[i], L::l2 + j * L::RS + i, acc[j]);
#pragma unroll
  for (int i = 0; i < HPR; ++i) h2[i] = relu2(f2{o2[2 * i], o2[2 * i + 1]});
}
template <int DP, int HPR>
__device__ __forceinline__ void lane_heads_dpp(const LaneDpp<DP, HPR>& W, int p, const f2 (&h2)[HPR], f2& S, f2& T, f2& Q) {
  using L = LaneL<DP, HPR>;
  const int hb = L::hd + p * L::HB, cb = hb + 12 * HPR;
  float z[6];                                    // (S, T, Q) x (dimension pair): six accumulators side by side
#pragma unroll
  for (int u = 0; u < 6; ++u) z[u] = W.get(cb + u);
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j) {
    const float hj = (j & 1) ? h2[j >> 1].y : h2[j >> 1].x;
#pragma unroll
    for (int u = 0; u < 6; ++u) W.fmac(z[u], hb + (u >> 1) * 4 * HPR + 2 * j + (u & 1), hj);
  }
  S = f2{W.get(cb + 6), W.get(cb + 7)} * tanh2(f2{z[0], z[1]});
  T = f2{z[2], z[3]};
  Q = f2{W.get(cb + 8), W.get(cb + 9)} * tanh2(f2{z[4], z[5]});
}

// grad U (and U) of one chain held by one lane; every parameter is wave-uniform.  prec / mu as the fused kernels get
// them: diagonal precisions (d); MFMA-packed symmetric precisions (pack_gauss_kernel) for the dense kinds.
// wave-uniform read-only tables: separate `const float* __restrict__` kernel parameters, so that the compiler may use
// scalar loads for them (pointers inside the by-value argument block are loaded with per-lane vector loads)
template <int EK, int DP>
__device__ __forceinline__ float lane_grad(const KArgs& A, const float* __restrict__ MU, const float* __restrict__ PREC,
                                           const float* __restrict__ LOGC, const float (&x)[DP], float (&g)[DP], bool wantU) {
  float U = 0.f;
  const int d = A.d;
  // (energy parameter arrays have d entries: padded dimensions read a clamped index and contribute nothing)
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
#pragma unroll
    for (int k = 0; k < DP; ++k) {            // (grad U itself is formed where it is used: g_of)
      const int kk = k < d ? k : d - 1;
      const float pk = k < d ? PREC[kk] : 0.f, dx = x[k] - MU[kk];
      U = fmaf(0.5f * dx, pk * dx, U);
    }
  } else if constexpr (EK == L2HMC_ENERGY_ROUGHWELL) {
    const float eta = A.eta, den = A.den, scale = eta / den;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const float arg = x[k] / den, lv = k < d ? 1.f : 0.f;
      g[k] = x[k] - scale * rw_sin1(arg);        // (a padded dimension holds x = 0: g = 0)
      if (wantU) U += 0.5f * x[k] * x[k] + lv * eta * rw_cos1(arg);
    }
  } else {
    // dense Gaussian / mixture, DP <= 16 (one 16 x 16 zero-padded tile of the packed symmetric precision per component):
    // G[a][b] at ((a + 16 (b >> 2)) * 4 + (b & 3))
    const int nc = EK == L2HMC_ENERGY_GMM ? A.ncomp : 1;
    float m = -INFINITY, ssum = 0.f;
    float ga[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) ga[k] = 0.f;
    for (int comp = 0; comp < nc; ++comp) {
      LANE_NO_HOIST();
      const float* G = PREC + (size_t)comp * gauss_floats(1);
      const float* mu = MU + comp * d;
      float dx[DP], y[DP], q = 0.f;
#pragma unroll
      for (int b = 0; b < DP; ++b) dx[b] = x[b] - mu[b < d ? b : d - 1];
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        y[a] = 0.f;
#pragma unroll
        for (int b = 0; b < DP; ++b) y[a] = fmaf(G[(a + 16 * (b >> 2)) * 4 + (b & 3)], dx[b], y[a]);
        q = fmaf(dx[a], y[a], q);
      }
      if constexpr (EK == L2HMC_ENERGY_GAUSS_DENSE) {
#pragma unroll
        for (int a = 0; a < DP; ++a) g[a] = y[a];
        U = 0.5f * q;
      } else {
        // online softmax over the components (reduce_logsumexp semantics: a -inf log-weight contributes nothing)
        const float V = -0.5f * q + LOGC[comp];
        const float mn = fmaxf(m, V);
        const float sc = (m == mn) ? 1.f : expf(m - mn), wi = (V == -INFINITY) ? 0.f : expf(V - mn);
        ssum = ssum * sc + wi;
#pragma unroll
        for (int a = 0; a < DP; ++a) ga[a] = ga[a] * sc + wi * y[a];
        m = mn;
      }
    }
    if constexpr (EK == L2HMC_ENERGY_GMM) {
      const float inv = 1.f / ssum;
#pragma unroll
      for (int a = 0; a < DP; ++a) g[a] = ga[a] * inv;
      U = -(m + logf(ssum));
    }
  }
  if (A.temperature != 1.f) {
    U = U / A.temperature;
    if constexpr (EK != L2HMC_ENERGY_GAUSS_DIAG) {
#pragma unroll
      for (int k = 0; k < DP; ++k) g[k] = g[k] / A.temperature;
    }
  }
  return U;
}

// hidden activations h2 (HPR pairs) of net W at inputs (a_k, b_k) and the lane's time encoding (tc, ts); every bound is a
// compile-time constant (the layout is zero padded to DP rows and 2 HPR units)
template <int DP, int HPR, int RES = 0, class FA, class FB>
__device__ __forceinline__ void lane_hidden(const float* __restrict__ W, FA&& a_of, FB&& b_of, float tc, float ts,
                                            f2 (&h2)[HPR], const LaneRes<HPR, DP / 2, RES>* R = nullptr) {
  using L = LaneL<DP, HPR>;
  f2 acc[HPR];
  {
    const float* tb = W + L::tb;
#pragma unroll
    for (int j = 0; j < HPR; ++j)
      acc[j] = fma2(ld2(tb + 2 * j), splat2(tc), fma2(ld2(tb + L::RS + 2 * j), splat2(ts), ld2(tb + 2 * L::RS + 2 * j)));
  }
#pragma unroll
  for (int k = 0; k < DP; ++k) {
    LANE_FENCE_FINE(DP);
    const f2 ak = splat2(a_of(k)), bk = splat2(b_of(k));
    const float* ra = W + L::l1a + k * L::RS;
    const float* rb = W + L::l1b + k * L::RS;
#pragma unroll
    for (int j = 0; j < HPR; ++j) acc[j] = fma2(ld2(ra + 2 * j), ak, fma2(ld2(rb + 2 * j), bk, acc[j]));
  }
#pragma unroll
  for (int j = 0; j < HPR; ++j) acc[j] = relu2(acc[j]);
#pragma unroll
  for (int i = 0; i < HPR; ++i) h2[i] = ld2(W + L::b4 + 2 * i);
#pragma unroll
  for (int j = 0; j < 2 * HPR; ++j) {
    LANE_FENCE_FINE(DP);
    const f2 hj = splat2((j & 1) ? acc[j >> 1].y : acc[j >> 1].x);
    const float* r = W + L::l2 + j * L::RS;
#pragma unroll
    for (int i = 0; i < HPR; ++i) {
      if constexpr (RES == 1) h2[i] = fma2(R->l2[j][i], hj, h2[i]);
      else h2[i] = fma2(ld2(r + 2 * i), hj, h2[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < HPR; ++i) h2[i] = relu2(h2[i]);
}

// heads of dimension pair p: S = e^{lam_s} tanh(.), T, Q = e^{lam_q} tanh(.); one head at a time (2 HPR SGPR pairs live)
template <int DP, int HPR, int RES = 0>
__device__ __forceinline__ void lane_heads(const float* __restrict__ W, int p, const f2 (&h2)[HPR], f2& S, f2& T, f2& Q,
                                           const LaneRes<HPR, DP / 2, RES>* R = nullptr) {
  using L = LaneL<DP, HPR>;
  const float* hb = W + L::hd + p * L::HB;
  const float* cb = hb + 12 * HPR;
  f2 z[3];
#pragma unroll
  for (int hd = 0; hd < 3; ++hd) {
    LANE_FENCE_FINE(DP);
    f2 acc = ld2(cb + 2 * hd);
#pragma unroll
    for (int j = 0; j < 2 * HPR; ++j) {
      const f2 hj = splat2((j & 1) ? h2[j >> 1].y : h2[j >> 1].x);
      if constexpr (RES == 1) acc = fma2(R->hd[p][hd][j], hj, acc);
      else acc = fma2(ld2(hb + hd * 4 * HPR + 2 * j), hj, acc);
    }
    z[hd] = acc;
  }
  LANE_FENCE_FINE(DP);
  S = ld2(cb + 6) * tanh2(z[0]);
  T = z[1];
  Q = ld2(cb + 8) * tanh2(z[2]);
}

template <int EK, int DP, int HPR, int RES = 0>
__global__ __launch_bounds__(64, RES != 0 ? 2 : (DP <= 8 ? 4 : 2)) void traj_lane_kernel(const KArgs A, const float* __restrict__ WX,
                                                                        const float* __restrict__ WV,
                                                                        const float* __restrict__ MASKS,
                                                                        const float* __restrict__ TRIG,
                                                                        const float* __restrict__ MU,
                                                                        const float* __restrict__ PREC,
                                                                        const float* __restrict__ LOGC) {
  static_assert(DP % 2 == 0, "dimension pairs");
  const int lane = threadIdx.x;
  const long long chain = (long long)blockIdx.x * 64 + lane;
  const bool live = chain < A.N;
  const long long row = live ? chain : 0;          // dead lanes shadow chain 0 and never store
  const int d = A.d;
  const float LOG2E = 1.4426950408889634f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr || A.u != nullptr || (A.rng_flags & L2HMC_RNG_U) != 0;
  const bool rng_v = (A.rng_flags & L2HMC_RNG_V) != 0, rng_d = (A.rng_flags & L2HMC_RNG_DIR) != 0;
  const bool rng_u = (A.rng_flags & L2HMC_RNG_U) != 0;
  const bool have_u = A.u != nullptr || rng_u;
  const long long gchain = A.chain_off + chain;

  float x[DP], v[DP], g[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) x[k] = (k < d) ? A.x[row * d + k] : 0.f;
  float U_cur = lane_grad<EK, DP>(A, MU, PREC, LOGC, x, g, need_p);
  LaneRes<HPR, DP / 2, RES == 1 ? 1 : 0> RX;
  lane_res_load<DP, HPR, RES == 1 ? 1 : 0>(WX, RX);
  LaneDpp<DP, RES == 2 ? HPR : 1> PX, PV;             // (RES = 2; otherwise never touched: no registers)
  if constexpr (RES == 2) { PX.load(WX, lane); PV.load(WV, lane); }
  if constexpr (DP > 16) {                 // x_next doubles as the current-state copy a rejected chain resumes from
    if (A.x_next != nullptr && live)
#pragma unroll
      for (int k = 0; k < DP; ++k)
        if (k < d) A.x_next[chain * d + k] = x[k];
  }

  for (int m = 0; m < A.M; ++m) {
    const long long moff = (long long)m * A.N;
    const unsigned long long prop = A.rng_prop0 + (unsigned long long)m;
    // ---- draws: momentum, direction, accept uniform ----------------------------------------------------------
    if (rng_v) {
#pragma unroll
      for (int b = 0; b < (DP + 3) / 4; ++b) {
        if (4 * b < d) {
          const f4 n4 = philox_normal4(A.rng_seed, gchain, (unsigned)b, prop);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * b + r < DP) v[4 * b + r] = (4 * b + r < d) ? n4[r] : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * b + r < DP) v[4 * b + r] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < DP; ++k) v[k] = (k < d) ? A.v[(moff + row) * d + k] : 0.f;
    }
    bool fwd = (A.dir != nullptr && !rng_d) ? (A.dir[moff + row] != 0) : (A.dir_all != 0);
    float u_m = (A.u != nullptr && !rng_u) ? A.u[moff + row] : 0.f;
    if (rng_d || rng_u) {
      bool fr;
      float ur;
      philox_dir_u(A.rng_seed, gchain, prop, fr, ur);
      if (rng_d) fwd = fr;
      if (rng_u) u_m = ur;
    }
    const float ff = fwd ? 1.f : 0.f, nf = ff - 1.f, sgn = fwd ? 1.f : -1.f;
    float K0 = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) K0 = fmaf(0.5f * v[k], v[k], K0);
    const float U0 = U_cur;
    float U1 = U_cur;
    float ld = 0.f;                         // log-det, natural log units
    float x0[DP <= 16 ? DP : 1];            // start point of the proposal (small DP: registers; else re-read from x_next)
    if constexpr (DP <= 16) {
#pragma unroll
      for (int k = 0; k < DP; ++k) x0[k] = x[k];
    }

    for (int it = 0; it < A.n_steps; ++it) {
      const int sf = A.step_begin + it, sb = A.T - 1 - sf;       // schedule rows of the two directions
      // both directions' scalars are loaded (wave-uniform), the lane selects VALUES (a select between the two
      // addresses would turn every one of these into a per-lane vector load)
      const float tcf = TRIG[2 * sf], tsf = TRIG[2 * sf + 1], tcb = TRIG[2 * sb], tsb = TRIG[2 * sb + 1];
      const float tc = fwd ? tcf : tcb, ts = fwd ? tsf : tsb;
      const float* mf = MASKS + sf * d;
      const float* mb = MASKS + sb * d;
      auto k1_of = [&](int k) {                                    // forward keeps m first, backward 1 - m
        const int kk = k < d ? k : d - 1;                          // (padded dimensions never move: any value does)
        const float a = mf[kk], bq = 1.f - mb[kk];
        return fwd ? a : bq;
      };
      f2 h2[HPR];
      // grad U of dimension k at the current x: the diagonal Gaussian's is two wave-uniform scalars away, no array
      auto g_of = [&](int k) {
        if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
          const int kk = k < d ? k : d - 1;
          const float pk = k < d ? PREC[kk] : 0.f;
          return pk * (x[k] - MU[kk]);       // (the dispatcher takes this kernel at temperature 1 only: l2hmc_abi.hip, lane_able)
        } else {
          return g[k];
        }
      };
      // momentum half-update with V(z, grad U(z)) (dynamics.py:118-125,147-153 / :162-170,192-199)
      auto v_half = [&]() {
        LANE_EVAL_FENCE();
        if constexpr (RES == 2) lane_hidden_dpp<DP, HPR>(PV, [&](int k) { return x[k]; }, g_of, tc, ts, h2);
        else lane_hidden<DP, HPR>(WV, [&](int k) { return x[k]; }, g_of, tc, ts, h2);
#pragma unroll
        for (int p = 0; p < DP / 2; ++p) {
          f2 S, T, Q;
          if constexpr (RES == 2) lane_heads_dpp<DP, HPR>(PV, p, h2, S, T, Q);
          else lane_heads<DP, HPR>(WV, p, h2, S, T, Q);
          const f2 sv = S * (sgn * heps);
          const f2 ES = ex2_2(sv * LOG2E), EQ = ex2_2(Q * (eps * LOG2E));
          const f2 gg = f2{g_of(2 * p), g_of(2 * p + 1)}, vv = f2{v[2 * p], v[2 * p + 1]};
          const f2 tr = (T - EQ * gg) * heps;
          const f2 vn = ES * (vv + nf * tr) + ff * tr;
          v[2 * p] = vn.x;
          v[2 * p + 1] = vn.y;
          ld += sv.x + sv.y;
        }
      };
      // masked position update with X(v_h, kept x) (dynamics.py:127-145 / :172-190)
      auto x_half = [&](bool first) {
        float kq[DP];                                              // the kept-mask of this update
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          const float k1 = k1_of(k);
          kq[k] = first ? k1 : 1.f - k1;
        }
        LANE_EVAL_FENCE();
        if constexpr (RES == 2) lane_hidden_dpp<DP, HPR>(PX, [&](int k) { return v[k]; }, [&](int k) { return kq[k] * x[k]; }, tc, ts, h2);
        else lane_hidden<DP, HPR, RES == 1 ? 1 : 0>(WX, [&](int k) { return v[k]; }, [&](int k) { return kq[k] * x[k]; }, tc, ts, h2, &RX);
#pragma unroll
        for (int p = 0; p < DP / 2; ++p) {
          f2 S, T, Q;
          if constexpr (RES == 2) lane_heads_dpp<DP, HPR>(PX, p, h2, S, T, Q);
          else lane_heads<DP, HPR, RES == 1 ? 1 : 0>(WX, p, h2, S, T, Q, &RX);
          const f2 up = 1.f - f2{kq[2 * p], kq[2 * p + 1]};
          const f2 sx = up * S * (sgn * eps);
          const f2 ES = ex2_2(sx * LOG2E), EQ = ex2_2(Q * (eps * LOG2E));
          const f2 vv = f2{v[2 * p], v[2 * p + 1]}, xx = f2{x[2 * p], x[2 * p + 1]};
          const f2 tr = up * (EQ * vv + T) * eps;
          const f2 xn = ES * (xx + nf * tr) + ff * tr;
          x[2 * p] = xn.x;
          x[2 * p + 1] = xn.y;
          ld += sx.x + sx.y;
        }
      };
      v_half();
      x_half(true);
      x_half(false);
      const bool lastU = need_p && it == A.n_steps - 1;
      const float Un = lane_grad<EK, DP>(A, MU, PREC, LOGC, x, g, lastU);     // (only the Rough Well's U costs extra work)
      if (lastU) U1 = Un;
      v_half();
    }

    // ---- per-proposal epilogue: proposal, log-det, accept probability, MH select ---------------------------------
    const bool last = m == A.M - 1;
    if (last && live) {
      if (A.x_out != nullptr)
#pragma unroll
        for (int k = 0; k < DP; ++k)
          if (k < d) A.x_out[chain * d + k] = x[k];
      if (A.v_out != nullptr)
#pragma unroll
        for (int k = 0; k < DP; ++k)
          if (k < d) A.v_out[chain * d + k] = v[k];
    }
    float K1 = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) K1 = fmaf(0.5f * v[k], v[k], K1);
    if (A.logjac_out != nullptr && live) A.logjac_out[moff + chain] = ld;
    if (need_p) {
      const float val = (U0 + K0) - (U1 + K1) + ld;                    // dynamics.py:302-309
      const float p = accept_prob(val);
      if (A.p_out != nullptr && live) A.p_out[moff + chain] = p;
      if (have_u) {
        const bool acc = (p - u_m) >= 0.f;                              // sampler.py:53-55
        if (!acc) {
          if constexpr (DP <= 16) {
#pragma unroll
            for (int k = 0; k < DP; ++k) x[k] = x0[k];
          } else {
#pragma unroll
            for (int k = 0; k < DP; ++k) x[k] = (k < d) ? A.x_next[row * d + k] : 0.f;   // the current-state copy
          }
        }
        // grad U (and U) at the state the chain continues from
        U_cur = lane_grad<EK, DP>(A, MU, PREC, LOGC, x, g, true);
        if constexpr (DP > 16) {
          if (live && acc)
#pragma unroll
            for (int k = 0; k < DP; ++k)
              if (k < d) A.x_next[chain * d + k] = x[k];
        }
      } else {
        U_cur = U1;
      }
    } else {
      U_cur = U1;
    }
    if (A.x_hist != nullptr && live)
#pragma unroll
      for (int k = 0; k < DP; ++k)
        if (k < d) A.x_hist[(moff + chain) * d + k] = x[k];
  }
  if (A.x_next != nullptr && live) {
    if constexpr (DP <= 16) {
#pragma unroll
      for (int k = 0; k < DP; ++k)
        if (k < d) A.x_next[chain * d + k] = x[k];
    }
  }
}

// the kernels this build instantiates: returns 0 if (ek, d, H) has none
int launch_lane(const KArgs& k, hipStream_t s);
bool lane_supported(int ek, int d, int H, int ncomp);

}  // namespace l2hmc
