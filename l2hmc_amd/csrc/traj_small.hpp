// traj_small.hpp -- the fused trajectory kernel for SMALL targets, d <= 4 (gfx950 / CDNA4).
//
// BASELINE.json configs 1 and 3 (SCG-2D, MoG-2D) have d = 2: in the 16-wide dimension tiles of
// traj_fast_kernel 14 of 16 MFMA rows and 3 of 4 values per lane are padding.  This form keeps the tiling
// over chains (one wave owns 16 chains = the N of v_mfma_f32_16x16x4_f32, the persistent sampler loop, the
// schedule records, the folded constants of traj_fast.hpp) but gives every lane ONE dimension:
//
//   lane l = (c = l & 15, q = l >> 4) holds chain c, dimension q (lanes with q >= d carry zeros);
//   * layer 1: the d <= 4 dimensions are exactly ONE k-step (k = q): 1 MFMA per input instead of 4;
//   * heads: the rows of ONE 16-row MFMA block are (dimension q', head h) at row 4 q' + h, h = S, T, Q --
//     so after 3 MFMAs (k-steps over the hidden units) lane (c, q) holds z_S, z_T, z_Q of ITS dimension in
//     acc[0..2]: 3 MFMAs per net evaluation instead of 9, and the elementwise update is scalar, not float4;
//   * dense precisions (SCG, the mixture components): y = G dx as ONE MFMA whose row 4 j carries output
//     dimension j, so y_q lands in lane q's acc[0].
// Per leapfrog step and wave: 29 MFMAs (5 layer 1 + 12 layer 2 + 12 heads) instead of 68, 24 transcendentals
// instead of 96.  Reference: utils/dynamics.py:115-309, utils/sampler.py:28-55 (same algorithm as traj_kernel).
#pragma once
#include <type_traits>
#include "traj_fast.hpp"

namespace l2hmc {

template <int EK>
struct SmallEnergy {
  float mu, prec;            // diagonal Gaussian: this lane's mean / precision entry
  float gf[8];               // dense Gaussian / GMM: A-operand of y = G dx per component (row 4 j <- G[j][q])
  float mus[8];              // per-component mean entry of this lane's dimension
};

// grad U for this lane's dimension and this lane's share of U (summing over the chain's 4 lanes gives U)
template <int EK>
__device__ __forceinline__ float grad_small(const KArgs& A, const float* smem, const SmallEnergy<EK>& E, int lane,
                                            float x, float& Upart, bool wantU) {
  const int q = lane >> 4;
  const bool livedim = q < A.d;
  float g = 0.f, U = 0.f;
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
    const float dx = x - E.mu;
    g = E.prec * dx;
    U = 0.5f * dx * g;
  } else if constexpr (EK == L2HMC_ENERGY_GAUSS_DENSE) {
    const float dx = x - E.mus[0];
    const f4 y = MFMA16(E.gf[0], dx, splat(0.f));
    g = y.x;
    U = 0.5f * dx * g;
  } else if constexpr (EK == L2HMC_ENERGY_GMM) {
    // U = -logsumexp_i(-q_i / 2 + log c_i); grad = sum_i softmax_i G_i (x - mu_i)  (online softmax, distributions.py:104-134)
    float m = -INFINITY, ssum = 0.f, gacc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < A.ncomp) {
        const float dx = x - E.mus[i];
        const f4 y = MFMA16(E.gf[i], dx, splat(0.f));
        float qq = dx * y.x;
        qq = chain4_sum(qq);
        const float V = -(0.5f * qq) + smem[A.o_logc + i];
        const float mn = fmaxf(m, V);
        const float sc = (m == mn) ? 1.f : expf(m - mn), wi = (V == -INFINITY) ? 0.f : expf(V - mn);
        ssum = ssum * sc + wi;
        gacc = gacc * sc + wi * y.x;
        m = mn;
      }
    }
    g = gacc / ssum;
    if (lane < 16) U = -(m + logf(ssum));
  } else if constexpr (EK == L2HMC_ENERGY_ROUGHWELL) {
    const float den = A.den;
    const float arg = x / den;
    g = x - (A.eta / den) * rw_sin1(arg);
    if (wantU) U = livedim ? 0.5f * x * x + A.eta * rw_cos1(arg) : 0.f;
  }
  Upart = U;
  return livedim ? g : 0.f;
}

// PK = 1 (round 6): the hidden layer and the head block as f16x2 (traj_fast.hpp) -- 4 x 4 f16 MFMAs of 16 cycles per step that leave
// the VALU free instead of 24 f32-input ones of 32 that block it; layer 1 is ONE k-step per input and keeps the f32-input MFMA (a
// K = 32 instruction would carry four live slots).  The end points of a proposal are held against L2HMC_F16_STATE_MAX.
// Register budget: two waves per SIMD (256 VGPRs) since round 6 -- with both nets' tail fragments resident (below) the f16x2 form
// needs ~150; the chain counts this kernel serves (up to 32 768 two-dimensional chains = two waves per SIMD) never held more.
#ifndef L2HMC_SMALL_WAVES
#define L2HMC_SMALL_WAVES 2
#endif
template <int EK, int KH, int PK = 0>
__global__ __launch_bounds__(64, L2HMC_SMALL_WAVES) void traj_small_kernel(const KArgs A) {
  constexpr bool F16 = PK == 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  const int lane = threadIdx.x;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N, livedim = q < A.d;
  const int NF = net_floats(1);
  const float LOG2E = 1.4426950408889634f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  constexpr int NTp = 1;
  const int FWN = F16 ? fast_fw_net(NTp, true) : fast_fw_net_f32(NTp), DPp = fast_dpp(NTp), FCN = fast_fc_net(NTp), R = fast_rec(NTp),
            RECD = fast_rec_dir(NTp, A.T);

  // ---- prologue: the same scaled tail fragments, constant tables and schedule records as traj_fast_kernel ----
  constexpr int FW4 = fast_fw_net_f32(NTp) / 4;             // float4 fragments per net: W4, then (S, T, Q) of the one slice
  for (int i = lane; i < 2 * FW4; i += 64) {
    const int net = i >= FW4, j = i - net * FW4, g = j >> 6;
    float sc = 1.f;
    if (g > 0) sc = ((g - 1) % 3 == 1) ? (net == 0 ? eps : heps) : 2.f * LOG2E;
    const f4 src = reinterpret_cast<const f4*>(A.packed + (size_t)net * NF + 3 * 256)[j];
    if constexpr (F16) {                                    // [first fragments of the net's four groups | second ones]
      const WF16 f = wsplit16(src * sc);
      reinterpret_cast<h8v*>(smem + A.o_fw + net * FWN)[j] = f.a1;
      reinterpret_cast<h8v*>(smem + A.o_fw + net * FWN + 4 * FW4)[j] = f.a2;
    } else {
      reinterpret_cast<f4*>(smem + A.o_fw + net * FWN)[j] = src * sc;
    }
  }
  for (int i = lane; i < 2 * 16; i += 64) {
    const int net = i / 16, dim = i % 16;
    const float* scl = A.packed + (size_t)net * NF + net_groups(1) * 256;
    const float epn = net == 0 ? eps : heps;
    const float cs = scl[dim] * epn * LOG2E, cq = scl[16 + dim] * eps * LOG2E;
    float* fc = smem + A.o_fc + net * FCN;
    fc[dim] = cs;
    fc[DPp + dim] = -cs;
    fc[2 * DPp + dim] = cq;
    fc[3 * DPp + dim] = cq + log2f(epn);
  }
  for (int i = lane; i < 2 * A.T * R; i += 64) {
    const int dr = i / (A.T * R), r = (i / R) % A.T, j = i % R;
    float val;
    if (j < 32) {
      const int net = j >> 4, u = j & 15;
      const float* tf = A.packed + (size_t)net * NF + (2 * 64) * 4;
      val = fmaf(tf[u * 4], A.trig[2 * r], fmaf(tf[(16 + u) * 4], A.trig[2 * r + 1], tf[(32 + u) * 4]));
    } else {
      const int dim = j - 32;
      const float m = dim < A.d ? A.masks[r * A.d + dim] : 0.f;
      val = dr ? m : 1.f - m;
    }
    smem[A.o_rec + dr * RECD + (r + 1) * R + j] = val;
  }
  if (EK == L2HMC_ENERGY_GMM)
    for (int i = lane; i < A.ncomp; i += 64) smem[A.o_logc + i] = A.logc[i];

  // this lane's energy constants (registers for the whole launch)
  SmallEnergy<EK> E;
  E.mu = E.prec = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) E.gf[i] = E.mus[i] = 0.f;
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
    if (livedim) { E.mu = A.mu[q]; E.prec = A.prec[q]; }
  } else if constexpr (EK == L2HMC_ENERGY_GAUSS_DENSE || EK == L2HMC_ENERGY_GMM) {
    const int nc = EK == L2HMC_ENERGY_GMM ? A.ncomp : 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < nc) {
        if (livedim) E.mus[i] = A.mu[i * A.d + q];
        // packed precision (pack_gauss_kernel, one 16x16 tile): element (row a, col b) at ((a + 16 (b / 4)) * 4 + b % 4);
        // A operand of lane (row i = c, k = q): G[j][q] on rows i = 4 j, zero elsewhere
        if ((c & 3) == 0 && livedim) E.gf[i] = A.prec[(size_t)i * gauss_floats(1) + (c >> 2) * 4 + q];
      }
    }
  }

  // layer-1 A operands: ONE k-step whose k = q is dimension q.  Packed layer-1 group (tile 0): element (lane (i, 0), r)
  // is W[dim r][unit row i], so this lane's scalar is component q of lane (c, 0)'s float4.
  float xa = A.packed[(0 * 64 + c) * 4 + q], xb = A.packed[(1 * 64 + c) * 4 + q];
  float va = A.packed[NF + (0 * 64 + c) * 4 + q], vb = A.packed[NF + (1 * 64 + c) * 4 + q];
  if (!livedim) xa = xb = va = vb = 0.f;

  float x = (live && livedim) ? A.x[chain * A.d + q] : 0.f;
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr || A.u != nullptr || (A.rng_flags & L2HMC_RNG_U) != 0;
  __syncthreads();

  const float* fwx = smem + A.o_fw;
  const float* fwv = fwx + FWN;
  const float* fcx = smem + A.o_fc;
  const float* fcv = fcx + FCN;
  const f4 Z = splat(0.f);
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {      // fold P into VNet's layer 1 (as traj_fast_kernel)
    va = va + E.prec * vb;
    const f4 cv = MFMA16(vb, -(E.prec * E.mu), Z);
    if (c == 0) {
      for (int i = 0; i < 2 * A.T; ++i) {
        float* tb = smem + A.o_rec + (i / A.T) * RECD + (i % A.T + 1) * R + 16 + 4 * q;
        *reinterpret_cast<f4*>(tb) = lds4(tb) + cv;
      }
    }
    __syncthreads();
  }
  auto vnet_l1 = [&](float xx, float gg) {
    if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) return MFMA16(va, xx, Z);
    else return MFMA16(vb, gg, MFMA16(va, xx, Z));
  };
  // tail of one net: hidden layers + the single (dimension, head) block; returns (z_S, z_T', z_Q) of this lane's dimension
  typedef std::conditional_t<F16, WF16, f4> FragS;
  struct TailS { FragS w2, hd; float cS, cQ, bQ; };
  auto load_frag_s = [&](const float* fw, int idx) -> FragS {          // fragment at float4 slot idx of the net
    if constexpr (F16) return WF16{*reinterpret_cast<const h8v*>(fw + idx * 4), *reinterpret_cast<const h8v*>(fw + 4 * FW4 + idx * 4)};
    else return lds4(fw + idx * 4);
  };
  auto load_tail_s = [&](TailS& t, const float* fw, const float* fc, int dofs) {
    t.w2 = load_frag_s(fw, lane);
    // head block row c = 4 q' + h: head h (S, T, Q) of dimension q'; fragment of head group h at lane (q', q)
    const int h = c & 3;
    t.hd = load_frag_s(fw, (1 + (h < 3 ? h : 0)) * 64 + (c >> 2) + 16 * q);
    if (h == 3) {                                                       // (rows 4 q' + 3 of the block: no head)
      if constexpr (F16) { t.hd.a1 = h8v{}; t.hd.a2 = h8v{}; } else { t.hd = Z; }
    }
    t.cS = fc[dofs + q];
    t.cQ = fc[2 * DPp + q];
    t.bQ = fc[3 * DPp + q];
  };
  auto tail_s = [&](const TailS& t, f4 hs_, float& aS, float& Tt, float& EQ) {
    f4 h = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(hs_[r]);
    f4 acc = Z, z = Z;
    if constexpr (F16) {
      acc = mfma16x2(t.w2, split16<true>(h), Z);
#pragma unroll
      for (int r = 0; r < KH; ++r) h[r] = relu_i(acc[r]);
      z = mfma16x2(t.hd, split16<true>(h), Z);
    } else {
#pragma unroll
      for (int r = 0; r < KH; ++r) acc = MFMA16(t.w2[r], h[r], acc);
#pragma unroll
      for (int r = 0; r < KH; ++r) h[r] = relu_i(acc[r]);
#pragma unroll
      for (int r = 0; r < KH; ++r) z = MFMA16(t.hd[r], h[r], z);
    }
    const float rS = __builtin_amdgcn_rcpf(-(__builtin_amdgcn_exp2f(z.x) * 0.5f + 0.5f));
    aS = rS * t.cS + t.cS;
    const float rQ = __builtin_amdgcn_rcpf(-(__builtin_amdgcn_exp2f(z.z) * 0.5f + 0.5f));
    EQ = __builtin_amdgcn_exp2f(rQ * t.cQ + t.bQ);
    Tt = z.y;
  };

  // two tails of ONE net on two inputs, stage by stage (the compiler keeps the source order of independent chains: written one
  // after the other they run one after the other, each MFMA pair followed by its own s_nop 7)
  auto tail_s2 = [&](const TailS& t, f4 hsa, f4 hsb, float& aSa, float& Ta, float& EQa, float& aSb, float& Tb, float& EQb) {
    f4 ha = Z, hb = Z, za = Z, zb = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) { ha[r] = relu_i(hsa[r]); hb[r] = relu_i(hsb[r]); }
    if constexpr (F16) {
      const h8v sa = split16<true>(ha), sb = split16<true>(hb);
      f4 acca = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.w2.a1, sa, Z, 0, 0, 0);
      f4 accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.w2.a1, sb, Z, 0, 0, 0);
      acca = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.w2.a2, sa, acca, 0, 0, 0);
      accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.w2.a2, sb, accb, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < KH; ++r) { ha[r] = relu_i(acca[r]); hb[r] = relu_i(accb[r]); }
      const h8v ta = split16<true>(ha), tb = split16<true>(hb);
      za = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.hd.a1, ta, Z, 0, 0, 0);
      zb = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.hd.a1, tb, Z, 0, 0, 0);
      za = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.hd.a2, ta, za, 0, 0, 0);
      zb = __builtin_amdgcn_mfma_f32_16x16x32_f16(t.hd.a2, tb, zb, 0, 0, 0);
    } else {
      f4 acca = Z, accb = Z;
#pragma unroll
      for (int r = 0; r < KH; ++r) { acca = MFMA16(t.w2[r], ha[r], acca); accb = MFMA16(t.w2[r], hb[r], accb); }
#pragma unroll
      for (int r = 0; r < KH; ++r) { ha[r] = relu_i(acca[r]); hb[r] = relu_i(accb[r]); }
#pragma unroll
      for (int r = 0; r < KH; ++r) { za = MFMA16(t.hd[r], ha[r], za); zb = MFMA16(t.hd[r], hb[r], zb); }
    }
    const float eSa = __builtin_amdgcn_exp2f(za.x), eSb = __builtin_amdgcn_exp2f(zb.x);
    const float eQa = __builtin_amdgcn_exp2f(za.z), eQb = __builtin_amdgcn_exp2f(zb.z);
    const float rSa = __builtin_amdgcn_rcpf(-(eSa * 0.5f + 0.5f)), rSb = __builtin_amdgcn_rcpf(-(eSb * 0.5f + 0.5f));
    const float rQa = __builtin_amdgcn_rcpf(-(eQa * 0.5f + 0.5f)), rQb = __builtin_amdgcn_rcpf(-(eQb * 0.5f + 0.5f));
    aSa = rSa * t.cS + t.cS;
    aSb = rSb * t.cS + t.cS;
    EQa = __builtin_amdgcn_exp2f(rQa * t.cQ + t.bQ);
    EQb = __builtin_amdgcn_exp2f(rQb * t.cQ + t.bQ);
    Ta = za.y;
    Tb = zb.y;
  };

  float U_start = 0.f;
  float g = grad_small<EK>(A, smem, E, lane, x, U_start, need_p);
  f4 pv = vnet_l1(x, g);

  // ---- persistent sampler loop (as traj_kernel / traj_fast_kernel) -------------------------------------------
  const long long gchain = A.chain_off + chain;
  const bool rng_v = (A.rng_flags & L2HMC_RNG_V) != 0, rng_d = (A.rng_flags & L2HMC_RNG_DIR) != 0;
  const bool rng_u = (A.rng_flags & L2HMC_RNG_U) != 0;
  const bool have_u = A.u != nullptr || rng_u;
  TailS tk;
  for (int m = 0; m < A.M; ++m) {
    const long long moff = (long long)m * A.N;
    const unsigned long long prop = A.rng_prop0 + (unsigned long long)m;
    float v;
    if (rng_v) {
      const f4 n4 = philox_normal4(A.rng_seed, gchain, 0u, prop);     // dims 0..3 = components of block 0
      v = livedim ? n4[q] : 0.f;
    } else {
      v = (live && livedim) ? A.v[(moff + chain) * A.d + q] : 0.f;
    }
    bool fwd = (A.dir != nullptr && !rng_d) ? (live ? A.dir[moff + chain] != 0 : true) : (A.dir_all != 0);
    float u_m = (A.u != nullptr && !rng_u && live) ? A.u[moff + chain] : 0.f;
    if (rng_d || rng_u) {
      bool fr;
      float ur;
      philox_dir_u(A.rng_seed, gchain, prop, fr, ur);
      if (rng_d) fwd = fr;
      if (rng_u) u_m = ur;
    }
    const float x0 = x, g0 = g;
    const f4 pv0 = pv;
    float amax_l = F16 ? fmaxf(fabsf(x), fmaxf(fabsf(v), fabsf(g))) : 0.f;      // f16x2: the proposal's end points against the operand range
    float red[5];
    red[0] = U_start;
    red[1] = 0.5f * v * v;
    red[2] = 0.f;
    float ldv = 0.f;
    const float ff = fwd ? 1.f : 0.f, nf = ff - 1.f;
    const int dofs = fwd ? 0 : DPp;
    const int row0 = fwd ? A.step_begin : (A.T - 1 - A.step_begin);
    const float* rec = smem + A.o_rec + (fwd ? RECD : 0) + (row0 + 1) * R;
    const int drec = fwd ? R : -R;
    f4 tbv = lds4(rec + 16 + 4 * q);
    load_tail_s(tk, fwv, fcv, dofs);
    // both nets' tail fragments and constants stay in registers for the whole proposal (round 6; -DL2HMC_SMALL_RELOAD_TAILS: re-read
    // from LDS twice per step as in rounds 2-5): MoG-2D / 8192 chains 36.1 -> 33.6 us per proposal, Rough Well d = 2 / 16 384 chains
    // 11.8 -> 11.2, SCG-2D unchanged (profiles/r06_resident_tails.txt).  Same values, same arithmetic: same bits.
#ifndef L2HMC_SMALL_RELOAD_TAILS
    TailS tkx;
    load_tail_s(tkx, fwx, fcx, dofs);
#define SMALL_TKX tkx
#else
#define SMALL_TKX tk
#endif
    // VNet is evaluated at the same (x, grad U(x)) at the end of step t and at the start of step t + 1 -- only the time row differs --
    // and its S, T, Q do not depend on the momentum: the two tails are INDEPENDENT given the shared layer-1 sum, so they are
    // issued side by side (round 6; -DL2HMC_SMALL_SERIAL_TAILS: one after the other as in rounds 2-5).  A lone wave per SIMD -- every
    // chain count this kernel serves -- is bound by the length of its dependent chain: four tails per step become three.  The last
    // step's look-ahead tail is computed and dropped.  Same operations on the same operands: results are bit-identical.
    float aS, T, EQ;
#ifndef L2HMC_SMALL_SERIAL_TAILS
    float aS_n, T_n, EQ_n;
    tail_s(tk, pv + tbv, aS_n, T_n, EQ_n);
#endif
    for (int it = 0; it < A.n_steps; ++it) {
      const f4 tbx = lds4(rec + 4 * q);
      const float k1 = rec[32 + q], up1 = 1.f - k1;
      rec += drec;
      const f4 tbv_n = lds4(rec + 16 + 4 * q);
      // ---- momentum half-update #1  (dynamics.py:118-125 / :162-170)
#ifdef L2HMC_SMALL_SERIAL_TAILS
      tail_s(tk, pv + tbv, aS, T, EQ);
#else
      aS = aS_n; T = T_n; EQ = EQ_n;
#endif
      float ES = __builtin_amdgcn_exp2f(aS);
      ldv += aS;
      float tr = T - EQ * g;
      const float vh = ES * (nf * tr + v) + ff * tr;
      // ---- two masked position updates  (:127-145 / :172-190)
#ifdef L2HMC_SMALL_RELOAD_TAILS
      load_tail_s(tk, fwx, fcx, dofs);
#endif
      const f4 pa = MFMA16(xa, vh, Z);
      tail_s(SMALL_TKX, MFMA16(xb, k1 * x, pa) + tbx, aS, T, EQ);
      float aSm = up1 * aS;
      ES = __builtin_amdgcn_exp2f(aSm);
      ldv += aSm;
      tr = up1 * (EQ * vh + T);
      const float y = ES * (nf * tr + x) + ff * tr;
      tail_s(SMALL_TKX, MFMA16(xb, up1 * y, pa) + tbx, aS, T, EQ);
      aSm = k1 * aS;
      ES = __builtin_amdgcn_exp2f(aSm);
      ldv += aSm;
      tr = k1 * (EQ * vh + T);
      x = ES * (nf * tr + y) + ff * tr;
      // ---- momentum half-update #2 at the new position  (:147-153 / :192-199)
#ifdef L2HMC_SMALL_RELOAD_TAILS
      load_tail_s(tk, fwv, fcv, dofs);
#endif
      g = grad_small<EK>(A, smem, E, lane, x, red[2], need_p && it == A.n_steps - 1);
      pv = vnet_l1(x, g);
#ifdef L2HMC_SMALL_SERIAL_TAILS
      tail_s(tk, pv + tbv, aS, T, EQ);
#else
      tail_s2(tk, pv + tbv, pv + tbv_n, aS, T, EQ, aS_n, T_n, EQ_n);   // + the next step's first half-update (its time row)
#endif
      ES = __builtin_amdgcn_exp2f(aS);
      ldv += aS;
      tr = T - EQ * g;
      v = ES * (nf * tr + vh) + ff * tr;
      tbv = tbv_n;
    }
    const bool last = m == A.M - 1;
    red[3] = 0.5f * v * v;
    red[4] = ldv * 0.6931471805599453f;
    const float U_end = red[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      red[i] += __shfl_xor(red[i], 16);
      red[i] += __shfl_xor(red[i], 32);
    }
    if constexpr (F16) {
      amax_l = fmaxf(amax_l, fmaxf(fabsf(x), fmaxf(fabsf(v), fabsf(g))));
      float oor = amax_l < L2HMC_F16_STATE_MAX ? 0.f : 1.f;
      oor += __shfl_xor(oor, 16);
      oor += __shfl_xor(oor, 32);
      if (oor > 0.f) {               // outside the f16x2 range: a loud non-result (traj_fast.hpp)
        x = v = red[4] = __uint_as_float(0x7fc00000u);
      }
    }
    if (last && live && livedim) {
      if (A.x_out != nullptr) A.x_out[chain * A.d + q] = x;
      if (A.v_out != nullptr) A.v_out[chain * A.d + q] = v;
    }
    const bool writer = live && lane < 16;
    if (A.logjac_out != nullptr && writer) A.logjac_out[moff + chain] = red[4];
    if (need_p) {
      const float p = accept_prob((red[0] + red[1]) - (red[2] + red[3]) + red[4]);     // dynamics.py:302-309
      if (A.p_out != nullptr && writer) A.p_out[moff + chain] = p;
      if (have_u) {
        const bool acc = live && (p - u_m) >= 0.f;                                     // sampler.py:53-55
        x = acc ? x : x0;
        g = acc ? g : g0;
        pv = sel4(acc, pv, pv0);
        U_start = acc ? U_end : U_start;
      } else {
        U_start = U_end;
      }
    } else {
      U_start = U_end;
    }
    if (A.x_hist != nullptr && live && livedim) A.x_hist[(moff + chain) * A.d + q] = x;
  }
  if (A.x_next != nullptr && live && livedim) A.x_next[chain * A.d + q] = x;
}

}  // namespace l2hmc
