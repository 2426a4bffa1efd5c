// traj_tile.hpp -- one wave per 16-chain tile, several tiles per workgroup: the many-chains form of the fused
// trajectory kernel for elementwise targets with 33 <= d <= 64 (gfx950 / CDNA4).
//
// traj_fast_kernel spreads a tile over 4 waves (one 16-dimension slice each): right when chains are scarce (4096
// chains = 256 tiles = one per CU), but every wave then repeats the hidden layer (12 of its 64 MFMAs per step), the
// relu / time-table work, and pays three LDS exchanges per step.  With >= 2 tiles per SIMD available the better
// decomposition is ONE wave per tile looping over the DT dimension slices: 220 instead of 256 MFMAs per tile-step,
// ~20 % fewer VALU instructions, no exchange, no barrier inside the trajectory.  On gfx950 an f32 MFMA blocks the
// SIMD's VALU (profiles/r02_ubench_issue.txt), so this instruction count IS the time.  TPW waves (tiles) share one
// staged copy of the weights in LDS: scaled tail fragments, layer-1 fragments (the diagonal precision folded into
// VNet's W1 as in traj_fast_kernel), constant tables and schedule records; per tile only the state lives in registers
// and every fragment is fetched from LDS where it is used.  Same algorithm and sampler loop as traj_kernel
// (utils/dynamics.py:115-309, utils/sampler.py:28-55).
#pragma once
#include <type_traits>
#include "traj_fast.hpp"

namespace l2hmc {

long long plan_lds_tile(KArgs& k, int DT);

// L2HMC_BFH_TILE: how the contractions run.
//   0  every contraction on the f32-input MFMA (round 3).
//   1  the head contractions -- 144 of the kernel's 212 MFMAs per tile-step -- as K-packed bf16x3 (bf3k.hpp, round 4): 3 bf16
//      MFMAs of 16 cycles that leave the VALU to the SIMD's other wave instead of 3 f32 MFMAs of 32 cycles that block it
//      (profiles/r04_bf16x3_heads.txt: 65 536 chains 224.8 -> 194 us per proposal); layer 1 and the hidden layer stay f32.
//   2  (the default since round 6) EVERY contraction as f16x2 (traj_fast.hpp): two v_mfma_f32_16x16x32_f16 per 16-k block on an
//      exact hd / lo split of the activation (8 VALU per float4 against bf16x3's 26, one 4-register operand against three) and
//      [64 w_hi | w_hi], [64 w_lo | w_lo] fragments split when they are staged (32 bytes per lane and block, as bf16x3's).
//      No f32-input MFMA is left in the step loop: 136 MFMAs of 16 cycles per tile-step (layer 1: 32, hidden: 8, heads: 96)
//      instead of 68 x 32 blocking + 144 x 16 (profiles/r06_f16x2.txt).
#ifndef L2HMC_BFH_TILE
#define L2HMC_BFH_TILE 2
#endif
// staged tail fragments per net: mode 0: W4 + 3 per slice, 16 bytes per lane; mode 1: W4 as f32 + 3 split blocks per slice;
// mode 2: (1 + 3 per slice) split blocks
__host__ __device__ constexpr int tile_fw_net(int NTp) {
  return L2HMC_BFH_TILE == 2 ? (6 * NTp + 2) * 256 : ((L2HMC_BFH_TILE ? 6 : 3) * NTp + 1) * 256;
}
// (The layer-1 contractions as bf16x3 too -- 20 more operand splits per tile-step -- were built and measured in round 4: 205.6 vs
//  195 us per proposal at 65 536 chains, the splits cost more VALU than the 56 f32 MFMAs they replace; commit d230bd6 has the code.
//  With f16x2's 8-instruction split they pay: mode 2.)
__host__ __device__ constexpr int tile_l1_floats(int DT) { return (L2HMC_BFH_TILE == 2 ? 8 : 4) * DT * 256; }
// mode 2: block b of a table = 2 x 64 x 16 bytes, a1 then a2, lane-major (the layout of bf3k.hpp's blocks)
__device__ __forceinline__ void wf16_store(float* base, int block, int lane, const WF16& w) {
  h8v* p = reinterpret_cast<h8v*>(base) + (size_t)block * 128 + lane;
  p[0] = w.a1;
  p[64] = w.a2;
}
__device__ __forceinline__ WF16 wf16_load(const float* base, int block, int lane) {
  const h8v* p = reinterpret_cast<const h8v*>(base) + (size_t)block * 128 + lane;
  return WF16{p[0], p[64]};
}

// the activation split of mode 2: 8 instructions with in-place halves (false) or 10 with independent ones (true, traj_fast.hpp)
#ifndef L2HMC_TILE_SPLIT_LAT
#define L2HMC_TILE_SPLIT_LAT false
#endif
template <int EK, int DT, int KH, int TPW, bool HALF>
#ifndef L2HMC_TILE_WPE
// waves per SIMD the register budget is set for (HIP's second launch-bounds argument).  3 was compiled in round 5: 168
// VGPRs with 78 of them spilled, and no third wave fits beside 80 KB of staged tables anyway.
#define L2HMC_TILE_WPE 2
#endif
__global__ __launch_bounds__(64 * TPW, L2HMC_TILE_WPE) void traj_tile_kernel(const KArgs A) {
  static_assert(EK == L2HMC_ENERGY_GAUSS_DIAG || EK == L2HMC_ENERGY_ROUGHWELL, "elementwise targets only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  const int tid = threadIdx.x, lane = tid & 63, nthr = 64 * TPW;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int NT = A.NT, NF = net_floats(NT);
  const float LOG2E = 1.4426950408889634f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  constexpr int NTp = DT;
  const int FWN = tile_fw_net(NTp), DPp = fast_dpp(NTp), FCN = fast_fc_net(NTp), R = fast_rec(NTp),
            RECD = fast_rec_dir(NTp, A.T);
  const f4 Z = splat(0.f);

  // ---- prologue (whole workgroup): scaled tail fragments, layer-1 fragments, constants, schedule records ----------
  constexpr int GN = 3 * NTp + 1;                  // groups per net: W4, then (S, T, Q) per dimension slice
  for (int i = tid; i < 2 * GN * 64; i += nthr) {
    const int net = i >= GN * 64, j = i - net * (GN * 64), g = j >> 6;
    float sc = 1.f;
    if (g > 0) sc = ((g - 1) % 3 == 1) ? (net == 0 ? eps : heps) : 2.f * LOG2E;
    f4 src = Z;
    if (g < 3 * NT + 1) src = reinterpret_cast<const f4*>(A.packed + (size_t)net * NF + (2 * NT + 1) * 256)[j];
#if L2HMC_BFH_TILE == 2
    wf16_store(smem + A.o_fw + net * FWN, g, j & 63, wsplit16(src * sc));
#elif L2HMC_BFH_TILE
    if (g == 0) reinterpret_cast<f4*>(smem + A.o_fw + net * FWN)[j] = src;
    else bfk_store(smem + A.o_fw + net * FWN + 256, g - 1, j & 63, bfk_wfrag(src * sc));
#else
    reinterpret_cast<f4*>(smem + A.o_fw + net * FWN)[j] = src * sc;
#endif
  }
  stage_energy<EK, false>(A, smem, tid, nthr);
  __syncthreads();                                               // (the fold below reads the staged precision)
  for (int i = tid; i < 4 * DT * 64; i += nthr) {                // layer-1 groups: [net][input][tile], one float4 per lane
    const int grp = i >> 6, ln = i & 63, net = grp / (2 * DT), inp = (grp / DT) & 1, tg = grp % DT;
    f4 val = Z;
    if (tg < NT) {
      val = reinterpret_cast<const f4*>(A.packed + (size_t)net * NF)[(inp * NT + tg) * 64 + ln];
      if (EK == L2HMC_ENERGY_GAUSS_DIAG && net == 1 && inp == 0) {   // W1 + P W2 (traj_fast.hpp)
        const f4 wb = reinterpret_cast<const f4*>(A.packed + (size_t)net * NF)[(NT + tg) * 64 + ln];
        val = val + lds4(smem + A.o_prec + 16 * tg + 4 * (ln >> 4)) * wb;
      }
    }
#if L2HMC_BFH_TILE == 2
    wf16_store(smem + A.o_state, grp, ln, wsplit16(val));
#else
    reinterpret_cast<f4*>(smem + A.o_state)[i] = val;
#endif
  }
  for (int i = tid; i < 2 * 16 * NTp; i += nthr) {
    const int net = i / (16 * NTp), dim = i % (16 * NTp);
    const float* scl = A.packed + (size_t)net * NF + net_groups(NT) * 256;
    const float epn = net == 0 ? eps : heps;
    const float es = dim < 16 * NT ? scl[dim] : 0.f, eq = dim < 16 * NT ? scl[16 * NT + dim] : 0.f;
    const float cs = es * epn * LOG2E, cq = eq * eps * LOG2E;
    float* fc = smem + A.o_fc + net * FCN;
    fc[dim] = cs;
    fc[DPp + dim] = -cs;
    fc[2 * DPp + dim] = cq;
    fc[3 * DPp + dim] = cq + log2f(epn);
  }
  for (int i = tid; i < 2 * A.T * R; i += nthr) {
    const int dr = i / (A.T * R), r = (i / R) % A.T, j = i % R;
    float val;
    if (j < 32) {
      const int net = j >> 4, u = j & 15;
      const float* tf = A.packed + (size_t)net * NF + (2 * NT * 64) * 4;
      val = fmaf(tf[u * 4], A.trig[2 * r], fmaf(tf[(16 + u) * 4], A.trig[2 * r + 1], tf[(32 + u) * 4]));
    } else {
      const int dim = j - 32;
      const float m = dim < A.d ? A.masks[r * A.d + dim] : 0.f;
      val = dr ? m : 1.f - m;
    }
    smem[A.o_rec + dr * RECD + (r + 1) * R + j] = val;
  }
  __syncthreads();
  auto chain4 = [&](f4 W, f4 in, f4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = MFMA16(W[r], in[r], acc);
    return acc;
  };
  [[maybe_unused]] auto l1frag = [&](int net, int inp, int t) { return lds4(smem + A.o_state + (((net * 2 + inp) * DT + t) * 64 + lane) * 4); };
  // layer-1 contraction of dimension slice t: k-step r covers the dimensions 16 t + 4 q + r, live only while 16 t + r < d --
  // the last slice of d = 50 has two live k-steps of four (wave-uniform bound: scalar branches)
  const int klast = A.d - 16 * (DT - 1);
  [[maybe_unused]] auto chain4t = [&](int t, f4 W, f4 in, f4 acc) {
    if (t < DT - 1 || klast >= 4) return chain4(W, in, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < klast) acc = MFMA16(W[r], in[r], acc);
    return acc;
  };
  // (+2.7 % at 65 536 chains for d = 50)
#if L2HMC_BFH_TILE == 2
  auto l1dot = [&](int net, int inp, int t, f4 in, f4 acc) {
    return mfma16x2(wf16_load(smem + A.o_state, (net * 2 + inp) * DT + t, lane), split16<L2HMC_TILE_SPLIT_LAT>(in), acc);
  };
#else
  auto l1dot = [&](int net, int inp, int t, f4 in, f4 acc) { return chain4t(t, l1frag(net, inp, t), in, acc); };
#endif
  auto mu_of = [&](int t) { return lds4(smem + A.o_mu + 16 * t + 4 * q); };
  auto prec_of = [&](int t) { return lds4(smem + A.o_prec + 16 * t + 4 * q); };
  if (EK == L2HMC_ENERGY_GAUSS_DIAG) {           // constant -W2^T P mu of the fold -> VNet time/bias table
    if (wv == 0) {
      f4 cv = Z;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const f4 wb = t < NT ? reinterpret_cast<const f4*>(A.packed + (size_t)NF)[(NT + t) * 64 + lane] : Z;
        cv = chain4(wb, -(prec_of(t) * mu_of(t)), cv);
      }
      if (c == 0) {
        for (int i = 0; i < 2 * A.T; ++i) {
          float* tb = smem + A.o_rec + (i / A.T) * RECD + (i % A.T + 1) * R + 16 + 4 * q;
          *reinterpret_cast<f4*>(tb) = lds4(tb) + cv;
        }
      }
    }
    __syncthreads();
  }

  // ---- from here on every wave works alone on its own tile (no barriers) -------------------------------------------
  const long long tile = (long long)blockIdx.x * TPW + wv;
  if (tile * 16 >= A.N) return;
  const long long chain = tile * 16 + c;
  const bool live = chain < A.N;
  const float* fwx = smem + A.o_fw;
  const float* fwv = fwx + FWN;
  const float* fcx = smem + A.o_fc;
  const float* fcv = fcx + FCN;
  const float rw_den = A.den;

  auto grad_t = [&](f4 xx, int t) {
    if (EK == L2HMC_ENERGY_GAUSS_DIAG) return prec_of(t) * (xx - mu_of(t));
    return xx - (A.eta / rw_den) * rw_sin4(xx / rw_den);
  };
  auto energy_part = [&](const f4 (&xx)[DT], const f4 (&gg)[DT]) {
    float U = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      if (EK == L2HMC_ENERGY_GAUSS_DIAG) {
        U += 0.5f * hsum((xx[t] - mu_of(t)) * gg[t]);
      } else {
        const int dim0 = 16 * t + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (dim0 + r < A.d) U += 0.5f * xx[t][r] * xx[t][r] + A.eta * cosf(xx[t][r] / rw_den);
      }
    }
    return U;
  };
  // hidden layers of one net: h2 (unit rows) from the layer-1 sum + time/bias row
#if L2HMC_BFH_TILE == 2
  auto hidden = [&](const float* fw, f4 hs_) {
    const WF16 w2 = wf16_load(fw, 0, lane);
    f4 h = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(hs_[r]);
    const f4 acc = mfma16x2(w2, split16<L2HMC_TILE_SPLIT_LAT>(h), Z);
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(acc[r]);
    return h;
  };
#else
  auto hidden = [&](const float* fw, f4 hs_) {
    const f4 w2 = lds4(fw + lane * 4);
    f4 h = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(hs_[r]);
    f4 acc = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) acc = MFMA16(w2[r], h[r], acc);
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(acc[r]);
    return h;
  };
#endif
  // heads of dimension slice t: aS = log2 of the scale factor, T' = step T, EQ' = step e^{eps Q}  (traj_fast.hpp)
#if L2HMC_BFH_TILE == 2
  typedef h8v HidT;                       // the second hidden activation as the split B operand of the heads
  auto hidden_b = [&](const float* fw, f4 hs_) { return split16<L2HMC_TILE_SPLIT_LAT>(hidden(fw, hs_)); };
#elif L2HMC_BFH_TILE
  typedef BfkA HidT;                      // the second hidden activation as the split B operand of the heads
  auto hidden_b = [&](const float* fw, f4 hs_) { return bfk_afrag(hidden(fw, hs_)); };
#else
  typedef f4 HidT;
  auto hidden_b = [&](const float* fw, f4 hs_) { return hidden(fw, hs_); };
#endif
  // The fragments and constants of a slice's heads are REQUESTED one slice ahead (heads_load) and consumed by heads_eval: fetched
  // where they are used, every slice waited out an LDS round trip in front of its MFMAs -- 16 times per leapfrog step with only
  // one other wave on the SIMD to cover it.  The request for slice t + 1 is issued right after slice t's MFMAs (whose operand
  // registers it re-uses), so it flies under slice t's transcendental chain; the compiler barrier keeps it there.
  struct HeadF {
#if L2HMC_BFH_TILE == 2
    WF16 ws, wq, wt;
#elif L2HMC_BFH_TILE
    BfkW ws, wq, wt;
#else
    f4 ws, wq, wt;
#endif
  };
  auto heads_load = [&](const float* fw, int t) {
    HeadF f;
#if L2HMC_BFH_TILE == 2
    f.ws = wf16_load(fw, 1 + 3 * t + 0, lane);
    f.wq = wf16_load(fw, 1 + 3 * t + 2, lane);
    f.wt = wf16_load(fw, 1 + 3 * t + 1, lane);
#elif L2HMC_BFH_TILE
    f.ws = bfk_load(fw + 256, 3 * t + 0, lane);
    f.wq = bfk_load(fw + 256, 3 * t + 2, lane);
    f.wt = bfk_load(fw + 256, 3 * t + 1, lane);
#else
    f.ws = lds4(fw + ((1 + 3 * t + 0) * 64 + lane) * 4);
    f.wt = lds4(fw + ((1 + 3 * t + 1) * 64 + lane) * 4);
    f.wq = lds4(fw + ((1 + 3 * t + 2) * 64 + lane) * 4);
#endif
    return f;
  };
  auto heads_mfma = [&](const HeadF& f, const HidT& h, f4& zs, f4& zq, f4& zt) {
    zs = Z; zq = Z; zt = Z;
#if L2HMC_BFH_TILE == 2
    zs = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ws.a1, h, zs, 0, 0, 0);
    zq = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wq.a1, h, zq, 0, 0, 0);
    zt = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wt.a1, h, zt, 0, 0, 0);
    zs = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ws.a2, h, zs, 0, 0, 0);
    zq = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wq.a2, h, zq, 0, 0, 0);
    zt = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.wt.a2, h, zt, 0, 0, 0);
#elif L2HMC_BFH_TILE
    zs = bfk_dot(f.ws, h, zs);
    zq = bfk_dot(f.wq, h, zq);
    zt = bfk_dot(f.wt, h, zt);
#else
#pragma unroll
    for (int r = 0; r < KH; ++r) {
      zs = MFMA16(f.ws[r], h[r], zs);
      zq = MFMA16(f.wq[r], h[r], zq);
      zt = MFMA16(f.wt[r], h[r], zt);
    }
    // (all nine head MFMAs before the first transcendental, as in traj_fast.hpp: +1 % at 65 536 chains)
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  // The last slice of d = 50 holds dimensions 48, 49: components 2, 3 of every lane's float4 (and the lanes q > 0 altogether) are
  // padding whose state is identically 0 and whose weights are 0 -- z = 0, aS = 0, T = 0 there, and any finite EQ is multiplied
  // by 0.  Transcendentals are the dearest instructions of the chain (8 cycles each, 24 per slice and evaluation: 46 % of the
  // VALU time of a tile-step by the counters, profiles/r04_tile_pmc.txt), so that slice evaluates them on its two live components
  // only and substitutes the padding's values (HALF: a compile-time form -- as a run-time branch both chains stay live and the
  // kernel spills; the launcher picks it when d - 16 (DT - 1) <= 2).
  constexpr bool half_last = HALF;
  typedef float f2t __attribute__((ext_vector_type(2)));
  auto lo2 = [](f4 a) { return f2t{a.x, a.y}; };
  auto wd2 = [](f2t a) { return f4{a.x, a.y, 0.f, 0.f}; };
  auto add_lo = [](f4 a, f2t b) { return f4{a.x + b.x, a.y + b.y, a.z, a.w}; };
  auto ex2_live = [&](f4 a, bool two) {           // 2^a; two: components 0, 1 only, the others read 1
    if (two) return f4{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y), 1.f, 1.f};
    return ex2_4(a);
  };
  auto heads_chain = [&](f4 cS, f4 cQ, f4 bQ, f4 zs, f4 zq, f4 zt, f4& aS, f4& Tt, f4& EQ, bool two) {
    if (two) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 es = {__builtin_amdgcn_exp2f(zs.x), __builtin_amdgcn_exp2f(zs.y)}, eq = {__builtin_amdgcn_exp2f(zq.x), __builtin_amdgcn_exp2f(zq.y)};
      const f2 ds = -(es * 0.5f + 0.5f), dq = -(eq * 0.5f + 0.5f);
      const f2 rS = {__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y)}, rQ = {__builtin_amdgcn_rcpf(dq.x), __builtin_amdgcn_rcpf(dq.y)};
      const f2 a2 = rS * f2{cS.x, cS.y} + f2{cS.x, cS.y}, q2 = rQ * f2{cQ.x, cQ.y} + f2{bQ.x, bQ.y};
      aS = f4{a2.x, a2.y, 0.f, 0.f};
      EQ = f4{__builtin_amdgcn_exp2f(q2.x), __builtin_amdgcn_exp2f(q2.y), 0.f, 0.f};
      Tt = f4{zt.x, zt.y, 0.f, 0.f};
      return;
    }
    const f4 rS = rcp4(-(ex2_4(zs) * 0.5f + 0.5f));
    aS = rS * cS + cS;
    const f4 rQ = rcp4(-(ex2_4(zq) * 0.5f + 0.5f));
    EQ = ex2_4(rQ * cQ + bQ);
    Tt = zt;
  };
  // one net evaluation's heads over the DT slices: body(t, aS, T', EQ') consumes slice t (the half-update and the layer-1
  // contribution of the next evaluation)
  // L2HMC_TILE_XNET_SLICES = n < DT (TIMING-ONLY experiment of round 6, profiles/r06_masked_heads.txt; results are wrong): the two
  // XNet evaluations of a step run their heads, transcendental chains, half-updates and layer-1 contributions on the first n
  // dimension slices only -- what a form that packs the dimensions a sub-update really moves (dynamics.py:127-145: update 1 the
  // 1 - m ones, update 2 the m ones) into n slices could save AT MOST, with the gather / scatter it needs priced at zero.
#ifndef L2HMC_TILE_XNET_SLICES
#define L2HMC_TILE_XNET_SLICES DT
#endif
  auto net_heads = [&](const float* fw, const float* fc, int dofs, const HidT& h, auto&& body, auto ns_tag) {
    constexpr int NS = decltype(ns_tag)::value;
#ifndef L2HMC_TILE_NO_PREFETCH
    HeadF cur = heads_load(fw, 0);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      f4 zs, zq, zt, aS, Tt, EQ;
      // (the slice's constants: requested in front of its MFMAs, needed behind them)
      const f4 cS = lds4(fc + dofs + 16 * t + 4 * q), cQ = lds4(fc + 2 * DPp + 16 * t + 4 * q), bQ = lds4(fc + 3 * DPp + 16 * t + 4 * q);
      heads_mfma(cur, h, zs, zq, zt);
      if (t + 1 < NS) cur = heads_load(fw, t + 1);
      asm volatile("" ::: "memory");
      const bool two = t == DT - 1 && half_last;
      heads_chain(cS, cQ, bQ, zs, zq, zt, aS, Tt, EQ, two);
      body(t, aS, Tt, EQ, two);
    }
#else
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      f4 zs, zq, zt, aS, Tt, EQ;
      const f4 cS = lds4(fc + dofs + 16 * t + 4 * q), cQ = lds4(fc + 2 * DPp + 16 * t + 4 * q), bQ = lds4(fc + 3 * DPp + 16 * t + 4 * q);
      const HeadF cur = heads_load(fw, t);
      heads_mfma(cur, h, zs, zq, zt);
      const bool two = t == DT - 1 && half_last;
      heads_chain(cS, cQ, bQ, zs, zq, zt, aS, Tt, EQ, two);
      body(t, aS, Tt, EQ, two);
    }
#endif
  };

  constexpr int XS = (L2HMC_TILE_XNET_SLICES) < DT ? (L2HMC_TILE_XNET_SLICES) : DT;
  f4 x[DT], v[DT], g[DT];
  load_state<DT, 1>(A.x, A, chain, live, 0, q, x);
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr || A.u != nullptr || (A.rng_flags & L2HMC_RNG_U) != 0;
  f4 pv = Z;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    g[t] = grad_t(x[t], t);
    pv = l1dot(1, 0, t, x[t], pv);
    if (EK != L2HMC_ENERGY_GAUSS_DIAG) pv = l1dot(1, 1, t, g[t], pv);
  }
  float U_start = energy_part(x, g);

  // ---- persistent sampler loop ----------------------------------------------------------------------------------------
  const long long gchain = A.chain_off + chain;
  const bool rng_v = (A.rng_flags & L2HMC_RNG_V) != 0, rng_d = (A.rng_flags & L2HMC_RNG_DIR) != 0;
  const bool rng_u = (A.rng_flags & L2HMC_RNG_U) != 0;
  const bool have_u = A.u != nullptr || rng_u;
  for (int m = 0; m < A.M; ++m) {
    const long long moff = (long long)m * A.N;
    const unsigned long long prop = A.rng_prop0 + (unsigned long long)m;
    if (rng_v) rng_state<DT, 1>(A, gchain, prop, 0, q, v);
    else load_state<DT, 1>(A.v + moff * A.d, A, chain, live, 0, q, v);
    bool fwd = (A.dir != nullptr && !rng_d) ? (live ? A.dir[moff + chain] != 0 : true) : (A.dir_all != 0);
    float u_m = (A.u != nullptr && !rng_u && live) ? A.u[moff + chain] : 0.f;
    if (rng_d || rng_u) {
      bool fr;
      float ur;
      philox_dir_u(A.rng_seed, gchain, prop, fr, ur);
      if (rng_d) fwd = fr;
      if (rng_u) u_m = ur;
    }
    // the start point, for the chains that reject (sampler.py:53-55): parked in x_next (a coalesced store per proposal) rather
    // than in 2 DT float4 of registers per lane -- the dispatcher only takes this kernel when the caller gave x_next with u
    if (have_u) store_state<DT, 1>(A.x_next, A, chain, live, 0, q, x);
    const f4 pv0 = pv;
    float red[5];
    red[0] = U_start;
    red[1] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) red[1] += 0.5f * hsum(v[t] * v[t]);
    float amax_l = 0.f;            // f16x2: the end points of the proposal against L2HMC_F16_STATE_MAX (traj_fast.hpp)
#if L2HMC_BFH_TILE == 2
#pragma unroll
    for (int t = 0; t < DT; ++t) amax_l = fmaxf(amax_l, fmaxf(amax4(x[t]), fmaxf(amax4(v[t]), amax4(g[t]))));
#endif
    f4 ldv = Z;
    const float ff = fwd ? 1.f : 0.f, nf = ff - 1.f;
    const int dofs = fwd ? 0 : DPp;
    const int row0 = fwd ? A.step_begin : (A.T - 1 - A.step_begin);
    const float* rec = smem + A.o_rec + (fwd ? RECD : 0) + (row0 + 1) * R + 4 * q;
    const int drec = fwd ? R : -R;

    for (int it = 0; it < A.n_steps; ++it) {
      // (compiler barrier: the weight fragments are loop-invariant LDS loads; hoisted out of the step loop they
      //  would need ~350 VGPRs and spill -- they are meant to be fetched where they are used)
      asm volatile("" ::: "memory");
      const f4 tbx = lds4(rec), tbv = lds4(rec + 16);
      f4 k1[DT], vh[DT], y[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) k1[t] = lds4(rec + 32 + 16 * t);
      rec += drec;
      // ---- momentum half-update #1 + the XNet layer-1 sums of (v_h, k1 x)  (dynamics.py:118-131 / :162-176)
      HidT h = hidden_b(fwv, pv + tbv);
      f4 pa = Z, pq = Z;
      net_heads(fwv, fcv, dofs, h, [&](int t, f4 aS, f4 Tt, f4 EQ, bool two) {
        const f4 ES = ex2_live(aS, two);
        if (two) {                       // the padded slice: the arithmetic on its two live components as well (the others stay 0)
          ldv = add_lo(ldv, lo2(aS));
          const f2t tr = lo2(Tt) - lo2(EQ) * lo2(g[t]);
          vh[t] = wd2(lo2(ES) * (nf * tr + lo2(v[t])) + ff * tr);
          pa = l1dot(0, 0, t, vh[t], pa);
          pq = l1dot(0, 1, t, wd2(lo2(k1[t]) * lo2(x[t])), pq);
          return;
        }
        ldv += aS;
        const f4 tr = Tt - EQ * g[t];
        vh[t] = ES * (nf * tr + v[t]) + ff * tr;
        pa = l1dot(0, 0, t, vh[t], pa);
        pq = l1dot(0, 1, t, k1[t] * x[t], pq);
      }, std::integral_constant<int, DT>());
      // ---- first masked position update (:131-137 / :176-182) + the layer-1 sum of (1 - k1) y
      asm volatile("" ::: "memory");
      h = hidden_b(fwx, pa + pq + tbx);
      pq = Z;
      net_heads(fwx, fcx, dofs, h, [&](int t, f4 aS, f4 Tt, f4 EQ, bool two) {
        if (two) {
          const f2t up = 1.f - lo2(k1[t]), aSm = up * lo2(aS);
          const f2t ES = {__builtin_amdgcn_exp2f(aSm.x), __builtin_amdgcn_exp2f(aSm.y)};
          ldv = add_lo(ldv, aSm);
          const f2t tr = up * (lo2(EQ) * lo2(vh[t]) + lo2(Tt));
          const f2t yy = ES * (nf * tr + lo2(x[t])) + ff * tr;
          y[t] = wd2(yy);
          pq = l1dot(0, 1, t, wd2(up * yy), pq);
          return;
        }
        const f4 up = 1.f - k1[t];
        const f4 aSm = up * aS;
        const f4 ES = ex2_live(aSm, two);
        ldv += aSm;
        const f4 tr = up * (EQ * vh[t] + Tt);
        y[t] = ES * (nf * tr + x[t]) + ff * tr;
        pq = l1dot(0, 1, t, up * y[t], pq);
      }, std::integral_constant<int, XS>());
#pragma unroll
      for (int t = XS; t < DT; ++t) y[t] = x[t];          // (timing experiment only: XS == DT in the product)
      // ---- second masked position update (:139-145 / :184-190), grad U and VNet's layer-1 sum at the new position
      asm volatile("" ::: "memory");
      h = hidden_b(fwx, pa + pq + tbx);
      pv = Z;
      net_heads(fwx, fcx, dofs, h, [&](int t, f4 aS, f4 Tt, f4 EQ, bool two) {
        if (two) {
          const f2t kk = lo2(k1[t]), aSm = kk * lo2(aS);
          const f2t ES = {__builtin_amdgcn_exp2f(aSm.x), __builtin_amdgcn_exp2f(aSm.y)};
          ldv = add_lo(ldv, aSm);
          const f2t tr = kk * (lo2(EQ) * lo2(vh[t]) + lo2(Tt));
          x[t] = wd2(ES * (nf * tr + lo2(y[t])) + ff * tr);
        } else {
          const f4 aSm = k1[t] * aS;
          const f4 ES = ex2_live(aSm, two);
          ldv += aSm;
          const f4 tr = k1[t] * (EQ * vh[t] + Tt);
          x[t] = ES * (nf * tr + y[t]) + ff * tr;
        }
        g[t] = grad_t(x[t], t);
        pv = l1dot(1, 0, t, x[t], pv);
        if (EK != L2HMC_ENERGY_GAUSS_DIAG) pv = l1dot(1, 1, t, g[t], pv);
      }, std::integral_constant<int, XS>());
#pragma unroll
      for (int t = XS; t < DT; ++t) {                      // (timing experiment only)
        x[t] = y[t];
        g[t] = grad_t(x[t], t);
        pv = l1dot(1, 0, t, x[t], pv);
        if (EK != L2HMC_ENERGY_GAUSS_DIAG) pv = l1dot(1, 1, t, g[t], pv);
      }
      // ---- momentum half-update #2  (:147-153 / :192-199)
      asm volatile("" ::: "memory");
      h = hidden_b(fwv, pv + tbv);
      net_heads(fwv, fcv, dofs, h, [&](int t, f4 aS, f4 Tt, f4 EQ, bool two) {
        const f4 ES = ex2_live(aS, two);
        if (two) {
          ldv = add_lo(ldv, lo2(aS));
          const f2t tr = lo2(Tt) - lo2(EQ) * lo2(g[t]);
          v[t] = wd2(lo2(ES) * (nf * tr + lo2(vh[t])) + ff * tr);
          return;
        }
        ldv += aS;
        const f4 tr = Tt - EQ * g[t];
        v[t] = ES * (nf * tr + vh[t]) + ff * tr;
      }, std::integral_constant<int, DT>());
    }
    const bool last = m == A.M - 1;
    red[2] = energy_part(x, g);
    red[3] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) red[3] += 0.5f * hsum(v[t] * v[t]);
    red[4] = hsum(ldv) * 0.6931471805599453f;
    const float U_end = red[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      red[i] += __shfl_xor(red[i], 16);
      red[i] += __shfl_xor(red[i], 32);
    }
#if L2HMC_BFH_TILE == 2
    {
#pragma unroll
      for (int t = 0; t < DT; ++t) amax_l = fmaxf(amax_l, fmaxf(amax4(x[t]), fmaxf(amax4(v[t]), amax4(g[t]))));
      float oor = amax_l < L2HMC_F16_STATE_MAX ? 0.f : 1.f;
      oor += __shfl_xor(oor, 16);
      oor += __shfl_xor(oor, 32);
      if (oor > 0.f) {               // outside the f16x2 range: a loud non-result
        const float qnan = __uint_as_float(0x7fc00000u);
#pragma unroll
        for (int t = 0; t < DT; ++t) { x[t] = splat(qnan); v[t] = splat(qnan); }
        red[4] = qnan;
      }
    }
#endif
    if (last) {
      store_state<DT, 1>(A.x_out, A, chain, live, 0, q, x);
      store_state<DT, 1>(A.v_out, A, chain, live, 0, q, v);
    }
    const bool writer = live && lane < 16;
    if (A.logjac_out != nullptr && writer) A.logjac_out[moff + chain] = red[4];
    if (need_p) {
      const float p = accept_prob((red[0] + red[1]) - (red[2] + red[3]) + red[4]);      // dynamics.py:302-309
      if (A.p_out != nullptr && writer) A.p_out[moff + chain] = p;
      if (have_u) {
        const bool acc = live && (p - u_m) >= 0.f;                                      // sampler.py:53-55
        f4 x0[DT];
        load_state<DT, 1>(A.x_next, A, chain, live, 0, q, x0);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          x[t] = sel4(acc, x[t], x0[t]);
          g[t] = sel4(acc, g[t], grad_t(x0[t], t));
        }
        pv = sel4(acc, pv, pv0);
        U_start = acc ? U_end : U_start;
      } else {
        U_start = U_end;
      }
    } else {
      U_start = U_end;
    }
    if (A.x_hist != nullptr) store_state<DT, 1>(A.x_hist + moff * A.d, A, chain, live, 0, q, x);
  }
  store_state<DT, 1>(A.x_next, A, chain, live, 0, q, x);
}

}  // namespace l2hmc
