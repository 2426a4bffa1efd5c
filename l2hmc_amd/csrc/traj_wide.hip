// traj_wide_kernel -- the fused trajectory / sampler-loop kernel for WIDE targets (d > 256).
//
// Same algorithm, tiling and S-layout as traj_kernel (l2hmc_kernels.hpp; dynamics.py:115-309,
// sampler.py:28-55), but the chain state lives in LDS instead of registers: with 8 dim-tiles per wave
// the register-resident form needs ~1000 VGPRs of state and spills.  A workgroup (4 waves) owns 16
// chains; x, v and grad U are three (NT, 64 lanes, 4) LDS arrays in S-layout order -- lane l of the
// wave that owns tile tg reads / writes ONE conflict-free ds_*_b128 per vector -- and every phase is a
// loop over the wave's own tiles:
//     heads of tile tg (9 MFMAs, fragments streamed from L2)  ->  elementwise update of that tile
//     ->  its layer-1 contribution to the NEXT net evaluation (4-8 MFMAs), fused in the same pass.
// Per leapfrog step: 4 tile passes, 3 LDS exchanges of one 16x16 partial (as in traj_kernel), layer 2
// once per net evaluation.  Elementwise energies (diagonal Gaussian, Rough Well) evaluate grad U inside the
// position pass; the DENSE Gaussian (distributions.py:41-57) needs the whole new position first: after a barrier
// every wave forms G (x' - mu) for its own tiles as NT x 4 MFMAs per tile, the packed precision fragments
// (l2hmc_pack_gaussian) streamed from L2 four tiles ahead, the other waves' x' tiles read from the LDS state.  The
// mixture of Gaussians (:104-134) does that per component, with the online softmax of traj_kernel.
#include <type_traits>
#include "traj_fast.hpp"

namespace l2hmc {

__device__ __forceinline__ f4 tl(const float* S, int tg, int lane) { return lds4(S + (tg * 64 + lane) * 4); }
__device__ __forceinline__ void ts(float* S, int tg, int lane, f4 v) { *reinterpret_cast<f4*>(S + (tg * 64 + lane) * 4) = v; }

// AIS bridge / temperature on a raw (grad U, U) pair (utils/ais.py:46-47, dynamics.py:203-212)
__device__ __forceinline__ f4 wide_finish(const KArgs& A, f4 x, f4 g, float& u, bool wantU) {
  if (A.beta != 1.f) {
    g = x * (1.f - A.beta) + g * A.beta;
    if (wantU) u = (1.f - A.beta) * 0.5f * hsum(x * x) + A.beta * u;
  }
  if (A.temperature != 1.f) {
    g = g / A.temperature;
    u = u / A.temperature;
  }
  return g;
}

// grad U of one tile and this lane's share of U (distributions.py:31-32,41-57 diagonal case; :84-97)
template <int EK>
__device__ __forceinline__ f4 wide_grad(const KArgs& A, const float* smem, int tg, int q, f4 x, float& U,
                                        bool wantU) {
  f4 g;
  float u = 0.f;
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
    const f4 mu = lds4(smem + A.o_mu + 16 * tg + 4 * q), s = lds4(smem + A.o_prec + 16 * tg + 4 * q);
    const f4 dx = x - mu;
    g = s * dx;
    if (wantU) u = 0.5f * hsum(dx * g);
  } else {
    const float eta = A.eta, den = A.den, scale = eta / den;
    const f4 arg = x / den;
    g = x - scale * rw_sin4(arg);
    if (wantU) {                           // (wave-uniform: only the end points of a trajectory need U)
      const int dim0 = 16 * tg + 4 * q;    // padded dims hold x = 0 and would add eta * cos(0): mask them out
      const f4 cs = rw_cos4(arg);
      const f4 lv = f4{dim0 < A.d ? 1.f : 0.f, dim0 + 1 < A.d ? 1.f : 0.f, dim0 + 2 < A.d ? 1.f : 0.f,
                       dim0 + 3 < A.d ? 1.f : 0.f};
      u = 0.5f * hsum(x * x) + eta * hsum(lv * cs);
    }
  }
  g = wide_finish(A, x, g, u, wantU);
  U += u;
  return g;
}

// PK = 1 (round 6, the elementwise targets): every contraction as f16x2 (traj_fast.hpp) -- the fragments stream from L2 already
// split (KArgs.packed16, l2hmc_pack_nets), an activation is split by the wave that consumes it (the second hidden activation
// once per net evaluation for all the wave's tiles), and the end points of a proposal are held against L2HMC_F16_STATE_MAX.
template <int EK, int KH, int NW, int PK = 0>
// Register budget of the four-wave form (d <= 256: two workgroups per CU by its LDS plan, `wide_waves`).  With `__launch_bounds__(256)`
// alone the compiler took 314-324 registers -- ONE workgroup per CU, half the waves the design counts on to hide the L2 latency of the
// streamed fragments -- for all of rounds 3-6.  Two waves per SIMD (256 VGPRs, 192-240 bytes of scratch in the f16x2 form): Rough Well
// d = 144 ... 256 / 16 384 chains 378 ... 499 -> 286 ... 378 us per proposal (profiles/r06_wide_handover.txt).
#ifndef L2HMC_WIDE4_WAVES
#define L2HMC_WIDE4_WAVES 2
#endif
__global__ __launch_bounds__(64 * NW, NW == 4 ? L2HMC_WIDE4_WAVES : 1) void traj_wide_kernel(const KArgs A) {
  constexpr bool GMMK = EK == L2HMC_ENERGY_GMM;                 // (both: grad U couples all dimensions)
  constexpr bool DENSE = EK == L2HMC_ENERGY_GAUSS_DENSE || GMMK;
  constexpr bool F16 = PK == 1;
  static_assert(!(F16 && DENSE), "f16x2: elementwise targets only");
  typedef std::conditional_t<F16, WF16, f4> Frag;                // a weight fragment
  typedef std::conditional_t<F16, h8v, f4> HidB;                 // the second hidden activation as the heads' B operand
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  constexpr int NTHR = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  const int NT = A.NT, DP = 16 * NT, NF = net_floats(NT);
  const int DTW = (NT + NW - 1) / NW;
  const int t_lo = w * DTW, t_hi = (t_lo + DTW < NT) ? t_lo + DTW : NT;     // this wave's tiles
  const float* wx = A.packed;            // XNet fragments (global, L2-hot)
  const float* wv = A.packed + NF;       // VNet fragments
  const int NG = net_groups(NT);
  const float* wx16 = A.packed16;        // (F16) the same groups as f16x2 pairs
  const float* wv16 = A.packed16 + (F16 ? net_f16_floats(NT) : 0);

  // ---- prologue: masks / time table / energy parameters into LDS -----------------------------------
  for (int i = tid; i < A.T * DP; i += NTHR) {
    const int row = i / DP, dim = i % DP;
    smem[A.o_mask + i] = dim < A.d ? A.masks[row * A.d + dim] : 0.f;
  }
  for (int i = tid; i < 2 * A.T; i += NTHR) smem[A.o_trig + i] = A.trig[i];
  stage_energy<EK, true>(A, smem, tid, NTHR);
  float* SX = smem + A.o_state;
  float* SV = SX + NT * 256;
  float* SG = SV + NT * 256;

  // (chain, dims 16 tg + 4 q + r) of a row-major (N, d) array  <->  this lane's f4 of tile tg
  auto gload = [&](const float* p, int tg) {
    const int dim0 = 16 * tg + 4 * q;
    f4 r = splat(0.f);
    if (live && p != nullptr) {
      const float* row = p + chain * A.d;
      if (dim0 + 0 < A.d) r.x = row[dim0 + 0];
      if (dim0 + 1 < A.d) r.y = row[dim0 + 1];
      if (dim0 + 2 < A.d) r.z = row[dim0 + 2];
      if (dim0 + 3 < A.d) r.w = row[dim0 + 3];
    }
    return r;
  };
  auto gstore = [&](float* p, int tg, f4 v) {
    if (!live || p == nullptr) return;
    const int dim0 = 16 * tg + 4 * q;
    float* row = p + chain * A.d;
    if (dim0 + 0 < A.d) row[dim0 + 0] = v.x;
    if (dim0 + 1 < A.d) row[dim0 + 1] = v.y;
    if (dim0 + 2 < A.d) row[dim0 + 2] = v.z;
    if (dim0 + 3 < A.d) row[dim0 + 3] = v.w;
  };
  for (int tg = t_lo; tg < t_hi; ++tg) {
    const f4 x = gload(A.x, tg);
    ts(SX, tg, lane, x);
    gstore(A.x_next, tg, x);             // x_next doubles as the current-state copy a rejected chain resumes from
  }
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr || A.u != nullptr || (A.rng_flags & L2HMC_RNG_U) != 0;
  __syncthreads();
  // time-embedding table TB[net][row s][unit row i] (as in traj_kernel)
  for (int idx = tid; idx < 2 * A.T * 16; idx += NTHR) {
    const int net = idx / (A.T * 16), srow = (idx / 16) % A.T, i = idx & 15;
    const float* tf = (net == 0 ? wx : wv) + (2 * NT * 64) * 4;
    const float ct = smem[A.o_trig + 2 * srow], st = smem[A.o_trig + 2 * srow + 1];
    smem[A.o_tb + idx] = fmaf(tf[i * 4], ct, fmaf(tf[(16 + i) * 4], st, tf[(32 + i) * 4]));
  }
  __syncthreads();

  const f4 Z = splat(0.f), O = splat(1.f);
  int pb = 0;
  // Weight fragments stream from L2 (several hundred cycles away) and a runtime tile loop is not
  // software-pipelined by the compiler: every pass fetches the fragments of tile tg + 1 while it works on tg.
  // `net`: 0 = XNet, 1 = VNet
  // (round 6, last session: the wave-uniform part of a fragment's address -- net, group -- stays a scalar base and the lane its
  //  16-byte index: `global_load_dwordx4 v, v_off, s[base]`; as (grp * 64 + lane) * 4 on a VGPR base every load paid a v_add_u32 and a
  //  v_lshl_add_u64 -- 26 of a slice's ~210 VALU instructions)
  auto frag = [&](int net, int grp) -> Frag {
    if constexpr (F16) {
      const h8v* b = reinterpret_cast<const h8v*>((net ? wv16 : wx16) + (size_t)grp * 256);
      const h8v* b2 = reinterpret_cast<const h8v*>((net ? wv16 : wx16) + (size_t)grp * 256 + (size_t)NG * 256);
      return WF16{b[lane], b2[lane]};
    } else {
      return reinterpret_cast<const f4*>((net ? wv : wx) + (size_t)grp * 256)[lane];
    }
  };
  const Frag w2x = frag(0, 2 * NT + 1), w2v = frag(1, 2 * NT + 1);
  struct HeadW { Frag Ws, Wt, Wq; f4 es, eq; };
  auto head_frag = [&](int net, int tg) {
    const float* sc = (net ? wv : wx) + net_groups(NT) * 256;
    HeadW hw;
    hw.Ws = frag(net, 2 * NT + 2 + 3 * tg + 0);
    hw.Wt = frag(net, 2 * NT + 2 + 3 * tg + 1);
    hw.Wq = frag(net, 2 * NT + 2 + 3 * tg + 2);
    hw.es = lds4(sc + 16 * tg + 4 * q);
    hw.eq = lds4(sc + 16 * NT + 16 * tg + 4 * q);
    return hw;
  };
  // layer-1 contribution of one tile: acc += W^T z
  auto l1 = [&](f4 acc, const Frag& W, f4 z) {
    if constexpr (F16) {
      return mfma16x2(W, split16<false>(z), acc);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = MFMA16(W[r], z[r], acc);
      return acc;
    }
  };
  // hidden activations after layer 2 from the exchanged layer-1 sum (F16: already split for the heads)
  auto layer2 = [&](const Frag& w2, f4 hpre, f4 tb) -> HidB {
    const f4 h = relu4(hpre + tb);
    if constexpr (F16) {
      return split16<false>(relu4(mfma16x2(w2, split16<false>(h), Z)));
    } else {
      f4 acc = Z;
#pragma unroll
      for (int r = 0; r < KH; ++r) acc = MFMA16(w2[r], h[r], acc);
      return relu4(acc);
    }
  };
  // heads of tile tg: ES = 2^aS, aS = kS e^{lam_s} tanh(zs), T, EQ = 2^{kQ e^{lam_q} tanh(zq)}
  auto heads = [&](const HeadW& hw, const HidB& h, float kS, float kQ, f4& ES, f4& aS, f4& Tt, f4& EQ) {
    f4 zs = Z, zt = Z, zq = Z;
    if constexpr (F16) {
      zs = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw.Ws.a1, h, zs, 0, 0, 0);
      zq = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw.Wq.a1, h, zq, 0, 0, 0);
      zt = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw.Wt.a1, h, zt, 0, 0, 0);
      zs = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw.Ws.a2, h, zs, 0, 0, 0);
      zq = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw.Wq.a2, h, zq, 0, 0, 0);
      zt = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw.Wt.a2, h, zt, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < KH; ++r) {
        zs = MFMA16(hw.Ws[r], h[r], zs);
        zq = MFMA16(hw.Wq[r], h[r], zq);
        zt = MFMA16(hw.Wt[r], h[r], zt);
      }
    }
    aS = ctanh4(hw.es * kS, zs);
    ES = exp2_4(aS);
    Tt = zt;
    EQ = exp2_4(ctanh4(hw.eq * kQ, zq));
  };
  auto nxt = [&](int tg) { return tg + 1 < t_hi ? tg + 1 : tg; };      // tile whose fragments to prefetch

  // y = G (x - mu) of tile tg from the complete position in SX: NT x 4 MFMAs, the packed fragments of row-tile tg
  // (l2hmc_pack_gaussian order) streamed from L2 four tiles ahead
  auto l1f = [&](f4 acc, f4 W, f4 z) {             // (dense precisions: f32-input MFMA)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = MFMA16(W[r], z[r], acc);
    return acc;
  };
  auto matvec_tile = [&](const float* Gp, const float* mu, int tg) {
    const float* Grow = Gp + (size_t)tg * NT * 256;
    auto gfrag = [&](int ti) { return lds4(Grow + ((ti < NT ? ti : NT - 1) * 64 + lane) * 4); };
    f4 acc = Z;
    f4 G0 = gfrag(0), G1 = gfrag(1), G2 = gfrag(2), G3 = gfrag(3);
    for (int t0 = 0; t0 < NT; t0 += 4) {
      const f4 N0 = gfrag(t0 + 4), N1 = gfrag(t0 + 5), N2 = gfrag(t0 + 6), N3 = gfrag(t0 + 7);
      acc = l1f(acc, G0, tl(SX, t0, lane) - lds4(mu + 16 * t0 + 4 * q));
      if (t0 + 1 < NT) acc = l1f(acc, G1, tl(SX, t0 + 1, lane) - lds4(mu + 16 * (t0 + 1) + 4 * q));
      if (t0 + 2 < NT) acc = l1f(acc, G2, tl(SX, t0 + 2, lane) - lds4(mu + 16 * (t0 + 2) + 4 * q));
      if (t0 + 3 < NT) acc = l1f(acc, G3, tl(SX, t0 + 3, lane) - lds4(mu + 16 * (t0 + 3) + 4 * q));
      G0 = N0; G1 = N1; G2 = N2; G3 = N3;
    }
    return acc;
  };
  // DENSE: grad U = G (x - mu) of this wave's tiles (barrier first: every wave's tiles of the new position), into SG; adds
  // this lane's share of U and the VNet layer-1 contribution of grad U (a1)
  auto dense_pass = [&](float& U, bool wantU, f4& a1) {
    __syncthreads();
    const float* mu = smem + A.o_mu;
    for (int tg = t_lo; tg < t_hi; ++tg) {
      const Frag Wb = frag(1, NT + tg);
      const f4 acc = matvec_tile(A.prec, mu, tg);
      const f4 x = tl(SX, tg, lane);
      float u = wantU ? 0.5f * hsum((x - lds4(mu + 16 * tg + 4 * q)) * acc) : 0.f;
      const f4 g = wide_finish(A, x, acc, u, wantU);
      U += u;
      ts(SG, tg, lane, g);
      a1 = l1(a1, Wb, g);
    }
  };
  // MIXTURE (distributions.py:104-134): per component y_i = G_i (x - mu_i) of this wave's (<= 4) tiles, the quadratic form
  // summed over the chain (all waves), online softmax of V_i = log c_i - q_i / 2 exactly as traj_kernel does
  auto gmm_pass = [&](float& U, bool wantU, f4& a1) {
    __syncthreads();
    float m = -INFINITY, ssum = 0.f;
    f4 gacc[4] = {Z, Z, Z, Z};
    for (int i = 0; i < A.ncomp; ++i) {
      const float* mu = smem + A.o_mu + i * DP;
      const float* Gp = A.prec + (size_t)i * gauss_floats(NT);
      f4 y[4];
      float qq[1] = {0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = Z;
        const int tg = t_lo + j;
        if (tg < t_hi) {
          y[j] = matvec_tile(Gp, mu, tg);
          qq[0] += hsum((tl(SX, tg, lane) - lds4(mu + 16 * tg + 4 * q)) * y[j]);
        }
      }
      chain_allreduce<NW, 1>(qq, smem + A.o_red, w, lane);
      const float V = -(0.5f * qq[0]) + smem[A.o_logc + i];
      const float mn = fmaxf(m, V);
      const float sc = (m == mn) ? 1.f : expf(m - mn), wi = (V == -INFINITY) ? 0.f : expf(V - mn);
      ssum = ssum * sc + wi;
#pragma unroll
      for (int j = 0; j < 4; ++j) gacc[j] = gacc[j] * sc + wi * y[j];
      m = mn;
    }
    const float inv = 1.f / ssum;
    const float it = 1.f / A.temperature;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tg = t_lo + j;
      if (tg < t_hi) {
        const f4 x = tl(SX, tg, lane);
        f4 g = gacc[j] * inv;
        if (A.beta != 1.f) {
          g = x * (1.f - A.beta) + g * A.beta;
          if (wantU) U += (1.f - A.beta) * 0.5f * hsum(x * x) * it;
        }
        g = g * it;
        ts(SG, tg, lane, g);
        a1 = l1(a1, frag(1, NT + tg), g);
      }
    }
    if (wantU && w == 0 && lane < 16) U += A.beta * -(m + logf(ssum)) * it;      // once per chain
  };

  // grad U at the start state, and the VNet layer-1 partial there (shared by consecutive half-updates)
  float U_start = 0.f;
  f4 pv[1];
  auto refresh = [&]() {                  // SG, U_start, pv from SX
    U_start = 0.f;
    f4 a0 = Z, a1 = Z;
    Frag Wa = frag(1, t_lo), Wb = frag(1, NT + t_lo);
    for (int tg = t_lo; tg < t_hi; ++tg) {
      const Frag Wa_n = frag(1, nxt(tg)), Wb_n = frag(1, NT + nxt(tg));
      const f4 x = tl(SX, tg, lane);
      a0 = l1(a0, Wa, x);
      if (!DENSE) {
        const f4 g = wide_grad<EK>(A, smem, tg, q, x, U_start, true);
        ts(SG, tg, lane, g);
        a1 = l1(a1, Wb, g);
      }
      Wa = Wa_n; Wb = Wb_n;
    }
    if (GMMK) gmm_pass(U_start, true, a1);
    else if (DENSE) dense_pass(U_start, true, a1);
    pv[0] = a0 + a1;
    xchg<NW, 1>(pv, A, smem, w, lane, pb);
  };
  refresh();

  const long long gchain = A.chain_off + chain;
  const bool rng_v = (A.rng_flags & L2HMC_RNG_V) != 0, rng_d = (A.rng_flags & L2HMC_RNG_DIR) != 0;
  const bool rng_u = (A.rng_flags & L2HMC_RNG_U) != 0;
  const bool have_u = A.u != nullptr || rng_u;
  const float LOG2E = 1.4426950408889634f;

  for (int m = 0; m < A.M; ++m) {
    const long long moff = (long long)m * A.N;
    const unsigned long long prop = A.rng_prop0 + (unsigned long long)m;
    bool fwd = (A.dir != nullptr && !rng_d) ? (live ? A.dir[moff + chain] != 0 : true) : (A.dir_all != 0);
    float u_m = (A.u != nullptr && !rng_u && live) ? A.u[moff + chain] : 0.f;
    if (rng_d || rng_u) {
      bool fr;
      float ur;
      philox_dir_u(A.rng_seed, gchain, prop, fr, ur);
      if (rng_d) fwd = fr;
      if (rng_u) u_m = ur;
    }
    float red[F16 ? 6 : 5];                // U0, K0, U1, K1, logdet (per-lane partial sums); f16x2: + the out-of-range flag
    float amax_l = 0.f;
    red[0] = U_start;
    red[1] = 0.f;
    for (int tg = t_lo; tg < t_hi; ++tg) {
      f4 v;
      if (rng_v) {
        const int dim0 = 16 * tg + 4 * q;
        const f4 n = philox_normal4(A.rng_seed, gchain, (unsigned)(dim0 >> 2), prop);
        v = f4{dim0 + 0 < A.d ? n.x : 0.f, dim0 + 1 < A.d ? n.y : 0.f, dim0 + 2 < A.d ? n.z : 0.f,
               dim0 + 3 < A.d ? n.w : 0.f};
      } else {
        v = gload(A.v + moff * A.d, tg);
      }
      ts(SV, tg, lane, v);
      red[1] += 0.5f * hsum(v * v);
      if constexpr (F16) amax_l = fmaxf(amax_l, fmaxf(amax4(v), fmaxf(amax4(tl(SX, tg, lane)), amax4(tl(SG, tg, lane)))));
    }
    red[2] = U_start;                      // (n_steps == 0: the end point is the start point)
    f4 ldv = splat(0.f);
    const float sgn = fwd ? 1.f : -1.f;
    const float kSx = sgn * eps * LOG2E, kSv = sgn * heps * LOG2E, kQ = eps * LOG2E;

    for (int it = 0; it < A.n_steps; ++it) {
      const int sf = A.step_begin + it, s = fwd ? sf : (A.T - 1 - sf);
      const f4 tbx = lds4(smem + A.o_tb + s * 16 + 4 * q), tbv = lds4(smem + A.o_tb + (A.T + s) * 16 + 4 * q);
      const float* mrow = smem + A.o_mask + s * DP;
      auto k1_of = [&](int tg) {            // forward keeps m first, backward keeps 1 - m first
        const f4 mk = lds4(mrow + 16 * tg + 4 * q);
        return sel4(fwd, mk, O - mk);
      };
      // ---- momentum half-update #1 (dynamics.py:118-125 / :162-170) + the XNet layer-1 sums of (v_h, k1 x)
      HidB h = layer2(w2v, pv[0], tbv);
      f4 pa = Z, pq = Z;
      // Two register sets of prefetched fragments, used in turn (round 6, last session): as `hw = hw_n; Wa = Wa_n; Wb = Wb_n` at the end of a
      // runtime loop's body the hand-over was 64 register moves per tile (24 v_mov_b64 + 16 v_mov_b32 of a slice's ~210 VALU
      // instructions: the compiler cannot rename across a back edge); every tile loop below takes two tiles per trip instead.
      HeadW hw = head_frag(1, t_lo), hw2;
      Frag Wa = frag(0, t_lo), Wb = frag(0, NT + t_lo), Wa2, Wb2;
      auto tileA = [&](int tg, const HeadW& hc, const Frag& Wac, const Frag& Wbc, HeadW& hn, Frag& Wan, Frag& Wbn) {
        hn = head_frag(1, nxt(tg));
        Wan = frag(0, nxt(tg));
        Wbn = frag(0, NT + nxt(tg));
        f4 ES, aS, Tt, EQ;
        heads(hc, h, kSv, kQ, ES, aS, Tt, EQ);
        const f4 vh = v_half(tl(SV, tg, lane), tl(SG, tg, lane), ES, aS, Tt, EQ, heps, fwd, ldv);
        ts(SV, tg, lane, vh);
        pa = l1(pa, Wac, vh);
        pq = l1(pq, Wbc, k1_of(tg) * tl(SX, tg, lane));
      };
      for (int tg = t_lo; tg < t_hi; tg += 2) {
        tileA(tg, hw, Wa, Wb, hw2, Wa2, Wb2);
        if (tg + 1 < t_hi) tileA(tg + 1, hw2, Wa2, Wb2, hw, Wa, Wb);
      }
      f4 px[1] = {pa + pq};
      xchg<NW, 1>(px, A, smem, w, lane, pb);
      // ---- first masked position update (:127-137 / :172-182) + the layer-1 sum of k2 y
      h = layer2(w2x, px[0], tbx);
      pq = Z;
      hw = head_frag(0, t_lo);
      Wb = frag(0, NT + t_lo);
      auto tileB = [&](int tg, const HeadW& hc, const Frag& Wbc, HeadW& hn, Frag& Wbn) {
        hn = head_frag(0, nxt(tg));
        Wbn = frag(0, NT + nxt(tg));
        f4 ES, aS, Tt, EQ;
        heads(hc, h, kSx, kQ, ES, aS, Tt, EQ);
        const f4 k1 = k1_of(tg);
        const f4 y = x_half(tl(SX, tg, lane), k1, tl(SV, tg, lane), ES, aS, Tt, EQ, eps, fwd, ldv);
        ts(SX, tg, lane, y);
        pq = l1(pq, Wbc, (O - k1) * y);
      };
      for (int tg = t_lo; tg < t_hi; tg += 2) {
        tileB(tg, hw, Wb, hw2, Wb2);
        if (tg + 1 < t_hi) tileB(tg + 1, hw2, Wb2, hw, Wb);
      }
      f4 py[1] = {pa + pq};
      xchg<NW, 1>(py, A, smem, w, lane, pb);
      // ---- second masked position update (:139-145 / :184-190), grad U at x', VNet layer-1 sums there
      h = layer2(w2x, py[0], tbx);
      const bool lastU = need_p && it == A.n_steps - 1;
      float Uend = 0.f;
      f4 a0 = Z, a1 = Z;
      hw = head_frag(0, t_lo);
      Wa = frag(1, t_lo);
      Wb = frag(1, NT + t_lo);
      auto tileC = [&](int tg, const HeadW& hc, const Frag& Wac, const Frag& Wbc, HeadW& hn, Frag& Wan, Frag& Wbn) {
        hn = head_frag(0, nxt(tg));
        Wan = frag(1, nxt(tg));
        Wbn = frag(1, NT + nxt(tg));
        f4 ES, aS, Tt, EQ;
        heads(hc, h, kSx, kQ, ES, aS, Tt, EQ);
        const f4 xn = x_half(tl(SX, tg, lane), O - k1_of(tg), tl(SV, tg, lane), ES, aS, Tt, EQ, eps, fwd, ldv);
        ts(SX, tg, lane, xn);
        a0 = l1(a0, Wac, xn);
        if (!DENSE) {
          const f4 g = wide_grad<EK>(A, smem, tg, q, xn, Uend, lastU);
          ts(SG, tg, lane, g);
          a1 = l1(a1, Wbc, g);
        }
      };
      for (int tg = t_lo; tg < t_hi; tg += 2) {
        tileC(tg, hw, Wa, Wb, hw2, Wa2, Wb2);
        if (tg + 1 < t_hi) tileC(tg + 1, hw2, Wa2, Wb2, hw, Wa, Wb);
      }
      if (GMMK) gmm_pass(Uend, lastU, a1);
      else if (DENSE) dense_pass(Uend, lastU, a1);
      if (lastU) red[2] = Uend;
      pv[0] = a0 + a1;
      xchg<NW, 1>(pv, A, smem, w, lane, pb);
      // ---- momentum half-update #2 (:147-153 / :192-199)
      h = layer2(w2v, pv[0], tbv);
      hw = head_frag(1, t_lo);
      auto tileD = [&](int tg, const HeadW& hc, HeadW& hn) {
        hn = head_frag(1, nxt(tg));
        f4 ES, aS, Tt, EQ;
        heads(hc, h, kSv, kQ, ES, aS, Tt, EQ);
        ts(SV, tg, lane, v_half(tl(SV, tg, lane), tl(SG, tg, lane), ES, aS, Tt, EQ, heps, fwd, ldv));
      };
      for (int tg = t_lo; tg < t_hi; tg += 2) {
        tileD(tg, hw, hw2);
        if (tg + 1 < t_hi) tileD(tg + 1, hw2, hw);
      }
    }
    const float ld = hsum(ldv) * 0.6931471805599453f;      // the log-det was accumulated in log2 units

    // ---- per-proposal epilogue: proposal, log-det, accept probability, MH select -------------------------
    const bool last = m == A.M - 1;
    red[3] = 0.f;
    for (int tg = t_lo; tg < t_hi; ++tg) {
      const f4 v = tl(SV, tg, lane);
      red[3] += 0.5f * hsum(v * v);
      if constexpr (F16) amax_l = fmaxf(amax_l, fmaxf(amax4(v), fmaxf(amax4(tl(SX, tg, lane)), amax4(tl(SG, tg, lane)))));
    }
    red[4] = ld;
    const float U_end = red[2];
    if constexpr (F16) red[5] = amax_l < L2HMC_F16_STATE_MAX ? 0.f : 1.f;
    chain_allreduce<NW, F16 ? 6 : 5>(red, smem + A.o_red, w, lane);
    bool oor = false;                      // f16x2: a proposal outside the operand range is a loud non-result (traj_fast.hpp)
    if constexpr (F16) {
      oor = red[5] > 0.f;
      if (oor) red[4] = __uint_as_float(0x7fc00000u);
    }
    if (last) {
      for (int tg = t_lo; tg < t_hi; ++tg) {
        const f4 qn = splat(__uint_as_float(0x7fc00000u));
        gstore(A.x_out, tg, oor ? qn : tl(SX, tg, lane));
        gstore(A.v_out, tg, oor ? qn : tl(SV, tg, lane));
      }
    }
    const bool writer = live && w == 0 && lane < 16;
    if (A.logjac_out != nullptr && writer) A.logjac_out[moff + chain] = red[4];
    bool resumed = false;                  // some chain of this tile went back to its start point
    if (need_p) {
      const float val = (red[0] + red[1]) - (red[2] + red[3]) + red[4];       // dynamics.py:302-309
      const float p = accept_prob(val);
      if (A.p_out != nullptr && writer) A.p_out[moff + chain] = p;
      if (have_u) {
        const bool acc = live && (p - u_m) >= 0.f;                             // sampler.py:53-55
        for (int tg = t_lo; tg < t_hi; ++tg) {
          const f4 xs = sel4(acc, tl(SX, tg, lane), gload(A.x_next != nullptr ? A.x_next : A.x, tg));
          ts(SX, tg, lane, xs);
          gstore(A.x_next, tg, xs);
          if (A.x_hist != nullptr) gstore(A.x_hist + moff * A.d, tg, xs);
        }
        resumed = true;
        U_start = acc ? U_end : U_start;
      } else {
        U_start = U_end;
      }
    } else {
      U_start = U_end;
    }
    if (!resumed && A.x_hist != nullptr)
      for (int tg = t_lo; tg < t_hi; ++tg) gstore(A.x_hist + moff * A.d, tg, tl(SX, tg, lane));
    // the next proposal starts from SX: grad U, U and the VNet partial there (a tile with any rejected
    // chain must recompute them; doing it always keeps the waves in step)
    if (!last) refresh();
  }
  if (!(need_p && have_u))                 // no MH step: x_next (if asked for) is the end point
    for (int tg = t_lo; tg < t_hi; ++tg) gstore(A.x_next, tg, tl(SX, tg, lane));
}

// shared-memory plan of the wide kernel for NW waves (floats); fills the offsets it uses
static long long wide_lds_floats(KArgs& k, int NW) {
  const int NT = k.NT, DP = 16 * NT;
  long long o = 0;
  k.o_mask = (int)o; o += (long long)k.T * DP;
  k.o_trig = (int)o; o += (2 * k.T + 3) / 4 * 4;
  k.o_tb = (int)o; o += 2LL * k.T * 16;
  k.o_P = (int)o; o += 2LL * NW * 256;
  k.o_red = (int)o; o += (long long)NW * 16 * 8;
  k.o_mu = (int)o; o += (long long)(k.ekind == L2HMC_ENERGY_GMM ? k.ncomp : 1) * DP;
  k.o_prec = (int)o; if (k.ekind == L2HMC_ENERGY_GAUSS_DIAG) o += DP;
  k.o_logc = (int)o; if (k.ekind == L2HMC_ENERGY_GMM) o += (k.ncomp + 3) / 4 * 4;
  k.o_state = (int)o; o += 3LL * NT * 256;
  return o;
}
// waves per workgroup of the wide kernel: 4 with TWO workgroups per CU while two of them fit its LDS (round 6, last session: by the plan
// itself -- until then "up to 16 tiles", which left d = 257 ... 288 on the eight-wave form: 552 against 445 us per proposal at d = 272 --
// and with the four-wave form's registers bounded so that two workgroups really are resident, traj_wide_kernel above); else 8 waves (two
// per SIMD hide the L2 latency of the streamed fragments) in the one workgroup that fits.
#ifndef L2HMC_WIDE_LDS_PER_CU
#define L2HMC_WIDE_LDS_PER_CU (160 * 1024)
#endif
static int wide_waves(const KArgs& k) {
  KArgs t = k;
  return 2 * 4 * wide_lds_floats(t, 4) <= L2HMC_WIDE_LDS_PER_CU ? 4 : 8;
}

long long plan_lds_wide(KArgs& k) { return 4 * wide_lds_floats(k, wide_waves(k)); }

template <int EK, int KH, int NW, int PK = 0>
static int launch_wide_t(const KArgs& k, long long lds, hipStream_t s) {
  auto kern = traj_wide_kernel<EK, KH, NW, PK>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)((k.N + 15) / 16)), dim3(64 * NW), (size_t)lds, s, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int launch_wide(const KArgs& k, int KH, long long lds, hipStream_t s) {
  const bool diag = k.ekind == L2HMC_ENERGY_GAUSS_DIAG, dense = k.ekind == L2HMC_ENERGY_GAUSS_DENSE, w8 = wide_waves(k) == 8;
#define L2HMC_WIDE(EKv)                                                                      \
  (KH == 3 ? (w8 ? launch_wide_t<EKv, 3, 8>(k, lds, s) : launch_wide_t<EKv, 3, 4>(k, lds, s)) \
           : (w8 ? launch_wide_t<EKv, 4, 8>(k, lds, s) : launch_wide_t<EKv, 4, 4>(k, lds, s)))
  if (dense) return L2HMC_WIDE(L2HMC_ENERGY_GAUSS_DENSE);
  if (k.ekind == L2HMC_ENERGY_GMM) return L2HMC_WIDE(L2HMC_ENERGY_GMM);
  if (k.packed16 != nullptr) {            // the elementwise targets with f16x2 contractions (the dispatcher clears it for variant 200 + v)
#define L2HMC_WIDE16(EKv)                                                                          \
  (KH == 3 ? (w8 ? launch_wide_t<EKv, 3, 8, 1>(k, lds, s) : launch_wide_t<EKv, 3, 4, 1>(k, lds, s)) \
           : (w8 ? launch_wide_t<EKv, 4, 8, 1>(k, lds, s) : launch_wide_t<EKv, 4, 4, 1>(k, lds, s)))
    return diag ? L2HMC_WIDE16(L2HMC_ENERGY_GAUSS_DIAG) : L2HMC_WIDE16(L2HMC_ENERGY_ROUGHWELL);
#undef L2HMC_WIDE16
  }
  return diag ? L2HMC_WIDE(L2HMC_ENERGY_GAUSS_DIAG) : L2HMC_WIDE(L2HMC_ENERGY_ROUGHWELL);
#undef L2HMC_WIDE
}

}  // namespace l2hmc
