// traj_fast.hpp -- the instruction-lean form of the fused trajectory kernel (gfx950 / CDNA4).
//
// Same algorithm, tiling, S-layout, K-split exchange and sampler loop as traj_kernel
// (l2hmc_kernels.hpp; reference: utils/dynamics.py:115-309, utils/sampler.py:28-55) for the
// geometries whose weights are LDS/register resident (DT <= 2 dim-tiles per wave) with S/T/Q
// nets.  On gfx950 the f32 MFMA and the f32 VALU share the SIMD's issue time (measured:
// profiles/README.md, "many-chains regime"), so the lever is the instruction count per leapfrog
// step.  What this form removes (profiles/r02_traj_isa_hist.txt has the before / after counts):
//
//   * constants folded at staging time instead of multiplied in the loop: the S and Q head
//     fragments (weights and bias row) are scaled by 2 log2(e) when they are copied into LDS,
//     so 2^z is the first op on the MFMA result; the T head is scaled by the step size (eps for
//     XNet, eps/2 for VNet) and log2(step) is added to the Q exponent, so
//     tr = eps (e^{eps Q} v_h + T) resp. (eps/2)(T - e^{eps Q} grad U) is ONE packed fma;
//   * c tanh(z) = c + c * rcp(-(2^z + 1)/2): one per-lane constant, the sign of the direction
//     folded into it (two tables in LDS, each lane reads the one of its direction);
//   * no per-lane selects between the forward and the inverse update: with f = 1 (forward) / 0
//     and nf = f - 1,   z' = ES (z + nf tr) + f tr   is the forward update for f = 1 and the
//     inverse one for f = 0 (ES = e^{-eps S} there);
//   * masks enter as 0/1 factors of the exponent and of tr (z' = z exactly where kept), read
//     from per-direction tables ("forward keeps m first, backward keeps 1 - m first");
//   * relu on the integer view (one v_max_i32; fmaxf needs a canonicalising second v_max);
//   * the diagonal-Gaussian energy value is taken once at the end points, not every step.
#pragma once
#include "l2hmc_kernels.hpp"
#include "bf3k.hpp"

// (Round 4 measured the head contractions as K-packed bf16x3 -- bf3k.hpp, the form traj_tile.hpp uses -- in this kernel too: no
//  gain with one wave per SIMD, profiles/r04_bf16x3_heads.txt; the switch and its code were removed in round 5, commit history
//  has them.)

namespace l2hmc {

__device__ __forceinline__ float relu_i(float a) { return __int_as_float(max(__float_as_int(a), 0)); }
__device__ __forceinline__ f4 rcp4(f4 a) {
  return f4{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y), __builtin_amdgcn_rcpf(a.z),
            __builtin_amdgcn_rcpf(a.w)};
}
__device__ __forceinline__ f4 ex2_4(f4 a) {
  return f4{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y), __builtin_amdgcn_exp2f(a.z),
            __builtin_amdgcn_exp2f(a.w)};
}

// ---- f16x2: every 16-wide contraction of the step loop as TWO f16 MFMAs at fp32-level accuracy (round 6) --------------------------
// v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate and blocks the SIMD's VALU while it does (32 cycles per instruction,
// profiles/r03_ubench_issue.txt); v_mfma_f32_16x16x32_f16 takes 16 cycles for K = 32 and leaves the plain VALU free.
//   * w a = (64 w) (a / 64).  An activation a is split as  hd = f16(a / 64),  lo = f16(a - 64 hd)  (both round-to-nearest; the
//     fma in front of the second rounding is exact):  |a - 64 hd - lo| <= 2^-22 |a| (two 11-bit terms: within a factor four of
//     f32's own rounding of an operand) while lo is a normal f16, |a| >= 0.25; below that lo's error is
//     2^-25 ABSOLUTE.  Eight VALU instructions per float4: 2 v_pk_mul_f32,
//     2 v_cvt_pk_f16_f32, 4 v_fma_mixlo/hi_f16 (or ten: split16<true>).  hd overflows at |a| = 64 x 65504 = 4.2e6 (see L2HMC_F16_STATE_MAX
//     below: proposals whose end points exceed 2.5e5 are poisoned; variant 200 + v keeps the f32-input MFMA for such states).
//   * A weight w is split as  w_hi = f16(w),  w_lo = f16(64 (w - w_hi)) / 64  -- exact to 2^-22 |w| down to |w| ~ 4e-3, whatever
//     the size of the activation it multiplies -- and staged as two fragments  [64 w_hi | w_hi]  and  [64 w_lo | w_lo]
//     (|w| < 1023).
//   * The k-slots of one MFMA carry [hd(k) | lo(k)] of the lane's own four k (slot 8q + j: hd of k = 4q + j for j < 4, lo of k =
//     4q + j - 4 above): the first MFMA adds w_hi (64 hd + lo), the second one w_lo (64 hd + lo), ONE accumulate chain, no
//     rescaling anywhere.  Products of two f16 are exact in the f32 accumulator.
//   * Error of a K-term contraction: <= 2.5 x 2^-22 sum |w_k a_k| + 2^-25 sum over {k: |a_k| < 1/4} of |w_k| in the worst case
//     (oracle/f16x2_oracle.py, tests/test_f16x2_oracle.py); in the median within a factor two of an fp32 dot product's own error,
//     and end to end indistinguishable from the f32-input MFMA against float64 (profiles/r06_f16x2_accuracy.txt).
//   * The two MFMAs of a chain MUST be the same instruction: a v_mfma_f32_16x16x16_f16 issued right behind the
//     v_mfma_f32_16x16x32_f16 whose result it accumulates onto loses that result (roc-7.2.0 places no wait states between the
//     two pass counts; tools/ubench_f16x2.hip).
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f2v __attribute__((ext_vector_type(2)));
#define L2HMC_F16_SCALE 64.0f
struct WF16 {      // one weight fragment: [64 w_hi | w_hi] and [64 w_lo | w_lo], one K = 32 instruction each
  h8v a1;
  h8v a2;
};
__device__ __forceinline__ h2v cvt_pk16(float a, float b) { return __builtin_convertvector(f2v{a, b}, h2v); }   // v_cvt_pk_f16_f32
// { f16(a0 - 64 h.x), f16(a1 - 64 h.y) }: the fma is exact, one rounding each (no builtin; VALU -> VALU only: nothing to pad)
__device__ __forceinline__ h2v lo_pair16(h2v h, float a0, float a1) {
  h2v r;
  const float ns = -L2HMC_F16_SCALE;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(ns), "v"(a0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "s"(ns), "v"(a1));
  return r;
}
// LAT = true: the two residuals of a pair through independent v_fma_mix_f32 and one v_cvt_pk_f16_f32 -- ten instructions per
// float4 instead of eight, but no instruction waits for its neighbour's half of a register: measured faster where one wave owns
// its SIMD (traj_fast_kernel: 20.4 against 20.7 us per proposal at 4096 chains, 31.5 against 32.3 at 8192;
// profiles/r06_f16x2.txt).  LAT = false: v_fma_mixlo_f16 + v_fma_mixhi_f16 write the two halves of the pair in place.
template <bool LAT>
__device__ __forceinline__ h8v split16(f4 a) {
  const f4 ad = a * (1.f / L2HMC_F16_SCALE);
  const h2v h01 = cvt_pk16(ad.x, ad.y), h23 = cvt_pk16(ad.z, ad.w);
  h2v l01, l23;
  if constexpr (LAT) {
    float r0, r1, r2, r3;
    const float ns = -L2HMC_F16_SCALE;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "s"(ns), "v"(a.x));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "s"(ns), "v"(a.y));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "s"(ns), "v"(a.z));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "s"(ns), "v"(a.w));
    l01 = cvt_pk16(r0, r1);
    l23 = cvt_pk16(r2, r3);
  } else {
    l01 = lo_pair16(h01, a.x, a.y);
    l23 = lo_pair16(h23, a.z, a.w);
  }
  return h8v{h01.x, h01.y, h23.x, h23.y, l01.x, l01.y, l23.x, l23.y};
}
__device__ __forceinline__ WF16 wsplit16(f4 w) {
  const float sc = L2HMC_F16_SCALE;
  const h2v h01 = cvt_pk16(w.x, w.y), h23 = cvt_pk16(w.z, w.w);                       // w_hi
  const f4 r = f4{w.x - (float)h01.x, w.y - (float)h01.y, w.z - (float)h23.x, w.w - (float)h23.y} * sc;
  const h2v L01 = cvt_pk16(r.x, r.y), L23 = cvt_pk16(r.z, r.w);                       // 64 w_lo
  const h2v H01 = cvt_pk16((float)h01.x * sc, (float)h01.y * sc), H23 = cvt_pk16((float)h23.x * sc, (float)h23.y * sc);   // 64 w_hi
  const h2v l01 = cvt_pk16((float)L01.x / sc, (float)L01.y / sc), l23 = cvt_pk16((float)L23.x / sc, (float)L23.y / sc);   // w_lo
  WF16 f;
  f.a1 = h8v{H01.x, H01.y, H23.x, H23.y, h01.x, h01.y, h23.x, h23.y};
  f.a2 = h8v{L01.x, L01.y, L23.x, L23.y, l01.x, l01.y, l23.x, l23.y};
  return f;
}
// The states a proposal starts from and arrives at are checked against this bound (|x|, |v|, |grad U|; one chain-wide flag beside
// the energy sums): hd overflows at 4.2e6, the hidden layer's relu (an integer max on the bit pattern) can turn the NaN that
// follows into a plausible zero, and the proposal is therefore POISONED instead -- Lx, Lv, log-det NaN, accept probability 0.
// The bound sits a factor 16 under the operand's range because the HIDDEN activations share that range and are not watched:
// |h1| <= (row sum of |W1|) max |input| + |bias|, so nets whose rows sum to less than 16 in absolute value (the reference's
// initialisation: ~6 at d = 50) cannot overflow a hidden activation from a state inside the bound; others: variant 200 + v.
#define L2HMC_F16_STATE_MAX 2.5e5f
__device__ __forceinline__ float amax4(f4 a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); }
// acc += W^T a for the 16 logical k of one fragment
__device__ __forceinline__ f4 mfma16x2(const WF16& W, h8v b, f4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.a1, b, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(W.a2, b, acc, 0, 0, 0);
}

// (Measured and not kept, profiles/r03_exchange_variants.txt: moving the K-split partial as 96 bits -- only r < KH of the
//  float4 are live hidden units -- and keeping a wave's own partial in registers, 3 x ds_read_b96 instead of
//  4 x ds_read_b128 per wave.  The exchange is latency-, not LDS-bandwidth-bound: b96 is neutral, and the wave-uniform
//  switch that keeps the canonical summation order costs 2.5-6 % at 4096 chains.)
// LDS geometry of the fast kernel (floats).  Host and device agree through these helpers.
// NTp = NW * DT >= NT: every wave's tiles exist in the tables (zero-filled beyond NT), so the loop has no
// "is this tile live" branches.
__host__ __device__ constexpr int fast_fw_net_f32(int NTp) { return (3 * NTp + 1) * 256; }                   // f32 tail fragments per net
__host__ __device__ constexpr int fast_fw_net(int NTp, bool f16 = false) { return (3 * NTp + 1) * (f16 ? 512 : 256); }   // staged tail fragments per net (f16x2: 2 x 16 bytes per lane)
__host__ __device__ inline int fast_dpp(int NTp) { return 16 * NTp + 16; }            // padded row of a constant table
__host__ __device__ inline int fast_fc_net(int NTp) { return 4 * fast_dpp(NTp); }     // cS(fwd) cS(bwd) cQ bQ
__host__ __device__ inline int fast_rec(int NTp) { return 32 + 16 * NTp; }            // tbx(16) tbv(16) k1 mask
// T rows plus one never-used row on either side (the next-row prefetch of the last step lands there)
__host__ __device__ inline int fast_rec_dir(int NTp, int T) { return (T + 2) * fast_rec(NTp) + 16; }

long long plan_lds_fast(KArgs& k, int NW, int DT, bool f16 = false);

// which activation split the four-wave kernel's contractions take (split16 above): measured per place, profiles/r06_f16x2.txt section 3
// and profiles/r06_paired_tails.txt section 4
#ifndef L2HMC_FAST_SPLIT_LAT_TAIL
#define L2HMC_FAST_SPLIT_LAT_TAIL true
#endif
#ifndef L2HMC_FAST_SPLIT_LAT_PAIR
#define L2HMC_FAST_SPLIT_LAT_PAIR true
#endif
#ifndef L2HMC_FAST_SPLIT_LAT_L1
#define L2HMC_FAST_SPLIT_LAT_L1 false        // (layer 1's splits in the 8-instruction form since the paired tails: 18.98 -> 18.70 us at 4096 chains, 8192 level)
#endif
template <int DT, bool F16 = false>
struct TailK {
  f4 w2;
  f4 hs[DT], ht[DT], hq[DT];
  f4 cS[DT], cQ[DT], bQ[DT];
};
template <int DT>
struct TailK<DT, true> {
  WF16 w2;
  WF16 hs[DT], ht[DT], hq[DT];
  f4 cS[DT], cQ[DT], bQ[DT];
};

// fw: this net's staged fragments; fc: this net's constant tables; dofs: 0 (forward) / DPp (backward)
template <int DT>
__device__ __forceinline__ void load_tailk(TailK<DT, false>& tk, const float* fw, const float* fc, int dofs, int NTp,
                                           int w, int lane) {
  const int q = lane >> 4, DPp = fast_dpp(NTp);
  tk.w2 = lds4(fw + lane * 4);
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int tg = w * DT + t;
    tk.hs[t] = lds4(fw + ((1 + 3 * tg + 0) * 64 + lane) * 4);
    tk.ht[t] = lds4(fw + ((1 + 3 * tg + 1) * 64 + lane) * 4);
    tk.hq[t] = lds4(fw + ((1 + 3 * tg + 2) * 64 + lane) * 4);
    tk.cS[t] = lds4(fc + dofs + 16 * tg + 4 * q);
    tk.cQ[t] = lds4(fc + 2 * DPp + 16 * tg + 4 * q);
    tk.bQ[t] = lds4(fc + 3 * DPp + 16 * tg + 4 * q);
  }
}
// f16x2: group g's [w_hi | w_hi] fragment at 16-byte slot g * 64 + lane, its [w_lo | w_lo] fragment at the same slot behind
// all the first ones
__device__ __forceinline__ WF16 lds_wf16(const float* fw, int NTp, int g, int lane) {
  WF16 f;
  f.a1 = *reinterpret_cast<const h8v*>(fw + (g * 64 + lane) * 4);
  f.a2 = *reinterpret_cast<const h8v*>(fw + (3 * NTp + 1) * 256 + (g * 64 + lane) * 4);
  return f;
}
template <int DT>
__device__ __forceinline__ void load_tailk(TailK<DT, true>& tk, const float* fw, const float* fc, int dofs, int NTp,
                                           int w, int lane) {
  const int q = lane >> 4, DPp = fast_dpp(NTp);
  tk.w2 = lds_wf16(fw, NTp, 0, lane);
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int tg = w * DT + t;
    tk.hs[t] = lds_wf16(fw, NTp, 1 + 3 * tg + 0, lane);
    tk.ht[t] = lds_wf16(fw, NTp, 1 + 3 * tg + 1, lane);
    tk.hq[t] = lds_wf16(fw, NTp, 1 + 3 * tg + 2, lane);
    tk.cS[t] = lds4(fc + dofs + 16 * tg + 4 * q);
    tk.cQ[t] = lds4(fc + 2 * DPp + 16 * tg + 4 * q);
    tk.bQ[t] = lds4(fc + 3 * DPp + 16 * tg + 4 * q);
  }
}

// hsum = exchanged layer-1 sum + time/bias term.  apply(t, aS, T', EQ') with
//   aS = log2 of the (unmasked) scale factor,  T' = step * T,  EQ' = step * e^{eps Q}.
template <int DT, int KH, class F>
__device__ __forceinline__ void tail_fast(const TailK<DT, false>& tk, f4 hs_, F&& apply) {
  f4 h = splat(0.f);
#pragma unroll
  for (int r = 0; r < KH; ++r) h[r] = relu_i(hs_[r]);
  {
    f4 acc = splat(0.f);
#pragma unroll
    for (int r = 0; r < KH; ++r) acc = MFMA16(tk.w2[r], h[r], acc);
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(acc[r]);
  }
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    f4 zs = splat(0.f), zt = splat(0.f), zq = splat(0.f);
#pragma unroll
    for (int r = 0; r < KH; ++r) {
      zs = MFMA16(tk.hs[t][r], h[r], zs);
      zq = MFMA16(tk.hq[t][r], h[r], zq);
      zt = MFMA16(tk.ht[t][r], h[r], zt);
    }
#if !defined(L2HMC_NO_HEAD_FENCE)
    // all nine head MFMAs first: by the time they have issued, zs (whose chain ended two MFMAs ago) is readable
    // without hazard nops, then zq; zt is used last (the compiler otherwise starts the exps after two chains and pays
    // the MFMA -> VALU wait states in front of them)
    __builtin_amdgcn_sched_barrier(0);
#endif
    // c tanh(z) = c + c * rcp(-(2^{2 z log2 e} + 1) / 2); the head outputs arrive pre-scaled
    const f4 rS = rcp4(-(ex2_4(zs) * 0.5f + 0.5f));
    const f4 aS = rS * tk.cS[t] + tk.cS[t];
    const f4 rQ = rcp4(-(ex2_4(zq) * 0.5f + 0.5f));
    const f4 EQ = ex2_4(rQ * tk.cQ[t] + tk.bQ[t]);
    apply(t, aS, zt, EQ);
  }
}
// f16x2 form: two splits (the two hidden layers' activations: rows r >= KH are dead hidden units, zero), 2 + 6 MFMAs of
// 16 cycles that leave the VALU free -- no fence: the S chain's transcendentals run beside the Q and T products
template <int DT, int KH, class F>
__device__ __forceinline__ void tail_fast(const TailK<DT, true>& tk, f4 hs_, F&& apply) {
  f4 h = splat(0.f);
#pragma unroll
  for (int r = 0; r < KH; ++r) h[r] = relu_i(hs_[r]);
  {
    const f4 acc = mfma16x2(tk.w2, split16<L2HMC_FAST_SPLIT_LAT_TAIL>(h), splat(0.f));
#pragma unroll
    for (int r = 0; r < KH; ++r) h[r] = relu_i(acc[r]);
  }
  const h8v b = split16<L2HMC_FAST_SPLIT_LAT_TAIL>(h);
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    f4 zs = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hs[t].a1, b, splat(0.f), 0, 0, 0);
    f4 zq = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hq[t].a1, b, splat(0.f), 0, 0, 0);
    f4 zt = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.ht[t].a1, b, splat(0.f), 0, 0, 0);
    zs = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hs[t].a2, b, zs, 0, 0, 0);
    zq = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hq[t].a2, b, zq, 0, 0, 0);
    zt = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.ht[t].a2, b, zt, 0, 0, 0);
    const f4 rS = rcp4(-(ex2_4(zs) * 0.5f + 0.5f));
    const f4 aS = rS * tk.cS[t] + tk.cS[t];
    const f4 rQ = rcp4(-(ex2_4(zq) * 0.5f + 0.5f));
    const f4 EQ = ex2_4(rQ * tk.cQ[t] + tk.bQ[t]);
    apply(t, aS, zt, EQ);
  }
}

// TWO tails of one net on two inputs, stage by stage (round 6): VNet sees the same (x, grad U(x)) at the end of step t and at the start
// of step t + 1 -- only the time row differs, and S, T, Q do not depend on the momentum -- so the two evaluations are independent
// given the shared layer-1 sum.  The compiler keeps independent chains in source order (written one after the other, each MFMA pair
// is followed by its own wait states): here every stage is written for both.  A lone wave per SIMD is bound by the length of its
// dependent chain; four tails per step become three.  applyA consumes the first evaluation, keepB stores the second.
template <int DT, int KH, class FA, class FB>
__device__ __forceinline__ void tail_fast2(const TailK<DT, true>& tk, f4 hsa, f4 hsb, FA&& applyA, FB&& keepB) {
  f4 ha = splat(0.f), hb = splat(0.f);
#pragma unroll
  for (int r = 0; r < KH; ++r) { ha[r] = relu_i(hsa[r]); hb[r] = relu_i(hsb[r]); }
  {
    const h8v sa = split16<L2HMC_FAST_SPLIT_LAT_PAIR>(ha), sb = split16<L2HMC_FAST_SPLIT_LAT_PAIR>(hb);
    f4 acca = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.w2.a1, sa, splat(0.f), 0, 0, 0);
    f4 accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.w2.a1, sb, splat(0.f), 0, 0, 0);
    acca = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.w2.a2, sa, acca, 0, 0, 0);
    accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.w2.a2, sb, accb, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < KH; ++r) { ha[r] = relu_i(acca[r]); hb[r] = relu_i(accb[r]); }
  }
  const h8v ba = split16<L2HMC_FAST_SPLIT_LAT_PAIR>(ha), bb = split16<L2HMC_FAST_SPLIT_LAT_PAIR>(hb);
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    f4 zsa = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hs[t].a1, ba, splat(0.f), 0, 0, 0);
    f4 zqa = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hq[t].a1, ba, splat(0.f), 0, 0, 0);
    f4 zta = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.ht[t].a1, ba, splat(0.f), 0, 0, 0);
    f4 zsb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hs[t].a1, bb, splat(0.f), 0, 0, 0);
    f4 zqb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hq[t].a1, bb, splat(0.f), 0, 0, 0);
    f4 ztb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.ht[t].a1, bb, splat(0.f), 0, 0, 0);
    zsa = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hs[t].a2, ba, zsa, 0, 0, 0);
    zqa = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hq[t].a2, ba, zqa, 0, 0, 0);
    zta = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.ht[t].a2, ba, zta, 0, 0, 0);
    zsb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hs[t].a2, bb, zsb, 0, 0, 0);
    zqb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.hq[t].a2, bb, zqb, 0, 0, 0);
    ztb = __builtin_amdgcn_mfma_f32_16x16x32_f16(tk.ht[t].a2, bb, ztb, 0, 0, 0);
    const f4 rSa = rcp4(-(ex2_4(zsa) * 0.5f + 0.5f));
    const f4 aSa = rSa * tk.cS[t] + tk.cS[t];
    const f4 rQa = rcp4(-(ex2_4(zqa) * 0.5f + 0.5f));
    const f4 EQa = ex2_4(rQa * tk.cQ[t] + tk.bQ[t]);
    applyA(t, aSa, zta, EQa);
    const f4 rSb = rcp4(-(ex2_4(zsb) * 0.5f + 0.5f));
    const f4 aSb = rSb * tk.cS[t] + tk.cS[t];
    const f4 rQb = rcp4(-(ex2_4(zqb) * 0.5f + 0.5f));
    const f4 EQb = ex2_4(rQb * tk.cQ[t] + tk.bQ[t]);
    keepB(t, aSb, ztb, EQb);
  }
}

// layer-1 fragments of the f16x2 form (register-resident, split once per launch)
template <int DT>
struct L1W16 {
  WF16 xa[DT], xb[DT], va[DT], vb[DT];
};
template <int DT>
__device__ __forceinline__ f4 l1_part16(const f4 (&z)[DT], f4 acc, const WF16* W) {
#pragma unroll
  for (int t = 0; t < DT; ++t) acc = mfma16x2(W[t], split16<L2HMC_FAST_SPLIT_LAT_L1>(z[t]), acc);
  return acc;
}

// PK: 0 = f32-input MFMA (v_mfma_f32_16x16x4_f32), 1 = f16x2 (above)
template <int EK, int DT, int NW, int KH, int PK = 0>
#ifndef L2HMC_FAST_WAVES
#define L2HMC_FAST_WAVES 2
#endif
// Register budget (round 6, profiles/r06_fast_dt2_spills.txt): with two dimension tiles per wave the f16x2 form holds 120 VGPRs of
// weight fragments (8 layer-1 + 7 tail WF16) beside the state -- under the two-waves-per-SIMD bound (256 VGPRs) the allocator spilled
// 450-630 bytes per lane to SCRATCH inside the step loop (`.amdhsa_private_segment_fixed_size`).  One wave per SIMD for DT = 2: the
// overflow goes to AGPRs (v_accvgpr_*), no scratch -- Rough Well d = 32 / 16 384 chains 62.1 -> 39.4 us per proposal, d = 96 ... 128
// 207 -> 160 (it had been SLOWER than the f32-input form it replaced: 44.9 / 181).  DT = 1 keeps the two-wave bound (8192 chains =
// two workgroups per CU; and with 512 registers the scheduler's choices cost 5-7 % there), and so does the f32-input form of
// DT = 2 (44-184 bytes of scratch; one wave per SIMD measured 5-9 % slower: dense Gaussian d = 32 41.9 -> 45.6 us).
#ifndef L2HMC_FAST_WAVES_DT2
#define L2HMC_FAST_WAVES_DT2 1
#endif
__global__ __launch_bounds__(64 * NW, (DT >= 2 && PK == 1) ? L2HMC_FAST_WAVES_DT2 : L2HMC_FAST_WAVES) void traj_fast_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  static_assert(DT <= 2, "the fast kernel keeps layer-1 and tail fragments in registers");
  const int tid = threadIdx.x, lane = tid & 63, nthr = 64 * NW;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  const int NT = A.NT, DP = 16 * NT, NF = net_floats(NT);
  const float LOG2E = 1.4426950408889634f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  constexpr int NTp = NW * DT;
  constexpr bool F16 = PK == 1;
  const int FWN = fast_fw_net(NTp, F16), DPp = fast_dpp(NTp), FCN = fast_fc_net(NTp), R = fast_rec(NTp),
            RECD = fast_rec_dir(NTp, A.T);

  // ---- prologue: stage the tail fragments (scaled), the constant tables and the schedule records ----
  {
    // (eight loads in flight per thread: one element per trip, each waited for before the next, made the prologue ~7 us)
    constexpr int GN = 3 * NTp + 1;                 // groups per net: W4, then (S, T, Q) per dimension tile
    constexpr int NCP = 2 * GN * 64, NTRIP = (NCP + 64 * NW - 1) / (64 * NW);
    for (int t0 = 0; t0 < NTRIP; t0 += 8) {
      f4 buf[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = tid + (t0 + u) * nthr;
        const int net = i >= GN * 64, j = i - net * (GN * 64), g = j >> 6;
        buf[u] = splat(0.f);
        if (i < NCP && g < 3 * NT + 1) buf[u] = reinterpret_cast<const f4*>(A.packed + (size_t)net * NF + (2 * NT + 1) * 256)[j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = tid + (t0 + u) * nthr;
        const int net = i >= GN * 64, j = i - net * (GN * 64), g = j >> 6;
        float sc = 1.f;
        if (g > 0) sc = ((g - 1) % 3 == 1) ? (net == 0 ? eps : heps) : 2.f * LOG2E;
        if constexpr (F16) {
          if (i < NCP) {
            const WF16 f = wsplit16(buf[u] * sc);
            float* base = smem + A.o_fw + net * FWN;
            reinterpret_cast<h8v*>(base)[j] = f.a1;
            reinterpret_cast<h8v*>(base + GN * 256)[j] = f.a2;
          }
        } else {
          if (i < NCP) reinterpret_cast<f4*>(smem + A.o_fw + net * FWN)[j] = buf[u] * sc;
        }
      }
    }
  }
  for (int i = tid; i < 2 * 16 * NTp; i += nthr) {
    const int net = i / (16 * NTp), dim = i % (16 * NTp);
    const float* scl = A.packed + (size_t)net * NF + net_groups(NT) * 256;
    const float epn = net == 0 ? eps : heps;
    const float es = dim < DP ? scl[dim] : 0.f, eq = dim < DP ? scl[DP + dim] : 0.f;
    const float cs = es * epn * LOG2E, cq = eq * eps * LOG2E;
    float* fc = smem + A.o_fc + net * FCN;
    fc[dim] = cs;
    fc[DPp + dim] = -cs;
    fc[2 * DPp + dim] = cq;
    fc[3 * DPp + dim] = cq + log2f(epn);
  }
#pragma unroll 4
  for (int i = tid; i < 2 * A.T * R; i += nthr) {      // (branch-free body: clamped loads, so that unrolled trips overlap)
    const int dr = i / (A.T * R), r = (i / R) % A.T, j = i % R;
    const bool tb = j < 32;
    const int net = tb ? j >> 4 : 0, u = j & 15, dim = tb ? 0 : j - 32;
    const float* tf = A.packed + (size_t)net * NF + (2 * NT * 64) * 4;
    const float a0 = tf[u * 4], a1 = tf[(16 + u) * 4], a2 = tf[(32 + u) * 4], c0 = A.trig[2 * r], c1 = A.trig[2 * r + 1];
    const float mk = A.masks[r * A.d + (dim < A.d ? dim : 0)];
    const float m = dim < A.d ? mk : 0.f;
    // time / bias row, or the mask row: forward keeps m first, backward keeps 1 - m first
    smem[A.o_rec + dr * RECD + (r + 1) * R + j] = tb ? fmaf(a0, c0, fmaf(a1, c1, a2)) : (dr ? m : 1.f - m);
  }
  stage_energy<EK, false>(A, smem, tid, nthr);

  f4 x[DT], v[DT], g[DT];
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr || A.u != nullptr ||
                      (A.rng_flags & L2HMC_RNG_U) != 0;
  __syncthreads();

  const float* fwx = smem + A.o_fw;
  const float* fwv = fwx + FWN;
  const float* fcx = smem + A.o_fc;
  const float* fcv = fcx + FCN;
  int pb = 0;
  const f4 Z = splat(0.f);
  EnergyRegs<EK, DT> er;
  load_energy_regs<EK, DT, NW>(er, A, smem, w, lane);
  // grad U (and, at the end points only, this lane's share of U)
  auto grad = [&](const f4 (&xx)[DT], f4 (&gg)[DT], float& Up, bool wantU) {
    if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
#pragma unroll
      for (int t = 0; t < DT; ++t) gg[t] = er.prec[t] * (xx[t] - er.mu[t]);
    } else {
      grad_energy<EK, DT, NW>(A, smem, w, lane, xx, gg, Up, wantU, &er);
    }
  };
  auto diag_U = [&](const f4 (&xx)[DT], const f4 (&gg)[DT]) {
    float U = 0.f;
    if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
#pragma unroll
      for (int t = 0; t < DT; ++t) U += 0.5f * hsum((xx[t] - er.mu[t]) * gg[t]);
    }
    return U;
  };
  float U_start = 0.f;
  grad(x, g, U_start, need_p);
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) U_start = diag_U(x, g);

  L1W<DT> l1w;
  load_l1w<DT, NW>(l1w, A.packed, A.packed + NF, A, w, lane);
  TailK<DT, F16> tk;
  f4 pv[1];
  // VNet layer 1 at (x, grad U(x)).  Diagonal Gaussian: grad U = P (x - mu) is linear in x, so
  //   W1^T x + W2^T grad U = (W1 + P W2)^T x - W2^T P mu:
  // the precision is folded into the register-resident W1 fragments once per launch (fragment element
  // (lane, r) belongs to dimension 16 tg + 4 q + r -- this lane's own slice of P) and the constant goes
  // into the VNet time/bias table: one contraction per step instead of two.
  L1W16<F16 ? DT : 1> l1h;       // (filled below, after the precision has been folded into va)
  auto vnet_l1 = [&](const f4 (&xx)[DT], const f4 (&gg)[DT]) {
    if constexpr (F16) {
      if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) return l1_part16<DT>(xx, Z, l1h.va);
      else return l1_part16<DT>(gg, l1_part16<DT>(xx, Z, l1h.va), l1h.vb);
    } else if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
      return l1_part<DT, NW>(nullptr, 0, A, w, lane, xx, Z, l1w.va);
    } else {
      return l1_part<DT, NW>(nullptr, 0, A, w, lane, xx, Z, l1w.va) + l1_part<DT, NW>(nullptr, NT, A, w, lane, gg, Z, l1w.vb);
    }
  };
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
    f4 pm[DT], cv[1];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      l1w.va[t] = l1w.va[t] + er.prec[t] * l1w.vb[t];
      pm[t] = -(er.prec[t] * er.mu[t]);
    }
    cv[0] = l1_part<DT, NW>(nullptr, NT, A, w, lane, pm, Z, l1w.vb);
    xchg<NW, 1>(cv, A, smem, w, lane, pb);
    if (w == 0 && c == 0) {
      for (int i = 0; i < 2 * A.T; ++i) {
        float* tb = smem + A.o_rec + (i / A.T) * RECD + (i % A.T + 1) * R + 16 + 4 * q;
        *reinterpret_cast<f4*>(tb) = lds4(tb) + cv[0];
      }
    }
    __syncthreads();
  }
  if constexpr (F16) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      l1h.xa[t] = wsplit16(l1w.xa[t]);
      l1h.xb[t] = wsplit16(l1w.xb[t]);
      l1h.va[t] = wsplit16(l1w.va[t]);
      l1h.vb[t] = wsplit16(l1w.vb[t]);
    }
  }
  // (Measured and not kept, profiles/r04_bf16x3_heads.txt: a head start of 256-1024 cycles for every second workgroup at 8192 chains,
  //  so that one workgroup's MFMA-heavy tails fall into the other's exchange stalls -- 38.9-39.4 us per proposal against 38.1-39.4.)
  PT_DECL;
  PT_MARK(0);      // prologue
  pv[0] = vnet_l1(x, g);
  xchg<NW, 1>(pv, A, smem, w, lane, pb);

  // ---- persistent sampler loop (as traj_kernel) ------------------------------------------------------
  const long long gchain = A.chain_off + chain;
  const bool rng_v = (A.rng_flags & L2HMC_RNG_V) != 0, rng_d = (A.rng_flags & L2HMC_RNG_DIR) != 0;
  const bool rng_u = (A.rng_flags & L2HMC_RNG_U) != 0;
  f4 vn[DT];
  if (!rng_v) load_state<DT, NW>(A.v, A, chain, live, w, q, vn);
  bool fwd_n = (A.dir != nullptr && !rng_d) ? (live ? A.dir[chain] != 0 : true) : (A.dir_all != 0);
  float u_n = (A.u != nullptr && !rng_u && live) ? A.u[chain] : 0.f;
  const bool have_u = A.u != nullptr || rng_u;
  for (int m = 0; m < A.M; ++m) {
    const long long moff = (long long)m * A.N;
    const unsigned long long prop = A.rng_prop0 + (unsigned long long)m;
    if (rng_v) {
      rng_state<DT, NW>(A, gchain, prop, w, q, v);
    } else {
#pragma unroll
      for (int t = 0; t < DT; ++t) v[t] = vn[t];
    }
    bool fwd = fwd_n;
    float u_m = u_n;
    if (rng_d || rng_u) {
      bool fr;
      float ur;
      philox_dir_u(A.rng_seed, gchain, prop, fr, ur);
      if (rng_d) fwd = fr;
      if (rng_u) u_m = ur;
    }
    if (m + 1 < A.M) {
      if (!rng_v) load_state<DT, NW>(A.v + (moff + A.N) * A.d, A, chain, live, w, q, vn);
      if (A.dir != nullptr && !rng_d && live) fwd_n = A.dir[moff + A.N + chain] != 0;
      if (A.u != nullptr && !rng_u && live) u_n = A.u[moff + A.N + chain];
    }
    // the start point: a rejected chain resumes from it (sampler.py:53-55)
    // (the diagonal Gaussian's grad U is two packed instructions away from x: it is re-formed from the selected state at the end of
    //  the proposal -- the same expression on the same x, the same bits -- instead of being held in registers through the trajectory)
    constexpr bool KEEP_G0 = EK != L2HMC_ENERGY_GAUSS_DIAG;
    f4 x0[DT], g0[KEEP_G0 ? DT : 1];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      x0[t] = x[t];
      if constexpr (KEEP_G0) g0[t] = g[t];
    }
    const f4 pv0 = pv[0];
    float red[F16 ? 6 : 5];        // U0, K0, U1, K1, logdet (per-lane partial sums); f16x2: + the out-of-range flag
    float amax_l = 0.f;
    if constexpr (F16) {
#pragma unroll
      for (int t = 0; t < DT; ++t) amax_l = fmaxf(amax_l, fmaxf(amax4(x[t]), fmaxf(amax4(v[t]), amax4(g[t]))));
    }
    red[0] = U_start;
    red[1] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) red[1] += 0.5f * hsum(v[t] * v[t]);
    red[2] = 0.f;
    f4 ldv = splat(0.f);

    // per-lane direction constants and table addresses
    const float ff = fwd ? 1.f : 0.f, nf = ff - 1.f;
    const int dofs = fwd ? 0 : DPp;
    const int row0 = fwd ? A.step_begin : (A.T - 1 - A.step_begin);
    const float* rec = smem + A.o_rec + (fwd ? RECD : 0) + (row0 + 1) * R + 4 * q;   // this lane's record, + 4 q
    const int drec = fwd ? R : -R;
    f4 tbv = lds4(rec + 16);

    // Resident tails (round 6, profiles/r06_resident_tails.txt): XNet's tail fragments and constants stay in registers for the whole
    // proposal beside VNet's (44 VGPRs per net as f16x2) instead of being re-read from LDS twice per step -- 22 of a step's 37
    // ds_read_b128: ICG-50 20.4 -> 19.4-19.6 us per proposal at 4096 chains, 31.3 -> 30.0 at 8192 (251 VGPRs, no scratch).  Not for
    // the f32-input form of DT = 2, which spills under the two-wave bound already.  Same values, same arithmetic: same bits.
    constexpr bool RT = DT == 1 || PK == 1;
    TailK<DT, F16> tkx;
    if constexpr (RT) load_tailk<DT>(tkx, fwx, fcx, dofs, NTp, w, lane);
    const TailK<DT, F16>& TKX = *(RT ? &tkx : &tk);
    load_tailk<DT>(tk, fwv, fcv, dofs, NTp, w, lane);
    // VNet's evaluation for the next step's first half-update rides beside this step's second one (tail_fast2 above; f16x2 with
    // resident tails; -DL2HMC_FAST_SERIAL_TAILS: four tails per step, one after the other)
#ifndef L2HMC_FAST_SERIAL_TAILS
#ifndef L2HMC_FAST_PAIR_DT2
#define L2HMC_FAST_PAIR_DT2 0
#endif
    constexpr bool PAIR = F16 && RT && (DT == 1 || L2HMC_FAST_PAIR_DT2);      // (DT = 2: measured neutral, profiles/r06_paired_tails.txt)
#else
    constexpr bool PAIR = false;
#endif
    f4 aS_n[DT], T_n[DT], EQ_n[DT];
    if constexpr (PAIR) {
      tail_fast<DT, KH>(tk, pv[0] + tbv, [&](int t, f4 aS, f4 T, f4 EQ) { aS_n[t] = aS; T_n[t] = T; EQ_n[t] = EQ; });
    }
    for (int it = 0; it < A.n_steps; ++it) {
      f4 k1[DT], vh[DT], y[DT], xin[DT];
      const f4 tbx = lds4(rec);
#pragma unroll
      for (int t = 0; t < DT; ++t) k1[t] = lds4(rec + 32 + 16 * (w * DT + t));
      rec += drec;
      const f4 tbv_n = lds4(rec + 16);      // (the last step reads the unused pad row)

      PT_MARK(1);  // step head
      // ---- momentum half-update #1: VNet([x, grad U(x), t])  (dynamics.py:118-125 / :162-170)
      auto vhalf1 = [&](int t, f4 aS, f4 T, f4 EQ) {
        const f4 ES = ex2_4(aS);
        ldv += aS;
        const f4 tr = T - EQ * g[t];
        vh[t] = ES * (nf * tr + v[t]) + ff * tr;
      };
      if constexpr (PAIR) {
#pragma unroll
        for (int t = 0; t < DT; ++t) vhalf1(t, aS_n[t], T_n[t], EQ_n[t]);
      } else {
        tail_fast<DT, KH>(tk, pv[0] + tbv, vhalf1);
      }
      PT_MARK(2);  // VNet tail #1

      // ---- two masked position updates: XNet([v_h, kept * x, t])  (:127-145 / :172-190)
      if constexpr (!RT) load_tailk<DT>(tk, fwx, fcx, dofs, NTp, w, lane);
#pragma unroll
      for (int t = 0; t < DT; ++t) xin[t] = k1[t] * x[t];
      f4 pa, px[1];
      if constexpr (F16) {
        pa = l1_part16<DT>(vh, Z, l1h.xa);
        px[0] = l1_part16<DT>(xin, pa, l1h.xb);
      } else {
        pa = l1_part<DT, NW>(nullptr, 0, A, w, lane, vh, Z, l1w.xa);
        px[0] = pa + l1_part<DT, NW>(nullptr, NT, A, w, lane, xin, Z, l1w.xb);
      }
      PT_MARK(3);  // XNet layer-1 partials (a, b)
      xchg<NW, 1>(px, A, smem, w, lane, pb);
      PT_MARK(4);  // exchange
      tail_fast<DT, KH>(TKX, px[0] + tbx, [&](int t, f4 aS, f4 T, f4 EQ) {
        const f4 up = 1.f - k1[t];
        const f4 aSm = up * aS;
        const f4 ES = ex2_4(aSm);
        ldv += aSm;
        const f4 tr = up * (EQ * vh[t] + T);
        y[t] = ES * (nf * tr + x[t]) + ff * tr;
        xin[t] = up * y[t];
      });
      PT_MARK(5);  // XNet tail #1
      f4 py[1];
      if constexpr (F16) py[0] = l1_part16<DT>(xin, pa, l1h.xb);
      else py[0] = pa + l1_part<DT, NW>(nullptr, NT, A, w, lane, xin, Z, l1w.xb);
      PT_MARK(6);  // XNet layer-1 partial (b only)
      xchg<NW, 1>(py, A, smem, w, lane, pb);
      PT_MARK(7);  // exchange
      tail_fast<DT, KH>(TKX, py[0] + tbx, [&](int t, f4 aS, f4 T, f4 EQ) {
        const f4 aSm = k1[t] * aS;
        const f4 ES = ex2_4(aSm);
        ldv += aSm;
        const f4 tr = k1[t] * (EQ * vh[t] + T);
        x[t] = ES * (nf * tr + y[t]) + ff * tr;
      });
      PT_MARK(8);  // XNet tail #2

      // ---- momentum half-update #2 at the new position  (:147-153 / :192-199)
      if constexpr (!RT) load_tailk<DT>(tk, fwv, fcv, dofs, NTp, w, lane);
      grad(x, g, red[2], need_p && it == A.n_steps - 1);
      pv[0] = vnet_l1(x, g);
      PT_MARK(9);  // grad U + VNet layer-1 partials
      xchg<NW, 1>(pv, A, smem, w, lane, pb);
      PT_MARK(10); // exchange
      auto vhalf2 = [&](int t, f4 aS, f4 T, f4 EQ) {
        const f4 ES = ex2_4(aS);
        ldv += aS;
        const f4 tr = T - EQ * g[t];
        v[t] = ES * (nf * tr + vh[t]) + ff * tr;
      };
      if constexpr (PAIR) {
        tail_fast2<DT, KH>(tk, pv[0] + tbv, pv[0] + tbv_n, vhalf2, [&](int t, f4 aS, f4 T, f4 EQ) { aS_n[t] = aS; T_n[t] = T; EQ_n[t] = EQ; });
      } else {
        tail_fast<DT, KH>(tk, pv[0] + tbv, vhalf2);
      }
      PT_MARK(11); // VNet tail #2
      tbv = tbv_n;
    }
    if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) red[2] = diag_U(x, g);
    const float ld = hsum(ldv) * 0.6931471805599453f;   // the log-det was accumulated in log2 units

    // ---- per-proposal epilogue: proposal, log-det, accept probability, MH select -------------------
    const bool last = m == A.M - 1;
    red[3] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) red[3] += 0.5f * hsum(v[t] * v[t]);
    red[4] = ld;
    const float U_end = red[2];
    if constexpr (F16) {
#pragma unroll
      for (int t = 0; t < DT; ++t) amax_l = fmaxf(amax_l, fmaxf(amax4(x[t]), fmaxf(amax4(v[t]), amax4(g[t]))));
      red[5] = amax_l < L2HMC_F16_STATE_MAX ? 0.f : 1.f;
    }
    chain_allreduce<NW, F16 ? 6 : 5>(red, smem + A.o_red, w, lane);
    if constexpr (F16) {
      if (red[5] > 0.f) {            // outside the f16x2 range: a loud non-result (see L2HMC_F16_STATE_MAX)
        const float qnan = __uint_as_float(0x7fc00000u);
#pragma unroll
        for (int t = 0; t < DT; ++t) { x[t] = splat(qnan); v[t] = splat(qnan); }
        red[4] = qnan;
      }
    }
    if (last) {
      store_state<DT, NW>(A.x_out, A, chain, live, w, q, x);
      store_state<DT, NW>(A.v_out, A, chain, live, w, q, v);
    }
    const bool writer = live && w == 0 && lane < 16;
    if (A.logjac_out != nullptr && writer) A.logjac_out[moff + chain] = red[4];
    if (need_p) {
      const float val = (red[0] + red[1]) - (red[2] + red[3]) + red[4];       // dynamics.py:302-309
      const float p = accept_prob(val);
      if (A.p_out != nullptr && writer) A.p_out[moff + chain] = p;
      if (have_u) {
        const bool acc = live && (p - u_m) >= 0.f;                            // sampler.py:53-55
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          x[t] = sel4(acc, x[t], x0[t]);
          if constexpr (KEEP_G0) g[t] = sel4(acc, g[t], g0[t]);
          else g[t] = er.prec[t] * (x[t] - er.mu[t]);
        }
        pv[0] = sel4(acc, pv[0], pv0);
        U_start = acc ? U_end : U_start;
      } else {
        U_start = U_end;
      }
    } else {
      U_start = U_end;
    }
    if (A.x_hist != nullptr) store_state<DT, NW>(A.x_hist + moff * A.d, A, chain, live, w, q, x);
  }  // proposals
  PT_FLUSH(w, lane);
  store_state<DT, NW>(A.x_next, A, chain, live, w, q, x);
}

}  // namespace l2hmc
