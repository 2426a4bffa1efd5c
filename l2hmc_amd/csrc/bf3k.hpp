// bf3k.hpp -- fp32-accurate 16 x 16 x 16 contractions on the bf16 matrix pipe of gfx950, three MFMAs each ("bf16x3, K-packed").
//
// On gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32, 32 cycles) blocks every VALU instruction of its SIMD, the bf16 MFMAs
// (16.3 cycles for K = 16 and for K = 32 alike) do not (profiles/r03_ubench_issue.txt).  An fp32 number is exactly the sum of
// three bf16 numbers, x = h + m + l (round-to-nearest at each level), and of the nine cross products of two such sums the six
// with weight >= 2^-16 carry fp32 accuracy (the dropped ones are <= 3 x 2^-24 |x y|, one fp32 rounding).  Round 3 issued those
// six as six K = 16 MFMAs (96 cycles per 16 x 16 x 16 block against 128 for four f32 k-steps -- no gain, measured).  The K = 32
// instruction costs the same 16 cycles, so TWO of the six products ride in one instruction: physical k-slot 8 q + j of lane
// (., q) carries part one of logical k = 4 q + j for j < 4 and part two of logical k = 4 q + (j - 4) for j >= 4 -- both parts
// are split from the lane's OWN float4 (k-step r of the f32 form is element r of the same float4), no cross-lane traffic:
//
//      A operand (weights, split once when they are staged)        B operand (activations, split by the consuming wave)
//      P = [a_h | a_m]                                             hh = [b_h | b_h]    ->  a_h b_h + a_m b_h
//      P = [a_h | a_m]                                             mm = [b_m | b_m]    ->  a_h b_m + a_m b_m
//      Q = [a_l | a_h]                                             hl = [b_h | b_l]    ->  a_l b_h + a_h b_l
//
// = 3 MFMAs x 16 cycles per 16 x 16 x 16 block (f32 form: 4 x 32, or 3 x 32 for the K = 12 hidden contractions), none of them
// blocking the VALU.  The D layout is that of the 16x16x4 form (row 4 q + r, column c), so everything downstream is unchanged.
// The split costs ~20 VALU per float4 (6 v_cvt_pk_bf16_f32, 8 shift / and, 4 subtractions, tuple copies).
#pragma once
#include <hip/hip_runtime.h>

namespace l2hmc {

typedef float bfk_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfk_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned bfk_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bfk_pk(float a, float b) {          // v_cvt_pk_bf16_f32: a -> low half, b -> high half (RNE)
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
struct BfkHml { unsigned h[2], m[2], l[2]; };
__device__ __forceinline__ BfkHml bfk_split(bfk_f4 x) {
  BfkHml o;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    const unsigned ph = bfk_pk(a, b);
    const float ra = a - __uint_as_float(ph << 16), rb = b - __uint_as_float(ph & 0xffff0000u);
    const unsigned pm = bfk_pk(ra, rb);
    const float sa = ra - __uint_as_float(pm << 16), sb = rb - __uint_as_float(pm & 0xffff0000u);
    o.h[i] = ph; o.m[i] = pm; o.l[i] = bfk_pk(sa, sb);
  }
  return o;
}
struct BfkW { bfk_u4 P, Q; };            // A operand of one 16 x 16 x 16 block (32 bytes per lane)
struct BfkA { bfk_u4 hh, mm, hl; };      // B operand
__device__ __forceinline__ BfkW bfk_wfrag(bfk_f4 w) {
  const BfkHml s = bfk_split(w);
  return BfkW{bfk_u4{s.h[0], s.h[1], s.m[0], s.m[1]}, bfk_u4{s.l[0], s.l[1], s.h[0], s.h[1]}};
}
__device__ __forceinline__ BfkA bfk_afrag(bfk_f4 x) {
  const BfkHml s = bfk_split(x);
  return BfkA{bfk_u4{s.h[0], s.h[1], s.h[0], s.h[1]}, bfk_u4{s.m[0], s.m[1], s.m[0], s.m[1]}, bfk_u4{s.h[0], s.h[1], s.l[0], s.l[1]}};
}
__device__ __forceinline__ bfk_f4 bfk_mfma(bfk_u4 a, bfk_u4 b, bfk_f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfk_bf8, a), __builtin_bit_cast(bfk_bf8, b), c, 0, 0, 0);
}
// acc + W^T x over the block's 16 logical k (smallest terms first)
__device__ __forceinline__ bfk_f4 bfk_dot(const BfkW& w, const BfkA& a, bfk_f4 acc) {
  acc = bfk_mfma(w.Q, a.hl, acc);
  acc = bfk_mfma(w.P, a.mm, acc);
  return bfk_mfma(w.P, a.hh, acc);
}
// (Round 4 also had `bfk_heads3` here: the split of one float4 software-pipelined under the nine MFMAs of the three head blocks
//  that share it, for ONE wave per SIMD -- traj_fast_kernel's experiment.  Measured no gain (the split is 143 cycles of VALU issue
//  whatever runs beside it: profiles/r04_bf16x3_heads.txt); removed with its only caller in round 5.  The product's user of this
//  header is traj_tile_kernel: bfk_afrag once per net evaluation + bfk_dot per dimension slice.)

// fragment storage: block b of a table of blocks = 2 x 64 x 16 bytes, P then Q, lane-major (conflict-free ds_read_b128)
__device__ __forceinline__ void bfk_store(float* base, int block, int lane, const BfkW& w) {
  bfk_u4* p = reinterpret_cast<bfk_u4*>(base) + (size_t)block * 128 + lane;
  p[0] = w.P;
  p[64] = w.Q;
}
__device__ __forceinline__ BfkW bfk_load(const float* base, int block, int lane) {
  const bfk_u4* p = reinterpret_cast<const bfk_u4*>(base) + (size_t)block * 128 + lane;
  return BfkW{p[0], p[64]};
}

}  // namespace l2hmc
