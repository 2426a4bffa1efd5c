// l2hmc_kernels.hpp -- device templates of the fused L2HMC leapfrog kernels (gfx950 / CDNA4).
//
// One fused kernel runs a whole trajectory (T generalised leapfrog steps: grad U, VNet,
// momentum half-update, XNet x2 with masked position updates, grad U, VNet, momentum
// half-update, log|det J|) followed by the Metropolis accept probability and MH select.
// Replaces utils/dynamics.py:115-309 and utils/sampler.py:28-55 of the reference.
//
// Mapping:
//   * a workgroup owns a tile of 16 chains; NW waves (1 or 4) split the d dimensions in
//     16-wide tiles, DT tiles per wave;
//   * state layout ("S-layout"): lane l = (c = l & 15, q = l >> 4) holds, for each of its
//     tiles tg, the float4 {z[c][16 tg + 4 q + r]}, r = 0..3 -- which is at the same time
//       - the B operand (k = q) of v_mfma_f32_16x16x4_f32 for k-step r, and
//       - the C/D layout (row = 4 q + r, col = c) of a product whose rows are dimensions.
//     So every layer is computed TRANSPOSED, out^T[unit, chain] = W^T[unit, k] in^T[k, chain]
//     with the weights as the A operand, and activations flow layer to layer with no
//     cross-lane traffic at all;
//   * weights are pre-ordered into A-operand fragments (l2hmc_pack_nets), staged once per
//     workgroup into LDS and read with ds_read_b128 (4 k-steps per read);
//   * biases ride on a constant-1 hidden unit, so there are no bias adds;
//   * per-chain reductions (|v|^2, U, log-det) are per-lane partial sums, reduced once at
//     the end with two wave shuffles (+ one LDS hop when NW = 4);
//   * the energy kind EK is a template parameter: each target gets its own straight-line
//     kernel (compiled in its own translation unit, traj_ek<k>.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/l2hmc.h"

namespace l2hmc {

typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Geometries with DT >= 4 tiles per wave (d > 128 with NW = 4, or the one-wave-per-tile kernels
// for d > 32) read the packed weight / precision fragments straight from global memory (L2-hot,
// prefetched a phase ahead) instead of staging them in LDS: beyond d ~ 190 the two nets no longer
// fit the 160 KiB, and for the NW = 1 kernels a 45 KiB staging buffer per 64-thread workgroup
// would cap the occupancy.
__host__ __device__ constexpr bool weights_in_global(int DT) { return DT >= 4; }

int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0);
// Rough Well (distributions.py:93): the divisor of the cosine argument.  The reference forms the Python-DOUBLE product
// eps * eps and TF rounds it to float32 once; a binding that holds the double passes that float in L2hmcEnergy.den.  den = 0:
// derived here from the float eta (the exact product of two floats rounded once -- the same value whenever eta is a float).
inline float roughwell_den(const L2hmcEnergy* e) {
  if (e->den > 0.f) return e->den;
  return e->easy ? e->eta : (float)((double)e->eta * (double)e->eta);
}
void note_kernel(const char* fmt, long long a = 0, long long b = 0, long long c = 0, long long d = 0);   // -> l2hmc_last_kernel

// Debug builds with -DL2HMC_LDS_POISON (tools/build_variant_full.sh poison -DL2HMC_LDS_POISON): every kernel that works out of
// dynamic LDS first fills ALL of it with NaN bit patterns.  LDS is not cleared between workgroups, so a kernel that reads a word
// it never staged normally sees whatever the previous workgroup left there -- often plausible numbers, and parity tests pass by
// luck; with the poison such a read turns into NaN and the same tests fail (profiles/r04_lds_poison.txt: the pass of round 4).
// group_segment_size is read from the dispatch packet (hsa_kernel_dispatch_packet_t, byte offset 28).
__device__ __forceinline__ void lds_poison(float* smem) {
#if defined(L2HMC_LDS_POISON) && defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(4))) const unsigned* cptr_t;
  const unsigned total = ((cptr_t)__builtin_amdgcn_dispatch_ptr())[7];
  const unsigned words = (total - __builtin_amdgcn_groupstaticsize()) / 4;
  const unsigned nthr = blockDim.x * blockDim.y * blockDim.z;
  for (unsigned i = threadIdx.x; i < words; i += nthr) reinterpret_cast<unsigned*>(smem)[i] = 0xffffffffu;
  __syncthreads();
#endif
}
int check_energy(const L2hmcEnergy* e, int d);
struct KArgs;
// traj_wide.hip: the LDS-resident-state kernel for d > 256 (elementwise energies)
long long plan_lds_wide(KArgs& k);
int launch_wide(const KArgs& k, int KH, long long lds, hipStream_t s);

// ------------------------------------------------------------------------------------------
// Packed layouts (shared by host and device)
// ------------------------------------------------------------------------------------------
// A "group" is 4 A-operand fragments interleaved per lane: element (lane, r) at
// (group * 64 + lane) * 4 + r, so one ds_read_b128 fetches the operands of 4 k-steps.
//
// Hidden units live on D rows 4 q + r with r < KH ("live" rows); unit u <-> (q = u / KH,
// r = u % KH).  Unit H is the constant-1 bias unit.  KH = ceil((H + 1) / 4) k-steps contract
// over the hidden layer.
__host__ __device__ inline int net_groups(int NT) { return 5 * NT + 2; }
__host__ __device__ inline int net_floats(int NT) { return net_groups(NT) * 256 + 32 * NT; }
__host__ __device__ inline int gauss_floats(int NT) { return NT * NT * 256; }
// the same groups as f16x2 fragments (traj_fast.hpp: [64 w_hi | w_hi] then [64 w_lo | w_lo], 16 bytes per lane each; all first
// fragments of a net, then all second ones): what traj_wide_kernel streams from L2 for the elementwise targets
__host__ __device__ inline int net_f16_floats(int NT) { return net_groups(NT) * 512; }

inline int tiles_of(int d) { return (d + 15) / 16; }
inline int khid_of(int H) { return (H + 1 + 3) / 4; }

// ------------------------------------------------------------------------------------------
// Device-side argument block
// ------------------------------------------------------------------------------------------
struct KArgs {
  const float* packed;
  const float* packed16;     // the f16x2 fragments of both nets (net_f16_floats each) behind the lane layout, or NULL
  const float* masks;
  const float* trig;
  const float* alpha;
  float eps_host;
  long long N;
  int d, H, T, step_begin, n_steps, NT;
  const float *x, *v;
  const unsigned char* dir;
  int dir_all;
  const float* u;
  float *x_out, *v_out, *logjac_out, *p_out, *x_next, *x_hist;
  int M;                     // proposals per launch (persistent sampler loop)
  unsigned rng_flags;        // L2HMC_RNG_*: which draws come from the in-kernel Philox
  unsigned long long rng_seed, rng_prop0;
  long long chain_off;
  // energy
  int ekind, ncomp, easy;
  const float *mu, *prec, *logc;
  float eta, temperature;
  float den;                 // Rough Well: the divisor of the cosine argument (L2hmcEnergy.den, or derived from eta in float32)
  float beta;                // AIS bridge (utils/ais.py:46-47): U := (1 - beta) |x|^2 / 2 + beta U;  1 = off
  // p_accept-only kernel inputs
  const float *x1, *v1, *logjac_in;
  float *U_out, *grad_out;
  // LDS offsets (floats)
  int o_mask, o_trig, o_tb, o_P, o_XB, o_red, o_mu, o_prec, o_logc, xb_stride;
  int o_state;               // traj_wide_kernel: x, v, grad U of the tile (3 x NT x 256 floats)
  // AIS mode of the persistent loop (utils/ais.py:43-66): proposal m is anneal step m with beta = ais_beta[m]
  const float* ais_beta;     // (M) or NULL
  const float* ais_v0;       // (N, d) momentum before the first step (only read when refreshing), or NULL -> Philox
  float ais_dbeta, ais_refresh;   // refresh < 0: fresh momenta every step (ais.py:57)
  float *ais_w, *ais_alpha;  // (N) log-weights / summed accept probabilities, accumulated
  int o_fw, o_fc, o_rec;     // traj_fast_kernel: staged tail fragments, constant tables, schedule records
  unsigned long long* dbg;   // phase-timing buffer (profiling builds only, else NULL)
};

// Phase timers: compiled in only with -DL2HMC_PHASE_TIMING (tools/phase_timing.py).  Wave w of
// block 0 accumulates s_memtime deltas per phase into dbg[w * 16 + phase].
#ifdef L2HMC_PHASE_TIMING
#define PT_DECL unsigned long long pt_t0 = __builtin_amdgcn_s_memtime(), pt_acc[12] = {0}
#define PT_MARK(i)                                                     \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long pt_t1 = __builtin_amdgcn_s_memtime();     \
    pt_acc[i] += pt_t1 - pt_t0;                                        \
    pt_t0 = pt_t1;                                                     \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#define PT_FLUSH(w, lane)                                                                \
  do {                                                                                   \
    if (A.dbg != nullptr && blockIdx.x == 0 && (lane) == 0)                              \
      for (int pt_i = 0; pt_i < 12; ++pt_i) A.dbg[(w) * 16 + pt_i] = pt_acc[pt_i];       \
  } while (0)
#else
#define PT_DECL
#define PT_MARK(i)
#define PT_FLUSH(w, lane)
#endif

__device__ __forceinline__ f4 splat(float a) { return f4{a, a, a, a}; }
// Branch-free transcendental forms for the hot loop (ocml's expf / tanhf carry range and
// denormal branches that serialise the 12 independent chains of a tile):
//   exp(x)  = v_exp_f32(x * log2 e)                  rel. error <= ~1e-7 (1 + |x|)
//   tanh(z) = 1 - 2 / (1 + exp(2 z))  (v_exp + v_rcp) abs. error <= ~2e-7, exact limits +-1
// Both are applied to net outputs that only enter as eps * S, eps * Q (|.| < ~1); the parity
// tests against the golden vectors bound the effect ("numerics").
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float ftanh(float z) {
  const float t = __builtin_amdgcn_exp2f(z * 2.8853900817779268f);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + t), 1.f);
}
__device__ __forceinline__ f4 exp4(f4 a) { return f4{fexp(a.x), fexp(a.y), fexp(a.z), fexp(a.w)}; }
__device__ __forceinline__ f4 tanh4(f4 a) { return f4{ftanh(a.x), ftanh(a.y), ftanh(a.z), ftanh(a.w)}; }
__device__ __forceinline__ f4 exp2_4(f4 a) {
#ifdef L2HMC_ABL_NOTRANS
  return a + 1.f;
#endif
  return f4{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y), __builtin_amdgcn_exp2f(a.z),
            __builtin_amdgcn_exp2f(a.w)};
}
// c * tanh(z) with the constant folded in: c (1 - 2 / (1 + 2^(z * 2 log2 e)))
__device__ __forceinline__ f4 ctanh4(f4 c, f4 z) {
  const f4 t = exp2_4(z * 2.8853900817779268f);
  const f4 u = t + 1.f;
#ifdef L2HMC_ABL_NOTRANS
  return c * (u * -2.f + 1.f);
#endif
  const f4 r = f4{__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y), __builtin_amdgcn_rcpf(u.z),
                  __builtin_amdgcn_rcpf(u.w)};
  return c * (r * -2.f + 1.f);
}
// (an inline-asm single v_max_f32 was tried: it hides the MFMA->VALU read hazard from the
// compiler's hazard recogniser and produced wrong results -- keep the builtin)
__device__ __forceinline__ float relu1(float a) { return fmaxf(a, 0.f); }
__device__ __forceinline__ f4 relu4(f4 a) { return f4{relu1(a.x), relu1(a.y), relu1(a.z), relu1(a.w)}; }
__device__ __forceinline__ float hsum(f4 a) { return (a.x + a.y) + (a.z + a.w); }
// Sum over the four lanes (c, q = 0..3) = lanes c, c + 16, c + 32, c + 48 that hold one chain; every one of them gets the total.
// gfx950's row swaps do it on the VALU -- v_permlane16_swap_b32 exchanges the odd 16-lane rows of one register with the even rows of
// another, v_permlane32_swap_b32 the wave's halves -- where `v += __shfl_xor(v, 16); v += __shfl_xor(v, 32)` compiles to two
// dependent ds_bpermute_b32 (an LDS round trip each: ~100 cycles in front of a lone wave, four per leapfrog step in the mixtures'
// grad U).  Same pairs added in the same order, (r0 + r1) + (r2 + r3) in every lane: the same bits as the shuffle form (round 6).
typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float chain4_sum(float a) {
#ifdef L2HMC_CHAIN_SUM_BPERMUTE
  a += __shfl_xor(a, 16);
  a += __shfl_xor(a, 32);
  return a;
#else
  const u2v_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  const float b = __uint_as_float(r.x) + __uint_as_float(r.y);
  const u2v_ t = __builtin_amdgcn_permlane32_swap(__float_as_uint(b), __float_as_uint(b), false, false);
  return __uint_as_float(t.x) + __uint_as_float(t.y);
#endif
}
__device__ __forceinline__ f4 sel4(bool c, f4 a, f4 b) { return c ? a : b; }
__device__ __forceinline__ f4 lds4(const float* p) { return *reinterpret_cast<const f4*>(p); }

// ---- sin / cos of the Rough Well (distributions.py:84-97): arguments are x / eta (easy) or x / eta^2 ------------------------
// ocml's sinf / cosf carry the Payne-Hanek path for arguments up to 2^127; the `easy` sweep (BASELINE config 4) only ever sees
// |x / eta| of a few hundred.  For |a| < 8192 pi/2 the quadrant count n = rint(a 2/pi) fits 13 bits, so the three-term
// Cody-Waite reduction r = ((a - n P1) - n P2) - n P3 (P1, P2 with 8 / 11 significand bits: both products exact) is exact up to
// the last fma, and the two minimax polynomials on [-pi/4, pi/4] are good to 1 ulp (Cephes sinf / cosf constants): abs. error
// <= 1.2e-7 like ocml's, at a third of the instructions.  Larger arguments take ocml's path; the switch is WAVE-uniform
// (one lane outside the range sends its whole wave through ocml), so there is no divergence.
__device__ __forceinline__ void rw_sincos_core(float a, float& s, float& c) {
  const float n = __builtin_rintf(a * 0.6366197723675814f);
  float r = fmaf(n, -1.5703125f, a);
  r = fmaf(n, -4.837512969970703125e-4f, r);
  r = fmaf(n, -7.549789948768648e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                        fmaf(-0.5f, z, 1.f));
  const int k = (int)n;
  const float ss = (k & 1) ? pc : ps, cc = (k & 1) ? ps : pc;
  s = (k & 2) ? -ss : ss;
  c = ((k + 1) & 2) ? -cc : cc;
}
constexpr float RW_FAST_MAX = 12867.0f;          // 8192 * pi / 2
__device__ __forceinline__ f4 rw_sin4(f4 a) {
  const float mx = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
  if (__builtin_amdgcn_ballot_w64(!(mx < RW_FAST_MAX)) != 0) return f4{sinf(a.x), sinf(a.y), sinf(a.z), sinf(a.w)};
  float s0, s1, s2, s3, c0, c1, c2, c3;
  rw_sincos_core(a.x, s0, c0); rw_sincos_core(a.y, s1, c1); rw_sincos_core(a.z, s2, c2); rw_sincos_core(a.w, s3, c3);
  return f4{s0, s1, s2, s3};
}
__device__ __forceinline__ f4 rw_cos4(f4 a) {
  const float mx = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
  if (__builtin_amdgcn_ballot_w64(!(mx < RW_FAST_MAX)) != 0) return f4{cosf(a.x), cosf(a.y), cosf(a.z), cosf(a.w)};
  float s0, s1, s2, s3, c0, c1, c2, c3;
  rw_sincos_core(a.x, s0, c0); rw_sincos_core(a.y, s1, c1); rw_sincos_core(a.z, s2, c2); rw_sincos_core(a.w, s3, c3);
  return f4{c0, c1, c2, c3};
}
__device__ __forceinline__ float rw_sin1(float a) {
  if (__builtin_amdgcn_ballot_w64(!(fabsf(a) < RW_FAST_MAX)) != 0) return sinf(a);
  float s, c;
  rw_sincos_core(a, s, c);
  return s;
}
__device__ __forceinline__ float rw_cos1(float a) {
  if (__builtin_amdgcn_ballot_w64(!(fabsf(a) < RW_FAST_MAX)) != 0) return cosf(a);
  float s, c;
  rw_sincos_core(a, s, c);
  return c;
}

// dynamics.py:302-309: exp(min(dH + logjac, 0)) with non-finite results mapped to 0.  TF's
// `minimum` propagates NaN (fminf would not), so a NaN Hamiltonian difference gives p = 0.
__device__ __forceinline__ float accept_prob(float val) {
  const float mn = (val != val) ? val : fminf(val, 0.f);
  const float p = expf(mn);
  return (fabsf(p) <= 3.402823466e38f) ? p : 0.f;
}

// ---- counter-based RNG (K6): Philox4x32-10 (Salmon et al. 2011), same constants as
// Random123 / cuRAND.  counter = (global chain, dim / 4, proposal index, stream), key = seed.
struct U4 { unsigned x, y, z, w; };
__host__ __device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
    const U4 n = {(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// 4 standard normals from one Philox block (two Box-Muller pairs, 24-bit uniforms) on the hardware transcendentals:
// v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (which take the angle in REVOLUTIONS: the uniform itself) -- 8
// transcendentals + 6 VALU per block against ~200 instructions for libm's logf / sqrtf / sincosf with their range reductions.
// Normals within 2e-6 of the libm forms (tests/test_gpu_parity.py::test_in_kernel_philox_matches_oracle_stream, whose oracle uses
// numpy's).  Round 3 measured +2 % in the many-chain regime and left it (the draws feed the training runs of the ESS evidence);
// since round 4 the one-wave-per-tile kernel is bound by VALU issue (profiles/r04_tile_pmc.txt: the per-proposal random-number
// work was ~20 % of its VALU instructions), which changes the sum: profiles/r04_bf16x3_heads.txt.  -DL2HMC_LIBM_NORMALS restores
// the libm forms.
__device__ __forceinline__ f4 philox_normal4(unsigned long long seed, long long gchain, unsigned blk,
                                             unsigned long long prop) {
  const U4 r = philox4x32_10(U4{(unsigned)gchain, blk, (unsigned)prop, (unsigned)(prop >> 32) << 1},
                             (unsigned)seed, (unsigned)(seed >> 32));
  const float k = 5.9604644775390625e-08f;   // 2^-24
  const float u1 = ((r.x >> 8) + 1) * k, u2 = (r.y >> 8) * k, u3 = ((r.z >> 8) + 1) * k, u4 = (r.w >> 8) * k;
#ifdef L2HMC_LIBM_NORMALS
  const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
  float sa, ca, sb, cb;
  sincosf(6.283185307179586f * u2, &sa, &ca);
  sincosf(6.283185307179586f * u4, &sb, &cb);
#else
  const float m2ln2 = -1.3862943611198906f;   // -2 ln 2:  -2 ln u = (-2 ln 2) log2 u
  const float ra = __builtin_amdgcn_sqrtf(m2ln2 * __builtin_amdgcn_logf(u1)), rb = __builtin_amdgcn_sqrtf(m2ln2 * __builtin_amdgcn_logf(u3));
  const float sa = __builtin_amdgcn_sinf(u2), ca = __builtin_amdgcn_cosf(u2), sb = __builtin_amdgcn_sinf(u4), cb = __builtin_amdgcn_cosf(u4);
#endif
  return f4{ra * ca, ra * sa, rb * cb, rb * sb};
}
// direction bit and accept uniform of (chain, proposal): stream 1
__device__ __forceinline__ void philox_dir_u(unsigned long long seed, long long gchain,
                                             unsigned long long prop, bool& fwd, float& u) {
  const U4 r = philox4x32_10(U4{(unsigned)gchain, 0u, (unsigned)prop, ((unsigned)(prop >> 32) << 1) | 1u},
                             (unsigned)seed, (unsigned)(seed >> 32));
  fwd = (r.x & 1u) != 0;
  u = (r.y >> 8) * 5.9604644775390625e-08f;
}
template <int DT, int NW>
__device__ __forceinline__ void rng_state(const KArgs& A, long long gchain, unsigned long long prop,
                                          int w, int q, f4 (&z)[DT]) {
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int dim0 = 16 * (w * DT + t) + 4 * q;
    f4 n = philox_normal4(A.rng_seed, gchain, (unsigned)(dim0 >> 2), prop);
    z[t] = f4{dim0 + 0 < A.d ? n.x : 0.f, dim0 + 1 < A.d ? n.y : 0.f, dim0 + 2 < A.d ? n.z : 0.f,
              dim0 + 3 < A.d ? n.w : 0.f};
  }
}

// Sum over the lanes / waves that hold one chain; every lane of the chain gets the total.
template <int NW, int NV>
__device__ __forceinline__ void chain_allreduce(float (&v)[NV], float* red, int w, int lane) {
  // (NV = 1 -- the mixtures' quadratic form, once per component and leapfrog step, with nothing to run beside it -- on the row swaps;
  //  several values at once -- a proposal's epilogue -- through the LDS crossbar, whose requests pipeline: six swaps in a row measured
  //  1.5 % of the bench kernel's cycles more than the twelve ds_bpermute they replaced, profiles/r06_chain_sum.txt)
  if constexpr (NV == 1) {
    v[0] = chain4_sum(v[0]);
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] += __shfl_xor(v[i], 16);
      v[i] += __shfl_xor(v[i], 32);
    }
  }
  if (NW > 1) {
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < NV; ++i) red[(w * 16 + lane) * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s += red[(ww * 16 + (lane & 15)) * NV + i];
      v[i] = s;
    }
    __syncthreads();
  }
}

// y = G dx for a dense symmetric G packed by pack_gauss_kernel (fragments in LDS at Gp).
template <int DT, int NW>
__device__ __forceinline__ void dense_matvec(const float* Gp, const KArgs& A, float* smem, int w,
                                             int lane, const f4 (&dx)[DT], f4 (&y)[DT]) {
  const int NT = A.NT, d = A.d;
  if (NW == 1) {
#pragma unroll
    for (int to = 0; to < DT; ++to) {
      f4 acc = splat(0.f);
      if (16 * to < d) {
#pragma unroll
        for (int ti = 0; ti < DT; ++ti) {
          if (16 * ti < d) {
            const f4 G = lds4(Gp + ((to * NT + ti) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = MFMA16(G[r], dx[ti][r], acc);
          }
        }
      }
      y[to] = acc;
    }
  } else {
    float* XB = smem + A.o_XB;
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int tg = w * DT + t;
      if (tg < NT) *reinterpret_cast<f4*>(XB + c * A.xb_stride + 16 * tg + 4 * q) = dx[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int to = w * DT + t;
      f4 acc = splat(0.f);
      if (16 * to < d) {
        for (int ti = 0; ti < NT; ++ti) {
          const f4 B = lds4(XB + c * A.xb_stride + 16 * ti + 4 * q);
          const f4 G = lds4(Gp + ((to * NT + ti) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = MFMA16(G[r], B[r], acc);
        }
      }
      y[t] = acc;
    }
    __syncthreads();
  }
}

// grad U (S-layout) and this lane's share of U (summing `Upart` over the chain's lanes
// gives U).  dynamics.py:203-218 with the energies of distributions.py.
// Per-lane energy constants kept in registers across the whole launch (diagonal Gaussian: this
// lane's slice of the mean and of the precision diagonal).
template <int EK, int DT>
struct EnergyRegs {
  static constexpr int NR = (EK == L2HMC_ENERGY_GAUSS_DIAG && DT <= 2) ? DT : 1;
  f4 mu[NR], prec[NR];
  bool loaded;
};
template <int EK, int DT, int NW>
__device__ __forceinline__ void load_energy_regs(EnergyRegs<EK, DT>& er, const KArgs& A, const float* smem, int w,
                                                 int lane) {
  er.loaded = false;
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG && DT <= 2) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int off = 16 * (w * DT + t) + 4 * (lane >> 4);
      const bool ok = (w * DT + t) < A.NT;
      er.mu[t] = ok ? lds4(smem + A.o_mu + off) : splat(0.f);
      er.prec[t] = ok ? lds4(smem + A.o_prec + off) : splat(0.f);
    }
    er.loaded = true;
  }
}

template <int EK, int DT, int NW>
__device__ __forceinline__ void grad_energy(const KArgs& A, float* smem, int w, int lane,
                                            const f4 (&x)[DT], f4 (&g)[DT], float& Upart,
                                            bool wantU, const EnergyRegs<EK, DT>* er = nullptr,
                                            const float* beta_p = nullptr) {
  const int q = lane >> 4, DP = 16 * A.NT;
  const float beta_a = beta_p != nullptr ? *beta_p : A.beta;     // (AIS loop: the bridge moves every proposal)
  float U = 0.f;
  if constexpr (EK == L2HMC_ENERGY_GAUSS_DIAG) {
    {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f4 mu, s;
        if constexpr (DT <= 2) {
          if (er != nullptr) { mu = er->mu[t]; s = er->prec[t]; }
          else {
            const bool ok = (w * DT + t) < A.NT;
            mu = ok ? lds4(smem + A.o_mu + 16 * (w * DT + t) + 4 * q) : splat(0.f);
            s = ok ? lds4(smem + A.o_prec + 16 * (w * DT + t) + 4 * q) : splat(0.f);
          }
        } else {
          const bool ok = (w * DT + t) < A.NT;
          mu = ok ? lds4(smem + A.o_mu + 16 * (w * DT + t) + 4 * q) : splat(0.f);
          s = ok ? lds4(smem + A.o_prec + 16 * (w * DT + t) + 4 * q) : splat(0.f);
        }
        const f4 dx = x[t] - mu;
        g[t] = s * dx;
        U += 0.5f * hsum(dx * g[t]);      // (always: a wave-uniform `if (wantU)` here costs more than it saves)
      }
    }
  } else if constexpr (EK == L2HMC_ENERGY_GAUSS_DENSE) {
    {
      f4 dx[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const bool ok = (w * DT + t) < A.NT;
        const f4 mu = ok ? lds4(smem + A.o_mu + 16 * (w * DT + t) + 4 * q) : splat(0.f);
        dx[t] = x[t] - mu;
      }
      dense_matvec<DT, NW>(weights_in_global(DT) ? A.prec : smem + A.o_prec, A, smem, w, lane, dx, g);
#pragma unroll
      for (int t = 0; t < DT; ++t) U += 0.5f * hsum(dx[t] * g[t]);
    }
  } else if constexpr (EK == L2HMC_ENERGY_GMM) {
    {
      // U = -logsumexp_i(-q_i/2 + log c_i); grad = sum_i softmax_i G_i (x - mu_i).
      // Online softmax over components: no per-component storage.
      float m = -INFINITY, ssum = 0.f;
      f4 gacc[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) gacc[t] = splat(0.f);
      for (int i = 0; i < A.ncomp; ++i) {
        f4 dx[DT], y[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const bool ok = (w * DT + t) < A.NT;
          const f4 mu = ok ? lds4(smem + A.o_mu + i * DP + 16 * (w * DT + t) + 4 * q) : splat(0.f);
          dx[t] = x[t] - mu;
        }
        dense_matvec<DT, NW>((weights_in_global(DT) ? A.prec : smem + A.o_prec) + i * gauss_floats(A.NT), A, smem, w, lane, dx, y);
        float qq[1] = {0.f};
#pragma unroll
        for (int t = 0; t < DT; ++t) qq[0] += hsum(dx[t] * y[t]);
        chain_allreduce<NW, 1>(qq, smem + A.o_red, w, lane);
        const float V = -(0.5f * qq[0]) + smem[A.o_logc + i];
        // (a component with V = -inf -- zero weight, or an overflowed quadratic -- contributes nothing; without
        //  the guards m - mn = -inf - -inf = NaN would poison the running sums: reduce_logsumexp semantics)
        const float mn = fmaxf(m, V);
        const float sc = (m == mn) ? 1.f : expf(m - mn), wi = (V == -INFINITY) ? 0.f : expf(V - mn);
        ssum = ssum * sc + wi;
#pragma unroll
        for (int t = 0; t < DT; ++t) gacc[t] = gacc[t] * sc + wi * y[t];
        m = mn;
      }
      const float inv = 1.f / ssum;
#pragma unroll
      for (int t = 0; t < DT; ++t) g[t] = gacc[t] * inv;
      if (w == 0 && lane < 16) U = -(m + logf(ssum));
    }
  } else if constexpr (EK == L2HMC_ENERGY_ROUGHWELL) {
    {
      const float eta = A.eta;
      const float den = A.den;
      const float scale = eta / den;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f4 arg = x[t] / den;
        g[t] = x[t] - scale * rw_sin4(arg);
        if (wantU) {
          // padded dims hold x = 0 and would add eta * cos(0): mask them out
          const int dim0 = 16 * (w * DT + t) + 4 * q;
          f4 cs = rw_cos4(arg);
          f4 live = f4{dim0 < A.d ? 1.f : 0.f, dim0 + 1 < A.d ? 1.f : 0.f,
                       dim0 + 2 < A.d ? 1.f : 0.f, dim0 + 3 < A.d ? 1.f : 0.f};
          U += 0.5f * hsum(x[t] * x[t]) + eta * hsum(live * cs);
        }
      }
    }
  } else if constexpr (EK == L2HMC_ENERGY_FUNNEL) {
    {
      const bool has0 = (w == 0 && lane < 16);  // lane holding dim 0 (tile 0, q 0, r 0)
      float rv[2] = {has0 ? x[0].x : 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < DT; ++t) rv[1] += hsum(x[t] * x[t]);
      if (has0) rv[1] -= x[0].x * x[0].x;
      chain_allreduce<NW, 2>(rv, smem + A.o_red, w, lane);
      const float vv = rv[0], sum_sq = rv[1], sigma = A.eta, clip = 4.f * sigma;
      const float n = (float)(A.d - 1);
      const bool hi = vv > clip, lo = -clip > vv;
      const float s = expf(vv);
      const float s_eff = hi ? expf(clip) : (lo ? expf(-clip) : s);
      const float inv_s = 1.f / s_eff;
#pragma unroll
      for (int t = 0; t < DT; ++t) g[t] = x[t] * inv_s;
      if (has0) {
        const float gv = vv / (sigma * sigma) + ((hi || lo) ? 0.f : 0.5f * (-sum_sq / s + n));
        g[0].x = gv;
        const float lp = (vv / sigma) * (vv / sigma);
        U = 0.5f * (lp + sum_sq / s_eff + n * logf(6.283185307179586f * s_eff));
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < DT; ++t) g[t] = splat(0.f);
  }
  if (beta_a != 1.f) {       // annealed energy between N(0, I) and the target (wave-uniform branch)
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      q += hsum(x[t] * x[t]);
      g[t] = x[t] * (1.f - beta_a) + g[t] * beta_a;
    }
    U = (1.f - beta_a) * 0.5f * q + beta_a * U;
  }
  if (A.temperature != 1.f) {
    U = U / A.temperature;
#pragma unroll
    for (int t = 0; t < DT; ++t) g[t] = g[t] / A.temperature;
  }
  Upart = U;
}

// ---- S/T/Q network, split so that layer-1 partial products can be shared -------------------
// h1_pre = W1^T a + W2^T b + (W3^T tau + biases).  The (a, b) contractions are K-split over the
// NW waves (`l1_part` + `xchg`); the tau/bias k-step, layer 2 and the heads are `net_tail`.
// Sharing (DESIGN.md "layer-1 reuse"): VNet is evaluated at the same (x, grad U) at the end of
// step t and the start of step t+1, and both XNet calls of a step see the same v_h -- those
// partial products are computed and exchanged once.

// Layer-1 partial pre-activation from one input, summed over this wave's dimension tiles.
// grp0 = 0 selects the weights of input `a` (W1), grp0 = NT those of input `b` (W2).
// For DT <= 2 the wave's layer-1 A fragments (one float4 per tile per input per net) are loaded
// ONCE per launch and stay in registers (`L1W`): no LDS round trip in front of the MFMAs.
template <int DT>
struct L1W {
  static constexpr bool RES = DT <= 2;          // register-resident?
  f4 xa[RES ? DT : 1], xb[RES ? DT : 1], va[RES ? DT : 1], vb[RES ? DT : 1];
};
template <int DT, int NW>
__device__ __forceinline__ void load_l1w(L1W<DT>& l, const float* wx, const float* wv, const KArgs& A, int w,
                                         int lane) {
  if constexpr (L1W<DT>::RES) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int tg = w * DT + t;
      const bool ok = 16 * tg < A.d;
      l.xa[t] = ok ? lds4(wx + (tg * 64 + lane) * 4) : splat(0.f);
      l.xb[t] = ok ? lds4(wx + ((A.NT + tg) * 64 + lane) * 4) : splat(0.f);
      l.va[t] = ok ? lds4(wv + (tg * 64 + lane) * 4) : splat(0.f);
      l.vb[t] = ok ? lds4(wv + ((A.NT + tg) * 64 + lane) * 4) : splat(0.f);
    }
  }
}
template <int DT, int NW>
__device__ __forceinline__ f4 l1_part(const float* wn, int grp0, const KArgs& A, int w, int lane,
                                      const f4 (&z)[DT], f4 acc, const f4* Wres = nullptr) {
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int tg = w * DT + t;
    if constexpr (L1W<DT>::RES) {
      const f4 W = Wres[t];           // zero for dead tiles: no guard branch at all
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = MFMA16(W[r], z[t][r], acc);
    } else {
      if (16 * tg < A.d) {
        const f4 W = lds4(wn + ((grp0 + tg) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r)   // dead k-steps (dims >= d) multiply zeros: no guard branch
          acc = MFMA16(W[r], z[t][r], acc);
      }
    }
  }
  return acc;
}

// Sum NP partial vectors over the NW waves of the workgroup: one LDS hop, one barrier
// (double-buffered so the next exchange never overwrites a buffer still being read).
template <int NW, int NP>
__device__ __forceinline__ void xchg(f4 (&p)[NP], const KArgs& A, float* smem, int w, int lane,
                                     int& pb) {
#ifdef L2HMC_ABL_NOXCHG
  return;
#endif
  if (NW > 1) {
    float* P = smem + A.o_P + pb * (NW * NP * 256);
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<f4*>(P + ((w * NP + i) * 64 + lane) * 4) = p[i];
    __syncthreads();
    // all NW * NP reads are issued back to back into distinct registers and summed afterwards
    // (a running sum makes the compiler wait for each read before issuing the next)
    f4 part[NP][NW];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) part[i][ww] = lds4(P + ((ww * NP + i) * 64 + lane) * 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      f4 sum = part[i][0];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) sum += part[i][ww];
      p[i] = sum;
    }
    pb ^= 1;
  }
}

// Weight fragments of the tail (tau/bias k-step, layer 2, heads, ScaleTanh scales).  Loaded
// BEFORE the exchange barrier so their LDS latency hides under it.
// (Head fragments are register-prefetched only for DT <= 2; wider waves read them from LDS
// tile by tile inside net_tail to stay within the register budget.)
template <int DT>
struct TailW {
  static constexpr int NH = DT <= 2 ? DT : 1;
  f4 w2;
  f4 hs[NH], ht[NH], hq[NH], es[NH], eq[NH];
  const float* wn;
  int NT, w, lane;
};

template <int DT, int NW>
__device__ __forceinline__ void load_tail(TailW<DT>& tw, const float* wn, const KArgs& A, int w,
                                          int lane) {
  const int NT = A.NT, q = lane >> 4;
  tw.wn = wn; tw.NT = NT; tw.w = w; tw.lane = lane;
  tw.w2 = lds4(wn + ((2 * NT + 1) * 64 + lane) * 4);
  const float* sc = wn + net_groups(NT) * 256;
  if constexpr (DT > 2) return;
#pragma unroll
  for (int t = 0; t < TailW<DT>::NH; ++t) {
    const int tg = w * DT + t;
    if (tg < NT) {
      tw.hs[t] = lds4(wn + ((2 * NT + 2 + 3 * tg + 0) * 64 + lane) * 4);
      tw.ht[t] = lds4(wn + ((2 * NT + 2 + 3 * tg + 1) * 64 + lane) * 4);
      tw.hq[t] = lds4(wn + ((2 * NT + 2 + 3 * tg + 2) * 64 + lane) * 4);
      tw.es[t] = lds4(sc + 16 * tg + 4 * q);
      tw.eq[t] = lds4(sc + 16 * NT + 16 * tg + 4 * q);
    } else {
      tw.hs[t] = tw.ht[t] = tw.hq[t] = tw.es[t] = tw.eq[t] = splat(0.f);
    }
  }
}

// h1 = relu(hpre + tau/bias term); h2 = relu(W4^T h1); heads.  The heads' nonlinearities are
// emitted in "folded" form: with kS = (+-)eps log2(e) (eps/2 for VNet) and kQ = eps log2(e),
//   aS = kS e^{lam_s} tanh(z_s)   (= log2 of the scale factor; sums to the log-det / ln 2)
//   ES = 2^{aS} = exp(+-eps S),   EQ = 2^{kQ e^{lam_q} tanh(z_q)} = exp(eps Q),   T = z_t
// `apply(t, ES, aS, T, EQ)` consumes one 16-dimension tile at a time.
template <int DT, int KH, class F>
__device__ __forceinline__ void net_tail(const TailW<DT>& tw, f4 hpre, f4 tb, float kS,
                                         float kQ, F&& apply) {
  // tb = W3^T tau + (b1 + b2 + b3) for this chain's schedule row: a per-(net, row) table built
  // once per launch (it is the same for every chain), so no tau k-step MFMA on the critical path
  f4 h = relu4(hpre + tb);
  {
    f4 acc = splat(0.f);
#pragma unroll
    for (int r = 0; r < KH; ++r) acc = MFMA16(tw.w2[r], h[r], acc);
    h = relu4(acc);
  }
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    f4 Ws, Wt, Wq, es, eq;
    if constexpr (DT <= 2) {
      Ws = tw.hs[t]; Wt = tw.ht[t]; Wq = tw.hq[t]; es = tw.es[t]; eq = tw.eq[t];
    } else {
      const int tg = tw.w * DT + t, NT = tw.NT, lane = tw.lane;
      const float* sc = tw.wn + net_groups(NT) * 256;
      const bool ok = tg < NT;
      Ws = ok ? lds4(tw.wn + ((2 * NT + 2 + 3 * tg + 0) * 64 + lane) * 4) : splat(0.f);
      Wt = ok ? lds4(tw.wn + ((2 * NT + 2 + 3 * tg + 1) * 64 + lane) * 4) : splat(0.f);
      Wq = ok ? lds4(tw.wn + ((2 * NT + 2 + 3 * tg + 2) * 64 + lane) * 4) : splat(0.f);
      es = ok ? lds4(sc + 16 * tg + 4 * (lane >> 4)) : splat(0.f);
      eq = ok ? lds4(sc + 16 * NT + 16 * tg + 4 * (lane >> 4)) : splat(0.f);
    }
    f4 zs = splat(0.f), zt = splat(0.f), zq = splat(0.f);
#ifdef L2HMC_ABL_NOHEADS
    zs = h * Ws; zq = h * Wq; zt = h * Wt;
#else
#pragma unroll
    for (int r = 0; r < KH; ++r) {
      zs = MFMA16(Ws[r], h[r], zs);
      zq = MFMA16(Wq[r], h[r], zq);
      zt = MFMA16(Wt[r], h[r], zt);
    }
#endif
#ifdef L2HMC_IGLP
    __builtin_amdgcn_iglp_opt(L2HMC_IGLP);
#endif
    const f4 aS = ctanh4(es * kS, zs);
    apply(t, exp2_4(aS), aS, zt, exp2_4(ctanh4(eq * kQ, zq)));
  }
}

// One momentum half-update.  forward (dynamics.py:121-125,149-153):
//   v' = v e^{eps S / 2} + (eps/2)(T - e^{eps Q} grad);   backward (:164-170,194-199):
//   v' = (v - (eps/2)(T - e^{eps Q} grad)) e^{-eps S / 2}.
// ES = e^{+-eps S / 2}, EQ = e^{eps Q}, aS = log2(ES); `ld2` accumulates log2|det| as a 4-wide partial
// sum (one packed add per update; summed across its components once per trajectory).
__device__ __forceinline__ f4 v_half(f4 vin, f4 g, f4 ES, f4 aS, f4 T, f4 EQ, float heps, bool fwd,
                                     f4& ld2) {
  const f4 cc = heps * (T - EQ * g);
  ld2 += aS;
  return sel4(fwd, vin * ES + cc, (vin - cc) * ES);
}

// One masked position update; `kp` = kept coordinates (0/1).  forward (dynamics.py:131-145):
//   z' = kp z + (1-kp)(z e^{eps S} + eps (e^{eps Q} v_h + T));   backward (:176-190):
//   z' = kp z + (1-kp) e^{-eps S} (z - eps (e^{eps Q} v_h + T)).
__device__ __forceinline__ f4 x_half(f4 zin, f4 kp, f4 vh, f4 ES, f4 aS, f4 T, f4 EQ, float eps,
                                     bool fwd, f4& ld2) {
  const f4 up = splat(1.f) - kp;
  const f4 tr = eps * (EQ * vh + T);
  const f4 nw = sel4(fwd, zin * ES + tr, ES * (zin - tr));
  ld2 += up * aS;
  return kp * zin + up * nw;
}

template <int DT, int NW>
__device__ __forceinline__ void load_state(const float* p, const KArgs& A, long long chain,
                                           bool live, int w, int q, f4 (&z)[DT]) {
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int dim0 = 16 * (w * DT + t) + 4 * q;
    f4 r = splat(0.f);
    if (live && p != nullptr) {
      // rows are 16-byte aligned when d % 4 == 0 (one dwordx4 per lane and tile), 8-byte aligned when d is
      // even (two dwordx2); dim0 is a multiple of 4, so "dim0 < d" then covers the whole vector
      const float* row = p + chain * A.d + dim0;
      if ((A.d & 3) == 0) {
        if (dim0 < A.d) r = *reinterpret_cast<const f4*>(row);
      } else if ((A.d & 1) == 0) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        if (dim0 < A.d) { const f2 a = *reinterpret_cast<const f2*>(row); r.x = a.x; r.y = a.y; }
        if (dim0 + 2 < A.d) { const f2 b = *reinterpret_cast<const f2*>(row + 2); r.z = b.x; r.w = b.y; }
      } else {
        if (dim0 + 0 < A.d) r.x = row[0];
        if (dim0 + 1 < A.d) r.y = row[1];
        if (dim0 + 2 < A.d) r.z = row[2];
        if (dim0 + 3 < A.d) r.w = row[3];
      }
    }
    z[t] = r;
  }
}

template <int DT, int NW>
__device__ __forceinline__ void store_state(float* p, const KArgs& A, long long chain, bool live,
                                            int w, int q, const f4 (&z)[DT]) {
  if (p == nullptr || !live) return;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int dim0 = 16 * (w * DT + t) + 4 * q;
    float* row = p + chain * A.d + dim0;
    if ((A.d & 3) == 0) {
      if (dim0 < A.d) *reinterpret_cast<f4*>(row) = z[t];
    } else if ((A.d & 1) == 0) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      if (dim0 < A.d) *reinterpret_cast<f2*>(row) = f2{z[t].x, z[t].y};
      if (dim0 + 2 < A.d) *reinterpret_cast<f2*>(row + 2) = f2{z[t].z, z[t].w};
    } else {
      if (dim0 + 0 < A.d) row[0] = z[t].x;
      if (dim0 + 1 < A.d) row[1] = z[t].y;
      if (dim0 + 2 < A.d) row[2] = z[t].z;
      if (dim0 + 3 < A.d) row[3] = z[t].w;
    }
  }
}

// Stage energy parameters into LDS (padded with zeros to 16*NT dims).
template <int EK, bool WG = false>
__device__ __forceinline__ void stage_energy(const KArgs& A, float* smem, int tid, int nthr) {
  const int DP = 16 * A.NT;
  const int nc = EK == L2HMC_ENERGY_GMM ? A.ncomp : 1;
  if (EK == L2HMC_ENERGY_GAUSS_DIAG || EK == L2HMC_ENERGY_GAUSS_DENSE ||
      EK == L2HMC_ENERGY_GMM) {
    for (int i = tid; i < nc * DP; i += nthr) {
      const int comp = i / DP, dim = i % DP;
      smem[A.o_mu + i] = dim < A.d ? A.mu[comp * A.d + dim] : 0.f;
    }
  }
  if (EK == L2HMC_ENERGY_GAUSS_DIAG) {
    for (int i = tid; i < DP; i += nthr) smem[A.o_prec + i] = i < A.d ? A.prec[i] : 0.f;
  } else if (EK == L2HMC_ENERGY_GAUSS_DENSE || EK == L2HMC_ENERGY_GMM) {
    if (!WG) {
      const int n4 = nc * gauss_floats(A.NT) / 4;
      const f4* src = reinterpret_cast<const f4*>(A.prec);
      f4* dst = reinterpret_cast<f4*>(smem + A.o_prec);
      for (int i = tid; i < n4; i += nthr) dst[i] = src[i];
    }
    if (EK == L2HMC_ENERGY_GMM)
      for (int i = tid; i < nc; i += nthr) smem[A.o_logc + i] = A.logc[i];
  }
}

// ------------------------------------------------------------------------------------------
// The fused trajectory kernel
// ------------------------------------------------------------------------------------------
template <int EK, int DT, int NW, int KH>
// (waves-per-SIMD hint 2 for DT <= 2 caps the kernel at 256 VGPRs, which makes the compiler keep
// MFMA accumulators in VGPRs -- no v_accvgpr_read traffic; wide-DT kernels keep all 512.)
#ifndef L2HMC_WAVES_PER_SIMD
#define L2HMC_WAVES_PER_SIMD 2
#endif
__global__ __launch_bounds__(64 * NW, (DT <= 2 ? L2HMC_WAVES_PER_SIMD : 1)) void traj_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  const int tid = threadIdx.x, lane = tid & 63, nthr = 64 * NW;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  const int NT = A.NT, DP = 16 * NT;
  const bool has_nets = A.packed != nullptr;
  const int NF = net_floats(NT);

  // ---- prologue: stage weights / masks / time table / energy parameters into LDS ----------
  constexpr bool WG = weights_in_global(DT);
  // Layer-1 groups (the first 2 NT groups of each net) go straight from global memory into
  // registers when L1W is resident, so only the rest of each net is staged in LDS.
  const int skip = (L1W<DT>::RES && !WG) ? 2 * NT * 256 : 0;      // floats not staged per net
  if (has_nets && !WG) {
    const int per = (NF - skip) / 4;
    f4* dst = reinterpret_cast<f4*>(smem);
    for (int i = tid; i < 2 * per; i += nthr) {
      const int net = i >= per, j = i - net * per;
      dst[i] = reinterpret_cast<const f4*>(A.packed + (size_t)net * NF + skip)[j];
    }
  }
  for (int i = tid; i < A.T * DP; i += nthr) {
    const int row = i / DP, dim = i % DP;
    smem[A.o_mask + i] = dim < A.d ? A.masks[row * A.d + dim] : 0.f;
  }
  for (int i = tid; i < 2 * A.T; i += nthr) smem[A.o_trig + i] = A.trig[i];
  stage_energy<EK, weights_in_global(DT)>(A, smem, tid, nthr);

  f4 x[DT], v[DT], g[DT];
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr || A.u != nullptr ||
                      (A.rng_flags & L2HMC_RNG_U) != 0;
  __syncthreads();

  // bases such that `base + group * 256` addresses group `group` (the skipped layer-1 groups lie
  // before the staged region and are never dereferenced through these)
  const float* wx = WG ? A.packed : smem - skip;                    // XNet fragments
  const float* wv = WG ? A.packed + NF : smem + (NF - skip) - skip;  // VNet fragments
  if (has_nets) {
    // time-embedding table TB[net][row s][unit row i] = W3[0,u] cos_s + W3[1,u] sin_s + b1+b2+b3
    // from the packed tau fragment (lane (i, q): q = 0 -> W3[0], 1 -> W3[1], 2 -> biases)
    for (int idx = tid; idx < 2 * A.T * 16; idx += nthr) {
      const int net = idx / (A.T * 16), srow = (idx / 16) % A.T, i = idx & 15;
      const float* tf = (net == 0 ? wx : wv) + (2 * NT * 64) * 4;
      const float ct = smem[A.o_trig + 2 * srow], st = smem[A.o_trig + 2 * srow + 1];
      smem[A.o_tb + idx] = fmaf(tf[i * 4], ct, fmaf(tf[(16 + i) * 4], st, tf[(32 + i) * 4]));
    }
    __syncthreads();
  }
  int pb = 0;
  const f4 Z = splat(0.f);
  float U_start;                 // this lane's share of U at the current state
  EnergyRegs<EK, DT> er;
  load_energy_regs<EK, DT, NW>(er, A, smem, w, lane);
  grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, U_start, need_p, &er);

  // VNet layer-1 partial at the current (x, grad U): shared by the closing half-update of one
  // step and the opening half-update of the next, and kept across proposals.
  TailW<DT> tw;
  L1W<DT> l1w;
  if (has_nets) load_l1w<DT, NW>(l1w, A.packed, A.packed + NF, A, w, lane);
  f4 pv[1] = {Z};
  PT_DECL;
  PT_MARK(0);      // prologue (staging + first grad)
  if (has_nets && A.n_steps > 0) {
    load_tail<DT, NW>(tw, wv, A, w, lane);
    // (two independent accumulators: the MFMA chain is pipe-bound, not latency-bound)
    pv[0] = l1_part<DT, NW>(wv, 0, A, w, lane, x, Z, l1w.va) + l1_part<DT, NW>(wv, NT, A, w, lane, g, Z, l1w.vb);
    xchg<NW, 1>(pv, A, smem, w, lane, pb);
  }

  // ---- persistent sampler loop: M proposals per launch (M = 1: a single trajectory) ---------
  // This proposal's draws (momenta, direction bit, accept uniform): either injected from HBM --
  // then fetched one proposal ahead so the latency hides under the current trajectory -- or
  // drawn in-kernel from the counter-based Philox stream.
  const long long gchain = A.chain_off + chain;
  const bool rng_v = (A.rng_flags & L2HMC_RNG_V) != 0, rng_d = (A.rng_flags & L2HMC_RNG_DIR) != 0;
  const bool rng_u = (A.rng_flags & L2HMC_RNG_U) != 0;
  f4 vn[DT];
  if (!rng_v) load_state<DT, NW>(A.v, A, chain, live, w, q, vn);
  bool fwd_n = (A.dir != nullptr && !rng_d) ? (live ? A.dir[chain] != 0 : true) : (A.dir_all != 0);
  float u_n = (A.u != nullptr && !rng_u && live) ? A.u[chain] : 0.f;
  const bool have_u = A.u != nullptr || rng_u;
  // AIS mode (utils/ais.py:43-66, HMC transitions): per proposal the bridge moves to beta = ais_beta[m], the
  // log-weight takes dbeta (|x|^2/2 - U_final(x)) at the CURRENT state, the momentum is drawn fresh or partially
  // refreshed, and a rejected chain keeps its state with the NEGATED PROPOSED momentum (ais.py:63).
  const bool ais = A.ais_beta != nullptr;
  float ais_wacc = 0.f, ais_aacc = 0.f, beta_m = A.beta;
  f4 vprev[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) vprev[t] = Z;
  if (ais && A.ais_refresh >= 0.f) {
    if (A.ais_v0 != nullptr) load_state<DT, NW>(A.ais_v0, A, chain, live, w, q, vprev);
    else rng_state<DT, NW>(A, gchain, A.rng_prop0 - 1, w, q, vprev);
  }
  for (int m = 0; m < A.M; ++m) {
  const long long moff = (long long)m * A.N;
  const unsigned long long prop = A.rng_prop0 + (unsigned long long)m;
  if (rng_v) {
    rng_state<DT, NW>(A, gchain, prop, w, q, v);
  } else {
#pragma unroll
    for (int t = 0; t < DT; ++t) v[t] = vn[t];
  }
  if (ais) {
    beta_m = A.ais_beta[m];
    const float one = 1.f;
    float pr[2];
    grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, pr[0], true, &er, &one);      // U_final, grad U_final at x
    pr[1] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) pr[1] += 0.5f * hsum(x[t] * x[t]);
    U_start = (1.f - beta_m) * pr[1] + beta_m * pr[0];                              // this lane's share of U_beta(x)
#pragma unroll
    for (int t = 0; t < DT; ++t) g[t] = x[t] * (1.f - beta_m) + g[t] * beta_m;
    chain_allreduce<NW, 2>(pr, smem + A.o_red, w, lane);
    ais_wacc += A.ais_dbeta * (-pr[0] + pr[1]);                                     // ais.py:58-59
    if (A.ais_refresh >= 0.f) {                                                     // ais.py:55
      const float keep = sqrtf(1.f - A.ais_refresh), mix = sqrtf(A.ais_refresh);
#pragma unroll
      for (int t = 0; t < DT; ++t) v[t] = vprev[t] * keep + v[t] * mix;
    }
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
  const float sgn = fwd ? 1.f : -1.f;
  // the start point: a rejected chain resumes from it (sampler.py:53-55)
  f4 x0[DT], g0[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t) { x0[t] = x[t]; g0[t] = g[t]; }
  const f4 pv0 = pv[0];
  float red[5];                  // U0, K0, U1, K1, logdet (per-lane partial sums)
  red[0] = U_start;
  red[1] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[1] += 0.5f * hsum(v[t] * v[t]);
  red[2] = 0.f;
  f4 ldv = splat(0.f);

  // folded constants: sgn eps log2(e) scales S of XNet, sgn (eps/2) log2(e) S of VNet, eps log2(e) Q
  const float LOG2E = 1.4426950408889634f;
  const float kSx = sgn * eps * LOG2E, kSv = sgn * heps * LOG2E, kQ = eps * LOG2E;
  const f4 O = splat(1.f);

  // schedule row of this chain at iteration `it`: forward chains walk 0..T-1, backward T-1..0
  auto row_of = [&](int it) { const int sf = A.step_begin + it; return fwd ? sf : (A.T - 1 - sf); };
  // time-embedding terms (XNet, VNet) and the first-kept mask of that row; all are PREFETCHED
  // one step ahead so their LDS latency never sits on the critical path
  auto tbx_of = [&](int s) { return lds4(smem + A.o_tb + s * 16 + 4 * q); };
  auto tbv_of = [&](int s) { return lds4(smem + A.o_tb + (A.T + s) * 16 + 4 * q); };
  auto mask_of = [&](int s, f4 (&k)[DT]) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const bool ok = (w * DT + t) < NT;
      const f4 m = ok ? lds4(smem + A.o_mask + s * DP + 16 * (w * DT + t) + 4 * q) : Z;
      k[t] = sel4(fwd, m, O - m);             // forward keeps m first, backward keeps 1-m first
    }
  };
  f4 k1[DT], k1n[DT];
  f4 tbx = Z, tbv = Z, tbxn = Z, tbvn = Z;
  if (A.n_steps > 0) {
    if (has_nets) { tbx = tbx_of(row_of(0)); tbv = tbv_of(row_of(0)); }
    mask_of(row_of(0), k1);
  }

  for (int it = 0; it < A.n_steps; ++it) {
    f4 xin[DT], y[DT], vh[DT];
    if (it + 1 < A.n_steps) {                 // prefetch the next step's schedule row
      if (has_nets) { tbxn = tbx_of(row_of(it + 1)); tbvn = tbv_of(row_of(it + 1)); }
      mask_of(row_of(it + 1), k1n);
    }

    if (has_nets) {
      PT_MARK(1);  // step head
      // ---- momentum half-update #1: VNet([x, grad U(x), t])  (dynamics.py:118-125 / :162-170)
      net_tail<DT, KH>(tw, pv[0], tbv, kSv, kQ, [&](int t, f4 ES, f4 aS, f4 T, f4 EQ) {
        vh[t] = v_half(v[t], g[t], ES, aS, T, EQ, heps, fwd, ldv);
      });
      PT_MARK(2);  // VNet tail #1

      // ---- two masked position updates: XNet([v_h, kept * x, t])  (:127-145 / :172-190);
      //      the v_h contraction is shared by both
      load_tail<DT, NW>(tw, wx, A, w, lane);
#pragma unroll
      for (int t = 0; t < DT; ++t) xin[t] = k1[t] * x[t];
      // the v_h contraction `pa` is computed once and enters both exchanges un-summed, so every
      // exchange carries ONE partial vector per wave
      const f4 pa = l1_part<DT, NW>(wx, 0, A, w, lane, vh, Z, l1w.xa);
      f4 px[1];
      px[0] = pa + l1_part<DT, NW>(wx, NT, A, w, lane, xin, Z, l1w.xb);
      PT_MARK(3);  // XNet layer-1 partials (a, b)
      xchg<NW, 1>(px, A, smem, w, lane, pb);
      PT_MARK(4);  // exchange
      net_tail<DT, KH>(tw, px[0], tbx, kSx, kQ, [&](int t, f4 ES, f4 aS, f4 T, f4 EQ) {
        y[t] = x_half(x[t], k1[t], vh[t], ES, aS, T, EQ, eps, fwd, ldv);
      });
      PT_MARK(5);  // XNet tail #1
#pragma unroll
      for (int t = 0; t < DT; ++t) xin[t] = (O - k1[t]) * y[t];
      f4 py[1];
      py[0] = pa + l1_part<DT, NW>(wx, NT, A, w, lane, xin, Z, l1w.xb);
      PT_MARK(6);  // XNet layer-1 partial (b only)
      xchg<NW, 1>(py, A, smem, w, lane, pb);
      PT_MARK(7);  // exchange
      net_tail<DT, KH>(tw, py[0], tbx, kSx, kQ, [&](int t, f4 ES, f4 aS, f4 T, f4 EQ) {
        x[t] = x_half(y[t], O - k1[t], vh[t], ES, aS, T, EQ, eps, fwd, ldv);
      });
      PT_MARK(8);  // XNet tail #2

      // ---- momentum half-update #2 at the new position  (:147-153 / :192-199); its layer-1
      //      partial is reused by half-update #1 of the next step
      load_tail<DT, NW>(tw, wv, A, w, lane);
      grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, red[2], need_p && it == A.n_steps - 1, &er);
      pv[0] = l1_part<DT, NW>(wv, 0, A, w, lane, x, Z, l1w.va) + l1_part<DT, NW>(wv, NT, A, w, lane, g, Z, l1w.vb);
      PT_MARK(9);  // grad U + VNet layer-1 partials
      xchg<NW, 1>(pv, A, smem, w, lane, pb);
      PT_MARK(10); // exchange
      net_tail<DT, KH>(tw, pv[0], tbv, kSv, kQ, [&](int t, f4 ES, f4 aS, f4 T, f4 EQ) {
        v[t] = v_half(vh[t], g[t], ES, aS, T, EQ, heps, fwd, ldv);
      });
      PT_MARK(11); // VNet tail #2
    } else {
      // HMC mode: S = T = Q = 0 (dynamics.py:73-76)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        vh[t] = v_half(v[t], g[t], O, Z, Z, O, heps, fwd, ldv);
        y[t] = x_half(x[t], k1[t], vh[t], O, Z, Z, O, eps, fwd, ldv);
        x[t] = x_half(y[t], O - k1[t], vh[t], O, Z, Z, O, eps, fwd, ldv);
      }
      grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, red[2], need_p && it == A.n_steps - 1, &er, &beta_m);
#pragma unroll
      for (int t = 0; t < DT; ++t) v[t] = v_half(vh[t], g[t], O, Z, Z, O, heps, fwd, ldv);
    }
    tbx = tbxn;
    tbv = tbvn;
#pragma unroll
    for (int t = 0; t < DT; ++t) k1[t] = k1n[t];
  }
  const float ld = hsum(ldv) * 0.6931471805599453f;   // the log-det was accumulated in log2 units

  // ---- per-proposal epilogue: proposal, log-det, accept probability, MH select ---------------
  const bool last = m == A.M - 1;
  if (last) {
    store_state<DT, NW>(A.x_out, A, chain, live, w, q, x);
    store_state<DT, NW>(A.v_out, A, chain, live, w, q, v);
  }
  if (A.n_steps == 0) red[2] = red[0];
  red[3] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[3] += 0.5f * hsum(v[t] * v[t]);
  red[4] = ld;
  const float U_end = red[2];
  chain_allreduce<NW, 5>(red, smem + A.o_red, w, lane);
  const bool writer = live && w == 0 && lane < 16;
  if (A.logjac_out != nullptr && writer) A.logjac_out[moff + chain] = red[4];
  if (need_p) {
    // dynamics.py:302-309
    const float e_new = red[2] + red[3], e_old = red[0] + red[1];
    const float val = e_old - e_new + red[4];
    const float p = accept_prob(val);
    if (A.p_out != nullptr && writer) A.p_out[moff + chain] = p;
    if (have_u) {
      const bool acc = live && (p - u_m) >= 0.f;                      // sampler.py:53-55
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        x[t] = sel4(acc, x[t], x0[t]);
        g[t] = sel4(acc, g[t], g0[t]);
      }
      pv[0] = sel4(acc, pv[0], pv0);
      U_start = acc ? U_end : U_start;
      if (ais) {
        ais_aacc += p;
#pragma unroll
        for (int t = 0; t < DT; ++t) vprev[t] = acc ? v[t] : -v[t];
      }
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
  if (ais && live && w == 0 && lane < 16) {
    if (A.ais_w != nullptr) A.ais_w[chain] += ais_wacc;
    if (A.ais_alpha != nullptr) A.ais_alpha[chain] += ais_aacc;
  }
}

// energy / grad only  (Dynamics.energy, Dynamics.grad_energy)
template <int EK, int DT, int NW>
__global__ __launch_bounds__(64 * NW) void energy_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  stage_energy<EK, weights_in_global(DT)>(A, smem, tid, 64 * NW);
  f4 x[DT], g[DT];
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  __syncthreads();
  float U[1];
  grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, U[0], true);
  store_state<DT, NW>(A.grad_out, A, chain, live, w, q, g);
  chain_allreduce<NW, 1>(U, smem + A.o_red, w, lane);
  if (A.U_out != nullptr && live && w == 0 && lane < 16) A.U_out[chain] = U[0];
}

// p_accept on arbitrary end points  (Dynamics.p_accept)
template <int EK, int DT, int NW>
__global__ __launch_bounds__(64 * NW) void paccept_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  stage_energy<EK, weights_in_global(DT)>(A, smem, tid, 64 * NW);
  f4 x[DT], v[DT], g[DT];
  float red[4];
  __syncthreads();
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  load_state<DT, NW>(A.v, A, chain, live, w, q, v);
  grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, red[0], true);
  red[1] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[1] += 0.5f * hsum(v[t] * v[t]);
  load_state<DT, NW>(A.x1, A, chain, live, w, q, x);
  load_state<DT, NW>(A.v1, A, chain, live, w, q, v);
  grad_energy<EK, DT, NW>(A, smem, w, lane, x, g, red[2], true);
  red[3] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[3] += 0.5f * hsum(v[t] * v[t]);
  chain_allreduce<NW, 4>(red, smem + A.o_red, w, lane);
  if (live && w == 0 && lane < 16) {
    const float val = (red[0] + red[1]) - (red[2] + red[3]) + A.logjac_in[chain];
    A.p_out[chain] = accept_prob(val);
  }
}


// ------------------------------------------------------------------------------------------
// Launchers (one explicit instantiation per energy kind, in traj_ek<k>.hip)
// ------------------------------------------------------------------------------------------
const int kMaxLdsBytes = 160 * 1024;

template <class K>
int launch(K kern, const KArgs& k, int NW, long long lds_bytes, hipStream_t s) {
  if (lds_bytes > kMaxLdsBytes)
    return fail(L2HMC_ERR_UNSUPPORTED, "needs %s%lld bytes of LDS (> 160 KiB): d too large for the LDS-resident weight path", "", lds_bytes);
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const long long blocks = (k.N + 15) / 16;
  if (blocks > 0x7fffffffLL) return fail(L2HMC_ERR_UNSUPPORTED, "too many chains%s");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), (size_t)lds_bytes, s, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

#define L2HMC_GEOM_SWITCH(DTv, NWv, CALL)                \
  if (DTv == 1 && NWv == 1) { CALL(1, 1) }               \
  else if (DTv == 2 && NWv == 1) { CALL(2, 1) }          \
  else if (DTv == 4 && NWv == 1) { CALL(4, 1) }          \
  else if (DTv == 1 && NWv == 4) { CALL(1, 4) }          \
  else if (DTv == 2 && NWv == 4) { CALL(2, 4) }          \
  else if (DTv == 4 && NWv == 4) { CALL(4, 4) }          \
  else if (DTv == 8 && NWv == 4) { CALL(8, 4) }          \
  else return fail(L2HMC_ERR_UNSUPPORTED, "no kernel for this geometry%s");

enum { OP_TRAJ = 0, OP_ENERGY = 1, OP_PACCEPT = 2, OP_TRAJ_FAST = 3, OP_TRAJ_SMALL = 4, OP_TRAJ_SMALL16 = 5 };

#define L2HMC_FAST_SWITCH(DTv, NWv, CALL)                \
  if (DTv == 1 && NWv == 1) { CALL(1, 1) }               \
  else if (DTv == 2 && NWv == 1) { CALL(2, 1) }          \
  else if (DTv == 1 && NWv == 4) { CALL(1, 4) }          \
  else if (DTv == 2 && NWv == 4) { CALL(2, 4) }          \
  else if (DTv == 2 && NWv == 2) { CALL(2, 2) }          \
  else return fail(L2HMC_ERR_UNSUPPORTED, "no fast kernel for this geometry%s");

// Declared here, defined (explicitly instantiated) once per energy kind.
template <int EK>
int launch_ek(int op, const KArgs& k, int DT, int NW, int KH, long long lds, hipStream_t s);
// traj_fast_kernel<EK, DT, NW, KH, 1>: the f16x2 form of the instruction-lean kernel (traj_f16_ek1.hip, traj_f16_ek4.hip)
template <int EK>
int launch_fast16_ek(const KArgs& k, int DT, int NW, int KH, long long lds, hipStream_t s);
// traj_tile_kernel (one wave per tile, 4 tiles per workgroup): elementwise targets only (traj_ek1.hip, traj_ek4.hip)
template <int EK>
int launch_tile_ek(const KArgs& k, int DT, int KH, int tpw, long long lds, hipStream_t s);

#define L2HMC_DEFINE_LAUNCH_EK(EKv)                                                              \
  template <>                                                                                    \
  int launch_ek<EKv>(int op, const KArgs& k, int DT, int NW, int KH, long long lds, hipStream_t s) { \
    if (op == OP_TRAJ) {                                                                         \
      _Pragma("clang diagnostic push")                                                           \
      L2HMC_GEOM_SWITCH(DT, NW, L2HMC_CALL_TRAJ_##EKv)                                           \
      _Pragma("clang diagnostic pop")                                                            \
    } else if (op == OP_TRAJ_FAST) {                                                             \
      L2HMC_FAST_SWITCH(DT, NW, L2HMC_CALL_FAST_##EKv)                                           \
    } else if (op == OP_TRAJ_SMALL) {                                                            \
      L2HMC_CALL_SMALL_##EKv                                                                     \
    } else if (op == OP_TRAJ_SMALL16) {                                                          \
      L2HMC_CALL_SMALL16_##EKv                                                                   \
    } else if (op == OP_ENERGY) {                                                                \
      L2HMC_GEOM_SWITCH(DT, NW, L2HMC_CALL_EN_##EKv)                                             \
    } else {                                                                                     \
      L2HMC_GEOM_SWITCH(DT, NW, L2HMC_CALL_PA_##EKv)                                             \
    }                                                                                            \
  }

}  // namespace l2hmc
