// gemm_xl.hpp -- the decoder-sized products of the split engine (config 5) on PRE-SPLIT bf16 planes: 256 x 128 workgroup tiles,
// eight waves, a main loop of loads, LDS traffic and bf16 MFMAs only (gfx950 / CDNA4).
//
// gemm_nt_kernel<.., BF3 = 1> (gemm_f32.hpp) splits every fp32 operand fragment into its three bf16 terms in the consuming wave:
// 3 VALU instructions per MFMA, an activation element split once per column tile that reads it, a weight element once per row
// tile.  Measured (profiles/r04_gemm_xl.txt; the experiments are commit 97db3a4): that loop is VALU-ISSUE bound -- with one wave
// per SIMD on 128 x 64 blocks (2.4 VALU per MFMA) a k-tile of 32 costs 5100-6000 cycles against 3072 of matrix pipe, because a
// wave's VALU instruction takes 6.5 issue cycles beside an MFMA and only two hide under it; with two waves per SIMD on 64 x 64
// blocks just as much.  Here both operands ARRIVE as three bf16 planes h | m | l (GemmArgs Ap / Bp): weights are converted once
// per parameter update (to_planes), activations are written as planes by the epilogue of the product that makes them (GemmArgs
// Cp), and a fragment is three ds_read_b128.  Same six products per 16 x 16 x 32 block in the same order as the BF3 = 1 form:
// results are bit-identical to it (tools/ubench_gemm_bf3.hip checks all 8.4 M elements).
//
//   * 256 x 128 tile, 4 x 2 waves of 64 x 64 blocks: two waves per SIMD, so one wave's LDS / global instructions issue while the
//     other's MFMAs run (a lone wave pays them in full: 22 cycles per ds_read_b128, 42 per ds_write_b128 of matrix-pipe idle time;
//     4 waves of 128 x 64 blocks: 5430 cycles per k-tile, 8 waves: 4650);
//   * LDS: [buffer][plane][row: TM activation rows, TN weight rows][4 chunks of 16 bytes = 8 bf16], UNPADDED (2 x 72 KB; padded
//     rows would not fit twice) with the chunk index XOR-swizzled by bits 2-3 of the row: the 16 lanes (row c, chunk q) of a
//     fragment read hit 16 distinct 16-byte bank groups, and so do the staging writes;
//   * staging: chunk (plane, row, quarter of the k-tile) = one dwordx4; the registers hold tile kt + 1 while tile kt is
//     multiplied; the first stages of a k-tile store them to the other buffer and re-issue their loads for tile kt + 2 (spread
//     over the stages: in one burst the waves of the CU queue behind one another at the texture addresser and the LDS pipe);
//   * the k-tile is six fenced STAGES (the compiler keeps the memory instructions between the MFMAs instead of collecting them)
//     with ONE workgroup barrier, in the MIDDLE: stages 0-2 store tile kt + 1 and request the rest of tile kt's fragments, after
//     stage 3 every wave has finished reading tile kt and storing tile kt + 1 -- barrier (LDS counter only: the global loads in
//     flight are not waited for) -- and stages 4-5 already request the first three fragments of tile kt + 1 while they multiply.
//     The tile boundary has no barrier and no LDS latency in front of its first MFMA: 3990 cycles per k-tile against 4650 with
//     the barrier at the end (the matrix pipe alone: 3280).  Hazards: tile t + 2 is stored into tile t's buffer only behind
//     the barrier of k-tile t, which every wave reaches with its reads of tile t complete; tile t + 1 is read only behind the
//     barrier that follows its stores.
//   * (round 5, measured and not kept -- profiles/r05_gemm_dma.txt, the code is one commit in the history: staging by gfx950's 16-byte
//     LDS DMA, `global_load_lds_dwordx4`, the swizzle moved into the global address.  Bit-identical, 36 registers and every
//     ds_write_b128 gone, and 4140 cycles per k-tile against 3990: nine wave-level DMAs cost more issue time than nine loads + nine
//     stores.  Config 5: a tie at 8192 chains, 2.6 % slower at 4096.  Without ANY staging the k-tile takes 3412 cycles.)
// 8192 x 1024 x 1024 standalone: 91 us against 121-125 us for the in-loop split in the same cold-clock run (x1.33-1.37; the chip
// clocks DOWN as the matrix pipe fills -- 1.74 GHz against 1.90 -- so cycles improve more than microseconds); inside config 5's
// chain of launches (where every epilogue also writes 6 bytes of planes per element): 3.79 -> 3.57 ms per proposal in an A/B on one
// box with the barrier at the end of the k-tile, 3.54-3.55 with it in the middle (another box).
#pragma once
#include <atomic>
#include <type_traits>

#include "gemm_f32.hpp"

namespace l2hmc {

template <int I, int N, class F>
__device__ __forceinline__ void xl_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    xl_static_for<I + 1, N>(f);
  }
}

#ifdef L2HMC_XL_TIMING      // tools/ubench_gemm_bf3.hip: shader cycles of the k loop of wave 0, summed over the workgroups
__device__ unsigned long long xl_ticks[4];
#endif
#ifndef L2HMC_XL_NO_FENCE
#define XL_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define XL_FENCE()
#endif
// PM = 1: f16x2 planes (gemm_f32.hpp): three products per block pair instead of six
template <int EPI, int WMB, int WNB, int WAVES_M, int WAVES_N, int PM = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_xlp_kernel(const GemmArgs g) {
  constexpr int NT = 64 * WAVES_M * WAVES_N, RPS = NT / 4;            // threads; rows per staging pass
  constexpr int TM = 16 * WMB * WAVES_M, TN = 16 * WNB * WAVES_N, TR = TM + TN;
  constexpr int NPL = PM == 1 ? 2 : 3;                                // planes per operand (f16x2: X1 | X2, X1 / 64 formed in registers)
  constexpr int BUFC = NPL * TR * 4;                                  // 16-byte chunks per buffer
  extern __shared__ __attribute__((aligned(16))) u4v xlp_smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int wm = (w / WAVES_N) * 16 * WMB, wn = (w % WAVES_N) * 16 * WNB;
  int bx = blockIdx.x, by = blockIdx.y;
  if ((gridDim.y & 7) == 0 && gridDim.y >= 16) {                      // XCD-aware tile order (gemm_nt_kernel)
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, k = lin >> 3;
    by = (int)((k / gridDim.x) * 8 + xcd);
    bx = (int)(k % gridDim.x);
  }
  const long long m0 = (long long)by * TM;
  const int n0 = bx * TN;

  constexpr int PA = (TM + RPS - 1) / RPS, PB = (TN + RPS - 1) / RPS, NA = NPL * PA, NB = NPL * PB, NCH = NA + NB;
  u4v rch[NCH];
  const int srow = tid >> 2, sch = tid & 3;
  const int s_lds = srow * 4 + (sch ^ ((srow >> 2) & 3));             // this thread's chunk inside a pass of RPS rows
  const bool fast = m0 + TM <= g.M;          // (the weight planes hold whole tiles of rows and both row strides whole k-tiles)
  auto gload1 = [&](auto ci_c, int k0, auto fast_c) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr bool isA = ci < NA;
    constexpr int pl = isA ? ci / PA : (ci - NA) / PB, pass = isA ? ci % PA : (ci - NA) % PB;
    const int row = srow + RPS * pass;
    const unsigned short* base = isA ? g.Ap + pl * g.ap_plane : g.Bp + pl * g.bp_plane;
    const long long ld = isA ? g.ldap : g.ldbp;
    const long long r0 = isA ? m0 : (long long)n0;
    const int k = k0 + 8 * sch;
    long long r = r0 + row;
    // the last row tile of the activations: rows beyond the matrix read its last row (their results are never stored)
    if constexpr (isA && !decltype(fast_c)::value) r = r < g.M - 1 ? r : (long long)g.M - 1;
    rch[ci] = *reinterpret_cast<const u4v*>(base + r * ld + k);
  };
  auto sstore1 = [&](auto ci_c, int buf) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr bool isA = ci < NA;
    constexpr int pl = isA ? ci / PA : (ci - NA) / PB, pass = isA ? ci % PA : (ci - NA) % PB;
    constexpr int T = isA ? TM : TN, rbase = isA ? 0 : TM;
    if (T % RPS == 0 || srow + RPS * pass < T) xlp_smem[buf * BUFC + (pl * TR + rbase + RPS * pass) * 4 + s_lds] = rch[ci];
  };

  f4 acc[WNB][WMB];
#pragma unroll
  for (int i = 0; i < WNB; ++i)
#pragma unroll
    for (int j = 0; j < WMB; ++j) acc[i][j] = splat(0.f);

  const int nk = (g.K + 31) / 32;
  // staging chunks per stage, in the first three stages of the k-tile: the compiler waits for EVERY outstanding load before the
  // first ds_write of the next k-tile (its counter bookkeeping across the loop edge is conservative), so the youngest load must
  // be old by then (3 stages 4647 cycles per k-tile, 6 stages 4817, 9 stages 4677 -- measured on the end-of-tile-barrier form)
  constexpr int CPS = (NCH + 2) / 3;
  const int f_lds = c * 4 + (q ^ ((c >> 2) & 3));                     // this lane's chunk inside a 16-row fragment
  // The k-tile (header): fragment order W0 A0 A1 W1 W2 W3 A2 A3 -- A0 / A1 are dead after the fourth stage and take the next tile's
  // behind the barrier; its W0 waits in 12 extra registers (sw0n).  The loop exists twice: whole row tiles load without the row clamp.
  static_assert(WMB == 4 && WNB == 4, "the mid-barrier schedule is written for 4 x 4 blocks");
  auto main_loop = [&](auto fast_c) {
  Split3 sa[4], sw[4], sw0n;
  // (f16x2: plane 0 = X1, plane 1 = X2; the middle term X1 / 64 is four v_pk_mul_f16 on the fragment -- exact: a power of two)
  auto d64 = [](u4v h) { return __builtin_bit_cast(u4v, __builtin_bit_cast(hf8, h) * (_Float16)0.015625f); };
  auto rdA = [&](Split3& dst, const u4v* base, int j) {
    const u4v* pa = base + (wm + 16 * j) * 4;
    if constexpr (PM == 1) { dst.h = pa[0]; dst.l = pa[TR * 4]; dst.m = d64(dst.h); }
    else { dst.h = pa[0]; dst.m = pa[TR * 4]; dst.l = pa[2 * TR * 4]; }
  };
  auto rdW = [&](Split3& dst, const u4v* base, int i) {
    const u4v* pw = base + (TM + wn + 16 * i) * 4;
    if constexpr (PM == 1) { dst.h = pw[0]; dst.l = pw[TR * 4]; dst.m = d64(dst.h); }
    else { dst.h = pw[0]; dst.m = pw[TR * 4]; dst.l = pw[2 * TR * 4]; }
  };
  auto mm = [&](const Split3& w_, const Split3& a0, const Split3& a1, f4& c0, f4& c1) {       // two accumulators, product-major
    if constexpr (PM == 1) {
      c0 = mfma_f16(w_.l, a0.m, c0); c1 = mfma_f16(w_.l, a1.m, c1);
      c0 = mfma_f16(w_.m, a0.l, c0); c1 = mfma_f16(w_.m, a1.l, c1);
      c0 = mfma_f16(w_.h, a0.h, c0); c1 = mfma_f16(w_.h, a1.h, c1);
      return;
    }
    c0 = mfma_bf16(w_.l, a0.h, c0); c1 = mfma_bf16(w_.l, a1.h, c1);
    c0 = mfma_bf16(w_.h, a0.l, c0); c1 = mfma_bf16(w_.h, a1.l, c1);
    c0 = mfma_bf16(w_.m, a0.m, c0); c1 = mfma_bf16(w_.m, a1.m, c1);
    c0 = mfma_bf16(w_.m, a0.h, c0); c1 = mfma_bf16(w_.m, a1.h, c1);
    c0 = mfma_bf16(w_.h, a0.m, c0); c1 = mfma_bf16(w_.h, a1.m, c1);
    c0 = mfma_bf16(w_.h, a0.h, c0); c1 = mfma_bf16(w_.h, a1.h, c1);
  };
  auto mm4 = [&](const Split3& a_, int j) {                                                     // four accumulators (all W, one A)
    if constexpr (PM == 1) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (p == 0) acc[i][j] = mfma_f16(sw[i].l, a_.m, acc[i][j]);
          if (p == 1) acc[i][j] = mfma_f16(sw[i].m, a_.l, acc[i][j]);
          if (p == 2) acc[i][j] = mfma_f16(sw[i].h, a_.h, acc[i][j]);
        }
      return;
    }
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (p == 0) acc[i][j] = mfma_bf16(sw[i].l, a_.h, acc[i][j]);
        if (p == 1) acc[i][j] = mfma_bf16(sw[i].h, a_.l, acc[i][j]);
        if (p == 2) acc[i][j] = mfma_bf16(sw[i].m, a_.m, acc[i][j]);
        if (p == 3) acc[i][j] = mfma_bf16(sw[i].m, a_.h, acc[i][j]);
        if (p == 4) acc[i][j] = mfma_bf16(sw[i].h, a_.m, acc[i][j]);
        if (p == 5) acc[i][j] = mfma_bf16(sw[i].h, a_.h, acc[i][j]);
      }
  };
  auto stage_chunks = [&](auto tc, int buf, int k_next) {
    constexpr int t = decltype(tc)::value;
    xl_static_for<0, CPS>([&](auto uc) {
      constexpr int ci = CPS * t + decltype(uc)::value;
      if constexpr (ci < NCH) {
        sstore1(std::integral_constant<int, ci>{}, buf ^ 1);
        gload1(std::integral_constant<int, ci>{}, k_next, fast_c);
      }
    });
  };
  xl_static_for<0, NCH>([&](auto ci) { gload1(ci, 0, fast_c); });
  xl_static_for<0, NCH>([&](auto ci) { sstore1(ci, 0); });
  __syncthreads();
  xl_static_for<0, NCH>([&](auto ci) { gload1(ci, (nk > 1 ? 1 : 0) * 32, fast_c); });
  rdW(sw0n, xlp_smem + f_lds, 0); rdA(sa[0], xlp_smem + f_lds, 0); rdA(sa[1], xlp_smem + f_lds, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const int k_next = (kt + 2 < nk ? kt + 2 : nk - 1) * 32;
    const u4v* sb = xlp_smem + buf * BUFC + f_lds;
    const u4v* sbn = xlp_smem + (buf ^ 1) * BUFC + f_lds;
    sw[0] = sw0n;
    rdW(sw[1], sb, 1); rdW(sw[2], sb, 2);
    XL_FENCE();
    stage_chunks(std::integral_constant<int, 0>{}, buf, k_next); rdW(sw[3], sb, 3);
    mm(sw[0], sa[0], sa[1], acc[0][0], acc[0][1]);
    XL_FENCE();
    stage_chunks(std::integral_constant<int, 1>{}, buf, k_next); rdA(sa[2], sb, 2);
    mm(sw[1], sa[0], sa[1], acc[1][0], acc[1][1]);
    XL_FENCE();
    stage_chunks(std::integral_constant<int, 2>{}, buf, k_next); rdA(sa[3], sb, 3);
    mm(sw[2], sa[0], sa[1], acc[2][0], acc[2][1]);
    XL_FENCE();
    mm(sw[3], sa[0], sa[1], acc[3][0], acc[3][1]);
    XL_FENCE();
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    XL_FENCE();
    rdA(sa[0], sbn, 0); rdA(sa[1], sbn, 1); rdW(sw0n, sbn, 0);
    mm4(sa[2], 2);
    XL_FENCE();
    mm4(sa[3], 3);
    XL_FENCE();
  }
  };
#ifdef L2HMC_XL_TIMING
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#endif
  if (fast) main_loop(std::true_type{});
  else main_loop(std::false_type{});
#ifdef L2HMC_XL_TIMING
  if (tid == 0) {
    atomicAdd(&xl_ticks[0], __builtin_readcyclecounter() - t0);
    atomicAdd(&xl_ticks[1], 1ull);
    atomicAdd(&xl_ticks[2], wall_clock64() - w0);
  }
#endif
  gemm_epilogue<EPI, WMB, WNB, WAVES_N>(g, acc, m0, n0, wm, wn, w, c, q);
}

template <int WMB, int WNB, int WAVES_M, int WAVES_N, int PM = 0>
constexpr size_t gemm_xlp_lds_bytes() { return (size_t)2 * (PM == 1 ? 2 : 3) * (16 * WMB * WAVES_M + 16 * WNB * WAVES_N) * 64; }

constexpr int XLP_TM = 256, XLP_TN = 128;
#ifndef XLP_MIN_TILES
#define XLP_MIN_TILES 84             // measured (profiles/r04_gemm_xl.txt): the planes form wins from 3072 chains (12 x 7 tiles) up
#endif
inline int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

// The kernel's 144 KB of dynamic LDS have to be asked for once per instantiation: an entry point that is going to use the
// planes form calls this FIRST and returns its error (the launches below cannot report one to their void callers)
constexpr int XLP_WMB = 4, XLP_WNB = 4, XLP_WAVES_M = 4, XLP_WAVES_N = 2;
template <int EPI, int PM = 0>
int gemm_planes_prepare() {
  constexpr size_t lds = gemm_xlp_lds_bytes<XLP_WMB, XLP_WNB, XLP_WAVES_M, XLP_WAVES_N, PM>();
  static_assert(lds <= 160 * 1024, "two plane buffers must fit the CU's LDS");
  // The attribute is per DEVICE: remembered per device id (one bit each, atomically -- the only process-wide word the library
  // keeps, and it only ever says "already asked"); ids beyond 63 simply ask every time.  (Round 4 kept ONE flag: a second GPU of
  // the same process would have launched without the attribute.)
  static std::atomic<unsigned long long> asked{0ull};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return fail(L2HMC_ERR_HIP, "gemm planes: %s", "hipGetDevice failed");
  const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
  if (bit && (asked.load(std::memory_order_acquire) & bit)) return L2HMC_OK;
  auto kern = gemm_xlp_kernel<EPI, XLP_WMB, XLP_WNB, XLP_WAVES_M, XLP_WAVES_N, PM>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(L2HMC_ERR_HIP, "gemm planes: %s", "the device refuses the LDS size of the 256-row plane tiles");
  asked.fetch_or(bit, std::memory_order_release);
  return L2HMC_OK;
}
// C = epilogue(A B^T) on planes; GemmArgs as for launch_gemm with Ap / Bp instead of A / B (contract at GemmArgs)
template <int EPI>
int launch_gemm_planes(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return L2HMC_OK;
  if (g.ldap < ceil_to(g.K, 32) || g.ldbp < ceil_to(g.K, 32) || (g.ldap & 7) || (g.ldbp & 7))
    return fail(L2HMC_ERR_ARG, "gemm planes: row strides must cover ceil32(K) (zero-padded) in multiples of 8%s");
  const int rc = g.pm ? gemm_planes_prepare<EPI, 1>() : gemm_planes_prepare<EPI, 0>();
  if (rc != L2HMC_OK) return rc;
  const dim3 grid((unsigned)((g.N + XLP_TN - 1) / XLP_TN), (unsigned)((g.M + XLP_TM - 1) / XLP_TM));
  constexpr size_t lds = gemm_xlp_lds_bytes<XLP_WMB, XLP_WNB, XLP_WAVES_M, XLP_WAVES_N, 0>();
  constexpr size_t lds1 = gemm_xlp_lds_bytes<XLP_WMB, XLP_WNB, XLP_WAVES_M, XLP_WAVES_N, 1>();
  if (g.pm) hipLaunchKernelGGL((gemm_xlp_kernel<EPI, XLP_WMB, XLP_WNB, XLP_WAVES_M, XLP_WAVES_N, 1>), grid, dim3(64 * XLP_WAVES_M * XLP_WAVES_N), lds1, s, g);
  else hipLaunchKernelGGL((gemm_xlp_kernel<EPI, XLP_WMB, XLP_WNB, XLP_WAVES_M, XLP_WAVES_N, 0>), grid, dim3(64 * XLP_WAVES_M * XLP_WAVES_N), lds, s, g);
  return L2HMC_OK;
}
// can this product take the pre-split form?  (decoder-sized: from 3072 chains at the widths of config 5 -- a third of the CUs get a
// 256 x 128 tile, and the k loop is as long whether 84 or 256 tiles run it)
inline bool gemm_planes_ok(long long M, int N, int K) {
#ifdef L2HMC_NO_PLANES            // A/B builds (tools/build_variant_full.sh): every product keeps the in-loop split
  return false;
#endif
  return ((M + XLP_TM - 1) / XLP_TM) * ((N + XLP_TN - 1) / XLP_TN) >= XLP_MIN_TILES && K >= 256 && K % 8 == 0 && N % 4 == 0;
}
// EPI_BCE on planes: row partials per chain = 2 per 128-wide column tile
inline int bce_tiles_planes(int n_pix) { return (n_pix + XLP_TN - 1) / XLP_TN; }

// fp32 matrix (rows x K, row stride ld) -> its three bf16 planes of rows_pad x ldp elements each (rows_pad >= rows, ldp >= K, both
// zero-filled beyond the matrix): weights, once per parameter update
__global__ void to_planes_kernel(const float* W, int ld, long long rows, int K, unsigned short* P, long long rows_pad, int ldp, int pm) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 4 consecutive k
  const int K4 = ldp / 4;
  if (i >= rows_pad * K4) return;
  const long long r = i / K4;
  const int k = (int)(i % K4) * 4;
  f4 v = splat(0.f);
  if (r < rows) {
    const float* p = W + r * ld + k;
    if (k + 0 < K) v.x = p[0];
    if (k + 1 < K) v.y = p[1];
    if (k + 2 < K) v.z = p[2];
    if (k + 3 < K) v.w = p[3];
  }
  const Split4 sp = pm ? split4_f16(v) : split4(v);
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  unsigned short* o = P + r * ldp + k;
  const long long plane = rows_pad * (long long)ldp;
  *reinterpret_cast<u2v*>(o) = u2v{sp.h[0], sp.h[1]};
  if (pm) {                                                      // f16x2: X1 | X2 (the third plane of the allocation stays unused)
    *reinterpret_cast<u2v*>(o + plane) = u2v{sp.l[0], sp.l[1]};
    return;
  }
  *reinterpret_cast<u2v*>(o + plane) = u2v{sp.m[0], sp.m[1]};
  *reinterpret_cast<u2v*>(o + 2 * plane) = u2v{sp.l[0], sp.l[1]};
}
inline void to_planes(hipStream_t s, const float* W, int ld, long long rows, int K, unsigned short* P, long long rows_pad, int ldp,
                      int pm = 0) {
  const long long n = rows_pad * (ldp / 4);
  hipLaunchKernelGGL(to_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ld, rows, K, P, rows_pad, ldp, pm);
}
// zero the columns [N, ldp) of the three planes of an activation the epilogues write (they only touch columns < N)
__global__ void planes_zero_pad_kernel(unsigned short* P, long long rows, int N, int ldp) {
  const int padc = ldp - N;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * rows * padc) return;
  const long long r = i / padc;                                                   // (plane, row) flattened: planes are contiguous
  P[r * ldp + N + (int)(i % padc)] = 0;
}
inline void planes_zero_pad(hipStream_t s, unsigned short* P, long long rows, int N, int ldp) {
  if (ldp == N) return;
  const long long n = 3 * rows * (ldp - N);
  hipLaunchKernelGGL(planes_zero_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P, rows, N, ldp);
}

}  // namespace l2hmc
