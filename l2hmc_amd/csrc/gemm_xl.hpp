// gemm_xl.hpp -- the bf16x3 NT GEMM of gemm_f32.hpp on 256-row workgroup tiles with ONE wave per SIMD (gfx950 / CDNA4).
//
// Why (profiles/r04_gemm_planes.txt, r04_gemm_xl.txt): gemm_nt_kernel<.., BF3 = 1> on 128 x 128 tiles spends 7100 cycles per
// k-tile of 32 on a CU (two workgroups, two waves per SIMD) of which the matrix pipe works 3070: every wave splits 4 + 4
// fragments for 96 MFMAs (3 VALU per MFMA, and a wave's VALU instruction costs 5.5 issue cycles), the eight waves between the
// same two barriers do so at the same time, and a 128 x 128 tile re-reads 256 operand rows per k-tile from L2.  Here a wave
// owns a 128 x 64 (or 64 x 112) block of C:
//   * 8 + 4 fragments split for 192 MFMAs = 2.25 VALU per MFMA -- the two VALU instructions a wave issues for free in the
//     shadow of one 16-cycle bf16 MFMA (profiles/r03_ubench_issue.txt: 1 MFMA + 2 v_fma_f32 = 16.8 cycles);
//   * 0.75x the LDS fragment reads and 0.75x the L2 -> LDS bytes per flop (384 operand rows per 256 x 128 tile);
//   * the k-tile is ONE software pipeline per wave: the fragments are split in the order W0 A0 A1 W1 A2 A3 ... and the MFMAs
//     of the (W_i, A_j) pairs a split makes possible are issued beside the NEXT split (sched_group_barrier keeps that
//     interleaving), so only the first two splits of a k-tile run without matrix work next to them.
// One 256-thread workgroup (110 KB of LDS) per CU; the accumulators (128 registers) live in the upper half of the 512-entry
// register file a lone wave owns.  Same staging, LDS layout, product order per accumulator and epilogues as gemm_nt_kernel:
// results are bit-identical to the 128 x 128 form.
#pragma once
#include <type_traits>

#include "gemm_f32.hpp"

namespace l2hmc {

// position of a fragment in the split order: the operand with more blocks ("major") takes slots 1, 2 of every three, the
// other one slot 0; leftovers of either kind follow in index order.  8 x 4 blocks: W0 A0 A1 W1 A2 A3 W2 A4 A5 W3 A6 A7 --
// the splits complete 1 1 2 2 2 4 3 3 6 4 4 (W, A) pairs, whose MFMAs run beside the following split.
template <int WMB, int WNB>
struct XlSched {
  static constexpr bool A_MAJOR = WMB >= WNB;
  static constexpr int NMAJ = A_MAJOR ? WMB : WNB, NMIN = A_MAJOR ? WNB : WMB;
  static constexpr int triples = (NMAJ / 2 < NMIN) ? NMAJ / 2 : NMIN;           // complete (minor, major, major) groups
  static constexpr int pos_min(int i) { return i < triples ? 3 * i : 3 * triples + (NMAJ - 2 * triples) + (i - triples); }
  static constexpr int pos_maj(int j) { return j < 2 * triples ? 3 * (j / 2) + 1 + (j % 2) : 3 * triples + (j - 2 * triples); }
  static constexpr int posA(int j) { return A_MAJOR ? pos_maj(j) : pos_min(j); }
  static constexpr int posW(int i) { return A_MAJOR ? pos_min(i) : pos_maj(i); }
  static constexpr int ready(int i, int j) { return posW(i) > posA(j) ? posW(i) : posA(j); }
  static constexpr int pairs(int t) {                                           // pairs completed by the split of stage t
    int n = 0;
    for (int i = 0; i < WNB; ++i)
      for (int j = 0; j < WMB; ++j) n += ready(i, j) == t;
    return n;
  }
  static constexpr int n = WMB + WNB;
};
#ifndef L2HMC_XL_NO_FENCE
#define XL_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define XL_FENCE()
#endif
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
template <int EPI, int WMB, int WNB, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_xl_kernel(const GemmArgs g) {
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int TM = 16 * WMB * WAVES_M, TN = 16 * WNB * WAVES_N;
  constexpr int GK = 32, GP = GK + 4;
  extern __shared__ __attribute__((aligned(16))) float xl_smem[];
  auto sA = [&](int buf) { return xl_smem + buf * (TM + TN) * GP; };              // activations [m][k]
  auto sB = [&](int buf) { return xl_smem + buf * (TM + TN) * GP + TM * GP; };    // weights [n][k]
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

  // global -> register -> LDS staging in CHUNKS of 32 rows x 32 k (one dwordx4 per thread): chunks 0 .. NA-1 are rows of the
  // activation tile, NA .. NA+NB-1 rows of the weight tile
  constexpr int QPR = GK / 4, RPP = 256 / QPR;                        // 8 quads per row, 32 rows per chunk
  const int lr = tid / QPR, lk = (tid % QPR) * 4;
  constexpr int NA = (TM + RPP - 1) / RPP, NB = (TN + RPP - 1) / RPP, NCH = NA + NB;
  f4 rch[NCH];
  const bool interior = TM % RPP == 0 && TN % RPP == 0 && m0 + TM <= g.M && n0 + TN <= g.N && g.K % GK == 0;
  auto gload1 = [&](auto ci_c, int k0, auto interior_c) {
    constexpr int ci = decltype(ci_c)::value;
    const int k = k0 + lk;
    if constexpr (ci < NA) {
      const long long m = m0 + lr + RPP * ci;
      if constexpr (decltype(interior_c)::value) rch[ci] = *reinterpret_cast<const f4*>(g.A + m * g.lda + k);
      else rch[ci] = (lr + RPP * ci < TM && m < g.M) ? load_kquad<4>(g.A + m * g.lda + k, k, g.K) : splat(0.f);
    } else {
      const int n = n0 + lr + RPP * (ci - NA);
      if constexpr (decltype(interior_c)::value) rch[ci] = *reinterpret_cast<const f4*>(g.B + (long long)n * g.ldb + k);
      else rch[ci] = (lr + RPP * (ci - NA) < TN && n < g.N) ? load_kquad<4>(g.B + (long long)n * g.ldb + k, k, g.K) : splat(0.f);
    }
  };
  auto sstore1 = [&](auto ci_c, int buf) {
    constexpr int ci = decltype(ci_c)::value;
    if constexpr (ci < NA) {
      if (TM % RPP == 0 || lr + RPP * ci < TM) *reinterpret_cast<f4*>(&sA(buf)[(lr + RPP * ci) * GP + lk]) = rch[ci];
    } else {
      if (TN % RPP == 0 || lr + RPP * (ci - NA) < TN) *reinterpret_cast<f4*>(&sB(buf)[(lr + RPP * (ci - NA)) * GP + lk]) = rch[ci];
    }
  };

  f4 acc[WNB][WMB];
#pragma unroll
  for (int i = 0; i < WNB; ++i)
#pragma unroll
    for (int j = 0; j < WMB; ++j) acc[i][j] = splat(0.f);

  using Ord = XlSched<WMB, WNB>;

  const int nk = (g.K + GK - 1) / GK;
  constexpr int AHEAD = 4;                 // fragments requested from LDS before the first split of a k-tile
  static_assert(NCH <= Ord::n + 1, "one staging chunk per stage");
  // The main loop exists twice: interior tiles (every row of both operand tiles exists, K a multiple of the k-tile) load with
  // plain dwordx4 and no predicates -- one basic block per k-tile; edge tiles take the guarded loads.
  // Memory instructions are SPREAD over the stages of a k-tile (measured, profiles/r04_gemm_xl.txt: with the 12 global loads and
  // the 24 LDS reads of a wave in one burst at the top of the k-tile, the four waves of the CU queue behind one another at the
  // texture addresser and the LDS pipe: +565 and +660 cycles per k-tile): stage s < 6 stores chunks 2 s, 2 s + 1 of tile kt + 1
  // (loaded one k-tile ago) to the other LDS buffer and re-issues those registers' loads for tile kt + 2; stage s requests
  // fragment s + AHEAD.
  auto main_loop = [&](auto interior_c) {
  xl_static_for<0, NCH>([&](auto ci) { gload1(ci, 0, interior_c); });
  xl_static_for<0, NCH>([&](auto ci) { sstore1(ci, 0); });
  __syncthreads();
  xl_static_for<0, NCH>([&](auto ci) { gload1(ci, (nk > 1 ? 1 : 0) * GK, interior_c); });
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // (past the end the loop fetches / stores the last tile again instead of branching: the body stays ONE basic block, which is
    //  what the interleaving below is pinned in)
    const int k_next = (kt + 2 < nk ? kt + 2 : nk - 1) * GK;
    f4 rawA[WMB][2], rawW[WNB][2];
    auto lds_read = [&](int t) {
#pragma unroll
      for (int j = 0; j < WMB; ++j)
        if (Ord::posA(j) == t) {
          const float* pa = &sA(buf)[(wm + 16 * j + c) * GP + 8 * q];
          rawA[j][0] = *reinterpret_cast<const f4*>(pa); rawA[j][1] = *reinterpret_cast<const f4*>(pa + 4);
        }
#pragma unroll
      for (int i = 0; i < WNB; ++i)
        if (Ord::posW(i) == t) {
          const float* pw = &sB(buf)[(wn + 16 * i + c) * GP + 8 * q];
          rawW[i][0] = *reinterpret_cast<const f4*>(pw); rawW[i][1] = *reinterpret_cast<const f4*>(pw + 4);
        }
    };
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) lds_read(t);
    Split3 sa[WMB], sw[WNB];
    xl_static_for<0, Ord::n + 1>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      // a stage = one chunk of staging + the split of fragment t (VALU) + the MFMAs of the pairs the split of stage t - 1
      // completed; nothing moves across a stage boundary, inside a stage the kinds alternate
      XL_FENCE();
      // (two chunks per stage in the FIRST half of the k-tile: the compiler waits for every outstanding load before the first
      //  ds_write of the next k-tile -- its counter bookkeeping across the loop edge is conservative -- so the youngest load
      //  has to be half a k-tile old by then)
#ifndef L2HMC_XL_ABL_NOLOAD
      if constexpr (2 * t < NCH) {
        sstore1(std::integral_constant<int, 2 * t>{}, buf ^ 1);
        gload1(std::integral_constant<int, 2 * t>{}, k_next, interior_c);
      }
      if constexpr (2 * t + 1 < NCH) {
        sstore1(std::integral_constant<int, 2 * t + 1>{}, buf ^ 1);
        gload1(std::integral_constant<int, 2 * t + 1>{}, k_next, interior_c);
      }
#endif
      if constexpr (t + AHEAD < Ord::n) lds_read(t + AHEAD);
      if constexpr (t < Ord::n) {
#pragma unroll
        for (int j = 0; j < WMB; ++j)
          if (Ord::posA(j) == t) sa[j] = split3(rawA[j][0], rawA[j][1]);
#pragma unroll
        for (int i = 0; i < WNB; ++i)
          if (Ord::posW(i) == t) sw[i] = split3(rawW[i][0], rawW[i][1]);
      }
      // product-major; smallest terms first per accumulator, as in gemm_nt_kernel
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int i = 0; i < WNB; ++i)
#pragma unroll
          for (int j = 0; j < WMB; ++j) {
            if (Ord::ready(i, j) != t - 1) continue;
            if (p == 0) acc[i][j] = mfma_bf16(sw[i].l, sa[j].h, acc[i][j]);
            if (p == 1) acc[i][j] = mfma_bf16(sw[i].h, sa[j].l, acc[i][j]);
            if (p == 2) acc[i][j] = mfma_bf16(sw[i].m, sa[j].m, acc[i][j]);
            if (p == 3) acc[i][j] = mfma_bf16(sw[i].m, sa[j].h, acc[i][j]);
            if (p == 4) acc[i][j] = mfma_bf16(sw[i].h, sa[j].m, acc[i][j]);
            if (p == 5) acc[i][j] = mfma_bf16(sw[i].h, sa[j].h, acc[i][j]);
          }
#ifndef L2HMC_XL_NO_SGB
      constexpr int pairs = t >= 1 ? Ord::pairs(t - 1) : 0;
      if constexpr (pairs > 0) {
        // the memory instructions of the stage go first, then MFMA and VALU alternate; a split is 34-39 VALU instructions and a
        // request beyond what the stage holds is harmless (the fences bound the stage)
#ifdef L2HMC_XL_SGB_MEM
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);         // ds_write (+ its address)
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // global load
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // ds_read x 2
#endif
        constexpr int per = t < Ord::n ? (40 + 6 * pairs - 1) / (6 * pairs) : 1;
#pragma unroll
        for (int r = 0; r < 6 * pairs; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, per, 0);     // its share of the split
        }
      }
#endif
    });
    XL_FENCE();
    __syncthreads();
  }
  };
#ifdef L2HMC_XL_TIMING
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#endif
  if (interior) main_loop(std::true_type{});
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

template <int WMB, int WNB, int WAVES_N>
constexpr size_t gemm_xl_lds_bytes() { return (size_t)2 * (16 * WMB * (4 / WAVES_N) + 16 * WNB * WAVES_N) * 36 * sizeof(float); }

// 256 x 128 tiles (2 x 2 waves of 8 x 4 MFMA tiles)
template <int EPI, int WMB = 8, int WNB = 4, int WAVES_N = 2>
int launch_gemm_xl(const GemmArgs& g, hipStream_t s) {
  constexpr int TM = 16 * WMB * (4 / WAVES_N), TN = 16 * WNB * WAVES_N;
  constexpr size_t lds = gemm_xl_lds_bytes<WMB, WNB, WAVES_N>();
  static bool once = false;                                           // (> 64 KB of dynamic LDS has to be asked for)
  auto kern = gemm_xl_kernel<EPI, WMB, WNB, WAVES_N>;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(L2HMC_ERR_HIP, "gemm_xl: %s", "the device refuses the LDS size of the 256-row tiles");
    once = true;
  }
  const dim3 grid((unsigned)((g.N + TN - 1) / TN), (unsigned)((g.M + TM - 1) / TM));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);
  return L2HMC_OK;
}

// ---- the same tile on PRE-SPLIT operands: no VALU in the main loop ---------------------------------------------------------
// What the in-loop split costs a lone wave (profiles/r04_gemm_xl.txt): a k-tile is 192 MFMAs (3072 cycles of the matrix pipe) and
// 470 VALU instructions at 6.5 cycles each beside them -- the wave is VALU-ISSUE bound (5100-6000 cycles per k-tile), with two
// waves per SIMD and 3 VALU per MFMA just as much (gemm_nt_kernel).  Here both operands arrive as three bf16 planes (GemmArgs
// Ap / Bp: weights converted once per parameter update, activations written as planes by the epilogue that produces them), a
// fragment is three ds_read_b128 and the k-tile is loads, LDS traffic and MFMAs only.
// LDS: [buffer][plane][row: TM activation rows, TN weight rows][4 chunks of 16 bytes = 8 bf16], UNPADDED (2 x 72 KB for 256 x 128
// tiles; padded rows would not fit twice) with the chunk index XOR-swizzled by bits 2-3 of the row: the 16 lanes (row c, chunk
// q) of a fragment read hit 16 distinct 16-byte bank groups, and so do the staging writes.
// Staging: chunk (plane, row, quarter) = one dwordx4; thread t takes quarter t & 3 of rows (t >> 2) + 64 pass.  As in
// gemm_xl_kernel the registers hold tile kt + 1 while tile kt is multiplied; the first six stages store three chunks each to the
// other buffer and re-issue their loads for tile kt + 2.
template <int EPI, int WMB, int WNB, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_xlp_kernel(const GemmArgs g) {
  constexpr int NT = 64 * WAVES_M * WAVES_N, RPS = NT / 4;            // threads; rows per staging pass
  constexpr int TM = 16 * WMB * WAVES_M, TN = 16 * WNB * WAVES_N, TR = TM + TN;
  constexpr int BUFC = 3 * TR * 4;                                    // 16-byte chunks per buffer
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

  constexpr int PA = (TM + RPS - 1) / RPS, PB = (TN + RPS - 1) / RPS, NA = 3 * PA, NB = 3 * PB, NCH = NA + NB;
  u4v rch[NCH];
  const int srow = tid >> 2, sch = tid & 3;
  const int s_lds = srow * 4 + (sch ^ ((srow >> 2) & 3));             // this thread's chunk inside a 64-row pass
  const bool fast = m0 + TM <= g.M && n0 + TN <= g.N && g.K % 32 == 0;
  auto gload1 = [&](auto ci_c, int k0, auto fast_c) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr bool isA = ci < NA;
    constexpr int pl = isA ? ci / PA : (ci - NA) / PB, pass = isA ? ci % PA : (ci - NA) % PB;
    const int row = srow + RPS * pass;
    const unsigned short* base = isA ? g.Ap + pl * g.ap_plane : g.Bp + pl * g.bp_plane;
    const long long ld = isA ? g.ldap : g.ldbp;
    const long long r0 = isA ? m0 : (long long)n0, rmax = isA ? (long long)g.M - 1 : (long long)g.N - 1;
    int k = k0 + 8 * sch;
    if constexpr (decltype(fast_c)::value) {
      rch[ci] = *reinterpret_cast<const u4v*>(base + (r0 + row) * ld + k);
    } else {
      // edge tiles / a ragged last k-tile: rows beyond the matrix read its last row (their results are never stored), k beyond
      // K reads as zero
      long long r = r0 + row;
      r = r < rmax ? r : rmax;
      const bool kok = k < g.K;
      k = kok ? k : g.K - 8;
      const u4v v = *reinterpret_cast<const u4v*>(base + r * ld + k);
      rch[ci] = kok ? v : u4v{0u, 0u, 0u, 0u};
    }
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

  using Ord = XlSched<WMB, WNB>;
  const int nk = (g.K + 31) / 32;
  constexpr int AHEAD = 4;
  constexpr int CPS = (NCH + 5) / 6;                                  // staging chunks per stage (stages 0 .. 5)
  const int f_lds = c * 4 + (q ^ ((c >> 2) & 3));                     // this lane's chunk inside a 16-row fragment
#ifdef L2HMC_XL_ABL_NOREAD
  Split3 sa[WMB], sw[WNB];
#endif
  auto main_loop = [&](auto fast_c) {
  xl_static_for<0, NCH>([&](auto ci) { gload1(ci, 0, fast_c); });
  xl_static_for<0, NCH>([&](auto ci) { sstore1(ci, 0); });
  __syncthreads();
  xl_static_for<0, NCH>([&](auto ci) { gload1(ci, (nk > 1 ? 1 : 0) * 32, fast_c); });
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const int k_next = (kt + 2 < nk ? kt + 2 : nk - 1) * 32;          // (past the end: the last tile again, no branch)
    const u4v* sb = xlp_smem + buf * BUFC + f_lds;
#ifndef L2HMC_XL_ABL_NOREAD
    Split3 sa[WMB], sw[WNB];
#endif
    auto lds_read = [&](int t) {
#ifdef L2HMC_XL_ABL_NOREAD
      if (kt > 0) return;
#endif
#pragma unroll
      for (int j = 0; j < WMB; ++j)
        if (Ord::posA(j) == t) {
          const u4v* pa = sb + (wm + 16 * j) * 4;
          sa[j].h = pa[0]; sa[j].m = pa[TR * 4]; sa[j].l = pa[2 * TR * 4];
        }
#pragma unroll
      for (int i = 0; i < WNB; ++i)
        if (Ord::posW(i) == t) {
          const u4v* pw = sb + (TM + wn + 16 * i) * 4;
          sw[i].h = pw[0]; sw[i].m = pw[TR * 4]; sw[i].l = pw[2 * TR * 4];
        }
    };
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) lds_read(t);
    xl_static_for<0, Ord::n + 1>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      XL_FENCE();
      xl_static_for<0, CPS>([&](auto uc) {
        constexpr int ci = CPS * t + decltype(uc)::value;
        if constexpr (t < 6 && ci < NCH) {
#ifndef L2HMC_XL_ABL_NOSTORE
          sstore1(std::integral_constant<int, ci>{}, buf ^ 1);
#endif
#ifndef L2HMC_XL_ABL_NOLOAD
          gload1(std::integral_constant<int, ci>{}, k_next, fast_c);
#endif
        }
      });
      if constexpr (t + AHEAD < Ord::n) lds_read(t + AHEAD);
      // the pairs fragment t - 1 completed; product-major, smallest terms first per accumulator (as gemm_nt_kernel)
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int i = 0; i < WNB; ++i)
#pragma unroll
          for (int j = 0; j < WMB; ++j) {
            if (Ord::ready(i, j) != t - 1) continue;
            if (p == 0) acc[i][j] = mfma_bf16(sw[i].l, sa[j].h, acc[i][j]);
            if (p == 1) acc[i][j] = mfma_bf16(sw[i].h, sa[j].l, acc[i][j]);
            if (p == 2) acc[i][j] = mfma_bf16(sw[i].m, sa[j].m, acc[i][j]);
            if (p == 3) acc[i][j] = mfma_bf16(sw[i].m, sa[j].h, acc[i][j]);
            if (p == 4) acc[i][j] = mfma_bf16(sw[i].h, sa[j].m, acc[i][j]);
            if (p == 5) acc[i][j] = mfma_bf16(sw[i].h, sa[j].h, acc[i][j]);
          }
    });
    XL_FENCE();
    __syncthreads();
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

template <int WMB, int WNB, int WAVES_M, int WAVES_N>
constexpr size_t gemm_xlp_lds_bytes() { return (size_t)2 * 3 * (16 * WMB * WAVES_M + 16 * WNB * WAVES_N) * 64; }

// 256 x 128 tiles: 2 x 2 waves of 128 x 64 blocks (one wave per SIMD) or 4 x 2 waves of 64 x 64 blocks (two per SIMD)
template <int EPI, int WMB = 8, int WNB = 4, int WAVES_M = 2, int WAVES_N = 2>
int launch_gemm_xlp(const GemmArgs& g, hipStream_t s) {
  constexpr int TM = 16 * WMB * WAVES_M, TN = 16 * WNB * WAVES_N;
  constexpr size_t lds = gemm_xlp_lds_bytes<WMB, WNB, WAVES_M, WAVES_N>();
  static_assert(lds <= 160 * 1024, "two plane buffers must fit the CU's LDS");
  static bool once = false;
  auto kern = gemm_xlp_kernel<EPI, WMB, WNB, WAVES_M, WAVES_N>;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(L2HMC_ERR_HIP, "gemm_xlp: %s", "the device refuses the LDS size of the 256-row plane tiles");
    once = true;
  }
  const dim3 grid((unsigned)((g.N + TN - 1) / TN), (unsigned)((g.M + TM - 1) / TM));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, g);
  return L2HMC_OK;
}


}  // namespace l2hmc
