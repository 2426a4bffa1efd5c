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
// K is consumed in tiles of 16: both tiles are staged in LDS as [row][16 k + 4 pad] (80-byte rows: one
// conflict-free ds_read_b128 per lane fetches the operands of the 4 k-steps of a 16-wide tile), double
// buffered, the next tile's global loads issued before the 64 MFMAs of the current one.
//
// Epilogues (fused, so no activation makes an extra HBM round trip):
//   EPI_BIAS            C = acc + b
//   EPI_BIAS_SOFTPLUS   C = softplus(acc + b),  C2 = sigmoid(acc + b)   (C2 feeds the backward pass)
//   EPI_BIAS_RELU       C = relu(acc + b)
//   EPI_MUL             C = acc * E[m][n]                               (backward through softplus)
//   EPI_ADD             C = acc + E[m][n]                               (d/dz of the prior term)
//   EPI_BCE             l = acc + b;  C = beta (sigmoid(l) - t),  rowsum[m][tile] = beta sum_n bce(l, t)
//                       (mnist_vae.py:122-126, TF's stable form max(l,0) - l t + log1p(e^{-|l|}))
//   EPI_NET1            C = relu(acc + tb[row(m)][n] + auxh[m][n])      (first hidden layer of an S/T/Q net:
//                       time/bias table row of the chain's schedule row, image-branch term)
#pragma once
#include <hip/hip_runtime.h>

#include "l2hmc_kernels.hpp"

namespace l2hmc {

enum { EPI_BIAS = 0, EPI_BIAS_SOFTPLUS = 1, EPI_BIAS_RELU = 2, EPI_MUL = 3, EPI_ADD = 4, EPI_BCE = 5, EPI_NET1 = 6 };

struct GemmArgs {
  const float* A; int lda;       // (M, K)
  const float* B; int ldb;       // (N, K)
  float* C; int ldc;             // (M, N)
  int M, N, K;
  const float* bias;             // (N) or NULL
  const float* E; int lde;       // (M, N) second operand of the epilogue (sigmoid / aux / z / auxh) or NULL
  float* C2; int ldc2;           // second output (sigmoid) or NULL
  float* rowsum; int n_tiles;    // EPI_BCE: (M, 2 * n_tiles) partial sums, one per (column tile, wave column)
  float beta;                    // EPI_BCE scale
  // EPI_NET1
  const float* tb;               // (T, N) time/bias table of this net
  const unsigned char* dir; int dir_all, it, T;
};

constexpr int GK = 16, GP = 20;               // k-tile, padded LDS row (floats)

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

// WMB x WNB: 16 x 16 MFMA tiles per wave along m and n; the 2 x 2 waves of a workgroup cover a
// (32 WMB) x (32 WNB) tile of C.  4 x 4 (128 x 128) for the big decoder products, 2 x 2 (64 x 64) for the
// H = 200 net layers (fills the chip at M = 8192), 1 x 2 (32 x 64) for the N = d = 50 latent gradient.
template <int EPI, int KV, int WMB, int WNB>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs g) {
  constexpr int TM = 32 * WMB, TN = 32 * WNB;
  __shared__ __attribute__((aligned(16))) float sA[2][TM * GP];   // activations  [m][k]
  __shared__ __attribute__((aligned(16))) float sB[2][TN * GP];   // weights      [n][k]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int wm = (w >> 1) * 16 * WMB, wn = (w & 1) * 16 * WNB;    // this wave's quadrant
  const long long m0 = (long long)blockIdx.y * TM;
  const int n0 = blockIdx.x * TN;

  // global -> register staging: thread loads k quad (tid & 3) of rows (tid >> 2) + 64 i
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  constexpr int NA = (TM + 63) / 64, NB = (TN + 63) / 64;
  f4 ra[NA], rb[NB];
  auto gload = [&](int k0) {
    const int k = k0 + lk;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const long long m = m0 + lr + 64 * i;
      ra[i] = (lr + 64 * i < TM && m < g.M) ? load_kquad<KV>(g.A + m * g.lda + k, k, g.K) : splat(0.f);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + lr + 64 * i;
      rb[i] = (lr + 64 * i < TN && n < g.N) ? load_kquad<KV>(g.B + (long long)n * g.ldb + k, k, g.K) : splat(0.f);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (lr + 64 * i < TM) *reinterpret_cast<f4*>(&sA[buf][(lr + 64 * i) * GP + lk]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (lr + 64 * i < TN) *reinterpret_cast<f4*>(&sB[buf][(lr + 64 * i) * GP + lk]) = rb[i];
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
    f4 fw[WNB], fa[WMB];
#pragma unroll
    for (int i = 0; i < WNB; ++i) fw[i] = *reinterpret_cast<const f4*>(&sB[buf][(wn + 16 * i + c) * GP + 4 * q]);
#pragma unroll
    for (int j = 0; j < WMB; ++j) fa[j] = *reinterpret_cast<const f4*>(&sA[buf][(wm + 16 * j + c) * GP + 4 * q]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < WNB; ++i)
#pragma unroll
        for (int j = 0; j < WMB; ++j) acc[i][j] = MFMA16(fw[i][s], fa[j][s], acc[i][j]);
    if (kt + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

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
      st4(g.C + m * g.ldc + n, out);
      if (EPI == EPI_BIAS_SOFTPLUS && g.C2 != nullptr) st4(g.C2 + m * g.ldc2 + n, out2);
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
      if (q == 0 && m < g.M) g.rowsum[m * (2 * g.n_tiles) + 2 * blockIdx.x + (w & 1)] = g.beta * s;
    }
  }
}

// rowsum layout of EPI_BCE: (M, 2 * n_tiles) with n_tiles = ceil(N / (32 WNB)) of the shape the launcher picks
template <int EPI, int WMB, int WNB>
int launch_gemm_shape(const GemmArgs& g, hipStream_t s) {
  const dim3 grid((unsigned)((g.N + 32 * WNB - 1) / (32 * WNB)), (unsigned)((g.M + 32 * WMB - 1) / (32 * WMB)));
  const bool al16 = ((reinterpret_cast<size_t>(g.A) | reinterpret_cast<size_t>(g.B)) & 15) == 0;
  const bool al8 = ((reinterpret_cast<size_t>(g.A) | reinterpret_cast<size_t>(g.B)) & 7) == 0;
  if (g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && al16)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, 4, WMB, WNB>), grid, dim3(256), 0, s, g);
  else if (g.K % 2 == 0 && g.lda % 2 == 0 && g.ldb % 2 == 0 && al8)
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, 2, WMB, WNB>), grid, dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, 1, WMB, WNB>), grid, dim3(256), 0, s, g);
  return L2HMC_OK;
}

enum { SHAPE_BIG = 0, SHAPE_MID = 1, SHAPE_SKINNY = 2 };     // 128 x 128, 64 x 64, 32 x 64 workgroup tiles
inline int gemm_tile_n(int shape) { return shape == SHAPE_BIG ? 128 : 64; }

template <int EPI>
int launch_gemm(const GemmArgs& g, hipStream_t s, int shape = SHAPE_BIG) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return L2HMC_OK;
  if (shape == SHAPE_BIG) return launch_gemm_shape<EPI, 4, 4>(g, s);
  if (shape == SHAPE_MID) return launch_gemm_shape<EPI, 2, 2>(g, s);
  return launch_gemm_shape<EPI, 1, 2>(g, s);
}

}  // namespace l2hmc
