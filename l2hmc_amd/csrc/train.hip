// l2hmc_train_propose_grad -- one direction-mixed proposal (sampler.py:28-51) together with the
// gradient of its training-loss term (SCGExperiment.ipynb raw lines 156-169) w.r.t. every net
// parameter and the step size, by hand-derived reverse mode through the generalised leapfrog
// trajectory (dynamics.py:115-201, 246-309), including the Hessian-vector path through
// `grad_energy` (TF1 differentiates through tf.gradients).  Derivation = oracle/l2hmc_train_oracle.py.
//
// Tile form: a workgroup of 4 waves owns 16 chains (= the N of v_mfma_f32_16x16x4_f32).  Every
// per-chain vector lives in LDS as a (16, ld) matrix; every matrix product of the forward AND the
// reverse sweep is a set of 16x16 fp32 MFMA tiles whose operands are gathered from those LDS
// matrices with strides:
//     forward layers            out(c, i)  = sum_k in(c, k) W(k, i)         K = d or H
//     input adjoints            din(c, k)  = sum_i dout(c, i) W(k, i)       K = H or d
//     weight gradients          dW(k, i)  += sum_c in(c, k) dout(c, i)      K = 16 chains
// Bias / scale gradients are column sums over the 16 chains.  Gradients accumulate in an LDS
// image of the flat parameter vector (each element owned by one lane per pass: no atomics); every
// workgroup writes its image to the workspace and a second tiny kernel adds them to the caller's
// buffer in block order -- the gradient is bitwise reproducible.  The forward trajectory checkpoints
// (x, v, v_half, y, x') per step to a caller workspace; the reverse sweep re-evaluates each net
// right before back-propagating through it, so only ONE net's activations are resident.
// Thread t of the 256 owns chain t & 15 and dims / hidden units (t >> 4) + 16 j in every elementwise
// phase (no index divisions; its chain's sign, step index and seeds sit in registers); with the
// row pitch below those accesses and the MFMA operand gathers are bank-conflict free.
#include "l2hmc_kernels.hpp"

namespace l2hmc {

// Phase timers (profiling builds only: -DL2HMC_TRAIN_TIMING, tools/train_phase_timing.py): lane 0 of wave 0
// of block 0 accumulates s_memtime deltas per phase kind into dbg[kind].
#ifdef L2HMC_TRAIN_TIMING
__device__ unsigned long long tt_acc[16];
__device__ unsigned long long tt_t0;
#define TT_MARK(i)                                                                      \
  do {                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                          \
      const unsigned long long tt_t1 = __builtin_amdgcn_s_memtime();                    \
      tt_acc[i] += tt_t1 - tt_t0;                                                       \
      tt_t0 = tt_t1;                                                                    \
    }                                                                                   \
  } while (0)
#define TT_START()                                                                      \
  do {                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0) tt_t0 = __builtin_amdgcn_s_memtime();     \
  } while (0)
#else
#define TT_MARK(i)
#define TT_START()
#endif

struct TArgs {
  L2hmcNet xnet, vnet;
  const float *masks, *trig, *alpha;
  float eps_host;
  long long N;
  int d, H, T;
  const float *x, *v;
  const unsigned char* dir;
  int dir_all;
  int ekind;                 // GAUSS_DIAG: prec = (d) | GAUSS_DENSE: raw (d, d) | GMM: raw (k, d, d)
  int ncomp, easy;
  const float *mu, *prec, *logc;
  float eta, den;            // den: Rough-Well divisor (L2hmcEnergy.den, or derived from eta in float32)
  float scale, inv_n;
  float *Lx, *p, *v1, *grad, *ws;
  // l2hmc_train_step: chains [0, n_head) start from x_head (n_head = 0: all from x)
  const float* x_head;
  long long n_head;
#ifdef L2HMC_DBG_EPILOGUE_SELECT     // round-4 experiment: the Metropolis select in the gradient kernel's epilogue
  const float* u;
  float* x_next;
#endif
};
__device__ __forceinline__ const float* x_row0(const TArgs& A, long long n) { return n < A.n_head ? A.x_head : A.x; }

constexpr int TC = 16;       // chains per workgroup
constexpr int TNW = 4;       // waves per workgroup
constexpr int TTHREADS = 64 * TNW;
constexpr int KC = 8;        // max mixture components
constexpr int CKPT = 5;      // checkpointed vectors per chain-step: x, v, v_half, y, x'

__host__ __device__ inline int net_params(int d, int H) { return 5 * d * H + H * H + 6 * H + 5 * d; }
// flat parameter layout of one net == NET_FIELDS order of include/l2hmc.h
struct NetOff { int W1, b1, W2, b2, W3, b3, W4, b4, Ws, bs, Wt, bt, Wq, bq, ls, lq; };
__host__ __device__ inline NetOff net_off(int d, int H) {
  NetOff o;
  int p = 0;
  o.W1 = p; p += d * H; o.b1 = p; p += H; o.W2 = p; p += d * H; o.b2 = p; p += H;
  o.W3 = p; p += 2 * H; o.b3 = p; p += H; o.W4 = p; p += H * H; o.b4 = p; p += H;
  o.Ws = p; p += H * d; o.bs = p; p += d; o.Wt = p; p += H * d; o.bt = p; p += d;
  o.Wq = p; p += H * d; o.bq = p; p += d; o.ls = p; p += d; o.lq = p; p += d;
  return o;
}

// row pitch of a (16, n) LDS matrix: a multiple of 4 floats (float4 epilogue stores) whose
// quarter is odd, so the 16 chains x 4 k-groups of an MFMA operand gather hit 64 distinct banks
__host__ __device__ inline int pitch(int n) {
  int p = (n + 3) / 4 * 4;
  if (((p / 4) & 1) == 0) p += 4;
  return p;
}

// the (16, ldd) matrices
enum { MX, MV, MVH, MY, MXO, MG, MLX, MLV, MDVH, MDZ, MTMP, MDS, MDT, MDQ, MDA, MDB, MDG, MTS, MT, MTQ, N_MD };
static_assert(MDT == MDS + 1 && MDQ == MDS + 2 && MDA == MDS + 3 && MDB == MDS + 4 && MT == MTS + 1 && MTQ == MTS + 2,
              "train.hip indexes these matrices arithmetically");
// the (16, ldh) matrices
enum { MH1, MH2, MDA2, MDA1, MPART, N_MH = MPART + TNW };

struct TLayout {
  int Wx, Wv, Gx, Gv, Ge, Msk, Trg, Mu, Pr, Lc, Ex, md, mh, wv, yv, cs, total;
  int ldd, ldh, P;
};
__host__ __device__ inline TLayout train_layout(int d, int H, int T, int ek, int nc) {
  TLayout L;
  auto up4 = [](int n) { return (n + 3) / 4 * 4; };
  L.ldd = pitch(d); L.ldh = pitch(H); L.P = net_params(d, H);
  int p = 0;
  L.Wx = p; p += up4(L.P);
  L.Wv = p; p += up4(L.P);
  L.Gx = p; p += up4(L.P);                 // Gx, Gv, Ge contiguous: zeroed / flushed together
  L.Gv = L.Gx + L.P; p += up4(L.P);
  L.Ge = L.Gx + 2 * L.P; p += 4;
  L.Msk = p; p += up4(T * d);
  L.Trg = p; p += up4(2 * T);
  L.Mu = p; p += up4(nc * d);
  const int npr = ek == L2HMC_ENERGY_GAUSS_DIAG ? d : (ek == L2HMC_ENERGY_ROUGHWELL ? 0 : nc * d * d);
  L.Pr = p; p += up4(npr);
  L.Lc = p; p += up4(nc);
  L.Ex = p; p += 4 * up4(d);               // exp(lam_s), exp(lam_q) of the X net, then of the V net
  L.md = p; p += N_MD * TC * L.ldd;
  L.mh = p; p += N_MH * TC * L.ldh;
  L.wv = p; p += TC * KC;
  L.yv = p; p += TC * KC;
  L.cs = p; p += 16 * TC;                  // per-chain scalars
  L.total = p;
  return L;
}
// per-chain scalar slots
enum { CS_SGN, CS_LIVE, CS_U0, CS_K0, CS_LAM, CS_DV1P, CS_OK, CS_U1 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// One 16x16 output tile: acc(m, n) += sum_{kk in [k0, k1)} A(m0 + m, kk) B(kk, n0 + n) with
// A(m, kk) = Ap[m * a_sm + kk * a_sk] (m < M) and B(kk, n) = Bp[kk * b_sk + n * b_sn] (n < N).
// Result layout (v_mfma_f32_16x16x4_f32): lane l holds rows m0 + 4 (l >> 4) + r, column n0 + (l & 15).
__device__ __forceinline__ f4 mm_tile(f4 acc, const float* Ap, int a_sm, int a_sk, int M, int m0,
                                      const float* Bp, int b_sk, int b_sn, int N, int n0, int k0,
                                      int k1, int lane) {
  const int r = lane & 15, g = lane >> 4;
  const bool mok = m0 + r < M, nok = n0 + r < N;
  // out-of-range rows / k are read from a clamped (valid) address and zeroed by a select, so the
  // gathers of a chunk of 4 k-steps issue back to back instead of one branch + wait per element
  const float* ap = Ap + (mok ? m0 + r : M - 1) * a_sm;
  const float* bp = Bp + (nok ? n0 + r : N - 1) * b_sn;
  for (int ks = k0; ks < k1; ks += 16) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = ks + 4 * u + g;
      const bool kok = kk < k1;
      const int kc = kok ? kk : k1 - 1;
      const float av = ap[kc * a_sk], bv = bp[kc * b_sk];
      a[u] = (mok && kok) ? av : 0.f;
      b[u] = (nok && kok) ? bv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ks + 4 * u < k1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
  }
  return acc;
}

// G(m0 + rr, n) += acc[rr], rr = 0..3 (m < M, n < N): the four old values are loaded first
__device__ __forceinline__ void acc_tile(float* G, int s_m, int s_n, int M, int N, int m0, int n, f4 acc) {
  const bool nok = n < N;
  const int nc = nok ? n : N - 1;
  float* p[4];
  float old[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    p[rr] = G + (m0 + rr < M ? m0 + rr : M - 1) * s_m + nc * s_n;
    old[rr] = *p[rr];
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
    if (nok && m0 + rr < M) *p[rr] = old[rr] + acc[rr];
}

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// 4 consecutive entries of an unaligned vector of length n (indices clamped: the tail repeats the last one)
__device__ __forceinline__ f4 ld4c(const float* p, int i0, int n) {
  return f4{p[i0 < n ? i0 : n - 1], p[i0 + 1 < n ? i0 + 1 : n - 1], p[i0 + 2 < n ? i0 + 2 : n - 1],
            p[i0 + 3 < n ? i0 + 3 : n - 1]};
}

// t = a * nb + b with 0 <= b < nb, for tiny a: avoids a ~40-instruction runtime integer division per tile
__device__ __forceinline__ void split_idx(int t, int nb, int& a, int& b) {
  a = 0;
  b = t;
  while (b >= nb) { b -= nb; ++a; }
}

struct TCtx {
  int tid, lane, wave, g, r;       // g = lane >> 4, r = lane & 15
  int d, H, T, ldd, ldh, tD, tH;   // tD, tH = 16-tiles over d and H
  NetOff o;
  float* md;                       // (N_MD, 16, ldd)
  float* mh;                       // (N_MH, 16, ldh)
  float* cs;                       // per-chain scalars [slot][16]
  const float *Msk, *Trg;
  int it;                          // trajectory step being processed
  int c, kb;                       // this thread's chain (tid & 15) and first dim / hidden unit (tid >> 4)
  float t0, t1;                    // time encoding of this thread's chain at step `it`
  __device__ __forceinline__ float* D(int m) const { return md + m * TC * ldd; }
  __device__ __forceinline__ float* Hm(int m) const { return mh + m * TC * ldh; }
  __device__ __forceinline__ bool fwd(int c_) const { return cs[CS_SGN * TC + c_] > 0.f; }
  __device__ __forceinline__ int step_of(int c_) const { return fwd(c_) ? it : (T - 1 - it); }
};

// ---- [ts, T, tq] caches and h1, h2 of net W at inputs (a, b, tau): 4 phases ---------------------------
// (t_net_fwd / t_net_bwd are force-inlined: with the default heuristic the compiler inlined one and
//  called the other, and that build produced wrong gradients and intermittent hangs on gfx950,
//  while the all-inlined and the all-outlined builds of the same source are both correct)
__device__ __forceinline__ void t_net_fwd(const TCtx& X, const float* W, const float* a, const float* b) {
  const NetOff& o = X.o;
  const int d = X.d, H = X.H, ldd = X.ldd, ldh = X.ldh, lane = X.lane;
  {  // layer 1, K split over the 4 waves: waves 0,1 take W1^T a (two halves of k), waves 2,3 W2^T b
    const int w = X.wave;
    const float* in = (w < 2) ? a : b;
    const float* Wt = W + ((w < 2) ? o.W1 : o.W2);
    const int kh = (d + 7) / 8 * 4;
    const int k0 = (w & 1) ? kh : 0, k1 = (w & 1) ? d : (kh < d ? kh : d);
    float* part = X.Hm(MPART + w);
    for (int tm = 0; tm < X.tH; ++tm) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mm_tile(acc, Wt, 1, H, H, 16 * tm, in, 1, ldd, TC, 0, k0, k1, lane);
      const int i0 = 16 * tm + 4 * X.g;
      if (i0 < ldh) *reinterpret_cast<f4*>(part + X.r * ldh + i0) = acc;
    }
  }
  __syncthreads();
  TT_MARK(1);
  {
    float* h1 = X.Hm(MH1);
    const float *p0 = X.Hm(MPART), *p1 = X.Hm(MPART + 1), *p2 = X.Hm(MPART + 2), *p3 = X.Hm(MPART + 3);
    for (int i = X.kb; i < H; i += 16) {
      const int q = X.c * ldh + i;
      const float acc = (W[o.b1 + i] + W[o.b2 + i]) + W[o.b3 + i] + X.t0 * W[o.W3 + i] + X.t1 * W[o.W3 + H + i];
      h1[q] = fmaxf(acc + ((p0[q] + p1[q]) + (p2[q] + p3[q])), 0.f);
    }
  }
  __syncthreads();
  TT_MARK(2);
  {  // layer 2
    const float* h1 = X.Hm(MH1);
    float* h2 = X.Hm(MH2);
    for (int tm = X.wave; tm < X.tH; tm += TNW) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const int j0 = 16 * tm + 4 * X.g;
      const f4 bias = ld4c(W + o.b4, j0, H);
      acc = mm_tile(acc, W + o.W4, 1, H, H, 16 * tm, h1, 1, ldh, TC, 0, 0, H, lane);
      if (j0 < ldh) st4(h2 + X.r * ldh + j0, relu4(acc + bias));     // (pad columns: never read)
    }
  }
  __syncthreads();
  TT_MARK(3);
  {  // heads
    const float* h2 = X.Hm(MH2);
    for (int t = X.wave; t < 3 * X.tD; t += TNW) {
      int head, tm;
      split_idx(t, X.tD, head, tm);
      // (offsets by arithmetic: the three heads are laid out [Ws bs Wt bt Wq bq]; a 3-way select of
      //  runtime values would be turned into a lookup table in scratch memory)
      const int hs = H * d + d;
      const float* Wh = W + o.Ws + head * hs;
      const int bo = o.bs + head * hs;
      float* out = X.D(MTS + head);
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const int k0 = 16 * tm + 4 * X.g;
      const f4 bias = ld4c(W + bo, k0, d);
      acc = mm_tile(acc, Wh, 1, d, d, 16 * tm, h2, 1, ldh, TC, 0, 0, H, lane);
      const f4 z = acc + bias;
      if (k0 < ldd) st4(out + X.r * ldd + k0, head == 1 ? z : tanh4(z));
    }
  }
  __syncthreads();
  TT_MARK(4);
}

// ---- reverse of t_net_fwd: consumes dS, dT, dQ (MDS, MDT, MDQ), accumulates parameter gradients into
// G, leaves the input adjoints in MDA (w.r.t. a) and MDB (w.r.t. b): 4 phases -------------------------------
__device__ __forceinline__ void t_net_bwd(const TCtx& X, const float* W, float* G, const float* a, const float* b) {
  const NetOff& o = X.o;
  const int d = X.d, H = X.H, ldd = X.ldd, ldh = X.ldh, lane = X.lane;
  // on entry (written by the *_half_bwd phase): MDS, MDT, MDQ = d zs, d zt, d zq;  MDA = dS * S, MDB = dQ * Q
  {
    const float *h2 = X.Hm(MH2);
    // head weight gradients  dWh(j, k) += sum_c h2(c, j) dz_h(c, k)
    for (int t = X.wave; t < 3 * X.tH * X.tD; t += TNW) {
      int head, u, tm, tn;
      split_idx(t, X.tH * X.tD, head, u);
      split_idx(u, X.tD, tm, tn);
      const float* dz = X.D(MDS + head);
      float* Gh = G + o.Ws + head * (H * d + d);
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mm_tile(acc, h2, 1, ldh, H, 16 * tm, dz, ldd, 1, d, 16 * tn, 0, TC, lane);
      acc_tile(Gh, d, 1, H, d, 16 * tm + 4 * X.g, 16 * tn + X.r, acc);
    }
    // d h2 partial of head w:  part_w(c, j) = sum_k Wh(j, k) dz_h(c, k)      (wave 3: zero)
    {
      const int w = X.wave;
      const float* dz = X.D(MDS + (w < 3 ? w : 0));
      const float* Wh = W + o.Ws + (w < 3 ? w : 0) * (H * d + d);
      float* part = X.Hm(MPART + w);
      for (int tm = 0; tm < X.tH; ++tm) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        if (w < 3) acc = mm_tile(acc, Wh, d, 1, H, 16 * tm, dz, 1, ldd, TC, 0, 0, d, lane);
        const int j0 = 16 * tm + 4 * X.g;
        if (j0 < ldh) *reinterpret_cast<f4*>(part + X.r * ldh + j0) = acc;
      }
    }
    // column sums over the chains: bs, bt, bq, lam_s, lam_q
    for (int e = X.tid; e < 5 * d; e += TTHREADS) {
      int which, k;
      split_idx(e, d, which, k);
      const float* src = X.D(MDS + which);                     // MDS, MDT, MDQ, MDA, MDB are consecutive
      const int dst = which < 3 ? o.bs + which * (H * d + d) : o.ls + (which - 3) * d;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < TC; ++c) s += src[c * ldd + k];
      G[dst + k] += s;
    }
  }
  __syncthreads();
  TT_MARK(5);
  {
    const float* h2 = X.Hm(MH2);
    float* da2 = X.Hm(MDA2);
    const float *p0 = X.Hm(MPART), *p1 = X.Hm(MPART + 1), *p2 = X.Hm(MPART + 2);
    for (int j = X.kb; j < H; j += 16) {
      const int q = X.c * ldh + j;
      da2[q] = h2[q] > 0.f ? (p0[q] + p1[q]) + p2[q] : 0.f;
    }
  }
  __syncthreads();
  TT_MARK(6);
  {
    const float *h1 = X.Hm(MH1), *da2 = X.Hm(MDA2);
    float* da1 = X.Hm(MDA1);
    const int nW4 = X.tH * X.tH;
    for (int t = X.wave; t < nW4 + X.tH; t += TNW) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      if (t < nW4) {          // dW4(i, j) += sum_c h1(c, i) da2(c, j)
        int tm, tn;
        split_idx(t, X.tH, tm, tn);
        acc = mm_tile(acc, h1, 1, ldh, H, 16 * tm, da2, ldh, 1, H, 16 * tn, 0, TC, lane);
        acc_tile(G + o.W4, H, 1, H, H, 16 * tm + 4 * X.g, 16 * tn + X.r, acc);
      } else {                // da1(c, i) = [h1 > 0] sum_j W4(i, j) da2(c, j)
        const int tm = t - nW4;
        acc = mm_tile(acc, W + o.W4, H, 1, H, 16 * tm, da2, 1, ldh, TC, 0, 0, H, lane);
        const int i0 = 16 * tm + 4 * X.g;
        if (i0 < ldh) {
          const f4 hh = ld4(h1 + X.r * ldh + i0);
          st4(da1 + X.r * ldh + i0, f4{hh.x > 0.f ? acc.x : 0.f, hh.y > 0.f ? acc.y : 0.f,
                                       hh.z > 0.f ? acc.z : 0.f, hh.w > 0.f ? acc.w : 0.f});
        }
      }
    }
    for (int j = X.tid; j < H; j += TTHREADS) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < TC; ++c) s += da2[c * ldh + j];
      G[o.b4 + j] += s;
    }
  }
  __syncthreads();
  TT_MARK(7);
  {
    const float* da1 = X.Hm(MDA1);
    const int nWg = X.tD * X.tH;        // tiles of one layer-1 weight gradient
    for (int t = X.wave; t < 2 * nWg + 2 * X.tD; t += TNW) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      if (t < 2 * nWg) {      // dW1(k, i) += sum_c a(c, k) da1(c, i);  dW2 likewise with b
        int which, u, tm, tn;
        split_idx(t, nWg, which, u);
        split_idx(u, X.tH, tm, tn);
        acc = mm_tile(acc, which == 0 ? a : b, 1, ldd, d, 16 * tm, da1, ldh, 1, H, 16 * tn, 0, TC, lane);
        float* Gw = G + o.W1 + which * (d * H + H);
        acc_tile(Gw, H, 1, d, H, 16 * tm + 4 * X.g, 16 * tn + X.r, acc);
      } else {                // da(c, k) = sum_i W1(k, i) da1(c, i);  db with W2
        int which, tm;
        split_idx(t - 2 * nWg, X.tD, which, tm);
        acc = mm_tile(acc, W + o.W1 + which * (d * H + H), H, 1, d, 16 * tm, da1, 1, ldh, TC, 0, 0, H, lane);
        float* out = X.D(MDA + which);
        const int k0 = 16 * tm + 4 * X.g;
        if (k0 < ldd) st4(out + X.r * ldd + k0, acc);
      }
    }
    for (int i = X.tid; i < H; i += TTHREADS) {
      float s = 0.f, s0 = 0.f, s1 = 0.f;
      for (int c = 0; c < TC; ++c) {
        const float v = da1[c * ldh + i];
        const int st = X.step_of(c);
        s += v; s0 += X.Trg[2 * st] * v; s1 += X.Trg[2 * st + 1] * v;
      }
      G[o.b1 + i] += s; G[o.b2 + i] += s; G[o.b3 + i] += s;
      G[o.W3 + i] += s0; G[o.W3 + H + i] += s1;
    }
  }
  __syncthreads();
  TT_MARK(8);
}

__global__ __launch_bounds__(TTHREADS) void train_kernel(const TArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  const int tid = threadIdx.x;
  const int d = A.d, H = A.H, T = A.T;
  const int ek = A.ekind;
  const int nc = ek == L2HMC_ENERGY_GMM ? A.ncomp : 1;
  const TLayout L = train_layout(d, H, T, ek, nc);
  const int P = L.P, ldd = L.ldd;
  float *Wx = smem + L.Wx, *Wv = smem + L.Wv, *Gx = smem + L.Gx, *Gv = smem + L.Gv, *Ge = smem + L.Ge;
  float *Msk = smem + L.Msk, *Trg = smem + L.Trg, *Mu = smem + L.Mu, *Pr = smem + L.Pr, *Lc = smem + L.Lc;
  float *WV = smem + L.wv, *YV = smem + L.yv;
  const int dpad = (d + 3) / 4 * 4;
  float *ESx = smem + L.Ex, *EQx = ESx + dpad, *ESv = EQx + dpad, *EQv = ESv + dpad;
  TCtx X;
  X.tid = tid; X.lane = tid & 63; X.wave = __builtin_amdgcn_readfirstlane(tid >> 6); X.g = X.lane >> 4; X.r = X.lane & 15;
  X.d = d; X.H = H; X.T = T; X.ldd = ldd; X.ldh = L.ldh; X.tD = (d + 15) / 16; X.tH = (H + 15) / 16;
  X.o = net_off(d, H);
  X.md = smem + L.md; X.mh = smem + L.mh; X.cs = smem + L.cs; X.Msk = Msk; X.Trg = Trg; X.it = 0;
  X.c = tid & 15; X.kb = tid >> 4;
  const int c = X.c, kb = X.kb;
  const long long n = (long long)blockIdx.x * TC + c;          // this thread's chain
  const bool alive = n < A.N;
  const bool isf = A.dir != nullptr ? (alive ? A.dir[n] != 0 : true) : (A.dir_all != 0);
  const float sg = isf ? 1.f : -1.f;

  // ---- stage ------------------------------------------------------------------------------------
  {
    const NetOff& o = X.o;
    // (field by field: indexing the kernel-argument struct through a pointer table would pin the
    //  whole struct in scratch memory and turn every later argument read into a scratch load)
    auto stage_net = [&](const L2hmcNet& nw, float* W) {
      auto cp = [&](const float* src, int off, int cnt) {
        for (int i = tid; i < cnt; i += TTHREADS) W[off + i] = src[i];
      };
      cp(nw.W1, o.W1, d * H); cp(nw.b1, o.b1, H); cp(nw.W2, o.W2, d * H); cp(nw.b2, o.b2, H);
      cp(nw.W3, o.W3, 2 * H); cp(nw.b3, o.b3, H); cp(nw.W4, o.W4, H * H); cp(nw.b4, o.b4, H);
      cp(nw.Ws, o.Ws, H * d); cp(nw.bs, o.bs, d); cp(nw.Wt, o.Wt, H * d); cp(nw.bt, o.bt, d);
      cp(nw.Wq, o.Wq, H * d); cp(nw.bq, o.bq, d); cp(nw.lam_s, o.ls, d); cp(nw.lam_q, o.lq, d);
    };
    stage_net(A.xnet, Wx);
    stage_net(A.vnet, Wv);
    for (int i = tid; i < d; i += TTHREADS) {
      ESx[i] = expf(A.xnet.lam_s[i]); EQx[i] = expf(A.xnet.lam_q[i]);
      ESv[i] = expf(A.vnet.lam_s[i]); EQv[i] = expf(A.vnet.lam_q[i]);
    }
    for (int i = tid; i < 2 * P + 4; i += TTHREADS) Gx[i] = 0.f;
    for (int i = tid; i < T * d; i += TTHREADS) Msk[i] = A.masks[i];
    for (int i = tid; i < 2 * T; i += TTHREADS) Trg[i] = A.trig[i];
    if (ek != L2HMC_ENERGY_ROUGHWELL) {
      for (int i = tid; i < nc * d; i += TTHREADS) Mu[i] = A.mu[i];
      if (ek == L2HMC_ENERGY_GAUSS_DIAG) {
        for (int i = tid; i < d; i += TTHREADS) Pr[i] = A.prec[i];
      } else {                  // symmetric part G = (S + S^T) / 2 of each raw precision
        for (int i = tid; i < nc * d * d; i += TTHREADS) {
          const int cc = i / (d * d), u = i - cc * d * d, k = u / d, j = u - k * d;
          Pr[i] = 0.5f * (A.prec[cc * d * d + k * d + j] + A.prec[cc * d * d + j * d + k]);
        }
      }
    }
    if (ek == L2HMC_ENERGY_GMM)
      for (int i = tid; i < nc; i += TTHREADS) Lc[i] = A.logc[i];
    for (int i = tid; i < (N_MD * TC * ldd); i += TTHREADS) X.md[i] = 0.f;
    for (int i = tid; i < (N_MH * TC * L.ldh); i += TTHREADS) X.mh[i] = 0.f;
    if (kb == 0) X.cs[CS_SGN * TC + c] = sg;
  }
#ifdef L2HMC_TRAIN_TIMING
  if (blockIdx.x == 0 && threadIdx.x == 0) tt_t0 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  TT_MARK(0);

  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const float rw_den = A.den;
  const bool EL = ek == L2HMC_ENERGY_GAUSS_DIAG || ek == L2HMC_ENERGY_ROUGHWELL;   // elementwise grad / Hessian
  float deps = 0.f;           // this thread's share of d loss / d eps
  int s_me = 0;               // step index of this thread's chain at iteration X.it

  auto set_step = [&](int it) {
    X.it = it;
    s_me = isf ? it : (T - 1 - it);
    X.t0 = Trg[2 * s_me];
    X.t1 = Trg[2 * s_me + 1];
  };
  // k1 of dynamics.py:126,167: the mask of the chain's current step, complemented on the backward branch
  auto kin_of = [&](int k) {
    const float m = Msk[s_me * d + k];
    return isf ? m : 1.f - m;
  };
  auto g_elem = [&](float z, int k) {          // elementwise grad U
    return ek == L2HMC_ENERGY_GAUSS_DIAG ? Pr[k] * (z - Mu[k]) : z - (A.eta / rw_den) * sinf(z / rw_den);
  };
  auto h_elem = [&](float z, int k) {          // elementwise Hessian diagonal
    return ek == L2HMC_ENERGY_GAUSS_DIAG ? Pr[k] : 1.f - (A.eta / (rw_den * rw_den)) * cosf(z / rw_den);
  };
  // y_comp(k) = sum_j G_comp(k, j) (z_j - [center] mu_comp_j) for one chain row (dense / mixture kinds)
  auto matG = [&](int comp, const float* zrow, bool center, int k) {
    const float* mu = Mu + comp * d;
    const float* Grow = Pr + comp * d * d + k * d;
    float acc = 0.f;
    for (int j = 0; j < d; ++j) acc += Grow[j] * (zrow[j] - (center ? mu[j] : 0.f));
    return acc;
  };
  // mixture log-weights V_comp(z) of every chain -> WV (GMM only); ends with a barrier
  auto gmm_logw = [&](const float* z) {
    for (int comp = kb; comp < nc; comp += 16) {
      const float* zrow = z + c * ldd;
      float q = 0.f;
      for (int k = 0; k < d; ++k) q += (zrow[k] - Mu[comp * d + k]) * matG(comp, zrow, true, k);
      WV[c * KC + comp] = -0.5f * q + Lc[comp];
    }
    __syncthreads();
  };
  auto gmm_softmax = [&](int cc, float* w) {     // returns logsumexp
    float m = -INFINITY, sum = 0.f;
    for (int comp = 0; comp < nc; ++comp) m = fmaxf(m, WV[cc * KC + comp]);
    for (int comp = 0; comp < nc; ++comp) { w[comp] = expf(WV[cc * KC + comp] - m); sum += w[comp]; }
    for (int comp = 0; comp < nc; ++comp) w[comp] /= sum;
    return m + logf(sum);
  };
  // g <- grad U(z) for the dense / mixture kinds (oracle/l2hmc_train_oracle.py *Target.grad); ends with a barrier
  auto gradU_full = [&](const float* z, float* g) {
    if (ek == L2HMC_ENERGY_GMM) gmm_logw(z);
    const float* zrow = z + c * ldd;
    float w[KC];
    if (ek == L2HMC_ENERGY_GMM) gmm_softmax(c, w);
    for (int k = kb; k < d; k += 16) {
      float out = 0.f;
      if (ek == L2HMC_ENERGY_GMM) {
        for (int comp = 0; comp < nc; ++comp) out += w[comp] * matG(comp, zrow, true, k);
      } else {
        out = matG(0, zrow, true, k);
      }
      g[c * ldd + k] = out;
    }
    __syncthreads();
  };
  // U(z) of chain cc, given g = grad U(z) (and WV of the same z for the mixture); one thread per chain
  auto energyU = [&](const float* z, const float* g, int cc) {
    const float* zrow = z + cc * ldd;
    float u = 0.f;
    if (ek == L2HMC_ENERGY_ROUGHWELL) {
      for (int k = 0; k < d; ++k) u += 0.5f * zrow[k] * zrow[k] + A.eta * cosf(zrow[k] / rw_den);
    } else if (ek == L2HMC_ENERGY_GMM) {
      float w[KC];
      u = -gmm_softmax(cc, w);
    } else {
      for (int k = 0; k < d; ++k) u += 0.5f * (zrow[k] - Mu[k]) * g[cc * ldd + k];
    }
    return u;
  };
  // out <- Hessian(z) vec for the dense / mixture kinds (needs WV of the same z); ends with a barrier
  auto hessvec_full = [&](const float* z, const float* vec, float* out) {
    const float* zrow = z + c * ldd;
    const float* vrow = vec + c * ldd;
    if (ek == L2HMC_ENERGY_GMM) {
      for (int comp = kb; comp < nc; comp += 16) {
        float yv = 0.f;
        for (int k = 0; k < d; ++k) yv += matG(comp, zrow, true, k) * vrow[k];
        YV[c * KC + comp] = yv;
      }
      __syncthreads();
    }
    float w[KC];
    if (ek == L2HMC_ENERGY_GMM) gmm_softmax(c, w);
    for (int k = kb; k < d; k += 16) {
      float o = 0.f;
      if (ek == L2HMC_ENERGY_GMM) {
        float gz = 0.f, gv = 0.f;
        for (int comp = 0; comp < nc; ++comp) {
          const float y = matG(comp, zrow, true, k);
          gz += w[comp] * y;
          gv += w[comp] * YV[c * KC + comp];
          o += w[comp] * (matG(comp, vrow, false, k) - y * YV[c * KC + comp]);
        }
        o += gz * gv;
      } else {
        o = matG(0, vrow, false, k);
      }
      out[c * ldd + k] = o;
    }
    __syncthreads();
  };

  float *mx = X.D(MX), *mv = X.D(MV), *mvh = X.D(MVH), *my = X.D(MY), *mxo = X.D(MXO), *mg = X.D(MG);
  float *lx = X.D(MLX), *lv = X.D(MLV), *dvh = X.D(MDVH), *dz = X.D(MDZ), *tmp = X.D(MTMP);
  float *dS = X.D(MDS), *dT = X.D(MDT), *dQ = X.D(MDQ), *dA = X.D(MDA), *dB = X.D(MDB), *dg = X.D(MDG);
  const float *cTS = X.D(MTS), *cT = X.D(MT), *cTQ = X.D(MTQ);
  float* ldm = lx;              // log-det terms of the forward trajectory (lx is free until the seeds)
  auto ckpt = [&](int t, int slot) { return A.ws + (((long long)t * A.N + n) * CKPT + slot) * d; };
#define EW_BEGIN for (int k = kb; k < d; k += 16) { const int q = c * ldd + k;
#define EW_END } __syncthreads(); TT_MARK(9);

  // v_half (dynamics.py:129-141 / 183-196): out = v_half(vin; g, V-net caches), logging the log-det;
  // `also_tmp`: tmp <- k1 x, the X-net's second input of the next stage
  auto v_half_fwd = [&](const float* vin, float* out, bool also_tmp) {
    EW_BEGIN
      const float S = ESv[k] * cTS[q], Q = EQv[k] * cTQ[q];
      const float ES = fexp(sg * heps * S), EQ = fexp(eps * Q);
      const float cc = heps * (cT[q] - EQ * mg[q]);
      out[q] = isf ? vin[q] * ES + cc : (vin[q] - cc) * ES;
      ldm[q] += sg * heps * S;
      if (also_tmp) tmp[q] = kin_of(k) * mx[q];
    EW_END
  };
  // x_half (dynamics.py:143-161 / 170-181): out = kp zin + (1 - kp) x_update(zin; vh, X-net caches);
  // first: also tmp <- k2 y;  second: also the step's checkpoints and (elementwise kinds) g <- grad U(x')
  auto x_half_fwd = [&](const float* zin, bool first, float* out, int it) {
    EW_BEGIN
      const float kin = kin_of(k), kp = first ? kin : 1.f - kin;
      const float S = ESx[k] * cTS[q], Q = EQx[k] * cTQ[q];
      const float ES = fexp(sg * eps * S), EQ = fexp(eps * Q);
      const float tr = eps * (EQ * mvh[q] + cT[q]);
      const float nw = isf ? zin[q] * ES + tr : ES * (zin[q] - tr);
      const float o = kp * zin[q] + (1.f - kp) * nw;
      out[q] = o;
      ldm[q] += (1.f - kp) * sg * eps * S;
      if (first) {
        tmp[q] = (1.f - kin) * o;
      } else {
        if (alive) {
          ckpt(it, 0)[k] = mx[q]; ckpt(it, 1)[k] = mv[q]; ckpt(it, 2)[k] = mvh[q];
          ckpt(it, 3)[k] = my[q]; ckpt(it, 4)[k] = o;
        }
        if (EL) mg[q] = g_elem(o, k);
      }
    EW_END
  };

  // ---- load the start state -----------------------------------------------------------------------
  set_step(0);
  EW_BEGIN
    mx[q] = alive ? x_row0(A, n)[n * d + k] : 0.f;
    mv[q] = alive ? A.v[n * d + k] : 0.f;
    ldm[q] = 0.f;
    if (EL) mg[q] = g_elem(mx[q], k);
  EW_END
  if (!EL) gradU_full(mx, mg);
  if (tid < TC) {
    float K0 = 0.f;
    for (int k = 0; k < d; ++k) K0 += 0.5f * mv[tid * ldd + k] * mv[tid * ldd + k];
    X.cs[CS_K0 * TC + tid] = K0;
    X.cs[CS_U0 * TC + tid] = energyU(mx, mg, tid);
  }
  __syncthreads();

  // ---- forward trajectory with checkpoints --------------------------------------------------------
  for (int it = 0; it < T; ++it) {
    set_step(it);
    t_net_fwd(X, Wv, mx, mg);
    v_half_fwd(mv, mvh, true);
    t_net_fwd(X, Wx, mvh, tmp);
    x_half_fwd(mx, true, my, it);
    t_net_fwd(X, Wx, mvh, tmp);
    x_half_fwd(my, false, mxo, it);
    if (!EL) gradU_full(mxo, mg);
    t_net_fwd(X, Wv, mxo, mg);
    v_half_fwd(mvh, mv, false);
    { float* t = mx; mx = mxo; mxo = t; }            // x <- x'
  }

  // ---- accept probability, loss term and the adjoint seeds ------------------------------------------
  // (mg = grad U at the end point already: it was the V-net's input of the last half step)
  if (tid < TC) {
    const int cc = tid;
    const long long nn = (long long)blockIdx.x * TC + cc;
    const bool lv_ = nn < A.N;
    float K1 = 0.f, sq = 0.f, ld = 0.f;
    for (int k = 0; k < d; ++k) {
      const float x0 = lv_ ? x_row0(A, nn)[nn * d + k] : 0.f;
      K1 += 0.5f * mv[cc * ldd + k] * mv[cc * ldd + k];
      sq += (x0 - mx[cc * ldd + k]) * (x0 - mx[cc * ldd + k]);
      ld += ldm[cc * ldd + k];
    }
    const float U1 = energyU(mx, mg, cc);
    const float val = (X.cs[CS_U0 * TC + cc] + X.cs[CS_K0 * TC + cc]) - (U1 + K1) + ld;
    const float p = accept_prob(val);
    const float v1 = sq * p + 1e-4f;
    if (lv_) { A.p[nn] = p; A.v1[nn] = v1; }
    const float dv1 = lv_ ? (A.scale * (-1.f / (v1 * v1)) - 1.f / A.scale) * A.inv_n : 0.f;
    const bool pfin = (val == val) && p > 0.f;   // finite branch of dynamics.py:309 actually taken
    // (a diverged chain -- non-finite end point -- has p = 0 and contributes no gradient, instead of
    //  the reference's 0 * NaN)
    X.cs[CS_LAM * TC + cc] = (pfin && val < 0.f) ? dv1 * sq * p : 0.f;
    X.cs[CS_DV1P * TC + cc] = dv1 * p * 2.f;
    X.cs[CS_OK * TC + cc] = sq < 3.0e38f ? 1.f : 0.f;
  }
  __syncthreads();
  const float lam = X.cs[CS_LAM * TC + c];
  {
    const bool okc = X.cs[CS_OK * TC + c] != 0.f;
    const float dv1p = X.cs[CS_DV1P * TC + c];
    EW_BEGIN
      const float x0 = alive ? x_row0(A, n)[n * d + k] : 0.f;
      if (alive) A.Lx[n * d + k] = mx[q];
      lx[q] = okc ? dv1p * (mx[q] - x0) - lam * mg[q] : 0.f;
      lv[q] = okc ? -lam * mv[q] : 0.f;
    EW_END
  }

  // ---- reverse sweep ------------------------------------------------------------------------------------
  // adjoint of out = v_half(vin; g, caches): d vin -> dvin_out; leaves d zs, d zt, d zq, the lam-scale terms
  // (MDA, MDB) for t_net_bwd, and dg
  auto v_half_bwd = [&](const float* dout_m, const float* vin, float* dvin_out) {
    EW_BEGIN
      const float ts = cTS[q], tq = cTQ[q], Tt = cT[q];
      const float S = ESv[k] * ts, Q = EQv[k] * tq;
      const float ES = fexp(sg * heps * S), EQ = fexp(eps * Q);
      const float gq = mg[q];
      const float cc = heps * (Tt - EQ * gq);
      const float dout = dout_m[q];
      const float dES = isf ? dout * vin[q] : dout * (vin[q] - cc);
      const float dcc = isf ? dout : -dout * ES;
      const float ds = dES * ES + lam;
      const float dSr = ds * sg * heps;                 // d S
      const float dq = -dcc * heps * gq * EQ;
      const float dQr = dq * eps;                       // d Q
      dvin_out[q] = dout * ES;
      dg[q] = -dcc * heps * EQ;
      dA[q] = dSr * S;
      dB[q] = dQr * Q;
      dS[q] = dSr * ESv[k] * (1.f - ts * ts);
      dT[q] = dcc * heps;
      dQ[q] = dQr * EQv[k] * (1.f - tq * tq);
      deps += ds * sg * 0.5f * S + dcc * 0.5f * (Tt - EQ * gq) + dq * Q;
    EW_END
  };
  // adjoint of out = x_half(zin, first; vh, caches): d zin (direct part) -> dzin_out, dvh +=, same outputs
  auto x_half_bwd = [&](const float* dout_m, const float* zin, bool first, float* dzin_out) {
    EW_BEGIN
      const float kin = kin_of(k), kp = first ? kin : 1.f - kin, up = 1.f - kp;
      const float ts = cTS[q], tq = cTQ[q], Tt = cT[q];
      const float S = ESx[k] * ts, Q = EQx[k] * tq;
      const float ES = fexp(sg * eps * S), EQ = fexp(eps * Q);
      const float vhq = mvh[q];
      const float tr = eps * (EQ * vhq + Tt);
      const float dnw = up * dout_m[q];
      const float dES = isf ? dnw * zin[q] : dnw * (zin[q] - tr);
      const float dtr = isf ? dnw : -dnw * ES;
      const float dsx = dES * ES + up * lam;
      const float dSr = dsx * sg * eps;
      const float dq = dtr * eps * vhq * EQ;
      const float dQr = dq * eps;
      dzin_out[q] = kp * dout_m[q] + dnw * ES;
      dvh[q] += dtr * eps * EQ;
      dA[q] = dSr * S;
      dB[q] = dQr * Q;
      dS[q] = dSr * ESx[k] * (1.f - ts * ts);
      dT[q] = dtr * eps;
      dQ[q] = dQr * EQx[k] * (1.f - tq * tq);
      deps += dsx * sg * S + dtr * (EQ * vhq + Tt) + dq * Q;
    EW_END
  };
  // lx += da + Hessian(z) (dg + db): the V-net's inputs were (z, grad U(z))
  auto through_grad = [&](const float* z) {
    if (EL) {
      EW_BEGIN
        lx[q] = lx[q] + dA[q] + h_elem(z[q], k) * (dg[q] + dB[q]);
      EW_END
    } else {
      EW_BEGIN
        dg[q] += dB[q];
      EW_END
      hessvec_full(z, dg, dz);
      EW_BEGIN
        lx[q] = lx[q] + dA[q] + dz[q];
      EW_END
    }
  };

  for (int it = T - 1; it >= 0; --it) {
    set_step(it);
    EW_BEGIN
      mx[q] = alive ? ckpt(it, 0)[k] : 0.f; mv[q] = alive ? ckpt(it, 1)[k] : 0.f;
      mvh[q] = alive ? ckpt(it, 2)[k] : 0.f; my[q] = alive ? ckpt(it, 3)[k] : 0.f;
      mxo[q] = alive ? ckpt(it, 4)[k] : 0.f;
      tmp[q] = (1.f - kin_of(k)) * my[q];                    // X-net input of stage (2)
      if (EL) mg[q] = g_elem(mxo[q], k);
    EW_END
    // (1) v' = v_half(vh; g(x'), V(x', g(x')))
    if (!EL) gradU_full(mxo, mg);
    t_net_fwd(X, Wv, mxo, mg);
    v_half_bwd(lv, mvh, dvh);
    t_net_bwd(X, Wv, Gv, mxo, mg);
    through_grad(mxo);                                       // lx = d x'
    // (2) x' = x_half(y, k2; vh, X(vh, k2 y)),  k2 = 1 - k1
    t_net_fwd(X, Wx, mvh, tmp);
    x_half_bwd(lx, my, false, dz);                           // dz = d y (direct part)
    t_net_bwd(X, Wx, Gx, mvh, tmp);
    EW_BEGIN
      const float kin = kin_of(k);
      dvh[q] += dA[q];
      dz[q] += (1.f - kin) * dB[q];
      tmp[q] = kin * mx[q];                                  // X-net input of stage (3)
    EW_END
    // (3) y = x_half(x, k1; vh, X(vh, k1 x))
    t_net_fwd(X, Wx, mvh, tmp);
    x_half_bwd(dz, mx, true, lx);                            // lx = d x (direct part)
    t_net_bwd(X, Wx, Gx, mvh, tmp);
    EW_BEGIN
      dvh[q] += dA[q];
      lx[q] += kin_of(k) * dB[q];
      if (EL) mg[q] = g_elem(mx[q], k);
    EW_END
    // (4) vh = v_half(v; g(x), V(x, g(x)))
    if (!EL) gradU_full(mx, mg);
    t_net_fwd(X, Wv, mx, mg);
    v_half_bwd(dvh, mv, lv);
    t_net_bwd(X, Wv, Gv, mx, mg);
    through_grad(mx);
  }
#undef EW_BEGIN
#undef EW_END
  TT_MARK(10);
  {
    // d loss / d eps: wave sums combined in a fixed order (no atomics anywhere in this kernel: the
    // gradient is bitwise reproducible from run to run)
    const float s = wave_sum(deps);
    if (X.lane == 0) X.cs[CS_U1 * TC + X.wave] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int w = 0; w < TNW; ++w) s += X.cs[CS_U1 * TC + w];
    Ge[0] = s;
  }
  __syncthreads();
  // this workgroup's flat gradient [xnet (P) | vnet (P) | eps] -> its slot of the workspace; train_reduce_kernel
  // adds the slots to the caller's buffer in block order
  float* slot = A.ws + (long long)T * A.N * CKPT * d + (long long)blockIdx.x * (2 * P + 1);
  for (int i = tid; i < 2 * P + 1; i += TTHREADS) slot[i] = Gx[i];
}

// Fixed-order sum of per-workgroup partial gradients: thread (i, c) adds the slots [c * chunk, (c + 1) * chunk) of
// parameter i in slot order; `accumulate`: dst[i] += sum (the caller's buffer), else dst[c * n_grad + i] = sum.
constexpr int kReduceChunk = 32;
__global__ void train_reduce_kernel(const float* part, int n_slots, int n_grad, int chunk, float* dst, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (i >= n_grad) return;
  const int b0 = c * chunk, b1 = (b0 + chunk < n_slots) ? b0 + chunk : n_slots;
  float s = 0.f;
  // (the loads of eight slots are in flight together; the additions stay in slot order)
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[(long long)(b + u) * n_grad + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; b < b1; ++b) s += part[(long long)b * n_grad + i];
  if (accumulate) dst[i] += s; else dst[(long long)c * n_grad + i] = s;
}

// The LAST level of the slot reduction with the rest of an optimiser step behind it (l2hmc_train_step): thread i adds the
// slots of parameter i in slot order, stores the sum (overwrite) and, when the optimiser's buffers are given, applies Adam to
// that parameter at once; one extra workgroup reduces the loss terms of v1 in the fixed order of l2hmc_loss_terms.
struct FinalArgs {
  const float* v1; long long n_v1; float scale; double inv_n; long long n_head;
  float* terms; double* loss;
  const float *x0, *Lx, *p, *u; float* x_next; int d;        // the Metropolis select of chains [0, n_head) (sampler.py:53-55)
  float *theta, *m, *v; float lr_t, b1, b2, eps; int last_is_log_eps; int n_par;
};
__device__ __forceinline__ void adam_update(float* p, float gi, float* m, float* v, long long i, long long n, float lr_t,
                                            float b1, float b2, float eps, int last_is_log_eps) {
  if (last_is_log_eps && i == n - 1) gi *= expf(p[i]);      // d/d alpha = eps d/d eps (dynamics.py:50-58)
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}
__device__ __forceinline__ void split_hi_lo(double t, float* out) {
  const float hi = (float)t;
  out[0] = hi;
  out[1] = (float)(t - (double)hi);
}
__global__ __launch_bounds__(256) void train_final_kernel(const float* part, int n_slots, int n_grad, float* dst, FinalArgs f) {
  const int nb = (n_grad + 255) / 256;
  if ((int)blockIdx.x > nb) {                  // the Metropolis-select blocks: x_next = (p - u >= 0) ? Lx : x, thread = (chain, dim)
    const long long i = (long long)(blockIdx.x - nb - 1) * 256 + threadIdx.x;
    if (i < f.n_head * f.d) {
      const long long n = i / f.d;
      f.x_next[i] = (f.p[n] - f.u[n] >= 0.f) ? f.Lx[i] : f.x0[i];
    }
    return;
  }
  if ((int)blockIdx.x == nb) {                 // the loss block
    __shared__ double sa[4], sb[4];
    double a = 0.0, b = 0.0;
    for (long long i = threadIdx.x; i < f.n_v1; i += 256) {
      const float v = f.v1[i];
      a += 1.0 / (double)v;
      b += (double)v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a += __shfl_xor(a, off);
      b += __shfl_xor(b, off);
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double A_ = ((sa[0] + sa[1]) + sa[2]) + sa[3], B_ = ((sb[0] + sb[1]) + sb[2]) + sb[3];
      if (f.loss != nullptr) {
        f.loss[0] = A_; f.loss[1] = B_;
        f.loss[2] = f.inv_n * ((double)f.scale * A_ - B_ / (double)f.scale);
      }
      if (f.terms != nullptr) {
        split_hi_lo(A_, f.terms);
        split_hi_lo(B_, f.terms + 2);
        split_hi_lo((double)f.n_head, f.terms + 4);
      }
    }
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_grad) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= n_slots; b += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[(long long)(b + u) * n_grad + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; b < n_slots; ++b) s += part[(long long)b * n_grad + i];
  dst[i] = s;
  if (f.theta != nullptr && i < f.n_par) adam_update(f.theta, s, f.m, f.v, i, f.n_par, f.lr_t, f.b1, f.b2, f.eps, f.last_is_log_eps);
}
__device__ __forceinline__ float relu_f(float a) { return fmaxf(a, 0.f); }
// Phase timers of profiling builds (-DL2HMC_TRAIN_TIMING, tools/train_phase_timing.py): accumulated in REGISTERS and written
// once at the end (a global read-modify-write per mark would cost more than the phases it brackets).
#ifdef L2HMC_TRAIN_TIMING
#define TS_DECL unsigned long long ts_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ts_t0 = __builtin_amdgcn_s_memtime()
#define TS_MARK(i)                                                          \
  do {                                                                      \
    const unsigned long long ts_t1 = __builtin_amdgcn_s_memtime();          \
    ts_acc[i] += ts_t1 - ts_t0;                                             \
    ts_t0 = ts_t1;                                                          \
  } while (0)
#define TS_FLUSH()                                                          \
  do {                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                \
      for (int ts_i = 0; ts_i < 9; ++ts_i) tt_acc[ts_i] += ts_acc[ts_i];    \
  } while (0)
#else
#define TS_DECL
#define TS_MARK(i)
#define TS_FLUSH()
#endif
#include "train_fast.hpp"
#include "train_small.hpp"

// d <= 4 targets (the notebook's own SCG-2D training among them): one dimension per lane (train_small.hpp)
inline bool train_small_ok(int ek, int d, int H) {
  return d <= 4 && H <= 15 && (ek == L2HMC_ENERGY_GAUSS_DIAG || ek == L2HMC_ENERGY_GAUSS_DENSE || ek == L2HMC_ENERGY_ROUGHWELL ||
                               ek == L2HMC_ENERGY_GMM);
}
template <int EK>
int launch_train_small(const TArgs& k, int KH, unsigned blocks, long long lds, hipStream_t s) {
  if (KH <= 3) hipLaunchKernelGGL((train_small_kernel<EK, 3>), dim3(blocks), dim3(128), (size_t)lds, s, k);
  else hipLaunchKernelGGL((train_small_kernel<EK, 4>), dim3(blocks), dim3(128), (size_t)lds, s, k);
  return L2HMC_OK;
}

// geometry of the register-resident kernel for this problem, or 0 if it stays on train_kernel
inline int train_fast_waves(int ek, int d, int H) {
  if (H > 15) return 0;
  if (ek == L2HMC_ENERGY_GAUSS_DIAG || ek == L2HMC_ENERGY_ROUGHWELL) return d <= 16 ? 1 : (d <= 64 ? 4 : 0);
  if (ek == L2HMC_ENERGY_GAUSS_DENSE || ek == L2HMC_ENERGY_FUNNEL) return d <= 16 ? 1 : 0;
  return 0;
}

template <int EK, int NW>
int launch_train_fast(const TArgs& k, int KH, unsigned blocks, long long lds, hipStream_t s) {
  auto go = [&](auto kern) -> int {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), (size_t)lds, s, k);
    return L2HMC_OK;
  };
  if (KH <= 3) return go(train_fast_kernel<EK, NW, 3>);
  return go(train_fast_kernel<EK, NW, 4>);
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" {

#ifdef L2HMC_TRAIN_TIMING
// copies the 16 phase accumulators to host and clears them
void l2hmc_train_read_timers(unsigned long long* out) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(tt_acc), sizeof(unsigned long long) * 16);
  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(tt_acc), z, sizeof(z));
}
#endif

int64_t l2hmc_train_workspace_floats(int64_t n_chains, int32_t d, int32_t H, int32_t T) {
  if (n_chains < 0 || d < 1 || H < 1 || T < 1) return fail(L2HMC_ERR_ARG, "l2hmc_train_workspace_floats: bad argument%s");
  // per-step checkpoints of every chain, then one flat partial gradient per 16-chain workgroup
  const int64_t blocks = (n_chains + TC - 1) / TC, chunks = (blocks + kReduceChunk - 1) / kReduceChunk;
  // checkpoints: (T, N, 5, d) for train_kernel; (blocks, T, 5, NW x 64 lanes) float4 for train_fast_kernel
  const int64_t ck_old = (int64_t)T * n_chains * CKPT * d, ck_fast = blocks * T * TF_CK * (4 * 256);
  return (ck_old > ck_fast ? ck_old : ck_fast) + (blocks + chunks) * (2LL * net_params(d, H) + 1);
}

int64_t l2hmc_train_grad_floats(int32_t d, int32_t H) {
  if (d < 1 || H < 1) return fail(L2HMC_ERR_ARG, "l2hmc_train_grad_floats: bad argument%s");
  return 2LL * net_params(d, H) + 1;
}

int64_t l2hmc_train_fused_lds_bytes(int32_t ek, int32_t n_comp, int32_t d, int32_t H, int32_t T) {
  if (d < 1 || H < 1 || T < 1) return fail(L2HMC_ERR_ARG, "l2hmc_train_fused_lds_bytes: bad argument%s");
  if (ek != L2HMC_ENERGY_GAUSS_DIAG && ek != L2HMC_ENERGY_GAUSS_DENSE && ek != L2HMC_ENERGY_GMM &&
      ek != L2HMC_ENERGY_ROUGHWELL && ek != L2HMC_ENERGY_FUNNEL)
    return fail(L2HMC_ERR_UNSUPPORTED, "no fused training kernel for this energy kind%s");
  if (d > 4096 || H > 4096) return fail(L2HMC_ERR_UNSUPPORTED, "fused training kernel: d / H too large%s");
  const long long lds_small = 4LL * ts_layout(T).total;
  if (train_small_ok(ek, d, H) && lds_small <= 48 * 1024) return lds_small;
  const int fnw = train_fast_waves(ek, d, H);
  const long long lds_fast = fnw ? 4LL * tf_layout(T, fnw).total : 0;
  if (fnw && lds_fast <= 160 * 1024) return lds_fast;
  if (ek == L2HMC_ENERGY_FUNNEL) return fail(L2HMC_ERR_UNSUPPORTED, "funnel: the fused trainer holds 2 <= d <= 16, H <= 15%s");
  const long long lds = 4LL * train_layout(d, H, T, ek, ek == L2HMC_ENERGY_GMM ? (n_comp < 1 ? 1 : n_comp) : 1).total;
  if (lds > 160 * 1024)
    return fail(L2HMC_ERR_UNSUPPORTED, "fused training kernel needs %s%lld bytes of LDS (> 160 KiB)", "", lds);
  return lds;
}

static int train_launch(const L2hmcTrainArgs* a, const L2hmcTrainStep* st, void* stream) {
  if (!a) return fail(L2HMC_ERR_ARG, "args is NULL%s");
  if (st != nullptr) {
    if (st->n_head < 0 || st->n_head > a->n_chains || (st->n_head > 0 && !st->x_head))
      return fail(L2HMC_ERR_ARG, "l2hmc_train_step: bad x_head / n_head%s");
    if ((st->u != nullptr) != (st->x_next != nullptr)) return fail(L2HMC_ERR_ARG, "l2hmc_train_step: u and x_next go together%s");
    if (st->u != nullptr && st->n_head < 1) return fail(L2HMC_ERR_ARG, "l2hmc_train_step: the Metropolis select needs n_head >= 1%s");
    if (st->theta != nullptr && (!st->m || !st->v || st->step < 1 || !(st->lr >= 0.f) || !(st->beta1 >= 0.f && st->beta1 < 1.f) ||
                                 !(st->beta2 >= 0.f && st->beta2 < 1.f) || !(st->epsilon > 0.f)))
      return fail(L2HMC_ERR_ARG, "l2hmc_train_step: bad optimiser arguments%s");
  }
  if (a->n_chains < 0 || a->d < 1 || a->T < 1 || a->H < 1) return fail(L2HMC_ERR_ARG, "bad n_chains / d / H / T%s");
  if (a->n_chains == 0) {
    if (st == nullptr) return L2HMC_OK;
    // An EMPTY shard of a sharded optimiser step (fewer chains than ranks): its slice of the one all-reduce must still be a
    // zero gradient and zero loss terms -- the reduction overwrites its destination, nothing zeroes it beforehand (round 4's
    // early return left the previous step's already-reduced gradient there, which every rank then added again).
    if (st->theta != nullptr) return fail(L2HMC_ERR_ARG, "l2hmc_train_step: an optimiser update needs at least one chain%s");
    if (!a->grad || !(a->scale > 0.f) || !(a->inv_n > 0.f)) return fail(L2HMC_ERR_ARG, "l2hmc_train_step: NULL grad / bad scale, inv_n%s");
    const int n_grad0 = 2 * net_params(a->d, a->H) + 1;
    FinalArgs f;
    memset(&f, 0, sizeof(f));
    f.scale = a->scale; f.inv_n = (double)a->inv_n; f.terms = st->terms; f.loss = st->loss;
    hipLaunchKernelGGL(train_final_kernel, dim3((unsigned)((n_grad0 + 255) / 256 + 1)), dim3(256), 0, (hipStream_t)stream, a->grad, 0,
                       n_grad0, a->grad, f);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e0));
    return L2HMC_OK;
  }
  if (a->d > 4096 || a->H > 4096) return fail(L2HMC_ERR_UNSUPPORTED, "training kernel: d / H too large (got d = %s%lld, H = %lld)", "", a->d, a->H);
  if (!a->xnet || !a->vnet || !a->masks || !a->trig || !a->x || !a->v || !a->Lx || !a->p || !a->v1 ||
      !a->grad || !a->workspace)
    return fail(L2HMC_ERR_ARG, "l2hmc_train_propose_grad: NULL pointer%s");
  const int ek = a->energy.kind;
  if (ek != L2HMC_ENERGY_GAUSS_DIAG && ek != L2HMC_ENERGY_GAUSS_DENSE && ek != L2HMC_ENERGY_GMM &&
      ek != L2HMC_ENERGY_ROUGHWELL && ek != L2HMC_ENERGY_FUNNEL)
    return fail(L2HMC_ERR_UNSUPPORTED, "training supports the Gaussian, GMM, Rough-Well and funnel targets (analytic Hessian-vector products)%s");
  if (ek == L2HMC_ENERGY_FUNNEL && (a->d > 16 || a->d < 2 || a->H > 15 || a->variant >= 100 || !(a->energy.eta > 0.f)))
    return fail(L2HMC_ERR_UNSUPPORTED, "funnel training: 2 <= d <= 16, H <= 15, sigma > 0 (register-resident kernel only)%s");
  if (ek != L2HMC_ENERGY_ROUGHWELL && ek != L2HMC_ENERGY_FUNNEL && (!a->energy.mu || !a->energy.prec))
    return fail(L2HMC_ERR_ARG, "energy needs mu and prec (RAW (d,d) precisions for the dense / GMM kinds)%s");
  if (ek == L2HMC_ENERGY_GMM && (!a->energy.logc || a->energy.n_comp < 1 || a->energy.n_comp > KC))
    return fail(L2HMC_ERR_ARG, "GMM training needs logc and 1 <= n_comp <= 8%s");
  if (ek == L2HMC_ENERGY_ROUGHWELL && !(a->energy.eta > 0.f)) return fail(L2HMC_ERR_ARG, "roughwell needs eta > 0%s");
  if (!(a->energy.temperature == 1.f)) return fail(L2HMC_ERR_UNSUPPORTED, "training kernel: temperature must be 1%s");
  if (a->energy.anneal_beta != 0.f && a->energy.anneal_beta != 1.f)
    return fail(L2HMC_ERR_UNSUPPORTED, "training kernel: annealed energies are not supported%s");
  if (!a->alpha && !(a->eps_host > 0.f)) return fail(L2HMC_ERR_ARG, "eps must be > 0%s");
  if (!(a->scale > 0.f) || !(a->inv_n > 0.f)) return fail(L2HMC_ERR_ARG, "scale and inv_n must be > 0%s");
  TArgs k;
  k.xnet = *a->xnet; k.vnet = *a->vnet;
  k.masks = a->masks; k.trig = a->trig; k.alpha = a->alpha; k.eps_host = a->eps_host;
  k.N = a->n_chains; k.d = a->d; k.H = a->H; k.T = a->T; k.x = a->x; k.v = a->v;
  k.dir = a->direction; k.dir_all = a->direction_all; k.ekind = a->energy.kind;
  k.mu = a->energy.mu; k.prec = a->energy.prec; k.logc = a->energy.logc; k.eta = a->energy.eta;
  k.ncomp = ek == L2HMC_ENERGY_GMM ? a->energy.n_comp : 1; k.easy = a->energy.easy;
  k.den = roughwell_den(&a->energy);
  k.scale = a->scale; k.inv_n = a->inv_n;
  k.Lx = a->Lx; k.p = a->p; k.v1 = a->v1; k.grad = a->grad; k.ws = a->workspace;
  k.x_head = st ? st->x_head : nullptr; k.n_head = st ? st->n_head : 0;
#ifdef L2HMC_DBG_EPILOGUE_SELECT
  k.u = nullptr; k.x_next = nullptr;
#endif
  const unsigned blocks = (unsigned)((a->n_chains + TC - 1) / TC);
  hipStream_t s = (hipStream_t)stream;
  const int n_grad = 2 * net_params(a->d, a->H) + 1;
  float* part;
  const int fnw = a->variant >= 100 ? 0 : train_fast_waves(ek, a->d, a->H);     // variant 100: the general tile kernel
  const long long lds_fast = fnw ? 4LL * tf_layout(a->T, fnw).total : 0;
  const long long lds_small = 4LL * ts_layout(a->T).total;
  if (a->variant == 0 && train_small_ok(ek, a->d, a->H) && lds_small <= 48 * 1024) {     // d <= 4: one dimension per lane
    const int KH = khid_of(a->H);
    if (ek == L2HMC_ENERGY_GAUSS_DIAG) launch_train_small<L2HMC_ENERGY_GAUSS_DIAG>(k, KH, blocks, lds_small, s);
    else if (ek == L2HMC_ENERGY_GAUSS_DENSE) launch_train_small<L2HMC_ENERGY_GAUSS_DENSE>(k, KH, blocks, lds_small, s);
    else if (ek == L2HMC_ENERGY_GMM) launch_train_small<L2HMC_ENERGY_GMM>(k, KH, blocks, lds_small, s);
    else launch_train_small<L2HMC_ENERGY_ROUGHWELL>(k, KH, blocks, lds_small, s);
    part = a->workspace + (long long)blocks * a->T * TF_CK * 256;
    note_kernel("train_small_kernel<%lld, %lld>", ek, KH <= 3 ? 3 : 4);
  } else if (fnw && lds_fast <= 160 * 1024) {    // register-resident kernel (train_fast.hpp)
    const int KH = khid_of(a->H);
    int rc;
    if (ek == L2HMC_ENERGY_GAUSS_DIAG) rc = fnw == 1 ? launch_train_fast<L2HMC_ENERGY_GAUSS_DIAG, 1>(k, KH, blocks, lds_fast, s)
                                                     : launch_train_fast<L2HMC_ENERGY_GAUSS_DIAG, 4>(k, KH, blocks, lds_fast, s);
    else if (ek == L2HMC_ENERGY_ROUGHWELL) rc = fnw == 1 ? launch_train_fast<L2HMC_ENERGY_ROUGHWELL, 1>(k, KH, blocks, lds_fast, s)
                                                         : launch_train_fast<L2HMC_ENERGY_ROUGHWELL, 4>(k, KH, blocks, lds_fast, s);
    else if (ek == L2HMC_ENERGY_FUNNEL) rc = launch_train_fast<L2HMC_ENERGY_FUNNEL, 1>(k, KH, blocks, lds_fast, s);
    else rc = launch_train_fast<L2HMC_ENERGY_GAUSS_DENSE, 1>(k, KH, blocks, lds_fast, s);
    if (rc) return rc;
    part = a->workspace + (long long)blocks * a->T * TF_CK * (fnw * 256);
    note_kernel("train_fast_kernel<%lld, %lld, %lld>", ek, fnw, KH <= 3 ? 3 : 4);
  } else {
    if (ek == L2HMC_ENERGY_FUNNEL)               // the general tile kernel has no funnel Hessian-vector product
      return fail(L2HMC_ERR_UNSUPPORTED, "funnel training runs on the register-resident kernel only (d <= 16, H <= 15, "
                  "its LDS plan <= 160 KiB; needs %s%lld bytes here)", "", lds_fast);
    const TLayout L = train_layout(a->d, a->H, a->T, ek, k.ncomp);
    const long long lds = 4LL * L.total;
    if (lds > 160 * 1024)
      return fail(L2HMC_ERR_UNSUPPORTED, "training kernel needs %s%lld bytes of LDS (> 160 KiB): d / H too large for the 16-chain tile", "", lds);
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(train_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(train_kernel, dim3(blocks), dim3(TTHREADS), (size_t)lds, s, k);
    note_kernel("train_kernel");
    part = a->workspace + (long long)a->T * a->n_chains * CKPT * a->d;
  }
  const float* last = part;
  int last_slots = (int)blocks;
  if (blocks > (unsigned)kReduceChunk) {      // two levels, both in slot order: still deterministic
    const int chunks = (int)((blocks + kReduceChunk - 1) / kReduceChunk);
    float* part2 = part + (long long)blocks * n_grad;
    hipLaunchKernelGGL(train_reduce_kernel, dim3((n_grad + 255) / 256, chunks), dim3(256), 0, s, part, (int)blocks,
                       n_grad, kReduceChunk, part2, 0);
    last = part2;
    last_slots = chunks;
  }
  if (st == nullptr) {
    hipLaunchKernelGGL(train_reduce_kernel, dim3((n_grad + 255) / 256, 1), dim3(256), 0, s, last, last_slots, n_grad,
                       last_slots, a->grad, 1);
  } else {
    FinalArgs f;
    memset(&f, 0, sizeof(f));
    // (the loss block is always launched -- it idles without outputs -- so that the select blocks sit at fixed indices)
    const long long mh_blocks = st->u != nullptr ? (st->n_head * a->d + 255) / 256 : 0;
    f.x0 = st->x_head; f.Lx = a->Lx; f.p = a->p; f.u = st->u; f.x_next = st->x_next; f.d = a->d;
    f.v1 = a->v1; f.n_v1 = a->n_chains; f.scale = a->scale; f.n_head = st->n_head;
    // the args carry inv_n as the float the kernel differentiates with; the reported loss is a double (l2hmc_loss_terms takes a
    // double 1 / chains): when the float is the rounding of 1 / integer -- it always is from the host layer -- use that integer
    const double inv_f = (double)a->inv_n, cnt = nearbyint(1.0 / inv_f);
    f.inv_n = (cnt >= 1.0 && (float)(1.0 / cnt) == a->inv_n) ? 1.0 / cnt : inv_f;
    f.terms = st->terms; f.loss = st->loss;
    if (st->theta != nullptr) {
      // lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)   (TF1 Adam, as l2hmc_adam_step)
      const double t = (double)st->step;
      f.lr_t = (float)((double)st->lr * sqrt(1.0 - pow((double)st->beta2, t)) / (1.0 - pow((double)st->beta1, t)));
      f.theta = st->theta; f.m = st->m; f.v = st->v; f.b1 = st->beta1; f.b2 = st->beta2; f.eps = st->epsilon;
      f.last_is_log_eps = st->train_alpha != 0;
      f.n_par = st->train_alpha ? n_grad : n_grad - 1;
    }
    hipLaunchKernelGGL(train_final_kernel, dim3((unsigned)((n_grad + 255) / 256 + 1 + mh_blocks)), dim3(256), 0, s, last, last_slots,
                       n_grad, a->grad, f);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_train_propose_grad(const L2hmcTrainArgs* a, void* stream) { return train_launch(a, nullptr, stream); }

int l2hmc_train_step(const L2hmcTrainArgs* a, const L2hmcTrainStep* st, void* stream) {
  if (!st) return fail(L2HMC_ERR_ARG, "l2hmc_train_step: step is NULL%s");
  return train_launch(a, st, stream);
}

// Adam behind a sharded step's all-reduce, the global-batch loss from the reduced tail in the same launch
__global__ void adam_terms_kernel(float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1, float b2,
                                  float eps, int last_is_log_eps, const float* terms6, float scale, double* loss_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && terms6 != nullptr && loss_out != nullptr) {
    const double A_ = (double)terms6[0] + (double)terms6[1], B_ = (double)terms6[2] + (double)terms6[3];
    const double cnt = (double)terms6[4] + (double)terms6[5];
    loss_out[0] = A_; loss_out[1] = B_;
    loss_out[2] = ((double)scale * A_ - B_ / (double)scale) / cnt;
  }
  if (i >= n) return;
  adam_update(p, g[i], m, v, i, n, lr_t, b1, b2, eps, last_is_log_eps);
}
int l2hmc_adam_step_terms(float* params, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                          float beta2, float epsilon, int64_t step, int32_t last_is_log_eps, const float* terms6,
                          float scale, double* loss_out, void* stream) {
  if (!params || !grad || !m || !v || n < 1 || step < 1 || !(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f) || !(epsilon > 0.f) || (terms6 != nullptr) != (loss_out != nullptr) ||
      (terms6 != nullptr && !(scale > 0.f)))
    return fail(L2HMC_ERR_ARG, "l2hmc_adam_step_terms: bad argument%s");
  const double t = (double)step;
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  hipLaunchKernelGGL(adam_terms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad, m, v,
                     (long long)n, lr_t, beta1, beta2, epsilon, last_is_log_eps, terms6, scale, loss_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

}  // extern "C"
