// l2hmc_trajectory_split -- the generalised-leapfrog trajectory for models that do not fit the
// single fused kernel: wide S/T/Q nets (any H), an image-conditioned 4th branch
// (`encoder_sampler(aux)`, mnist_vae.py:134-150) and the VAE latent-posterior energy
// (mnist_vae.py:104-126: decoder 50 -> 1024 -> 1024 -> 784, BCE + prior).  BASELINE.json config 5.
//
// Everything GEMM-shaped here is a dense product over the chain batch -- (N x K)(K x M) with N = thousands
// of chains: our own fp32 MFMA GEMM (gemm_f32.hpp) with bias / softplus / sigmoid / relu / BCE-gradient /
// chain-rule epilogues fused in; what remains between the products (the masked leapfrog updates with
// tanh / exp, log-det, accept probability, MH select) is a handful of small kernels.  No BLAS library.
// One C-ABI call enqueues the whole trajectory on the caller's stream; all intermediates live in
// a caller-provided workspace.  Same per-chain direction mixing as the fused kernel: each chain
// runs only in its drawn direction.
#include "gemm_f32.hpp"
#include "gemm_xl.hpp"

namespace l2hmc {

// Wt[n][k] = W[k][n]  (W is (K, N) row-major): the forward products contract with W^T in the NT GEMM
__global__ void k_transpose(const float* W, int K, int N, float* Wt, int ldt, int col_off) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + threadIdx.x;
    tile[r][threadIdx.x] = (k < K && n < N) ? W[(long long)k * N + n] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + threadIdx.x;
    if (n < N && k < K) Wt[(long long)n * ldt + col_off + k] = tile[threadIdx.x][r];
  }
}
inline void transpose_into(hipStream_t s, const float* W, int K, int N, float* Wt, int ldt, int col_off) {
  hipLaunchKernelGGL(k_transpose, dim3((N + 31) / 32, (K + 31) / 32), dim3(32, 8), 0, s, W, K, N, Wt, ldt, col_off);
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// time-embedding table tb[net][s][j] = W3[0][j] cos_s + W3[1][j] sin_s + b1 + b2 + b3
__global__ void k_time_table(L2hmcNet xn, L2hmcNet vn, const float* trig, int T, int H, float* tb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * T * H) return;
  const int net = i / (T * H), s = (i / H) % T, j = i % H;
  const L2hmcNet& w = net == 0 ? xn : vn;
  tb[i] = w.W3[j] * trig[2 * s] + w.W3[H + j] * trig[2 * s + 1] + ((w.b1[j] + w.b2[j]) + w.b3[j]);
}
__device__ __forceinline__ int row_of(const unsigned char* dir, int dir_all, long long n, int it, int T, bool& fwd) {
  fwd = dir != nullptr ? dir[n] != 0 : (dir_all != 0);
  return fwd ? it : (T - 1 - it);
}
// U[n] = sum of the BCE row partials the logits GEMM left (fixed order) + |z|^2 / 2   (mnist_vae.py:122-126)
// (|U| ~ 550 for MNIST-sized images: one fp32 rounding of U is 3e-5, so the partials are added in double and the
//  trajectory keeps U0 / U1 as doubles for the energy DIFFERENCE of p_accept; the fp32 copy is for the API.)
__global__ void k_vae_U(const float* rowsum, int np, const float* z, int ldz, int d, float* U, double* Ud,
                        long long N) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double acc = 0.0;
  for (int k = 0; k < np; ++k) acc += (double)rowsum[n * np + k];
  double qd = 0.0;
  for (int k = 0; k < d; ++k) qd += (double)z[n * ldz + k] * (double)z[n * ldz + k];
  acc += 0.5 * qd;
  if (U != nullptr) U[n] = (float)acc;
  if (Ud != nullptr) Ud[n] = acc;
}
// The per-chain reductions of the engine (energy from its row partials, kinetic energy) as one thread per chain walk 50-odd
// strided words each: 8192 chains = 32 workgroups of pure latency (17 / 10 us, the proposal's MH select 29 us -- round 5 trace).
// Staged forms: a workgroup's ROW_CPB chains are read coalesced into LDS, then thread = chain sums ITS row in the same order as
// before (bit-identical results).  Taken while the rows fit 48 KB.
constexpr int ROW_CPB = 64;
__device__ __forceinline__ void stage_rows(float* sm, const float* src, long long ld, int cols, long long n0, long long N) {
  for (int i = threadIdx.x; i < ROW_CPB * cols; i += blockDim.x) {
    const int r = i / cols, k = i % cols;
    sm[i] = n0 + r < N ? src[(n0 + r) * ld + k] : 0.f;
  }
}
__global__ __launch_bounds__(256) void k_vae_U_staged(const float* rowsum, int np, const float* z, int ldz, int d, float* U,
                                                      double* Ud, long long N) {
  extern __shared__ float rsm[];
  float *sr = rsm, *sz = rsm + ROW_CPB * np;
  const long long n0 = (long long)blockIdx.x * ROW_CPB;
  stage_rows(sr, rowsum, np, np, n0, N);
  stage_rows(sz, z, ldz, d, n0, N);
  __syncthreads();
  const long long n = n0 + threadIdx.x;
  if (threadIdx.x >= ROW_CPB || n >= N) return;
  double acc = 0.0;
  for (int k = 0; k < np; ++k) acc += (double)sr[threadIdx.x * np + k];
  double qd = 0.0;
  for (int k = 0; k < d; ++k) qd += (double)sz[threadIdx.x * d + k] * (double)sz[threadIdx.x * d + k];
  acc += 0.5 * qd;
  if (U != nullptr) U[n] = (float)acc;
  if (Ud != nullptr) Ud[n] = acc;
}
inline void launch_vae_U(hipStream_t s, const float* rowsum, int np, const float* z, int ldz, int d, float* U, double* Ud, long long N) {
  const size_t lds = sizeof(float) * ROW_CPB * (size_t)(np + d);
  if (lds <= 48 * 1024)
    hipLaunchKernelGGL(k_vae_U_staged, dim3((unsigned)((N + ROW_CPB - 1) / ROW_CPB)), dim3(256), lds, s, rowsum, np, z, ldz, d, U, Ud, N);
  else
    hipLaunchKernelGGL(k_vae_U, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, rowsum, np, z, ldz, d, U, Ud, N);
}
// HMC mode (nets identically zero, dynamics.py:73-76): the generalised step is the plain leapfrog
//   v_h = v - (eps/2) g(x);  x' = x + eps v_h     [k_hmc_drift]      v' = v_h - (eps/2) g(x')   [k_hmc_kick]
// and its inverse (dynamics.py:159-201 with S = T = Q = 0: v_h = v + (eps/2) g(x'), x = x' - eps v_h, v = v_h + (eps/2) g(x))
// is the same step with eps -> -eps, per chain by its direction bit.
__device__ __forceinline__ float hmc_eps(const float* alpha, float eps_host, const unsigned char* dir, int dall, long long n) {
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host;
  return (dir != nullptr ? dir[n] != 0 : dall != 0) ? eps : -eps;
}
__global__ void k_hmc_drift(float* x, int ldx, const float* v, const float* g, int ldg, float* vh, const float* alpha,
                            float eps_host, const unsigned char* dir, int dall, long long N, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  const long long n = i / d;
  const int k = (int)(i % d);
  const float eps = hmc_eps(alpha, eps_host, dir, dall, n);
  const float h = v[i] + 0.5f * eps * (-g[n * ldg + k]);
  vh[i] = h;
  x[n * ldx + k] = x[n * ldx + k] + eps * h;
}
__global__ void k_hmc_kick(float* v, const float* vh, const float* g, int ldg, const float* alpha, float eps_host,
                           const unsigned char* dir, int dall, long long N, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  const float eps = hmc_eps(alpha, eps_host, dir, dall, i / d);
  v[i] = vh[i] + 0.5f * eps * (-g[(i / d) * ldg + (i % d)]);
}

// Update kernels: ONE WAVE PER CHAIN, lanes over the d dimensions (coalesced rows of every operand), the chain's
// log-det contribution a fixed-order wave reduction.  Every state array carries its row stride: the engine keeps
// [x | grad U] and [v_h | masked x] side by side (row stride 2 d) so that they ARE the K = 2 d inputs of the nets'
// first layer.
__device__ __forceinline__ float wave_sum(float a) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  return a;
}
// Momentum half-update.  out3 = h2 [Ws|Wt|Wq] (no biases yet).  forward: v' = v e^{eps S/2} + (eps/2)(T - e^{eps Q} g);
// backward: v' = (v - (eps/2)(T - e^{eps Q} g)) e^{-eps S/2}   (dynamics.py:121-125,149-153 / :164-170,194-199)
__global__ __launch_bounds__(256) void k_v_half(const float* out3, L2hmcNet w, const float* vin, int ldvi,
                                                const float* g, int ldg, float* vout, int ldvo, float* ld,
                                                const unsigned char* dir, int dir_all, const float* alpha,
                                                float eps_host, long long N, int d) {
  const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const bool fwd = dir != nullptr ? dir[n] != 0 : (dir_all != 0);
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host, heps = 0.5f * eps, sgn = fwd ? 1.f : -1.f;
  float acc = 0.f;
  for (int k = lane; k < d; k += 64) {
    // (w.lam_s == NULL: out3 already holds the final S | T | Q of a caller-supplied net, L2hmcNetCallback)
    const bool raw = w.lam_s == nullptr;
    const float S = raw ? out3[n * 3 * d + k] : expf(w.lam_s[k]) * tanhf(out3[n * 3 * d + k] + w.bs[k]);
    const float T = raw ? out3[n * 3 * d + d + k] : out3[n * 3 * d + d + k] + w.bt[k];
    const float Q = raw ? out3[n * 3 * d + 2 * d + k] : expf(w.lam_q[k]) * tanhf(out3[n * 3 * d + 2 * d + k] + w.bq[k]);
    const float sv = sgn * heps * S, ES = expf(sv), EQ = expf(eps * Q);
    const float cc = heps * (T - EQ * g[n * ldg + k]);
    const float vi = vin[n * ldvi + k];
    vout[n * ldvo + k] = fwd ? vi * ES + cc : (vi - cc) * ES;
    acc += sv;
  }
  acc = wave_sum(acc);
  if (lane == 0) ld[n] += acc;
}
// Masked position update + the masked input of the NEXT net evaluation.  second = 0: keeps
// k1 = (fwd ? m : 1-m) and emits xin = (1-k1) z'; second = 1: keeps 1-k1.
// (dynamics.py:131-145 / :176-190)
__global__ __launch_bounds__(256) void k_x_half(const float* out3, L2hmcNet w, const float* zin, int ldzi,
                                                const float* vh, int ldvh, float* zout, int ldzo, float* xin_next,
                                                int ldxn, float* ld, const float* masks, const unsigned char* dir,
                                                int dir_all, int it, int T, int second, const float* alpha,
                                                float eps_host, long long N, int d) {
  const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  bool fwd;
  const int s = row_of(dir, dir_all, n, it, T, fwd);
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host, sgn = fwd ? 1.f : -1.f;
  float acc = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float m = masks[s * d + k];
    const float k1 = fwd ? m : 1.f - m;
    const float kp = second ? 1.f - k1 : k1, up = 1.f - kp;
    const bool raw = w.lam_s == nullptr;
    const float S = raw ? out3[n * 3 * d + k] : expf(w.lam_s[k]) * tanhf(out3[n * 3 * d + k] + w.bs[k]);
    const float T_ = raw ? out3[n * 3 * d + d + k] : out3[n * 3 * d + d + k] + w.bt[k];
    const float Q = raw ? out3[n * 3 * d + 2 * d + k] : expf(w.lam_q[k]) * tanhf(out3[n * 3 * d + 2 * d + k] + w.bq[k]);
    const float sx = sgn * eps * S, ES = expf(sx), EQ = expf(eps * Q);
    const float tr = eps * (EQ * vh[n * ldvh + k] + T_);
    const float zi = zin[n * ldzi + k];
    const float nw = fwd ? zi * ES + tr : ES * (zi - tr);
    const float zo = kp * zi + up * nw;
    zout[n * ldzo + k] = zo;
    if (xin_next != nullptr) xin_next[n * ldxn + k] = up * zo;      // next kept mask = this update mask
    acc += up * sx;
  }
  acc = wave_sum(acc);
  if (lane == 0) ld[n] += acc;
}
// xin = k1 * x for the first XNet evaluation of a step
__global__ void k_mask_first(const float* x, int ldx, float* xin, int ldxi, const float* masks, const unsigned char* dir,
                             int dir_all, int it, int T, long long N, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  const long long n = i / d;
  const int k = (int)(i % d);
  bool fwd;
  const int s = row_of(dir, dir_all, n, it, T, fwd);
  const float m = masks[s * d + k];
  xin[n * ldxi + k] = (fwd ? m : 1.f - m) * x[n * ldx + k];
}
__global__ void k_kinetic(const float* v, float* K, float* ld, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < d; ++k) acc += 0.5f * v[n * d + k] * v[n * d + k];
  K[n] = acc;
  if (ld != nullptr) ld[n] = 0.f;
}
__global__ __launch_bounds__(256) void k_kinetic_staged(const float* v, float* K, float* ld, long long N, int d) {
  extern __shared__ float rsm[];
  const long long n0 = (long long)blockIdx.x * ROW_CPB;
  stage_rows(rsm, v, d, d, n0, N);
  __syncthreads();
  const long long n = n0 + threadIdx.x;
  if (threadIdx.x >= ROW_CPB || n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < d; ++k) acc += 0.5f * rsm[threadIdx.x * d + k] * rsm[threadIdx.x * d + k];
  K[n] = acc;
  if (ld != nullptr) ld[n] = 0.f;
}
inline void launch_kinetic(hipStream_t s, const float* v, float* K, float* ld, long long N, int d) {
  const size_t lds = sizeof(float) * ROW_CPB * (size_t)d;
  if (lds <= 48 * 1024) hipLaunchKernelGGL(k_kinetic_staged, dim3((unsigned)((N + ROW_CPB - 1) / ROW_CPB)), dim3(256), lds, s, v, K, ld, N, d);
  else hipLaunchKernelGGL(k_kinetic, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, v, K, ld, N, d);
}
// accept probability (dynamics.py:302-309) + MH select (sampler.py:53-55): one thread per ELEMENT of the state (every thread of a
// chain forms the same p from the same five words; the select is a coalesced copy), dimension 0 writes the chain's outputs
__global__ void k_finish(const double* U0, const float* K0, const double* U1, const float* K1, const float* ld,
                         const float* u, const float* x0, const float* x1, int ldx1, float* p_out, float* logjac_out,
                         float* x_next, long long N, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  const long long n = i / d;
  const int k = (int)(i % d);
  const float p = accept_prob((float)((U0[n] - U1[n]) + ((double)K0[n] - (double)K1[n]) + (double)ld[n]));
  if (k == 0) {
    if (p_out != nullptr) p_out[n] = p;
    if (logjac_out != nullptr) logjac_out[n] = ld[n];
  }
  if (x_next != nullptr) {
    const bool acc = (p - u[n]) >= 0.f;
    x_next[i] = acc ? x1[n * ldx1 + k] : x0[i];
  }
}

// accept probability from precomputed energies (dynamics.py:302-309): H = U + |v|^2 / 2
__global__ void k_accept_from_energies(const float* U0, const float* v0, const float* U1, const float* v1,
                                       const float* lj, float* p, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float K0 = 0.f, K1 = 0.f;
  for (int k = 0; k < d; ++k) {
    K0 += 0.5f * v0[n * d + k] * v0[n * d + k];
    K1 += 0.5f * v1[n * d + k] * v1[n * d + k];
  }
  p[n] = accept_prob((U0[n] + K0) - (U1[n] + K1) + lj[n]);
}

inline unsigned nblk(long long n) { return (unsigned)((n + 255) / 256); }

__global__ void k_f2d(const float* a, double* b, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = (double)a[i];
}

// Workspace slices of one 3-layer MLP evaluation: transposed weights (forward products), activations a1 / a2
// and their sigmoids s1 / s2 (= softplus', for the input gradient)
struct Mlp3Ws {
  float *w1t, *w2t, *w3t, *a1, *s1, *a2, *s2;
  // pre-split bf16 planes (gemm_pl_kernel), NULL = the fp32 path: the weights of the two decoder-sized layers in both
  // orientations, and the activations that only ever feed the next product (a1, a2, the BCE gradient, d a2)
  unsigned short *pw2t, *pw3t, *pw2, *pw3, *pa1, *pa2, *plg, *pda2;
  // the trainer's forward evaluations (vae_energy_keep): the same four weight matrices once more as f16x2 planes (two planes each),
  // NULL = the forward pass uses the bf16x3 planes above like the reverse sweep
  unsigned short *pw2t_h, *pw3t_h, *pw2_h, *pw3_h;
};

// L2hmcSplitArgs.gemm_mode / L2hmcTrainSplitArgs.gemm_mode of the call being served on this thread (set at the top of every
// entry point, so no state survives a call): 1 = the decoder-sized products run as bf16x3 (gemm_f32.hpp)
thread_local int t_gemm_bf3 = 0;
// 1 = the pre-split planes of the call being served are f16x2 (GemmArgs.pm; gemm_mode 3: the sampler only)
thread_local int t_plane_mode = 0;

inline GemmArgs gemm_args(const float* A, int lda, const float* B, int ldb, float* C, int ldc, long long M, int N, int K) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.bf3 = t_gemm_bf3;
  g.pm = t_plane_mode;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = (int)M; g.N = N; g.K = K; g.beta = 1.f;
  return g;
}

void mlp3_transposes(hipStream_t s, const L2hmcMlp3& m, const Mlp3Ws& ws) {
  transpose_into(s, m.W1, m.n_in, m.n_h1, ws.w1t, m.n_in, 0);
  transpose_into(s, m.W2, m.n_h1, m.n_h2, ws.w2t, m.n_h1, 0);
  transpose_into(s, m.W3, m.n_h2, m.n_out, ws.w3t, m.n_h2, 0);
}

// a2 = softplus(softplus(x W1 + b1) W2 + b2)  (sigmoids kept when ws.s1 / ws.s2 are given); x has row stride ldx
void mlp3_hidden(hipStream_t s, const L2hmcMlp3& m, const float* x, int ldx, long long N, const Mlp3Ws& ws) {
  GemmArgs g = gemm_args(x, ldx, ws.w1t, m.n_in, ws.a1, m.n_h1, N, m.n_h1, m.n_in);
  g.bias = m.b1; g.C2 = ws.s1; g.ldc2 = m.n_h1;
  launch_gemm<EPI_BIAS_SOFTPLUS>(g, s, m.n_in <= 64 ? SHAPE_MID : SHAPE_AUTO);
  g = gemm_args(ws.a1, m.n_h1, ws.w2t, m.n_h1, ws.a2, m.n_h2, N, m.n_h2, m.n_h1);
  g.bias = m.b2; g.C2 = ws.s2; g.ldc2 = m.n_h2;
  launch_gemm<EPI_BIAS_SOFTPLUS>(g, s);
}

// out (N x n_out) = Linear-softplus-Linear-softplus-Linear(x)
void mlp3_forward(hipStream_t s, const L2hmcMlp3& m, const float* x, long long N, const Mlp3Ws& ws, float* out) {
  mlp3_hidden(s, m, x, m.n_in, N, ws);
  GemmArgs g = gemm_args(ws.a2, m.n_h2, ws.w3t, m.n_h2, out, m.n_out, N, m.n_out, m.n_h2);
  g.bias = m.b3;
  launch_gemm<EPI_BIAS>(g, s, m.n_out <= 256 ? SHAPE_MID : SHAPE_AUTO);
}

// column tiles of the logits GEMM (EPI_BCE leaves 2 row partials per tile): the shape launch_gemm picks for (N, n_pix)
inline int bce_tiles(long long N, int n_pix) { return (n_pix + gemm_tile_n(gemm_auto_shape(N, n_pix)) - 1) / gemm_tile_n(gemm_auto_shape(N, n_pix)); }
inline int bce_partials(long long N, int n_pix) { return gemm_waves_n(gemm_auto_shape(N, n_pix)) * bce_tiles(N, n_pix); }   // per chain
inline int bce_tiles_max(int n_pix) { return (n_pix + 63) / 64; }
// bf16 planes (gemm_xl.hpp): row stride of a K-extent, rows of a weight matrix with N outputs
inline int pld(int K) { return ceil_to(K, 32); }
inline long long prows(int N) { return ceil_to(N, XLP_TN); }

// U (N) and grad (N x d, row stride ldg) of the VAE latent posterior at z (row stride ldz) (mnist_vae.py:122-126):
// six GEMMs, every bias / softplus / sigmoid / BCE / chain-rule product fused into their epilogues; lg (N x n_pix)
// and rowsum (N x 2 tiles) are scratch.  The transposed decoder weights must already be in ws (mlp3_transposes).
// Returns the first error of the pre-split launches (their LDS-size request can be refused by a device), L2HMC_OK otherwise.
int vae_energy(hipStream_t s, const L2hmcMlp3& dec, const float* aux, const float* z, int ldz, long long N, int d,
               const Mlp3Ws& ws, float* lg, float* rowsum, float* U, double* Ud, float* grad, int ldg, float beta = 1.f) {
  if (ws.pa1 != nullptr) {
    int rc = L2HMC_OK, r1;
    // ---- pre-split form (gemm_mode 1 at decoder sizes, gemm_xl.hpp): every activation that only feeds the next product is
    //      written as bf16 planes by its producer's epilogue; the four decoder-sized products read planes on both sides.
    //      Plane row strides are whole k-tiles (pld) and the weight planes whole tiles of rows (prows), zero-padded.
    const int l1 = pld(dec.n_h1), l2 = pld(dec.n_h2), lo = pld(dec.n_out);
    const long long n1 = N * l1, n2 = N * l2, no = N * lo;
    GemmArgs g = gemm_args(z, ldz, ws.w1t, dec.n_in, nullptr, dec.n_h1, N, dec.n_h1, dec.n_in);
    g.bias = dec.b1; g.C2 = ws.s1; g.ldc2 = dec.n_h1; g.Cp = ws.pa1; g.cp_plane = n1; g.ldcp = l1;
    launch_gemm<EPI_BIAS_SOFTPLUS>(g, s, dec.n_in <= 64 ? SHAPE_MID : SHAPE_AUTO);                 // a1 (planes), s1
    g = gemm_args(nullptr, 0, nullptr, 0, nullptr, dec.n_h2, N, dec.n_h2, dec.n_h1);
    g.Ap = ws.pa1; g.ap_plane = n1; g.ldap = l1; g.Bp = ws.pw2t; g.bp_plane = prows(dec.n_h2) * l1; g.ldbp = l1;
    g.bias = dec.b2; g.C2 = ws.s2; g.ldc2 = dec.n_h2; g.Cp = ws.pa2; g.cp_plane = n2; g.ldcp = l2;
    if ((r1 = launch_gemm_planes<EPI_BIAS_SOFTPLUS>(g, s)) != L2HMC_OK) rc = rc ? rc : r1;                                                     // a2 (planes), s2
    g = gemm_args(nullptr, 0, nullptr, 0, nullptr, dec.n_out, N, dec.n_out, dec.n_h2);
    g.Ap = ws.pa2; g.ap_plane = n2; g.ldap = l2; g.Bp = ws.pw3t; g.bp_plane = prows(dec.n_out) * l2; g.ldbp = l2;
    g.bias = dec.b3; g.E = aux; g.lde = dec.n_out; g.rowsum = rowsum; g.n_tiles = bce_tiles_planes(dec.n_out); g.beta = beta;
    g.Cp = ws.plg; g.cp_plane = no; g.ldcp = lo;
    if ((r1 = launch_gemm_planes<EPI_BCE>(g, s)) != L2HMC_OK) rc = rc ? rc : r1;                                                               // beta (sigmoid(logit) - aux) (planes)
    if (U != nullptr || Ud != nullptr)
      launch_vae_U(s, rowsum, 2 * bce_tiles_planes(dec.n_out), z, ldz, d, U, Ud, N);
    if (grad == nullptr) return rc;
    g = gemm_args(nullptr, 0, nullptr, 0, nullptr, dec.n_h2, N, dec.n_h2, dec.n_out);
    g.Ap = ws.plg; g.ap_plane = no; g.ldap = lo; g.Bp = ws.pw3; g.bp_plane = prows(dec.n_h2) * lo; g.ldbp = lo;
    g.E = ws.s2; g.lde = dec.n_h2; g.Cp = ws.pda2; g.cp_plane = n2; g.ldcp = l2;
    if ((r1 = launch_gemm_planes<EPI_MUL>(g, s)) != L2HMC_OK) rc = rc ? rc : r1;                                                               // d a2 (planes)
    g = gemm_args(nullptr, 0, nullptr, 0, ws.a1, dec.n_h1, N, dec.n_h1, dec.n_h2);
    g.Ap = ws.pda2; g.ap_plane = n2; g.ldap = l2; g.Bp = ws.pw2; g.bp_plane = prows(dec.n_h1) * l2; g.ldbp = l2;
    g.E = ws.s1; g.lde = dec.n_h1;
    if ((r1 = launch_gemm_planes<EPI_MUL>(g, s)) != L2HMC_OK) rc = rc ? rc : r1;                                                               // d a1 (fp32: the K = 1024, N = d product reads it)
    g = gemm_args(ws.a1, dec.n_h1, dec.W1, dec.n_h1, grad, ldg, N, d, dec.n_h1);
    g.E = z; g.lde = ldz;
    launch_gemm<EPI_ADD>(g, s, d <= 64 ? SHAPE_SKINNY : SHAPE_MID);
    return rc;
  }
  mlp3_hidden(s, dec, z, ldz, N, ws);
  GemmArgs g = gemm_args(ws.a2, dec.n_h2, ws.w3t, dec.n_h2, lg, dec.n_out, N, dec.n_out, dec.n_h2);
  g.bias = dec.b3; g.E = aux; g.lde = dec.n_out; g.rowsum = rowsum; g.n_tiles = bce_tiles(N, dec.n_out); g.beta = beta;
  launch_gemm<EPI_BCE>(g, s);                                     // lg := beta (sigmoid(logit) - aux)
  if (U != nullptr || Ud != nullptr)
    launch_vae_U(s, rowsum, bce_partials(N, dec.n_out), z, ldz, d, U, Ud, N);
  if (grad == nullptr) return L2HMC_OK;
  // d a2 = dl W3^T (.) sigmoid(p2);  d a1 = d a2 W2^T (.) sigmoid(p1);  d z = d a1 W1^T + z
  g = gemm_args(lg, dec.n_out, dec.W3, dec.n_out, ws.a2, dec.n_h2, N, dec.n_h2, dec.n_out);
  g.E = ws.s2; g.lde = dec.n_h2;
  launch_gemm<EPI_MUL>(g, s);
  g = gemm_args(ws.a2, dec.n_h2, dec.W2, dec.n_h2, ws.a1, dec.n_h1, N, dec.n_h1, dec.n_h2);
  g.E = ws.s1; g.lde = dec.n_h1;
  launch_gemm<EPI_MUL>(g, s);
  g = gemm_args(ws.a1, dec.n_h1, dec.W1, dec.n_h1, grad, ldg, N, d, dec.n_h1);
  g.E = z; g.lde = ldz;
  launch_gemm<EPI_ADD>(g, s, d <= 64 ? SHAPE_SKINNY : SHAPE_MID);
  return L2HMC_OK;
}

struct SplitPlan {
  long long total;
  long long abv, abx, vc, y, h1, h2, out3, aux_h, tb, U0, K0, U1, K1, ld, rowsum, lg;   // abv = [x | grad U], abx = [v_h | masked x]
  long long xp, gp, uf;                                          // built-in energies: contiguous x, grad U, fp32 U
  long long dw1t, dw2t, dw3t, a1, s1, a2, s2;                    // decoder
  long long ew1t, ew2t, ew3t, e1, e2;                            // image branch
  long long nx12t, nx4t, nxht, nv12t, nv4t, nvht;                // S/T/Q nets: [W1; W2]^T, W4^T, [Ws | Wt | Wq]^T
  bool planes;                                                   // the decoder products can take the pre-split form at this size
  long long pw2t, pw3t, pw2, pw3, pa1, pa2, plg, pda2;           // bf16 planes (3 x elements x 2 bytes each), in floats
};

SplitPlan plan_split(long long N, int d, int H, int T, const L2hmcMlp3* enc, const L2hmcMlp3* dec) {
  SplitPlan p;
  long long o = 0;
  auto take = [&](long long n) { const long long at = o; o += (n + 3) & ~3LL; return at; };
  p.abv = take(N * 2 * d); p.abx = take(N * 2 * d); p.vc = take(N * d); p.y = take(N * d);
  p.h1 = take(N * H); p.h2 = take(N * H); p.out3 = take(N * 3 * d);
  p.aux_h = take(enc ? N * H : 0); p.tb = take(2LL * T * H);
  p.U0 = take(2 * N); p.K0 = take(N); p.U1 = take(2 * N); p.K1 = take(N); p.ld = take(N);      // U0 / U1: doubles
  p.rowsum = take(dec ? N * 2 * bce_tiles_max(dec->n_out) : 0);
  p.lg = take(dec ? N * dec->n_out : 0);
  p.dw1t = take(dec ? (long long)dec->n_in * dec->n_h1 : 0); p.dw2t = take(dec ? (long long)dec->n_h1 * dec->n_h2 : 0);
  p.dw3t = take(dec ? (long long)dec->n_h2 * dec->n_out : 0);
  p.a1 = take(dec ? N * dec->n_h1 : 0); p.s1 = take(dec ? N * dec->n_h1 : 0);
  p.a2 = take(dec ? N * dec->n_h2 : 0); p.s2 = take(dec ? N * dec->n_h2 : 0);
  p.xp = take(dec ? 0 : N * d); p.gp = take(dec ? 0 : N * d); p.uf = take(dec ? 0 : N);
  p.ew1t = take(enc ? (long long)enc->n_in * enc->n_h1 : 0); p.ew2t = take(enc ? (long long)enc->n_h1 * enc->n_h2 : 0);
  p.ew3t = take(enc ? (long long)enc->n_h2 * enc->n_out : 0);
  p.e1 = take(enc ? N * enc->n_h1 : 0); p.e2 = take(enc ? N * enc->n_h2 : 0);
  // S/T/Q nets: transposed copies, rows and K zero-padded to multiples of 16 (net_eval_kernel has no guards)
  const long long n12 = (long long)ceil16(H) * ceil16(2 * d), n4 = (long long)ceil16(H) * ceil16(H), nh = (long long)ceil16(3 * d) * ceil16(H);
  p.nx12t = take(n12); p.nx4t = take(n4); p.nxht = take(nh);
  p.nv12t = take(n12); p.nv4t = take(n4); p.nvht = take(nh);
  // bf16 planes of the decoder's two big layers and of the activations between them (gemm_xl.hpp): 3 planes x 2 bytes
  p.planes = dec != nullptr && gemm_planes_ok(N, dec->n_h2, dec->n_h1) && gemm_planes_ok(N, dec->n_out, dec->n_h2) &&
             gemm_planes_ok(N, dec->n_h2, dec->n_out) && gemm_planes_ok(N, dec->n_h1, dec->n_h2) && dec->n_h1 % 4 == 0;
  auto takep = [&](long long elems) { return take(p.planes ? (elems * 3 + 1) / 2 : 0); };
  if (dec != nullptr) {
    p.pw2t = takep(prows(dec->n_h2) * pld(dec->n_h1)); p.pw3t = takep(prows(dec->n_out) * pld(dec->n_h2));
    p.pw2 = takep(prows(dec->n_h1) * pld(dec->n_h2)); p.pw3 = takep(prows(dec->n_h2) * pld(dec->n_out));
    p.pa1 = takep(N * pld(dec->n_h1)); p.pa2 = takep(N * pld(dec->n_h2));
    p.plg = takep(N * pld(dec->n_out)); p.pda2 = takep(N * pld(dec->n_h2));
  } else {
    p.pw2t = p.pw3t = p.pw2 = p.pw3 = p.pa1 = p.pa2 = p.plg = p.pda2 = o;
  }
  p.total = o;
  return p;
}

int check_mlp(const L2hmcMlp3* m, const char* what) {
  if (!m || !m->W1 || !m->b1 || !m->W2 || !m->b2 || !m->W3 || !m->b3 || m->n_in < 1 || m->n_h1 < 1 ||
      m->n_h2 < 1 || m->n_out < 1)
    return fail(L2HMC_ERR_ARG, "%s: incomplete 3-layer MLP description", what);
  return L2HMC_OK;
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" {

int64_t l2hmc_split_workspace_floats(int64_t n_chains, int32_t d, int32_t H, int32_t T,
                                     const L2hmcMlp3* aux_encoder, const L2hmcMlp3* decoder) {
  if (n_chains < 0 || d < 1 || H < 1 || T < 1) return fail(L2HMC_ERR_ARG, "l2hmc_split_workspace_floats: bad argument%s");
  return plan_split(n_chains, d, H, T, aux_encoder, decoder).total;
}

int l2hmc_bf16_planes(const float* W, int32_t ld, int64_t rows, int32_t K, uint16_t* planes, int64_t rows_pad, int32_t ld_planes,
                      void* stream) {
  if (!W || !planes || rows < 1 || K < 1 || ld < K || rows_pad < rows || ld_planes < K || (ld_planes & 3))
    return fail(L2HMC_ERR_ARG, "l2hmc_bf16_planes: bad argument%s");
  to_planes((hipStream_t)stream, W, ld, rows, K, planes, rows_pad, ld_planes);
  return L2HMC_OK;
}

int l2hmc_vae_energy(const L2hmcMlp3* decoder, const float* aux, const float* x, int64_t n_chains, int32_t d,
                     float* U_out, float* grad_out, float* workspace, float bce_scale, void* stream) {
  t_gemm_bf3 = 0;
  t_plane_mode = 0;
  int rc = check_mlp(decoder, "l2hmc_vae_energy");
  if (rc) return rc;
  if (!aux || !x || !workspace || n_chains < 0 || d != decoder->n_in) return fail(L2HMC_ERR_ARG, "l2hmc_vae_energy: bad argument%s");
  if (!(bce_scale >= 0.f && bce_scale <= 1.f)) return fail(L2HMC_ERR_ARG, "bce_scale must be in [0, 1] (0 = off)%s");
  const float beta = bce_scale > 0.f ? bce_scale : 1.f;
  if (n_chains == 0) return L2HMC_OK;
  hipStream_t s = (hipStream_t)stream;
  const SplitPlan p = plan_split(n_chains, d, 1, 1, nullptr, decoder);
  float* w = workspace;
  const Mlp3Ws ws = {w + p.dw1t, w + p.dw2t, w + p.dw3t, w + p.a1, w + p.s1, w + p.a2, w + p.s2};
  mlp3_transposes(s, *decoder, ws);
  rc = vae_energy(s, *decoder, aux, x, d, n_chains, d, ws, w + p.lg, w + p.rowsum, U_out, nullptr, grad_out, d, beta);
  if (rc) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_p_accept_energies(const float* U0, const float* v0, const float* U1, const float* v1, const float* log_jac,
                            int64_t n_chains, int32_t d, float* p_out, void* stream) {
  if (!U0 || !v0 || !U1 || !v1 || !log_jac || !p_out || n_chains < 0 || d < 1)
    return fail(L2HMC_ERR_ARG, "l2hmc_p_accept_energies: bad argument%s");
  if (n_chains == 0) return L2HMC_OK;
  hipLaunchKernelGGL(k_accept_from_energies, dim3(nblk(n_chains)), dim3(256), 0, (hipStream_t)stream, U0, v0, U1, v1,
                     log_jac, p_out, (long long)n_chains, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_trajectory_split(const L2hmcSplitArgs* a, void* stream) {
  if (!a) return fail(L2HMC_ERR_ARG, "args is NULL%s");
  if (a->gemm_mode < 0 || a->gemm_mode > 3) return fail(L2HMC_ERR_ARG, "gemm_mode must be 0 (f32 MFMA), 1 (bf16x3), 2 (bf16x3, split in the loop) or 3 (f16x2 planes)%s");
  t_gemm_bf3 = a->gemm_mode != 0;
  t_plane_mode = a->gemm_mode == 3;
  const bool user = a->energy_cb != nullptr;        // the caller's own energy, evaluated on the host between launches
  const bool builtin = a->energy != nullptr;        // a target of utils/distributions.py instead of the decoder posterior
  const bool unets = a->net_cb != nullptr;          // the caller's own S/T/Q nets (any callable, dynamics.py:69-79)
  int rc;
  // (round 6: also with the decoder posterior -- the caller's nets then receive the images themselves, e.g. two nets with
  //  SEPARATE image branches, which the fused form does not have)
  if (unets && (a->xnet || a->vnet || a->aux_encoder || a->hmc))
    return fail(L2HMC_ERR_ARG, "net_cb excludes xnet / vnet / aux_encoder / hmc%s");
  if (user) {
    if (a->decoder || a->energy) return fail(L2HMC_ERR_ARG, "energy_cb excludes decoder and energy%s");
    if (a->bce_scale != 0.f) return fail(L2HMC_ERR_UNSUPPORTED, "bce_scale with a caller-supplied energy (anneal it in the callback)%s");
    if (a->aux_encoder && !a->aux) return fail(L2HMC_ERR_ARG, "aux_encoder needs aux%s");
  } else if (builtin) {
    if (a->decoder || a->aux_encoder || a->hmc)
      return fail(L2HMC_ERR_UNSUPPORTED, "a built-in energy excludes decoder / aux_encoder / hmc (HMC mode runs on l2hmc_trajectory)%s");
    if ((rc = check_energy(a->energy, a->d))) return rc;       // (a tempered target: l2hmc_energy divides U and grad U, dynamics.py:203-212)
  } else if ((rc = check_mlp(a->decoder, "decoder"))) {
    return rc;
  }
  if (a->aux_encoder && (rc = check_mlp(a->aux_encoder, "aux_encoder"))) return rc;
  const long long N = a->n_chains;
  const int d = a->d, H = unets ? 4 : a->H, T = a->T;      // (caller-supplied nets: no hidden activations are planned for)
  if (N < 0 || d < 1 || H < 1 || T < 1) return fail(L2HMC_ERR_ARG, "bad n_chains / d / H / T%s");
  if (N == 0) return L2HMC_OK;
  const bool hmc = a->hmc != 0;
  if ((!hmc && ((!unets && (!a->xnet || !a->vnet)) || !a->masks || !a->trig)) || (!builtin && !user && !a->aux) || !a->x || !a->v || !a->workspace)
    return fail(L2HMC_ERR_ARG, "l2hmc_trajectory_split: NULL pointer%s");
  if (!(a->bce_scale >= 0.f && a->bce_scale <= 1.f)) return fail(L2HMC_ERR_ARG, "bce_scale must be in [0, 1] (0 = off)%s");
  const float beta = a->bce_scale > 0.f ? a->bce_scale : 1.f;
  if (!builtin && !user && a->decoder->n_in != d) return fail(L2HMC_ERR_ARG, "decoder input width != d%s");
  if (a->aux_encoder && (a->aux_encoder->n_out != H || (!user && a->aux_encoder->n_in != a->decoder->n_out)))
    return fail(L2HMC_ERR_ARG, "aux_encoder must map (N, n_pix) -> (N, H)%s");
  if (a->step_begin < 0 || a->n_steps < 0 || a->step_begin + a->n_steps > T) return fail(L2HMC_ERR_ARG, "steps outside the schedule%s");
  if (a->x_next && !a->u) return fail(L2HMC_ERR_ARG, "x_next needs u%s");
  if (!a->alpha && !(a->eps_host > 0.f)) return fail(L2HMC_ERR_ARG, "eps must be > 0%s");
  const SplitPlan p = plan_split(N, d, H, T, a->aux_encoder, a->decoder);
  if (a->workspace_floats < p.total) return fail(L2HMC_ERR_ARG, "workspace too small: need %s%lld floats", "", p.total);
  hipStream_t s = (hipStream_t)stream;
  float* w = a->workspace;
  static const L2hmcMlp3 no_dec = {};
  const L2hmcMlp3& dec = (builtin || user) ? no_dec : *a->decoder;
  Mlp3Ws dws = {w + p.dw1t, w + p.dw2t, w + p.dw3t, w + p.a1, w + p.s1, w + p.a2, w + p.s2};
  const bool use_planes = p.planes && (a->gemm_mode == 1 || a->gemm_mode == 3) && !builtin && !user;
  // (what l2hmc_last_kernel reports for this engine: the kernel of the decoder-sized products, or the net evaluation)
  note_kernel(builtin || user ? "net_eval_kernel" : use_planes ? "gemm_xlp_kernel" : "gemm_nt_kernel");
  if (use_planes) {     // the three epilogues vae_energy launches on planes
    if (t_plane_mode) {
      if ((rc = gemm_planes_prepare<EPI_BIAS_SOFTPLUS, 1>()) != L2HMC_OK || (rc = gemm_planes_prepare<EPI_BCE, 1>()) != L2HMC_OK ||
          (rc = gemm_planes_prepare<EPI_MUL, 1>()) != L2HMC_OK)
        return rc;
    } else if ((rc = gemm_planes_prepare<EPI_BIAS_SOFTPLUS>()) != L2HMC_OK || (rc = gemm_planes_prepare<EPI_BCE>()) != L2HMC_OK ||
               (rc = gemm_planes_prepare<EPI_MUL>()) != L2HMC_OK)
      return rc;
  }
  if (use_planes) {
    auto us = [&](long long off) { return reinterpret_cast<unsigned short*>(w + off); };
    dws.pw2t = us(p.pw2t); dws.pw3t = us(p.pw3t); dws.pw2 = us(p.pw2); dws.pw3 = us(p.pw3);
    dws.pa1 = us(p.pa1); dws.pa2 = us(p.pa2); dws.plg = us(p.plg); dws.pda2 = us(p.pda2);
  }
  // [x | grad U] and [v_h | masked x] live side by side (row stride L = 2 d): they are the first-layer inputs
  const int L = 2 * d;
  float *xc = w + p.abv, *g = w + p.abv + d, *vh = w + p.abx, *xin = w + p.abx + d, *vc = w + p.vc, *y = w + p.y;
  float *h1 = w + p.h1, *h2 = w + p.h2, *out3 = w + p.out3, *tb = w + p.tb, *ld = w + p.ld;
  float* aux_h = a->aux_encoder ? w + p.aux_h : nullptr;
  static const L2hmcNet no_net = {};
  const L2hmcNet &xn = (hmc || unets) ? no_net : *a->xnet, &vn = (hmc || unets) ? no_net : *a->vnet;
  const unsigned char* dir = a->direction;
  const int dall = a->direction_all;
  const unsigned nw4 = (unsigned)((N + 3) / 4);                  // one wave per chain, 4 per workgroup

  (void)hipMemcpy2DAsync(xc, sizeof(float) * L, a->x, sizeof(float) * d, sizeof(float) * d, (size_t)N, hipMemcpyDeviceToDevice, s);
  (void)hipMemcpyAsync(vc, a->v, sizeof(float) * N * d, hipMemcpyDeviceToDevice, s);
  const bool have_w = (a->reuse & 1) != 0, have_auxh = (a->reuse & 2) != 0;    // still in the workspace (caller vouches)
  if (!builtin && !user && !have_w) {
    mlp3_transposes(s, dec, dws);
    if (use_planes) {       // the two decoder-sized layers, both orientations, as bf16 planes: once per parameter update
      to_planes(s, dws.w2t, dec.n_h1, dec.n_h2, dec.n_h1, dws.pw2t, prows(dec.n_h2), pld(dec.n_h1), t_plane_mode);
      to_planes(s, dws.w3t, dec.n_h2, dec.n_out, dec.n_h2, dws.pw3t, prows(dec.n_out), pld(dec.n_h2), t_plane_mode);
      to_planes(s, dec.W2, dec.n_h2, dec.n_h1, dec.n_h2, dws.pw2, prows(dec.n_h1), pld(dec.n_h2), t_plane_mode);
      to_planes(s, dec.W3, dec.n_out, dec.n_h2, dec.n_out, dws.pw3, prows(dec.n_h2), pld(dec.n_out), t_plane_mode);
      // the epilogues write the columns of an activation only: its padding up to a whole k-tile is zeroed here
      planes_zero_pad(s, dws.pa1, N, dec.n_h1, pld(dec.n_h1));
      planes_zero_pad(s, dws.pa2, N, dec.n_h2, pld(dec.n_h2));
      planes_zero_pad(s, dws.plg, N, dec.n_out, pld(dec.n_out));
      planes_zero_pad(s, dws.pda2, N, dec.n_h2, pld(dec.n_h2));
    }
  }
  if (a->aux_encoder && !hmc) {      // the image branch is step-invariant: once per trajectory, not 4T times
    const L2hmcMlp3& enc = *a->aux_encoder;
    const Mlp3Ws ews = {w + p.ew1t, w + p.ew2t, w + p.ew3t, w + p.e1, nullptr, w + p.e2, nullptr};
    if (!have_w) mlp3_transposes(s, enc, ews);
    if (!have_auxh) mlp3_forward(s, enc, a->aux, N, ews, aux_h);
  }
  if (!hmc && !unets && !have_w) {
    hipLaunchKernelGGL(k_time_table, dim3(nblk(2LL * T * H)), dim3(256), 0, s, xn, vn, a->trig, T, H, tb);
    // per net: [W1; W2]^T (H x 2d), W4^T (H x H), [Ws | Wt | Wq]^T (3d x H)
    const L2hmcNet* nets[2] = {&xn, &vn};
    float* w12t[2] = {w + p.nx12t, w + p.nv12t};
    float* w4t[2] = {w + p.nx4t, w + p.nv4t};
    float* wht[2] = {w + p.nxht, w + p.nvht};
    const int K1p = ceil16(L), Hp = ceil16(H);
    (void)hipMemsetAsync(w + p.nx12t, 0, sizeof(float) * (size_t)(p.nvht + (long long)ceil16(3 * d) * Hp - p.nx12t), s);
    for (int i = 0; i < 2; ++i) {
      transpose_into(s, nets[i]->W1, d, H, w12t[i], K1p, 0);
      transpose_into(s, nets[i]->W2, d, H, w12t[i], K1p, d);
      transpose_into(s, nets[i]->W4, H, H, w4t[i], Hp, 0);
      transpose_into(s, nets[i]->Ws, H, d, wht[i], Hp, 0);
      transpose_into(s, nets[i]->Wt, H, d, wht[i] + (long long)d * Hp, Hp, 0);
      transpose_into(s, nets[i]->Wq, H, d, wht[i] + 2LL * d * Hp, Hp, 0);
    }
  }
  launch_kinetic(s, vc, w + p.K0, ld, N, d);
  double *U0d = reinterpret_cast<double*>(w + p.U0), *U1d = reinterpret_cast<double*>(w + p.U1);
  // U (double, optional) and grad U at the current x: the decoder posterior (six GEMMs) or one of the built-in
  // targets (the fused kernels' own energy kernel on a contiguous copy of x)
  auto energy_eval = [&](double* Ud) -> int {
    if (user) {      // the caller enqueues U / grad U of the (N, d) block at xc (row stride L) on this stream
      const int r = a->energy_cb(a->energy_cb_user, xc, L, N, d, Ud, g, L, stream);
      return r ? fail(L2HMC_ERR_ARG, "the energy callback failed (returned %s%lld)", "", (long long)r) : L2HMC_OK;
    }
    if (!builtin) {
      return vae_energy(s, dec, a->aux, xc, L, N, d, dws, w + p.lg, w + p.rowsum, nullptr, Ud, g, L, beta);
    }
    (void)hipMemcpy2DAsync(w + p.xp, sizeof(float) * d, xc, sizeof(float) * L, sizeof(float) * d, (size_t)N, hipMemcpyDeviceToDevice, s);
    const int r = l2hmc_energy(a->energy, w + p.xp, N, d, Ud ? w + p.uf : nullptr, w + p.gp, stream);
    if (r) return r;
    (void)hipMemcpy2DAsync(g, sizeof(float) * L, w + p.gp, sizeof(float) * d, sizeof(float) * d, (size_t)N, hipMemcpyDeviceToDevice, s);
    if (Ud) hipLaunchKernelGGL(k_f2d, dim3(nblk(N)), dim3(256), 0, s, w + p.uf, Ud, N);
    return L2HMC_OK;
  };
  if ((rc = energy_eval(U0d))) return rc;
  if (a->n_steps == 0) (void)hipMemcpyAsync(U1d, U0d, sizeof(double) * N, hipMemcpyDeviceToDevice, s);

  // one net evaluation: out3 = relu(relu([a | b] [W1; W2] + time + aux_h) W4 + b4) [Ws|Wt|Wq]   (the head biases are
  // added by the update kernels).  H % 4 == 0 and d even: ONE launch of net_eval_kernel (activations resident in
  // LDS); otherwise three GEMMs with fused epilogues on 64 x 64 tiles.
  // (32 chains per workgroup -- every weight fragment streamed from L2 feeding two MFMAs, net_eval_kernel<2> -- was
  //  measured twice at 8192 chains: with one workgroup per CU (88 KB of LDS) the evaluation is 25 % slower than with 16
  //  chains, with two per CU (69 KB after the head products moved into the dead first activation) still 8 % slower:
  //  occupancy, not L2 traffic, is what this kernel lives on)
  // 32 chains on 8 waves (one workgroup per CU, half the L2 weight traffic) once that still fills the chip
  int ne_dev = 0, ne_cus = 256;
  if (hipGetDevice(&ne_dev) != hipSuccess || hipDeviceGetAttribute(&ne_cus, hipDeviceAttributeMultiprocessorCount, ne_dev) != hipSuccess ||
      ne_cus <= 0)
    ne_cus = 256;
  const int ne_cb = N >= 32LL * ne_cus ? 2 : 1;            // (32 chains per CU = one 8-wave workgroup each; 8192 on 256 CUs)
  const size_t ne_lds = net_eval_lds_bytes(d, H, ne_cb);
  const bool ne_ok = !unets && (H % 4 == 0) && (d % 2 == 0) && ceil16(H) <= 16 * NE_MAXKT && ceil16(2 * d) <= 16 * NE_MAXKT && ne_lds <= 160 * 1024;
  if (ne_ok && !hmc && ne_lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(net_eval_fn(ne_cb, d, H), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ne_lds);
    if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  // `upd`: the half-update that consumes the evaluation; fused into net_eval_kernel when that kernel runs, else the
  // stand-alone update kernel(s) follow the three GEMMs
  auto net_eval = [&](const L2hmcNet& nw, int net, const float* ab, int it, NetEvalArgs::Update upd) -> int {
    upd.bs = nw.bs; upd.bt = nw.bt; upd.bq = nw.bq; upd.lam_s = nw.lam_s; upd.lam_q = nw.lam_q;
    upd.alpha = a->alpha; upd.eps_host = a->eps_host; upd.ld = ld; upd.masks = a->masks;
    if (ne_ok) {
      NetEvalArgs na = {};
      na.AB = ab; na.ldab = L; na.W12t = w + (net == 0 ? p.nx12t : p.nv12t); na.W4t = w + (net == 0 ? p.nx4t : p.nv4t);
      na.Wht = w + (net == 0 ? p.nxht : p.nvht); na.b4 = nw.b4; na.tb = tb + (long long)net * T * H; na.auxh = aux_h;
      na.dir = dir; na.dir_all = dall; na.it = it; na.T = T; na.out3 = out3; na.M = (int)N; na.d = d; na.H = H;
      na.upd = upd;
      const unsigned blocks = (unsigned)((N + 16 * ne_cb - 1) / (16 * ne_cb));
      launch_net_eval(ne_cb, blocks, ne_lds, s, na);
      return L2HMC_OK;
    }
    if (unets) {     // the caller's net writes the final S | T | Q into out3 (nw = no_net: the update kernels take them as they are)
      const int r = a->net_cb(a->net_cb_user, net, ab, L, N, d, it, dir, dall, out3, stream);
      if (r) return fail(L2HMC_ERR_ARG, "the net callback failed (returned %s%lld)", "", (long long)r);
    } else {
    GemmArgs ga = gemm_args(ab, L, w + (net == 0 ? p.nx12t : p.nv12t), ceil16(L), h1, H, N, H, L);
    ga.E = aux_h; ga.lde = H; ga.tb = tb + (long long)net * T * H; ga.dir = dir; ga.dir_all = dall; ga.it = it; ga.T = T;
    launch_gemm<EPI_NET1>(ga, s, SHAPE_MID);
    ga = gemm_args(h1, H, w + (net == 0 ? p.nx4t : p.nv4t), ceil16(H), h2, H, N, H, H);
    ga.bias = nw.b4;
    launch_gemm<EPI_BIAS_RELU>(ga, s, SHAPE_MID);
    ga = gemm_args(h2, H, w + (net == 0 ? p.nxht : p.nvht), ceil16(H), out3, 3 * d, N, 3 * d, H);
    launch_gemm<EPI_BIAS>(ga, s, SHAPE_MID);
    }
    if (upd.mode == 1) {
      hipLaunchKernelGGL(k_v_half, dim3(nw4), dim3(256), 0, s, out3, nw, upd.vin, upd.ldvi, upd.g, upd.ldg, upd.vout, upd.ldvo, ld,
                         dir, dall, a->alpha, a->eps_host, N, d);
      if (upd.xin != nullptr)
        hipLaunchKernelGGL(k_mask_first, dim3(nblk(N * d)), dim3(256), 0, s, upd.x, upd.ldx, upd.xin, upd.ldxi, a->masks, dir, dall,
                           it, T, N, d);
    } else {
      hipLaunchKernelGGL(k_x_half, dim3(nw4), dim3(256), 0, s, out3, nw, upd.zin, upd.ldzi, upd.vh, upd.ldvh, upd.zout, upd.ldzo,
                         upd.xin_next, upd.ldxn, ld, a->masks, dir, dall, it, T, upd.second, a->alpha, a->eps_host, N, d);
    }
    return L2HMC_OK;
  };
  auto v_update = [&](const float* vin, int ldvi, float* vout, int ldvo, bool with_mask) {
    NetEvalArgs::Update u = {};
    u.mode = 1; u.vin = vin; u.ldvi = ldvi; u.g = g; u.ldg = L; u.vout = vout; u.ldvo = ldvo;
    if (with_mask) { u.x = xc; u.ldx = L; u.xin = xin; u.ldxi = L; }
    return u;
  };
  auto x_update = [&](const float* zin, int ldzi, float* zout, int ldzo, float* xin_next, int second) {
    NetEvalArgs::Update u = {};
    u.mode = 2; u.zin = zin; u.ldzi = ldzi; u.vh = vh; u.ldvh = L; u.zout = zout; u.ldzo = ldzo;
    u.xin_next = xin_next; u.ldxn = L; u.second = second;
    return u;
  };

  for (int k = 0; k < a->n_steps; ++k) {
    const int it = a->step_begin + k;
    const bool last = k == a->n_steps - 1;
    if (hmc) {
      hipLaunchKernelGGL(k_hmc_drift, dim3(nblk(N * d)), dim3(256), 0, s, xc, L, vc, g, L, y, a->alpha, a->eps_host, dir, dall, N, d);
      if ((rc = energy_eval(last ? U1d : nullptr))) return rc;
      hipLaunchKernelGGL(k_hmc_kick, dim3(nblk(N * d)), dim3(256), 0, s, vc, y, g, L, a->alpha, a->eps_host, dir, dall, N, d);
      continue;
    }
    if ((rc = net_eval(vn, 1, xc, it, v_update(vc, d, vh, L, true)))) return rc;      // v_h, and xin = k1 x for the next evaluation
    if ((rc = net_eval(xn, 0, vh, it, x_update(xc, L, y, d, xin, 0)))) return rc;      // y, xin = (1 - k1) y
    if ((rc = net_eval(xn, 0, vh, it, x_update(y, d, xc, L, nullptr, 1)))) return rc;  // x'
    if ((rc = energy_eval(last ? U1d : nullptr))) return rc;
    if ((rc = net_eval(vn, 1, xc, it, v_update(vh, L, vc, d, false)))) return rc;      // v'
  }
  launch_kinetic(s, vc, w + p.K1, (float*)nullptr, N, d);
  if (a->x_out) (void)hipMemcpy2DAsync(a->x_out, sizeof(float) * d, xc, sizeof(float) * L, sizeof(float) * d, (size_t)N, hipMemcpyDeviceToDevice, s);
  if (a->v_out) (void)hipMemcpyAsync(a->v_out, vc, sizeof(float) * N * d, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(k_finish, dim3(nblk(N * d)), dim3(256), 0, s, U0d, w + p.K0, U1d, w + p.K1, ld, a->u, a->x,
                     xc, L, a->p_out, a->logjac_out, a->x_next, N, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

}  // extern "C"

#include "train_split.hpp"
