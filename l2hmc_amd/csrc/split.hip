// l2hmc_trajectory_split -- the generalised-leapfrog trajectory for models that do not fit the
// single fused kernel: wide S/T/Q nets (any H), an image-conditioned 4th branch
// (`encoder_sampler(aux)`, mnist_vae.py:134-150) and the VAE latent-posterior energy
// (mnist_vae.py:104-126: decoder 50 -> 1024 -> 1024 -> 784, BCE + prior).  BASELINE.json config 5.
//
// Everything GEMM-shaped here is a PLAIN dense product over the chain batch -- (N x K)(K x M) with
// N = thousands of chains -- so it goes to the library (rocBLAS sgemm, fp32); the glue between the
// products (bias + softplus / relu, the BCE gradient, the masked leapfrog updates with
// tanh / exp, log-det, accept probability, MH select) is a handful of small hand-written kernels.
// One C-ABI call enqueues the whole trajectory on the caller's stream; all intermediates live in
// a caller-provided workspace.  Same per-chain direction mixing as the fused kernel: each chain
// runs only in its drawn direction.
#include <rocblas/rocblas.h>

#include "l2hmc_kernels.hpp"

namespace l2hmc {

rocblas_handle g_blas = nullptr;

int blas_handle(hipStream_t s, rocblas_handle* out) {
  if (g_blas == nullptr) {
    if (rocblas_create_handle(&g_blas) != rocblas_status_success) return fail(L2HMC_ERR_HIP, "rocblas_create_handle failed%s");
    rocblas_set_pointer_mode(g_blas, rocblas_pointer_mode_host);
  }
  if (rocblas_set_stream(g_blas, s) != rocblas_status_success) return fail(L2HMC_ERR_HIP, "rocblas_set_stream failed%s");
  *out = g_blas;
  return L2HMC_OK;
}

// row-major C[M x N] (ldc) = A[M x K] (lda) . op(B) + beta C;  op(B) = B (K x N, ldb) or, with
// transB, B^T for B stored (N x K, ldb).  (row-major X is the column-major X^T.)
int gemm_rm(rocblas_handle h, bool transB, int M, int N, int K, const float* A, int lda, const float* B,
            int ldb, float* C, int ldc, float beta) {
  const float one = 1.f;
  const rocblas_status st = rocblas_sgemm(h, transB ? rocblas_operation_transpose : rocblas_operation_none,
                                          rocblas_operation_none, N, M, K, &one, B, ldb, A, lda, &beta, C, ldc);
  if (st != rocblas_status_success) return fail(L2HMC_ERR_HIP, "rocblas_sgemm failed (status %s%lld)", "", (long long)st);
  return L2HMC_OK;
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// pre[n][j] += b[j]; act[n][j] = softplus(pre)      (pre kept for the backward pass)
__global__ void k_bias_softplus(float* pre, float* act, const float* b, long long n, int w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * w) return;
  const float p = pre[i] + b[i % w];
  pre[i] = p;
  act[i] = softplus_f(p);
}
__global__ void k_bias_add(float* x, const float* b, long long n, int w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * w) x[i] += b[i % w];
}
__global__ void k_bias_relu(float* x, const float* b, long long n, int w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * w) x[i] = fmaxf(x[i] + b[i % w], 0.f);
}
// dh[i] *= sigmoid(pre[i])      (softplus' = sigmoid)
__global__ void k_mul_sigmoid(float* dh, const float* pre, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dh[i] *= sigmoid_f(pre[i]);
}
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
// h1[n][j] = relu(h1pre + tb[net][row(n)][j] + aux_h[n][j])
__global__ void k_layer1(float* h1, const float* tb_net, const float* aux_h, const unsigned char* dir,
                         int dir_all, int it, int T, long long n, int H) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * H) return;
  bool fwd;
  const int s = row_of(dir, dir_all, i / H, it, T, fwd);
  h1[i] = fmaxf(h1[i] + tb_net[s * H + (int)(i % H)] + (aux_h ? aux_h[i] : 0.f), 0.f);
}

// BCE part of the VAE energy, one workgroup per chain: U[n] = beta sum_pix bce(logit, aux) + |z|^2 / 2;
// the logits are overwritten by d U / d logit = beta (sigmoid(logit) - aux).  beta = 1 except on the AIS
// bridge (utils/ais.py:46-47 with the N(0, I) initial energy of eval_vae.py:55-62: only the BCE term anneals).
__global__ __launch_bounds__(256) void k_vae_out(float* lg, const float* aux, const float* z, int n_pix, int d,
                                                 float* U, float beta) {
  __shared__ float part[4];
  const long long n = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < n_pix; k += 256) {
    const float l = lg[n * n_pix + k], t = aux[n * n_pix + k];
    acc += fmaxf(l, 0.f) - l * t + log1pf(expf(-fabsf(l)));      // TF's stable form (mnist_vae.py:124)
    lg[n * n_pix + k] = beta * (sigmoid_f(l) - t);
  }
  acc *= beta;
  for (int k = threadIdx.x; k < d; k += 256) acc += 0.5f * z[n * d + k] * z[n * d + k];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && U != nullptr) U[n] = (part[0] + part[1]) + (part[2] + part[3]);
}
// HMC mode (nets identically zero, dynamics.py:73-76): the generalised step is the plain leapfrog
//   v_h = v - (eps/2) g(x);  x' = x + eps v_h     [k_hmc_drift]      v' = v_h - (eps/2) g(x')   [k_hmc_kick]
__global__ void k_hmc_drift(float* x, const float* v, const float* g, float* vh, const float* alpha, float eps_host,
                            long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host;
  const float h = v[i] + 0.5f * eps * (-g[i]);
  vh[i] = h;
  x[i] = x[i] + eps * h;
}
__global__ void k_hmc_kick(float* v, const float* vh, const float* g, const float* alpha, float eps_host, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host;
  v[i] = vh[i] + 0.5f * eps * (-g[i]);
}
__global__ void k_add(float* g, const float* z, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] += z[i];
}

// Momentum half-update, one thread per chain (deterministic log-det sum).  out3 = h2 [Ws|Wt|Wq]
// (no biases yet).  forward: v' = v e^{eps S/2} + (eps/2)(T - e^{eps Q} g); backward:
// v' = (v - (eps/2)(T - e^{eps Q} g)) e^{-eps S/2}   (dynamics.py:121-125,149-153 / :164-170,194-199)
__global__ void k_v_half(const float* out3, L2hmcNet w, const float* vin, const float* g, float* vout,
                         float* ld, const unsigned char* dir, int dir_all, const float* alpha, float eps_host,
                         long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const bool fwd = dir != nullptr ? dir[n] != 0 : (dir_all != 0);
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host, heps = 0.5f * eps, sgn = fwd ? 1.f : -1.f;
  float acc = 0.f;
  for (int k = 0; k < d; ++k) {
    const float S = expf(w.lam_s[k]) * tanhf(out3[n * 3 * d + k] + w.bs[k]);
    const float T = out3[n * 3 * d + d + k] + w.bt[k];
    const float Q = expf(w.lam_q[k]) * tanhf(out3[n * 3 * d + 2 * d + k] + w.bq[k]);
    const float sv = sgn * heps * S, ES = expf(sv), EQ = expf(eps * Q);
    const float cc = heps * (T - EQ * g[n * d + k]);
    const float vi = vin[n * d + k];
    vout[n * d + k] = fwd ? vi * ES + cc : (vi - cc) * ES;
    acc += sv;
  }
  ld[n] += acc;
}
// Masked position update + the masked input of the NEXT net evaluation.  second = 0: keeps
// k1 = (fwd ? m : 1-m) and emits xin = (1-k1) z'; second = 1: keeps 1-k1.
// (dynamics.py:131-145 / :176-190)
__global__ void k_x_half(const float* out3, L2hmcNet w, const float* zin, const float* vh, float* zout,
                         float* xin_next, float* ld, const float* masks, const unsigned char* dir, int dir_all,
                         int it, int T, int second, const float* alpha, float eps_host, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  bool fwd;
  const int s = row_of(dir, dir_all, n, it, T, fwd);
  const float eps = alpha != nullptr ? expf(*alpha) : eps_host, sgn = fwd ? 1.f : -1.f;
  float acc = 0.f;
  for (int k = 0; k < d; ++k) {
    const float m = masks[s * d + k];
    const float k1 = fwd ? m : 1.f - m;
    const float kp = second ? 1.f - k1 : k1, up = 1.f - kp;
    const float S = expf(w.lam_s[k]) * tanhf(out3[n * 3 * d + k] + w.bs[k]);
    const float T_ = out3[n * 3 * d + d + k] + w.bt[k];
    const float Q = expf(w.lam_q[k]) * tanhf(out3[n * 3 * d + 2 * d + k] + w.bq[k]);
    const float sx = sgn * eps * S, ES = expf(sx), EQ = expf(eps * Q);
    const float tr = eps * (EQ * vh[n * d + k] + T_);
    const float zi = zin[n * d + k];
    const float nw = fwd ? zi * ES + tr : ES * (zi - tr);
    const float zo = kp * zi + up * nw;
    zout[n * d + k] = zo;
    if (xin_next != nullptr) xin_next[n * d + k] = up * zo;      // next kept mask = this update mask
    acc += up * sx;
  }
  ld[n] += acc;
}
// xin = k1 * x for the first XNet evaluation of a step
__global__ void k_mask_first(const float* x, float* xin, const float* masks, const unsigned char* dir, int dir_all,
                             int it, int T, long long N, int d) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  bool fwd;
  const int s = row_of(dir, dir_all, i / d, it, T, fwd);
  const float m = masks[s * d + (int)(i % d)];
  xin[i] = (fwd ? m : 1.f - m) * x[i];
}
__global__ void k_kinetic(const float* v, float* K, float* ld, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < d; ++k) acc += 0.5f * v[n * d + k] * v[n * d + k];
  K[n] = acc;
  if (ld != nullptr) ld[n] = 0.f;
}
// accept probability (dynamics.py:302-309) + MH select (sampler.py:53-55)
__global__ void k_finish(const float* U0, const float* K0, const float* U1, const float* K1, const float* ld,
                         const float* u, const float* x0, const float* x1, float* p_out, float* logjac_out,
                         float* x_next, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float p = accept_prob((U0[n] + K0[n]) - (U1[n] + K1[n]) + ld[n]);
  if (p_out != nullptr) p_out[n] = p;
  if (logjac_out != nullptr) logjac_out[n] = ld[n];
  if (x_next != nullptr) {
    const bool acc = (p - u[n]) >= 0.f;
    for (int k = 0; k < d; ++k) x_next[n * d + k] = acc ? x1[n * d + k] : x0[n * d + k];
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

struct Mlp3Ws { float *p1, *a1, *p2, *a2; };

// out (N x n_out) = Linear-softplus-Linear-softplus-Linear(x); pre-activations kept in ws
int mlp3_forward(rocblas_handle h, hipStream_t s, const L2hmcMlp3& m, const float* x, long long N, const Mlp3Ws& ws,
                 float* out) {
  int rc;
  if ((rc = gemm_rm(h, false, (int)N, m.n_h1, m.n_in, x, m.n_in, m.W1, m.n_h1, ws.p1, m.n_h1, 0.f))) return rc;
  hipLaunchKernelGGL(k_bias_softplus, dim3(nblk(N * m.n_h1)), dim3(256), 0, s, ws.p1, ws.a1, m.b1, N, m.n_h1);
  if ((rc = gemm_rm(h, false, (int)N, m.n_h2, m.n_h1, ws.a1, m.n_h1, m.W2, m.n_h2, ws.p2, m.n_h2, 0.f))) return rc;
  hipLaunchKernelGGL(k_bias_softplus, dim3(nblk(N * m.n_h2)), dim3(256), 0, s, ws.p2, ws.a2, m.b2, N, m.n_h2);
  if ((rc = gemm_rm(h, false, (int)N, m.n_out, m.n_h2, ws.a2, m.n_h2, m.W3, m.n_out, out, m.n_out, 0.f))) return rc;
  hipLaunchKernelGGL(k_bias_add, dim3(nblk(N * m.n_out)), dim3(256), 0, s, out, m.b3, N, m.n_out);
  return L2HMC_OK;
}

// U (N) and grad (N x d) of the VAE latent posterior at z; lg is an (N x n_pix) scratch
int vae_energy(rocblas_handle h, hipStream_t s, const L2hmcMlp3& dec, const float* aux, const float* z, long long N,
               int d, const Mlp3Ws& ws, float* lg, float* U, float* grad, float beta = 1.f) {
  int rc;
  if ((rc = mlp3_forward(h, s, dec, z, N, ws, lg))) return rc;
  hipLaunchKernelGGL(k_vae_out, dim3((unsigned)N), dim3(256), 0, s, lg, aux, z, dec.n_out, d, U, beta);
  if (grad == nullptr) return L2HMC_OK;
  // d a2 = dl W3^T (.) sigmoid(p2);  d a1 = d a2 W2^T (.) sigmoid(p1);  d z = d a1 W1^T + z
  if ((rc = gemm_rm(h, true, (int)N, dec.n_h2, dec.n_out, lg, dec.n_out, dec.W3, dec.n_out, ws.a2, dec.n_h2, 0.f))) return rc;
  hipLaunchKernelGGL(k_mul_sigmoid, dim3(nblk(N * dec.n_h2)), dim3(256), 0, s, ws.a2, ws.p2, N * dec.n_h2);
  if ((rc = gemm_rm(h, true, (int)N, dec.n_h1, dec.n_h2, ws.a2, dec.n_h2, dec.W2, dec.n_h2, ws.a1, dec.n_h1, 0.f))) return rc;
  hipLaunchKernelGGL(k_mul_sigmoid, dim3(nblk(N * dec.n_h1)), dim3(256), 0, s, ws.a1, ws.p1, N * dec.n_h1);
  if ((rc = gemm_rm(h, true, (int)N, d, dec.n_h1, ws.a1, dec.n_h1, dec.W1, dec.n_h1, grad, d, 0.f))) return rc;
  hipLaunchKernelGGL(k_add, dim3(nblk(N * d)), dim3(256), 0, s, grad, z, N * d);
  return L2HMC_OK;
}

struct SplitPlan {
  long long total;
  long long xc, vc, g, vh, y, xin, h1, h2, out3, aux_h, tb, U0, K0, U1, K1, ld, p1, a1, p2, a2, lg, e1, e1a, e2, e2a;
};

SplitPlan plan_split(long long N, int d, int H, int T, const L2hmcMlp3* enc, const L2hmcMlp3* dec) {
  SplitPlan p;
  long long o = 0;
  auto take = [&](long long n) { const long long at = o; o += (n + 3) & ~3LL; return at; };
  p.xc = take(N * d); p.vc = take(N * d); p.g = take(N * d); p.vh = take(N * d); p.y = take(N * d);
  p.xin = take(N * d); p.h1 = take(N * H); p.h2 = take(N * H); p.out3 = take(N * 3 * d);
  p.aux_h = take(enc ? N * H : 0); p.tb = take(2LL * T * H);
  p.U0 = take(N); p.K0 = take(N); p.U1 = take(N); p.K1 = take(N); p.ld = take(N);
  p.p1 = take(N * dec->n_h1); p.a1 = take(N * dec->n_h1); p.p2 = take(N * dec->n_h2); p.a2 = take(N * dec->n_h2);
  p.lg = take(N * dec->n_out);
  p.e1 = take(enc ? N * enc->n_h1 : 0); p.e1a = take(enc ? N * enc->n_h1 : 0);
  p.e2 = take(enc ? N * enc->n_h2 : 0); p.e2a = take(enc ? N * enc->n_h2 : 0);
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
  if (n_chains < 0 || d < 1 || H < 1 || T < 1 || !decoder) return fail(L2HMC_ERR_ARG, "l2hmc_split_workspace_floats: bad argument%s");
  return plan_split(n_chains, d, H, T, aux_encoder, decoder).total;
}

int l2hmc_vae_energy(const L2hmcMlp3* decoder, const float* aux, const float* x, int64_t n_chains, int32_t d,
                     float* U_out, float* grad_out, float* workspace, float bce_scale, void* stream) {
  int rc = check_mlp(decoder, "l2hmc_vae_energy");
  if (rc) return rc;
  if (!aux || !x || !workspace || n_chains < 0 || d != decoder->n_in) return fail(L2HMC_ERR_ARG, "l2hmc_vae_energy: bad argument%s");
  if (!(bce_scale >= 0.f && bce_scale <= 1.f)) return fail(L2HMC_ERR_ARG, "bce_scale must be in [0, 1] (0 = off)%s");
  const float beta = bce_scale > 0.f ? bce_scale : 1.f;
  if (n_chains == 0) return L2HMC_OK;
  hipStream_t s = (hipStream_t)stream;
  rocblas_handle h;
  if ((rc = blas_handle(s, &h))) return rc;
  const SplitPlan p = plan_split(n_chains, d, 1, 1, nullptr, decoder);
  float* w = workspace;
  const Mlp3Ws ws = {w + p.p1, w + p.a1, w + p.p2, w + p.a2};
  if ((rc = vae_energy(h, s, *decoder, aux, x, n_chains, d, ws, w + p.lg, U_out, grad_out, beta))) return rc;
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
  int rc = check_mlp(a->decoder, "decoder");
  if (rc) return rc;
  if (a->aux_encoder && (rc = check_mlp(a->aux_encoder, "aux_encoder"))) return rc;
  const long long N = a->n_chains;
  const int d = a->d, H = a->H, T = a->T;
  if (N < 0 || d < 1 || H < 1 || T < 1) return fail(L2HMC_ERR_ARG, "bad n_chains / d / H / T%s");
  if (N == 0) return L2HMC_OK;
  const bool hmc = a->hmc != 0;
  if ((!hmc && (!a->xnet || !a->vnet || !a->masks || !a->trig)) || !a->aux || !a->x || !a->v || !a->workspace)
    return fail(L2HMC_ERR_ARG, "l2hmc_trajectory_split: NULL pointer%s");
  if (hmc && (a->direction != nullptr || a->direction_all == 0))
    return fail(L2HMC_ERR_UNSUPPORTED, "HMC mode runs forward only (sampler.py:29-31, ais.py:61)%s");
  if (!(a->bce_scale >= 0.f && a->bce_scale <= 1.f)) return fail(L2HMC_ERR_ARG, "bce_scale must be in [0, 1] (0 = off)%s");
  const float beta = a->bce_scale > 0.f ? a->bce_scale : 1.f;
  if (a->decoder->n_in != d) return fail(L2HMC_ERR_ARG, "decoder input width != d%s");
  if (a->aux_encoder && (a->aux_encoder->n_out != H || a->aux_encoder->n_in != a->decoder->n_out))
    return fail(L2HMC_ERR_ARG, "aux_encoder must map (N, n_pix) -> (N, H)%s");
  if (a->step_begin < 0 || a->n_steps < 0 || a->step_begin + a->n_steps > T) return fail(L2HMC_ERR_ARG, "steps outside the schedule%s");
  if (a->x_next && !a->u) return fail(L2HMC_ERR_ARG, "x_next needs u%s");
  if (!a->alpha && !(a->eps_host > 0.f)) return fail(L2HMC_ERR_ARG, "eps must be > 0%s");
  const SplitPlan p = plan_split(N, d, H, T, a->aux_encoder, a->decoder);
  if (a->workspace_floats < p.total) return fail(L2HMC_ERR_ARG, "workspace too small: need %s%lld floats", "", p.total);
  hipStream_t s = (hipStream_t)stream;
  rocblas_handle h;
  if ((rc = blas_handle(s, &h))) return rc;
  float* w = a->workspace;
  const Mlp3Ws dws = {w + p.p1, w + p.a1, w + p.p2, w + p.a2};
  float *xc = w + p.xc, *vc = w + p.vc, *g = w + p.g, *vh = w + p.vh, *y = w + p.y, *xin = w + p.xin;
  float *h1 = w + p.h1, *h2 = w + p.h2, *out3 = w + p.out3, *tb = w + p.tb, *ld = w + p.ld;
  float* aux_h = a->aux_encoder ? w + p.aux_h : nullptr;
  static const L2hmcNet no_net = {};
  const L2hmcNet &xn = hmc ? no_net : *a->xnet, &vn = hmc ? no_net : *a->vnet;
  const unsigned char* dir = a->direction;
  const int dall = a->direction_all;

  (void)hipMemcpyAsync(xc, a->x, sizeof(float) * N * d, hipMemcpyDeviceToDevice, s);
  (void)hipMemcpyAsync(vc, a->v, sizeof(float) * N * d, hipMemcpyDeviceToDevice, s);
  if (a->aux_encoder && !hmc) {      // the image branch is step-invariant: once per trajectory, not 4T times
    const Mlp3Ws ews = {w + p.e1, w + p.e1a, w + p.e2, w + p.e2a};
    if ((rc = mlp3_forward(h, s, *a->aux_encoder, a->aux, N, ews, aux_h))) return rc;
  }
  if (!hmc) hipLaunchKernelGGL(k_time_table, dim3(nblk(2LL * T * H)), dim3(256), 0, s, xn, vn, a->trig, T, H, tb);
  hipLaunchKernelGGL(k_kinetic, dim3(nblk(N)), dim3(256), 0, s, vc, w + p.K0, ld, N, d);
  if ((rc = vae_energy(h, s, *a->decoder, a->aux, xc, N, d, dws, w + p.lg, w + p.U0, g, beta))) return rc;
  if (a->n_steps == 0) (void)hipMemcpyAsync(w + p.U1, w + p.U0, sizeof(float) * N, hipMemcpyDeviceToDevice, s);

  // one net evaluation: out3 = relu(relu(a W1 + b W2 + time + aux_h) W4 + b4) [Ws|Wt|Wq]
  auto net_eval = [&](const L2hmcNet& nw, int net, const float* ain, const float* bin, int it) -> int {
    int r;
    if ((r = gemm_rm(h, false, (int)N, H, d, ain, d, nw.W1, H, h1, H, 0.f))) return r;
    if ((r = gemm_rm(h, false, (int)N, H, d, bin, d, nw.W2, H, h1, H, 1.f))) return r;
    hipLaunchKernelGGL(k_layer1, dim3(nblk(N * H)), dim3(256), 0, s, h1, tb + (long long)net * T * H, aux_h, dir, dall,
                       it, T, N, H);
    if ((r = gemm_rm(h, false, (int)N, H, H, h1, H, nw.W4, H, h2, H, 0.f))) return r;
    hipLaunchKernelGGL(k_bias_relu, dim3(nblk(N * H)), dim3(256), 0, s, h2, nw.b4, N, H);
    if ((r = gemm_rm(h, false, (int)N, d, H, h2, H, nw.Ws, d, out3, 3 * d, 0.f))) return r;
    if ((r = gemm_rm(h, false, (int)N, d, H, h2, H, nw.Wt, d, out3 + d, 3 * d, 0.f))) return r;
    if ((r = gemm_rm(h, false, (int)N, d, H, h2, H, nw.Wq, d, out3 + 2 * d, 3 * d, 0.f))) return r;
    return L2HMC_OK;
  };

  for (int k = 0; k < a->n_steps; ++k) {
    const int it = a->step_begin + k;
    const bool last = k == a->n_steps - 1;
    if (hmc) {
      hipLaunchKernelGGL(k_hmc_drift, dim3(nblk(N * d)), dim3(256), 0, s, xc, vc, g, vh, a->alpha, a->eps_host, N * d);
      if ((rc = vae_energy(h, s, *a->decoder, a->aux, xc, N, d, dws, w + p.lg, last ? w + p.U1 : nullptr, g, beta))) return rc;
      hipLaunchKernelGGL(k_hmc_kick, dim3(nblk(N * d)), dim3(256), 0, s, vc, vh, g, a->alpha, a->eps_host, N * d);
      continue;
    }
    if ((rc = net_eval(vn, 1, xc, g, it))) return rc;
    hipLaunchKernelGGL(k_v_half, dim3(nblk(N)), dim3(256), 0, s, out3, vn, vc, g, vh, ld, dir, dall, a->alpha,
                       a->eps_host, N, d);
    hipLaunchKernelGGL(k_mask_first, dim3(nblk(N * d)), dim3(256), 0, s, xc, xin, a->masks, dir, dall, it, T, N, d);
    if ((rc = net_eval(xn, 0, vh, xin, it))) return rc;
    hipLaunchKernelGGL(k_x_half, dim3(nblk(N)), dim3(256), 0, s, out3, xn, xc, vh, y, xin, ld, a->masks, dir, dall, it, T,
                       0, a->alpha, a->eps_host, N, d);
    if ((rc = net_eval(xn, 0, vh, xin, it))) return rc;
    hipLaunchKernelGGL(k_x_half, dim3(nblk(N)), dim3(256), 0, s, out3, xn, y, vh, xc, (float*)nullptr, ld, a->masks, dir,
                       dall, it, T, 1, a->alpha, a->eps_host, N, d);
    if ((rc = vae_energy(h, s, *a->decoder, a->aux, xc, N, d, dws, w + p.lg, last ? w + p.U1 : nullptr, g, beta))) return rc;
    if ((rc = net_eval(vn, 1, xc, g, it))) return rc;
    hipLaunchKernelGGL(k_v_half, dim3(nblk(N)), dim3(256), 0, s, out3, vn, vh, g, vc, ld, dir, dall, a->alpha,
                       a->eps_host, N, d);
  }
  hipLaunchKernelGGL(k_kinetic, dim3(nblk(N)), dim3(256), 0, s, vc, w + p.K1, (float*)nullptr, N, d);
  if (a->x_out) (void)hipMemcpyAsync(a->x_out, xc, sizeof(float) * N * d, hipMemcpyDeviceToDevice, s);
  if (a->v_out) (void)hipMemcpyAsync(a->v_out, vc, sizeof(float) * N * d, hipMemcpyDeviceToDevice, s);
  hipLaunchKernelGGL(k_finish, dim3(nblk(N)), dim3(256), 0, s, w + p.U0, w + p.K0, w + p.U1, w + p.K1, ld, a->u, a->x,
                     xc, a->p_out, a->logjac_out, a->x_next, N, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

}  // extern "C"
