// libl2hmc_hip.so -- the L2HMC generalised-leapfrog hot path for MI355X (gfx950 / CDNA4).
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
//     the end with two wave shuffles (+ one LDS hop when NW = 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/l2hmc.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

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

inline int tiles_of(int d) { return (d + 15) / 16; }
inline int khid_of(int H) { return (H + 1 + 3) / 4; }

__global__ void pack_net_kernel(L2hmcNet net, int d, int H, int KH, int NT, float* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int ngf = net_groups(NT) * 256;
  if (idx >= net_floats(NT)) return;
  float val = 0.f;
  if (idx < ngf) {
    const int g = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3;
    const int i = lane & 15, q = lane >> 4;
    const int ui = ((i & 3) < KH) ? (i >> 2) * KH + (i & 3) : -1;  // unit on output row i
    const int uk = (r < KH) ? q * KH + r : -1;                     // unit on k index (q, r)
    if (g < 2 * NT) {                                              // layer 1: embeds of a / b
      const int tg = g < NT ? g : g - NT;
      const float* W = g < NT ? net.W1 : net.W2;
      const int dim = 16 * tg + 4 * q + r;
      if (dim < d && ui >= 0 && ui < H) val = W[dim * H + ui];
    } else if (g == 2 * NT) {                                      // time embed + biases
      if (r == 0 && ui >= 0) {
        if (q == 0 && ui < H) val = net.W3[ui];
        if (q == 1 && ui < H) val = net.W3[H + ui];
        if (q == 2) val = ui < H ? (net.b1[ui] + net.b2[ui]) + net.b3[ui] : (ui == H ? 1.f : 0.f);
      }
    } else if (g == 2 * NT + 1) {                                  // layer 2 (+ b4, + 1 -> 1)
      if (uk >= 0 && ui >= 0) {
        if (uk < H && ui < H) val = net.W4[uk * H + ui];
        else if (uk == H && ui < H) val = net.b4[ui];
        else if (uk == H && ui == H) val = 1.f;
      }
    } else {                                                       // heads S, T, Q
      const int hg = g - (2 * NT + 2), tg = hg / 3, h = hg % 3;
      const float* W = h == 0 ? net.Ws : (h == 1 ? net.Wt : net.Wq);
      const float* b = h == 0 ? net.bs : (h == 1 ? net.bt : net.bq);
      const int dim = 16 * tg + i;
      if (dim < d && uk >= 0) {
        if (uk < H) val = W[uk * d + dim];
        else if (uk == H) val = b[dim];
      }
    }
  } else {                                                          // exp(log-scale) of ScaleTanh
    const int j = idx - ngf, which = j / (16 * NT), dim = j % (16 * NT);
    const float* lam = which == 0 ? net.lam_s : net.lam_q;
    if (dim < d) val = expf(lam[dim]);
  }
  out[idx] = val;
}

__global__ void pack_gauss_kernel(const float* S, int d, int NT, float* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gauss_floats(NT)) return;
  const int g = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3;
  const int to = g / NT, ti = g % NT;
  const int a = 16 * to + (lane & 15), b = 16 * ti + 4 * (lane >> 4) + r;
  out[idx] = (a < d && b < d) ? 0.5f * (S[a * d + b] + S[b * d + a]) : 0.f;
}

// ------------------------------------------------------------------------------------------
// Device-side argument block
// ------------------------------------------------------------------------------------------
struct KArgs {
  const float* packed;
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
  float *x_out, *v_out, *logjac_out, *p_out, *x_next;
  // energy
  int ekind, ncomp, easy;
  const float *mu, *prec, *logc;
  float eta, temperature;
  // p_accept-only kernel inputs
  const float *x1, *v1, *logjac_in;
  float *U_out, *grad_out;
  // LDS offsets (floats)
  int o_mask, o_trig, o_P, o_XB, o_red, o_mu, o_prec, o_logc, xb_stride;
};

__device__ __forceinline__ f4 splat(float a) { return f4{a, a, a, a}; }
__device__ __forceinline__ f4 exp4(f4 a) { return f4{expf(a.x), expf(a.y), expf(a.z), expf(a.w)}; }
__device__ __forceinline__ f4 tanh4(f4 a) { return f4{tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w)}; }
__device__ __forceinline__ f4 relu4(f4 a) {
  return f4{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)};
}
__device__ __forceinline__ float hsum(f4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ f4 sel4(bool c, f4 a, f4 b) { return c ? a : b; }
__device__ __forceinline__ f4 lds4(const float* p) { return *reinterpret_cast<const f4*>(p); }

// Sum over the lanes / waves that hold one chain; every lane of the chain gets the total.
template <int NW, int NV>
__device__ __forceinline__ void chain_allreduce(float (&v)[NV], float* red, int w, int lane) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] += __shfl_xor(v[i], 16);
    v[i] += __shfl_xor(v[i], 32);
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
            for (int r = 0; r < 4; ++r)
              if (16 * ti + r < d) acc = MFMA16(G[r], dx[ti][r], acc);
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
          for (int r = 0; r < 4; ++r)
            if (16 * ti + r < d) acc = MFMA16(G[r], B[r], acc);
        }
      }
      y[t] = acc;
    }
    __syncthreads();
  }
}

// grad U (S-layout) and this lane's share of U (summing `Upart` over the chain's lanes
// gives U).  dynamics.py:203-218 with the energies of distributions.py.
template <int DT, int NW>
__device__ __forceinline__ void grad_energy(const KArgs& A, float* smem, int w, int lane,
                                            const f4 (&x)[DT], f4 (&g)[DT], float& Upart,
                                            bool wantU) {
  const int q = lane >> 4, DP = 16 * A.NT;
  float U = 0.f;
  switch (A.ekind) {
    case L2HMC_ENERGY_GAUSS_DIAG: {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int off = 16 * (w * DT + t) + 4 * q;
        const bool ok = (w * DT + t) < A.NT;
        const f4 mu = ok ? lds4(smem + A.o_mu + off) : splat(0.f);
        const f4 s = ok ? lds4(smem + A.o_prec + off) : splat(0.f);
        const f4 dx = x[t] - mu;
        g[t] = s * dx;
        U += 0.5f * hsum(dx * g[t]);
      }
    } break;
    case L2HMC_ENERGY_GAUSS_DENSE: {
      f4 dx[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const bool ok = (w * DT + t) < A.NT;
        const f4 mu = ok ? lds4(smem + A.o_mu + 16 * (w * DT + t) + 4 * q) : splat(0.f);
        dx[t] = x[t] - mu;
      }
      dense_matvec<DT, NW>(smem + A.o_prec, A, smem, w, lane, dx, g);
#pragma unroll
      for (int t = 0; t < DT; ++t) U += 0.5f * hsum(dx[t] * g[t]);
    } break;
    case L2HMC_ENERGY_GMM: {
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
        dense_matvec<DT, NW>(smem + A.o_prec + i * gauss_floats(A.NT), A, smem, w, lane, dx, y);
        float qq[1] = {0.f};
#pragma unroll
        for (int t = 0; t < DT; ++t) qq[0] += hsum(dx[t] * y[t]);
        chain_allreduce<NW, 1>(qq, smem + A.o_red, w, lane);
        const float V = -(0.5f * qq[0]) + smem[A.o_logc + i];
        const float mn = fmaxf(m, V);
        const float sc = expf(m - mn), wi = expf(V - mn);
        ssum = ssum * sc + wi;
#pragma unroll
        for (int t = 0; t < DT; ++t) gacc[t] = gacc[t] * sc + wi * y[t];
        m = mn;
      }
      const float inv = 1.f / ssum;
#pragma unroll
      for (int t = 0; t < DT; ++t) g[t] = gacc[t] * inv;
      if (w == 0 && lane < 16) U = -(m + logf(ssum));
    } break;
    case L2HMC_ENERGY_ROUGHWELL: {
      const float eta = A.eta;
      const float den = A.easy ? eta : eta * eta;
      const float scale = eta / den;
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        f4 arg = x[t] / den;
        g[t] = x[t] - scale * f4{sinf(arg.x), sinf(arg.y), sinf(arg.z), sinf(arg.w)};
        if (wantU) {
          // padded dims hold x = 0 and would add eta * cos(0): mask them out
          const int dim0 = 16 * (w * DT + t) + 4 * q;
          f4 cs = f4{cosf(arg.x), cosf(arg.y), cosf(arg.z), cosf(arg.w)};
          f4 live = f4{dim0 < A.d ? 1.f : 0.f, dim0 + 1 < A.d ? 1.f : 0.f,
                       dim0 + 2 < A.d ? 1.f : 0.f, dim0 + 3 < A.d ? 1.f : 0.f};
          U += 0.5f * hsum(x[t] * x[t]) + eta * hsum(live * cs);
        }
      }
    } break;
    case L2HMC_ENERGY_FUNNEL: {
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
    } break;
    default:
#pragma unroll
      for (int t = 0; t < DT; ++t) g[t] = splat(0.f);
  }
  if (A.temperature != 1.f) {
    U = U / A.temperature;
#pragma unroll
    for (int t = 0; t < DT; ++t) g[t] = g[t] / A.temperature;
  }
  Upart = U;
}

// [S, T, Q] = net([a, b, tau]) for every tile of this wave; `apply(t, S, T, Q)` consumes one
// 16-dimension tile at a time so the head outputs never live longer than a tile.
template <int DT, int NW, int KH, class F>
__device__ __forceinline__ void net_eval(const float* wn, const KArgs& A, float* smem, int w,
                                         int lane, const f4 (&a)[DT], const f4 (&b)[DT],
                                         float tauB, int& pb, F&& apply) {
  const int NT = A.NT, d = A.d, q = lane >> 4;
  f4 acc0 = splat(0.f), acc1 = splat(0.f);
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int tg = w * DT + t;
    if (16 * tg < d) {
      const f4 Wa = lds4(wn + (tg * 64 + lane) * 4);
      const f4 Wb = lds4(wn + ((NT + tg) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (16 * tg + r < d) {
          acc0 = MFMA16(Wa[r], a[t][r], acc0);
          acc1 = MFMA16(Wb[r], b[t][r], acc1);
        }
      }
    }
  }
  if (w == NW - 1) acc0 = MFMA16(wn[(2 * NT * 64 + lane) * 4], tauB, acc0);
  f4 h = acc0 + acc1;
  if (NW > 1) {  // K-split over waves: exchange partial pre-activations through LDS
    float* P = smem + A.o_P + pb * (NW * 256);
    *reinterpret_cast<f4*>(P + (w * 64 + lane) * 4) = h;
    __syncthreads();
    h = lds4(P + lane * 4);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) h += lds4(P + (ww * 64 + lane) * 4);
    pb ^= 1;
  }
  h = relu4(h);
  {
    const f4 W2 = lds4(wn + ((2 * NT + 1) * 64 + lane) * 4);
    f4 acc = splat(0.f);
#pragma unroll
    for (int r = 0; r < KH; ++r) acc = MFMA16(W2[r], h[r], acc);
    h = relu4(acc);
  }
  const float* sc = wn + net_groups(NT) * 256;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const int tg = w * DT + t;
    if (16 * tg < d) {
      const f4 Ws = lds4(wn + ((2 * NT + 2 + 3 * tg + 0) * 64 + lane) * 4);
      const f4 Wt = lds4(wn + ((2 * NT + 2 + 3 * tg + 1) * 64 + lane) * 4);
      const f4 Wq = lds4(wn + ((2 * NT + 2 + 3 * tg + 2) * 64 + lane) * 4);
      f4 zs = splat(0.f), zt = splat(0.f), zq = splat(0.f);
#pragma unroll
      for (int r = 0; r < KH; ++r) {
        zs = MFMA16(Ws[r], h[r], zs);
        zt = MFMA16(Wt[r], h[r], zt);
        zq = MFMA16(Wq[r], h[r], zq);
      }
      const f4 es = lds4(sc + 16 * tg + 4 * q);
      const f4 eq = lds4(sc + 16 * NT + 16 * tg + 4 * q);
      apply(t, es * tanh4(zs), zt, eq * tanh4(zq));
    } else {
      apply(t, splat(0.f), splat(0.f), splat(0.f));
    }
  }
}

// One momentum half-update.  forward (dynamics.py:121-125,149-153):
//   v' = v e^{eps S / 2} + (eps/2)(T - e^{eps Q} grad);   backward (:164-170,194-199):
//   v' = (v - (eps/2)(T - e^{eps Q} grad)) e^{-eps S / 2}.   `ld` accumulates log|det|.
__device__ __forceinline__ f4 v_half(f4 vin, f4 g, f4 S, f4 T, f4 Q, float eps, float heps,
                                     float sgn, bool fwd, float& ld) {
  const f4 sv = (sgn * heps) * S;
  const f4 e = exp4(sv);
  const f4 cc = heps * (T - exp4(eps * Q) * g);
  ld += hsum(sv);
  return sel4(fwd, vin * e + cc, (vin - cc) * e);
}

// One masked position update; `kp` = kept coordinates (0/1).  forward (dynamics.py:131-145):
//   z' = kp z + (1-kp)(z e^{eps S} + eps (e^{eps Q} v_h + T));   backward (:176-190):
//   z' = kp z + (1-kp) e^{-eps S} (z - eps (e^{eps Q} v_h + T)).
__device__ __forceinline__ f4 x_half(f4 zin, f4 kp, f4 vh, f4 S, f4 T, f4 Q, float eps,
                                     float sgn, bool fwd, float& ld) {
  const f4 up = splat(1.f) - kp;
  const f4 sx = (sgn * eps) * S;
  const f4 e = exp4(sx);
  const f4 tr = eps * (exp4(eps * Q) * vh + T);
  const f4 nw = sel4(fwd, zin * e + tr, e * (zin - tr));
  ld += hsum(up * sx);
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
      const float* row = p + chain * A.d + dim0;
      if (dim0 + 0 < A.d) r.x = row[0];
      if (dim0 + 1 < A.d) r.y = row[1];
      if (dim0 + 2 < A.d) r.z = row[2];
      if (dim0 + 3 < A.d) r.w = row[3];
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
    if (dim0 + 0 < A.d) row[0] = z[t].x;
    if (dim0 + 1 < A.d) row[1] = z[t].y;
    if (dim0 + 2 < A.d) row[2] = z[t].z;
    if (dim0 + 3 < A.d) row[3] = z[t].w;
  }
}

// Stage energy parameters into LDS (padded with zeros to 16*NT dims).
__device__ __forceinline__ void stage_energy(const KArgs& A, float* smem, int tid, int nthr) {
  const int DP = 16 * A.NT;
  const int nc = A.ekind == L2HMC_ENERGY_GMM ? A.ncomp : 1;
  if (A.ekind == L2HMC_ENERGY_GAUSS_DIAG || A.ekind == L2HMC_ENERGY_GAUSS_DENSE ||
      A.ekind == L2HMC_ENERGY_GMM) {
    for (int i = tid; i < nc * DP; i += nthr) {
      const int comp = i / DP, dim = i % DP;
      smem[A.o_mu + i] = dim < A.d ? A.mu[comp * A.d + dim] : 0.f;
    }
  }
  if (A.ekind == L2HMC_ENERGY_GAUSS_DIAG) {
    for (int i = tid; i < DP; i += nthr) smem[A.o_prec + i] = i < A.d ? A.prec[i] : 0.f;
  } else if (A.ekind == L2HMC_ENERGY_GAUSS_DENSE || A.ekind == L2HMC_ENERGY_GMM) {
    const int n4 = nc * gauss_floats(A.NT) / 4;
    const f4* src = reinterpret_cast<const f4*>(A.prec);
    f4* dst = reinterpret_cast<f4*>(smem + A.o_prec);
    for (int i = tid; i < n4; i += nthr) dst[i] = src[i];
    if (A.ekind == L2HMC_ENERGY_GMM)
      for (int i = tid; i < nc; i += nthr) smem[A.o_logc + i] = A.logc[i];
  }
}

// ------------------------------------------------------------------------------------------
// The fused trajectory kernel
// ------------------------------------------------------------------------------------------
template <int DT, int NW, int KH>
__global__ __launch_bounds__(64 * NW) void traj_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, nthr = 64 * NW;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  const int NT = A.NT, DP = 16 * NT;
  const bool has_nets = A.packed != nullptr;
  const int NF = net_floats(NT);

  // ---- prologue: stage weights / masks / time table / energy parameters into LDS ----------
  if (has_nets) {
    const f4* src = reinterpret_cast<const f4*>(A.packed);
    f4* dst = reinterpret_cast<f4*>(smem);
    for (int i = tid; i < 2 * NF / 4; i += nthr) dst[i] = src[i];
  }
  for (int i = tid; i < A.T * DP; i += nthr) {
    const int row = i / DP, dim = i % DP;
    smem[A.o_mask + i] = dim < A.d ? A.masks[row * A.d + dim] : 0.f;
  }
  for (int i = tid; i < 2 * A.T; i += nthr) smem[A.o_trig + i] = A.trig[i];
  stage_energy(A, smem, tid, nthr);

  f4 x[DT], v[DT], g[DT];
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  load_state<DT, NW>(A.v, A, chain, live, w, q, v);
  const bool fwd = A.dir != nullptr ? (live ? A.dir[chain] != 0 : true) : (A.dir_all != 0);
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const float sgn = fwd ? 1.f : -1.f;
  const bool need_p = A.p_out != nullptr || A.x_next != nullptr;
  __syncthreads();

  const float* wx = smem;        // XNet fragments
  const float* wv = smem + NF;   // VNet fragments
  int pb = 0;
  float red[5];                  // U0, K0, U1, K1, logdet (per-lane partial sums)
  red[1] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[1] += 0.5f * hsum(v[t] * v[t]);
  grad_energy<DT, NW>(A, smem, w, lane, x, g, red[0], need_p);
  red[2] = 0.f;
  float ld = 0.f;
  const f4 Z = splat(0.f);

  for (int it = 0; it < A.n_steps; ++it) {
    const int sf = A.step_begin + it;
    const int s = fwd ? sf : (A.T - 1 - sf);
    const float ct = smem[A.o_trig + 2 * s], st = smem[A.o_trig + 2 * s + 1];
    const float tauB = q == 0 ? ct : (q == 1 ? st : (q == 2 ? 1.f : 0.f));

    // ---- momentum half-update #1: VNet([x, grad U(x), t])  (dynamics.py:118-125 / :162-170)
    f4 vh[DT];
    if (has_nets) {
      net_eval<DT, NW, KH>(wv, A, smem, w, lane, x, g, tauB, pb, [&](int t, f4 S, f4 T, f4 Q) {
        vh[t] = v_half(v[t], g[t], S, T, Q, eps, heps, sgn, fwd, ld);
      });
    } else {
#pragma unroll
      for (int t = 0; t < DT; ++t) vh[t] = v_half(v[t], g[t], Z, Z, Z, eps, heps, sgn, fwd, ld);
    }

    // ---- two masked position updates: XNet([v_h, kept * x, t])  (:127-145 / :172-190)
    f4 k1[DT], xin[DT], y[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const bool ok = (w * DT + t) < NT;
      const f4 m = ok ? lds4(smem + A.o_mask + s * DP + 16 * (w * DT + t) + 4 * q) : Z;
      k1[t] = sel4(fwd, m, splat(1.f) - m);   // forward keeps m first, backward keeps 1-m first
      xin[t] = k1[t] * x[t];
    }
    if (has_nets) {
      net_eval<DT, NW, KH>(wx, A, smem, w, lane, vh, xin, tauB, pb, [&](int t, f4 S, f4 T, f4 Q) {
        y[t] = x_half(x[t], k1[t], vh[t], S, T, Q, eps, sgn, fwd, ld);
      });
#pragma unroll
      for (int t = 0; t < DT; ++t) xin[t] = (splat(1.f) - k1[t]) * y[t];
      net_eval<DT, NW, KH>(wx, A, smem, w, lane, vh, xin, tauB, pb, [&](int t, f4 S, f4 T, f4 Q) {
        x[t] = x_half(y[t], splat(1.f) - k1[t], vh[t], S, T, Q, eps, sgn, fwd, ld);
      });
    } else {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        y[t] = x_half(x[t], k1[t], vh[t], Z, Z, Z, eps, sgn, fwd, ld);
        x[t] = x_half(y[t], splat(1.f) - k1[t], vh[t], Z, Z, Z, eps, sgn, fwd, ld);
      }
    }

    // ---- momentum half-update #2 at the new position  (:147-153 / :192-199)
    grad_energy<DT, NW>(A, smem, w, lane, x, g, red[2], need_p && it == A.n_steps - 1);
    if (has_nets) {
      net_eval<DT, NW, KH>(wv, A, smem, w, lane, x, g, tauB, pb, [&](int t, f4 S, f4 T, f4 Q) {
        v[t] = v_half(vh[t], g[t], S, T, Q, eps, heps, sgn, fwd, ld);
      });
    } else {
#pragma unroll
      for (int t = 0; t < DT; ++t) v[t] = v_half(vh[t], g[t], Z, Z, Z, eps, heps, sgn, fwd, ld);
    }
  }

  // ---- epilogue: proposal, log-det, accept probability, MH select ---------------------------
  store_state<DT, NW>(A.x_out, A, chain, live, w, q, x);
  store_state<DT, NW>(A.v_out, A, chain, live, w, q, v);
  if (A.n_steps == 0) red[2] = red[0];
  red[3] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[3] += 0.5f * hsum(v[t] * v[t]);
  red[4] = ld;
  chain_allreduce<NW, 5>(red, smem + A.o_red, w, lane);
  const bool writer = live && w == 0 && lane < 16;
  if (A.logjac_out != nullptr && writer) A.logjac_out[chain] = red[4];
  if (need_p) {
    // dynamics.py:302-309
    const float e_new = red[2] + red[3], e_old = red[0] + red[1];
    const float val = e_old - e_new + red[4];
    float p = expf(fminf(val, 0.f));
    if (!(fabsf(p) <= 3.402823466e38f)) p = 0.f;   // non-finite -> 0
    if (A.p_out != nullptr && writer) A.p_out[chain] = p;
    if (A.x_next != nullptr && live) {
      const bool acc = (p - A.u[chain]) >= 0.f;      // sampler.py:53-55
      if (!acc) load_state<DT, NW>(A.x, A, chain, live, w, q, x);
      store_state<DT, NW>(A.x_next, A, chain, live, w, q, x);
    }
  }
}

// energy / grad only  (Dynamics.energy, Dynamics.grad_energy)
template <int DT, int NW>
__global__ __launch_bounds__(64 * NW) void energy_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  stage_energy(A, smem, tid, 64 * NW);
  f4 x[DT], g[DT];
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  __syncthreads();
  float U[1];
  grad_energy<DT, NW>(A, smem, w, lane, x, g, U[0], true);
  store_state<DT, NW>(A.grad_out, A, chain, live, w, q, g);
  chain_allreduce<NW, 1>(U, smem + A.o_red, w, lane);
  if (A.U_out != nullptr && live && w == 0 && lane < 16) A.U_out[chain] = U[0];
}

// p_accept on arbitrary end points  (Dynamics.p_accept)
template <int DT, int NW>
__global__ __launch_bounds__(64 * NW) void paccept_kernel(const KArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const long long chain = (long long)blockIdx.x * 16 + c;
  const bool live = chain < A.N;
  stage_energy(A, smem, tid, 64 * NW);
  f4 x[DT], v[DT], g[DT];
  float red[4];
  __syncthreads();
  load_state<DT, NW>(A.x, A, chain, live, w, q, x);
  load_state<DT, NW>(A.v, A, chain, live, w, q, v);
  grad_energy<DT, NW>(A, smem, w, lane, x, g, red[0], true);
  red[1] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[1] += 0.5f * hsum(v[t] * v[t]);
  load_state<DT, NW>(A.x1, A, chain, live, w, q, x);
  load_state<DT, NW>(A.v1, A, chain, live, w, q, v);
  grad_energy<DT, NW>(A, smem, w, lane, x, g, red[2], true);
  red[3] = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) red[3] += 0.5f * hsum(v[t] * v[t]);
  chain_allreduce<NW, 4>(red, smem + A.o_red, w, lane);
  if (live && w == 0 && lane < 16) {
    const float val = (red[0] + red[1]) - (red[2] + red[3]) + A.logjac_in[chain];
    float p = expf(fminf(val, 0.f));
    if (!(fabsf(p) <= 3.402823466e38f)) p = 0.f;
    A.p_out[chain] = p;
  }
}

__global__ void mh_select_kernel(const float* x, const float* Lx, const float* px, const float* u,
                                 long long N, int d, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  const long long n = i / d;
  out[i] = (px[n] - u[n] >= 0.f) ? Lx[i] : x[i];
}

// ------------------------------------------------------------------------------------------
// Host side: LDS planning and dispatch
// ------------------------------------------------------------------------------------------
const int kMaxLdsBytes = 160 * 1024;

int round4(int v) { return (v + 3) & ~3; }

// Fills the LDS offsets of `k`; returns the dynamic LDS size in bytes.
long long plan_lds(KArgs& k, bool with_nets, bool with_schedule, int NW) {
  const int NT = k.NT, DP = 16 * NT;
  long long o = 0;
  if (with_nets) o += 2LL * net_floats(NT);
  k.o_mask = (int)o;
  if (with_schedule) o += (long long)k.T * DP;
  k.o_trig = (int)o;
  if (with_schedule) o += round4(2 * k.T);
  k.o_P = (int)o;
  if (NW > 1) o += 2LL * NW * 256;
  k.xb_stride = DP + 4;
  k.o_XB = (int)o;
  if (NW > 1 && (k.ekind == L2HMC_ENERGY_GAUSS_DENSE || k.ekind == L2HMC_ENERGY_GMM))
    o += 16LL * k.xb_stride;
  k.o_red = (int)o;
  o += (long long)NW * 16 * 8;
  const int nc = k.ekind == L2HMC_ENERGY_GMM ? k.ncomp : 1;
  k.o_mu = (int)o;
  o += (long long)nc * DP;
  k.o_prec = (int)o;
  if (k.ekind == L2HMC_ENERGY_GAUSS_DIAG) o += DP;
  if (k.ekind == L2HMC_ENERGY_GAUSS_DENSE || k.ekind == L2HMC_ENERGY_GMM)
    o += (long long)nc * gauss_floats(NT);
  k.o_logc = (int)o;
  o += round4(nc);
  return o * 4;
}

int check_energy(const L2hmcEnergy* e, int d) {
  if (e == nullptr) return fail(L2HMC_ERR_ARG, "energy is NULL%s");
  switch (e->kind) {
    case L2HMC_ENERGY_GAUSS_DIAG:
    case L2HMC_ENERGY_GAUSS_DENSE:
      if (!e->mu || !e->prec) return fail(L2HMC_ERR_ARG, "gaussian energy needs mu and prec%s");
      break;
    case L2HMC_ENERGY_GMM:
      if (!e->mu || !e->prec || !e->logc || e->n_comp < 1)
        return fail(L2HMC_ERR_ARG, "gmm energy needs mu, prec, logc, n_comp >= 1%s");
      break;
    case L2HMC_ENERGY_ROUGHWELL:
      if (!(e->eta > 0.f)) return fail(L2HMC_ERR_ARG, "roughwell needs eta > 0%s");
      break;
    case L2HMC_ENERGY_FUNNEL:
      if (!(e->eta > 0.f) || d < 2) return fail(L2HMC_ERR_ARG, "funnel needs sigma > 0 and d >= 2%s");
      break;
    default:
      return fail(L2HMC_ERR_ARG, "unknown energy kind %s%lld", "", e->kind);
  }
  if (!(e->temperature > 0.f)) return fail(L2HMC_ERR_ARG, "temperature must be > 0%s");
  return L2HMC_OK;
}

void fill_energy(KArgs& k, const L2hmcEnergy* e) {
  k.ekind = e->kind;
  k.ncomp = e->kind == L2HMC_ENERGY_GMM ? e->n_comp : 1;
  k.easy = e->easy;
  k.mu = e->mu;
  k.prec = e->prec;
  k.logc = e->logc;
  k.eta = e->eta;
  k.temperature = e->temperature;
}

// (DT, NW) geometry for d dimensions.  NW = 4 spreads a 16-chain tile over the 4 SIMDs of
// a CU (more parallelism per chain: right when there are few chains); NW = 1 keeps a tile
// in one wave (no LDS exchange, fewer MFMAs: right when chains are plentiful).
bool pick_geometry(int d, long long N, int variant, int& DT, int& NW) {
  const int NT = tiles_of(d);
  if (NT <= 1) { DT = 1; NW = 1; return variant == 0 || variant == 1; }
  if (NT <= 4) {
    const bool want4 = variant == 4 || (variant == 0 && N <= 16LL * 256 * 16);
    if (want4) { DT = 1; NW = 4; } else { DT = NT <= 2 ? 2 : 4; NW = 1; }
    return variant == 0 || variant == 1 || variant == 4;
  }
  if (variant == 1) return false;
  NW = 4;
  DT = NT <= 8 ? 2 : (NT <= 16 ? 4 : 8);
  return NT <= 32;
}

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

#define GEOM_SWITCH(DTv, NWv, CALL)                      \
  if (DTv == 1 && NWv == 1) { CALL(1, 1) }               \
  else if (DTv == 2 && NWv == 1) { CALL(2, 1) }          \
  else if (DTv == 4 && NWv == 1) { CALL(4, 1) }          \
  else if (DTv == 1 && NWv == 4) { CALL(1, 4) }          \
  else if (DTv == 2 && NWv == 4) { CALL(2, 4) }          \
  else if (DTv == 4 && NWv == 4) { CALL(4, 4) }          \
  else if (DTv == 8 && NWv == 4) { CALL(8, 4) }          \
  else return fail(L2HMC_ERR_UNSUPPORTED, "no kernel for this geometry%s");

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int l2hmc_abi_version(void) { return L2HMC_ABI_VERSION; }

const char* l2hmc_last_error(void) { return g_err; }

int64_t l2hmc_packed_nets_floats(int32_t d, int32_t H) {
  if (d < 1 || H < 1) return fail(L2HMC_ERR_ARG, "d and H must be >= 1%s");
  if (H > 15) return fail(L2HMC_ERR_UNSUPPORTED, "fused nets support H <= 15 (got %s%lld)", "", H);
  if (d > 512) return fail(L2HMC_ERR_UNSUPPORTED, "fused nets support d <= 512 (got %s%lld)", "", d);
  return 2LL * net_floats(tiles_of(d));
}

int l2hmc_pack_nets(const L2hmcNet* xnet, const L2hmcNet* vnet, int32_t d, int32_t H,
                    float* packed, void* stream) {
  const int64_t n = l2hmc_packed_nets_floats(d, H);
  if (n < 0) return (int)n;
  if (!xnet || !vnet || !packed) return fail(L2HMC_ERR_ARG, "l2hmc_pack_nets: NULL argument%s");
  const L2hmcNet* nets[2] = {xnet, vnet};
  const int NT = tiles_of(d), NF = net_floats(NT), KH = khid_of(H);
  for (int i = 0; i < 2; ++i) {
    const float* const* p = reinterpret_cast<const float* const*>(nets[i]);
    for (int j = 0; j < 16; ++j)
      if (p[j] == nullptr) return fail(L2HMC_ERR_ARG, "l2hmc_pack_nets: NULL weight pointer%s");
    hipLaunchKernelGGL(pack_net_kernel, dim3((NF + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       *nets[i], d, H, KH, NT, packed + (size_t)i * NF);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "pack launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int64_t l2hmc_packed_gaussian_floats(int32_t d) {
  if (d < 1) return fail(L2HMC_ERR_ARG, "d must be >= 1%s");
  return gauss_floats(tiles_of(d));
}

int l2hmc_pack_gaussian(const float* i_sigma, int32_t d, float* packed, void* stream) {
  if (!i_sigma || !packed || d < 1) return fail(L2HMC_ERR_ARG, "l2hmc_pack_gaussian: bad argument%s");
  const int NT = tiles_of(d), n = gauss_floats(NT);
  hipLaunchKernelGGL(pack_gauss_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     i_sigma, d, NT, packed);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "pack launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_trajectory(const L2hmcTrajectoryArgs* a, void* stream) {
  if (!a) return fail(L2HMC_ERR_ARG, "args is NULL%s");
  if (a->n_chains < 0 || a->d < 1 || a->T < 1) return fail(L2HMC_ERR_ARG, "bad n_chains / d / T%s");
  if (a->n_chains == 0) return L2HMC_OK;
  if (!a->x || !a->v || !a->masks || !a->trig) return fail(L2HMC_ERR_ARG, "x, v, masks, trig are required%s");
  if (a->step_begin < 0 || a->n_steps < 0 || a->step_begin + a->n_steps > a->T)
    return fail(L2HMC_ERR_ARG, "steps [%s%lld, +%lld) outside the T-step schedule", "", a->step_begin, a->n_steps);
  if (a->x_next && (!a->u)) return fail(L2HMC_ERR_ARG, "x_next needs u%s");
  if (a->x_out == a->x || a->x_next == a->x) return fail(L2HMC_ERR_ARG, "x_out / x_next must not alias x%s");
  if (!a->alpha && !(a->eps_host > 0.f)) return fail(L2HMC_ERR_ARG, "eps must be > 0%s");
  int rc = check_energy(&a->energy, a->d);
  if (rc) return rc;
  int KH = 3;
  if (a->packed_nets) {
    if (l2hmc_packed_nets_floats(a->d, a->H) < 0) return L2HMC_ERR_UNSUPPORTED;
    KH = khid_of(a->H);
  }
  int DT, NW;
  if (!pick_geometry(a->d, a->n_chains, a->variant, DT, NW))
    return fail(L2HMC_ERR_UNSUPPORTED, "d = %s%lld not supported with variant %lld", "", a->d, a->variant);
  KArgs k;
  memset(&k, 0, sizeof(k));
  k.packed = a->packed_nets; k.masks = a->masks; k.trig = a->trig; k.alpha = a->alpha;
  k.eps_host = a->eps_host; k.N = a->n_chains; k.d = a->d; k.H = a->H; k.T = a->T;
  k.step_begin = a->step_begin; k.n_steps = a->n_steps; k.NT = tiles_of(a->d);
  k.x = a->x; k.v = a->v; k.dir = a->direction; k.dir_all = a->direction_all; k.u = a->u;
  k.x_out = a->x_out; k.v_out = a->v_out; k.logjac_out = a->logjac_out; k.p_out = a->p_out;
  k.x_next = a->x_next;
  fill_energy(k, &a->energy);
  const long long lds = plan_lds(k, a->packed_nets != nullptr, true, NW);
  hipStream_t s = (hipStream_t)stream;
#define CALL_TRAJ(DTc, NWc)                                                     \
  if (KH <= 3) return launch(traj_kernel<DTc, NWc, 3>, k, NWc, lds, s);          \
  else return launch(traj_kernel<DTc, NWc, 4>, k, NWc, lds, s);
  GEOM_SWITCH(DT, NW, CALL_TRAJ)
#undef CALL_TRAJ
}

int l2hmc_energy(const L2hmcEnergy* energy, const float* x, int64_t n_chains, int32_t d,
                 float* U_out, float* grad_out, void* stream) {
  if (n_chains < 0 || d < 1 || !x) return fail(L2HMC_ERR_ARG, "l2hmc_energy: bad argument%s");
  if (n_chains == 0) return L2HMC_OK;
  int rc = check_energy(energy, d);
  if (rc) return rc;
  int DT, NW;
  if (!pick_geometry(d, n_chains, 0, DT, NW)) return fail(L2HMC_ERR_UNSUPPORTED, "d = %s%lld too large", "", d);
  KArgs k;
  memset(&k, 0, sizeof(k));
  k.N = n_chains; k.d = d; k.NT = tiles_of(d); k.x = x; k.U_out = U_out; k.grad_out = grad_out;
  fill_energy(k, energy);
  const long long lds = plan_lds(k, false, false, NW);
#define CALL_EN(DTc, NWc) return launch(energy_kernel<DTc, NWc>, k, NWc, lds, (hipStream_t)stream);
  GEOM_SWITCH(DT, NW, CALL_EN)
#undef CALL_EN
}

int l2hmc_p_accept(const L2hmcEnergy* energy, const float* x0, const float* v0, const float* x1,
                   const float* v1, const float* logjac, int64_t n_chains, int32_t d, float* p_out,
                   void* stream) {
  if (n_chains < 0 || d < 1 || !x0 || !v0 || !x1 || !v1 || !logjac || !p_out)
    return fail(L2HMC_ERR_ARG, "l2hmc_p_accept: bad argument%s");
  if (n_chains == 0) return L2HMC_OK;
  int rc = check_energy(energy, d);
  if (rc) return rc;
  int DT, NW;
  if (!pick_geometry(d, n_chains, 0, DT, NW)) return fail(L2HMC_ERR_UNSUPPORTED, "d = %s%lld too large", "", d);
  KArgs k;
  memset(&k, 0, sizeof(k));
  k.N = n_chains; k.d = d; k.NT = tiles_of(d);
  k.x = x0; k.v = v0; k.x1 = x1; k.v1 = v1; k.logjac_in = logjac; k.p_out = p_out;
  fill_energy(k, energy);
  const long long lds = plan_lds(k, false, false, NW);
#define CALL_PA(DTc, NWc) return launch(paccept_kernel<DTc, NWc>, k, NWc, lds, (hipStream_t)stream);
  GEOM_SWITCH(DT, NW, CALL_PA)
#undef CALL_PA
}

int l2hmc_mh_select(const float* x, const float* Lx, const float* px, const float* u,
                    int64_t n_chains, int32_t d, float* x_next, void* stream) {
  if (n_chains < 0 || d < 1 || !x || !Lx || !px || !u || !x_next)
    return fail(L2HMC_ERR_ARG, "l2hmc_mh_select: bad argument%s");
  if (n_chains == 0) return L2HMC_OK;
  const long long n = n_chains * (long long)d;
  hipLaunchKernelGGL(mh_select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, Lx, px, u, (long long)n_chains, d, x_next);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

}  // extern "C"
