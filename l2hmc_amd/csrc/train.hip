// l2hmc_train_propose_grad -- one direction-mixed proposal (sampler.py:28-51) together with the
// gradient of its training-loss term (SCGExperiment.ipynb raw lines 156-169) w.r.t. every net
// parameter and the step size, by hand-derived reverse mode through the generalised leapfrog
// trajectory (dynamics.py:115-201, 246-309), including the Hessian-vector path through
// `grad_energy` (TF1 differentiates through tf.gradients).  Derivation = oracle/l2hmc_train_oracle.py.
//
// Round-1 form, correctness first: ONE CHAIN PER LANE, scalar loops over d and H, raw weights
// staged in LDS (wave-uniform reads broadcast), per-step states checkpointed to a caller
// workspace, step intermediates re-computed in the reverse sweep, parameter gradients reduced
// over the wave with shuffles and accumulated in LDS, one global atomicAdd per parameter per
// workgroup.  (The sampling hot path is the MFMA kernel in l2hmc_kernels.hpp; an MFMA form of
// this kernel is future work.)  Gaussian targets (diagonal or dense precision), d <= 64, H <= 16.
#include "l2hmc_kernels.hpp"

namespace l2hmc {

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
  float eta;
  float scale, inv_n;
  float *Lx, *p, *v1, *grad, *ws;
};

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int DM, int HM>
struct NetCache {
  float h1[HM], h2[HM], ts[DM], tq[DM], S[DM], T[DM], Q[DM];
};

// [S, T, Q] = net([a, b, tau]) for this lane's chain; W = this net's flat weights in LDS.
template <int DM, int HM>
__device__ void net_fwd(const float* W, const NetOff& o, int d, int H, const float* a, const float* b,
                        float t0, float t1, NetCache<DM, HM>& c) {
  for (int i = 0; i < H; ++i) {
    float acc = (W[o.b1 + i] + W[o.b2 + i]) + W[o.b3 + i] + t0 * W[o.W3 + i] + t1 * W[o.W3 + H + i];
    for (int k = 0; k < d; ++k) acc += a[k] * W[o.W1 + k * H + i] + b[k] * W[o.W2 + k * H + i];
    c.h1[i] = fmaxf(acc, 0.f);
  }
  for (int j = 0; j < H; ++j) {
    float acc = W[o.b4 + j];
    for (int i = 0; i < H; ++i) acc += c.h1[i] * W[o.W4 + i * H + j];
    c.h2[j] = fmaxf(acc, 0.f);
  }
  for (int k = 0; k < d; ++k) {
    float zs = W[o.bs + k], zt = W[o.bt + k], zq = W[o.bq + k];
    for (int j = 0; j < H; ++j) {
      zs += c.h2[j] * W[o.Ws + j * d + k];
      zt += c.h2[j] * W[o.Wt + j * d + k];
      zq += c.h2[j] * W[o.Wq + j * d + k];
    }
    c.ts[k] = tanhf(zs);
    c.tq[k] = tanhf(zq);
    c.S[k] = expf(W[o.ls + k]) * c.ts[k];
    c.T[k] = zt;
    c.Q[k] = expf(W[o.lq + k]) * c.tq[k];
  }
}

// Reverse of net_fwd: parameter gradients (summed over the wave's chains) go to the LDS
// accumulator G (same flat layout), input gradients to da, db.  dS/dT/dQ are consumed.
template <int DM, int HM>
__device__ void net_bwd(const float* W, float* G, const NetOff& o, int d, int H, const float* a,
                        const float* b, float t0, float t1, const NetCache<DM, HM>& c, float* dS,
                        float* dT, float* dQ, float* da, float* db, int lane) {
  auto acc = [&](int idx, float v) {
    const float s = wave_sum(v);
    if (lane == 0) atomicAdd(&G[idx], s);
  };
  float dh[HM];
  for (int j = 0; j < H; ++j) dh[j] = 0.f;
  for (int k = 0; k < d; ++k) {
    const float es = expf(W[o.ls + k]), eq = expf(W[o.lq + k]);
    acc(o.ls + k, dS[k] * c.S[k]);
    acc(o.lq + k, dQ[k] * c.Q[k]);
    const float dzs = dS[k] * es * (1.f - c.ts[k] * c.ts[k]);
    const float dzq = dQ[k] * eq * (1.f - c.tq[k] * c.tq[k]);
    const float dzt = dT[k];
    acc(o.bs + k, dzs);
    acc(o.bt + k, dzt);
    acc(o.bq + k, dzq);
    for (int j = 0; j < H; ++j) {
      acc(o.Ws + j * d + k, c.h2[j] * dzs);
      acc(o.Wt + j * d + k, c.h2[j] * dzt);
      acc(o.Wq + j * d + k, c.h2[j] * dzq);
      dh[j] += W[o.Ws + j * d + k] * dzs + W[o.Wt + j * d + k] * dzt + W[o.Wq + j * d + k] * dzq;
    }
  }
  float d1[HM];
  for (int i = 0; i < H; ++i) d1[i] = 0.f;
  for (int j = 0; j < H; ++j) {
    const float da2 = c.h2[j] > 0.f ? dh[j] : 0.f;
    acc(o.b4 + j, da2);
    for (int i = 0; i < H; ++i) {
      acc(o.W4 + i * H + j, c.h1[i] * da2);
      d1[i] += W[o.W4 + i * H + j] * da2;
    }
  }
  for (int k = 0; k < d; ++k) { da[k] = 0.f; db[k] = 0.f; }
  for (int i = 0; i < H; ++i) {
    const float da1 = c.h1[i] > 0.f ? d1[i] : 0.f;
    acc(o.b1 + i, da1);
    acc(o.b2 + i, da1);
    acc(o.b3 + i, da1);
    acc(o.W3 + i, t0 * da1);
    acc(o.W3 + H + i, t1 * da1);
    for (int k = 0; k < d; ++k) {
      acc(o.W1 + k * H + i, a[k] * da1);
      acc(o.W2 + k * H + i, b[k] * da1);
      da[k] += W[o.W1 + k * H + i] * da1;
      db[k] += W[o.W2 + k * H + i] * da1;
    }
  }
}

template <int DM, int HM>
__global__ __launch_bounds__(256) void train_kernel(const TArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int d = A.d, H = A.H, T = A.T;
  const int P = net_params(d, H);
  const NetOff o = net_off(d, H);
  float* Wx = smem;                  // XNet weights (flat), then VNet
  float* Wv = smem + P;
  float* Gx = smem + 2 * P;          // gradient accumulators, same layout, + 1 for eps
  float* Gv = smem + 3 * P;
  float* Ge = smem + 4 * P;
  float* Msk = smem + 4 * P + 4;     // masks (T, d)
  float* Trg = Msk + T * d;          // trig (T, 2)
  const int nc = A.ekind == L2HMC_ENERGY_GMM ? A.ncomp : 1;
  float* Mu = Trg + 2 * T;           // means (nc, d); precision diag (d) / dense (d, d) / (nc, d, d); log c (nc)
  float* Pr = Mu + nc * d;
  const int npr = A.ekind == L2HMC_ENERGY_GAUSS_DIAG ? d : (A.ekind == L2HMC_ENERGY_ROUGHWELL ? 0 : nc * d * d);
  float* Lc = Pr + npr;

  // ---- stage ------------------------------------------------------------------------------------
  {
    const float* const* srcs[2] = {reinterpret_cast<const float* const*>(&A.xnet),
                                   reinterpret_cast<const float* const*>(&A.vnet)};
    const int offs[17] = {o.W1, o.b1, o.W2, o.b2, o.W3, o.b3, o.W4, o.b4, o.Ws, o.bs, o.Wt, o.bt, o.Wq,
                          o.bq, o.ls, o.lq, P};
    for (int n = 0; n < 2; ++n)
      for (int f = 0; f < 16; ++f) {
        const float* src = srcs[n][f];
        float* dst = (n == 0 ? Wx : Wv) + offs[f];
        for (int i = tid; i < offs[f + 1] - offs[f]; i += 256) dst[i] = src[i];
      }
    for (int i = tid; i < 2 * P + 4; i += 256) Gx[i] = 0.f;
    for (int i = tid; i < T * d; i += 256) Msk[i] = A.masks[i];
    for (int i = tid; i < 2 * T; i += 256) Trg[i] = A.trig[i];
    if (A.ekind != L2HMC_ENERGY_ROUGHWELL) {
      for (int i = tid; i < nc * d; i += 256) Mu[i] = A.mu[i];
      for (int i = tid; i < npr; i += 256) Pr[i] = A.prec[i];
    }
    if (A.ekind == L2HMC_ENERGY_GMM)
      for (int i = tid; i < nc; i += 256) Lc[i] = A.logc[i];
  }
  __syncthreads();

  const long long n = (long long)blockIdx.x * 256 + tid;
  const bool live = n < A.N;
  const long long nn = live ? n : 0;
  const bool fwd = A.dir != nullptr ? A.dir[nn] != 0 : (A.dir_all != 0);
  const float sgn = fwd ? 1.f : -1.f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const bool dense = A.ekind == L2HMC_ENERGY_GAUSS_DENSE;

  // ---- target: U, grad U, Hessian-vector product (oracle/l2hmc_train_oracle.py *Target classes) ----
  const int ek = A.ekind;
  const float rw_den = A.easy ? A.eta : A.eta * A.eta;
  // y = G_c (z - mu_c) for component c (G = (S + S^T)/2; diagonal kind: elementwise); center=false: G z
  auto matG = [&](int c, const float* z, bool center, float* out) {
    const float* mu = Mu + c * d;
    for (int k = 0; k < d; ++k) {
      if (ek == L2HMC_ENERGY_GAUSS_DIAG) {
        out[k] = Pr[k] * (z[k] - (center ? mu[k] : 0.f));
      } else {
        const float* S = Pr + c * d * d;
        float acc = 0.f;
        for (int j = 0; j < d; ++j) acc += 0.5f * (S[k * d + j] + S[j * d + k]) * (z[j] - (center ? mu[j] : 0.f));
        out[k] = acc;
      }
    }
  };
  constexpr int KC = 8;      // max mixture components
  // softmax weights w_c and y_c of the GMM at z; returns logsumexp
  auto gmm_parts = [&](const float* z, float (*ys)[DM], float* wts) {
    float V[KC], m = -INFINITY;
    for (int c = 0; c < nc; ++c) {
      matG(c, z, true, ys[c]);
      const float* S = Pr + c * d * d;
      float q = 0.f;                       // (z - mu)^T S (z - mu) == (z - mu)^T G (z - mu)
      for (int k = 0; k < d; ++k) q += (z[k] - Mu[c * d + k]) * ys[c][k];
      (void)S;
      V[c] = -0.5f * q + Lc[c];
      m = fmaxf(m, V[c]);
    }
    float sum = 0.f;
    for (int c = 0; c < nc; ++c) { wts[c] = expf(V[c] - m); sum += wts[c]; }
    for (int c = 0; c < nc; ++c) wts[c] /= sum;
    return m + logf(sum);
  };
  auto gradU = [&](const float* z, float* out) {
    if (ek == L2HMC_ENERGY_ROUGHWELL) {
      for (int k = 0; k < d; ++k) out[k] = z[k] - (A.eta / rw_den) * sinf(z[k] / rw_den);
    } else if (ek == L2HMC_ENERGY_GMM) {
      float ys[KC][DM], wts[KC];
      gmm_parts(z, ys, wts);
      for (int k = 0; k < d; ++k) {
        float acc = 0.f;
        for (int c = 0; c < nc; ++c) acc += wts[c] * ys[c][k];
        out[k] = acc;
      }
    } else {
      matG(0, z, true, out);
    }
  };
  auto energyU = [&](const float* z) {
    float u = 0.f;
    if (ek == L2HMC_ENERGY_ROUGHWELL) {
      for (int k = 0; k < d; ++k) u += 0.5f * z[k] * z[k] + A.eta * cosf(z[k] / rw_den);
    } else if (ek == L2HMC_ENERGY_GMM) {
      float ys[KC][DM], wts[KC];
      u = -gmm_parts(z, ys, wts);
    } else {
      float gz[DM];
      matG(0, z, true, gz);
      for (int k = 0; k < d; ++k) u += 0.5f * (z[k] - Mu[k]) * gz[k];
    }
    return u;
  };
  // out = H(z) vec
  auto hessvec = [&](const float* z, const float* vec, float* out) {
    if (ek == L2HMC_ENERGY_ROUGHWELL) {
      for (int k = 0; k < d; ++k) out[k] = (1.f - (A.eta / (rw_den * rw_den)) * cosf(z[k] / rw_den)) * vec[k];
    } else if (ek == L2HMC_ENERGY_GMM) {
      float ys[KC][DM], wts[KC], gz[DM], gv = 0.f, tmpv[DM];
      gmm_parts(z, ys, wts);
      for (int k = 0; k < d; ++k) {
        float acc = 0.f;
        for (int c = 0; c < nc; ++c) acc += wts[c] * ys[c][k];
        gz[k] = acc;
        gv += acc * vec[k];
      }
      for (int k = 0; k < d; ++k) out[k] = gz[k] * gv;
      for (int c = 0; c < nc; ++c) {
        float yv = 0.f;
        for (int k = 0; k < d; ++k) yv += ys[c][k] * vec[k];
        matG(c, vec, false, tmpv);
        for (int k = 0; k < d; ++k) out[k] += wts[c] * (tmpv[k] - ys[c][k] * yv);
      }
    } else {
      matG(0, vec, false, out);
    }
  };

  float x[DM], v[DM], g1[DM], vh[DM], y[DM], xo[DM], g2[DM], kin[DM], tmp[DM];
  NetCache<DM, HM> c1, ca, cb, c2;
  // checkpoints: ws[(t * N + n) * 2 d + {0..d-1: x_t, d..2d-1: v_t}]
  auto ckpt = [&](int t) { return A.ws + ((long long)t * A.N + nn) * 2 * d; };

  for (int k = 0; k < d; ++k) { x[k] = live ? A.x[nn * d + k] : 0.f; v[k] = live ? A.v[nn * d + k] : 0.f; }
  float x0[DM], U0, K0 = 0.f, ld = 0.f;
  for (int k = 0; k < d; ++k) { x0[k] = x[k]; K0 += 0.5f * v[k] * v[k]; }
  U0 = energyU(x);

  // one forward step from (x, v); fills vh, y, xo, g1, g2, the four caches and (x, v) <- new state
  auto step_fwd = [&](int it, bool keep_ld) {
    const int s = fwd ? it : (T - 1 - it);
    const float t0 = Trg[2 * s], t1 = Trg[2 * s + 1];
    const float* m = Msk + s * d;
    gradU(x, g1);
    net_fwd<DM, HM>(Wv, o, d, H, x, g1, t0, t1, c1);
    for (int k = 0; k < d; ++k) {
      const float ES = expf(sgn * heps * c1.S[k]), EQ = expf(eps * c1.Q[k]);
      const float cc = heps * (c1.T[k] - EQ * g1[k]);
      vh[k] = fwd ? v[k] * ES + cc : (v[k] - cc) * ES;
      if (keep_ld) ld += sgn * heps * c1.S[k];
    }
    for (int k = 0; k < d; ++k) { kin[k] = fwd ? m[k] : 1.f - m[k]; tmp[k] = kin[k] * x[k]; }   // kin = k1
    net_fwd<DM, HM>(Wx, o, d, H, vh, tmp, t0, t1, ca);
    for (int k = 0; k < d; ++k) {
      const float ES = expf(sgn * eps * ca.S[k]), EQ = expf(eps * ca.Q[k]);
      const float tr = eps * (EQ * vh[k] + ca.T[k]);
      const float nw = fwd ? x[k] * ES + tr : ES * (x[k] - tr);
      y[k] = kin[k] * x[k] + (1.f - kin[k]) * nw;
      if (keep_ld) ld += (1.f - kin[k]) * sgn * eps * ca.S[k];
    }
    for (int k = 0; k < d; ++k) tmp[k] = (1.f - kin[k]) * y[k];
    net_fwd<DM, HM>(Wx, o, d, H, vh, tmp, t0, t1, cb);
    for (int k = 0; k < d; ++k) {
      const float ES = expf(sgn * eps * cb.S[k]), EQ = expf(eps * cb.Q[k]);
      const float tr = eps * (EQ * vh[k] + cb.T[k]);
      const float nw = fwd ? y[k] * ES + tr : ES * (y[k] - tr);
      xo[k] = (1.f - kin[k]) * y[k] + kin[k] * nw;
      if (keep_ld) ld += kin[k] * sgn * eps * cb.S[k];
    }
    gradU(xo, g2);
    net_fwd<DM, HM>(Wv, o, d, H, xo, g2, t0, t1, c2);
  };

  // ---- forward trajectory with checkpoints --------------------------------------------------------
  for (int it = 0; it < T; ++it) {
    float* ck = ckpt(it);
    if (live) for (int k = 0; k < d; ++k) { ck[k] = x[k]; ck[d + k] = v[k]; }
    step_fwd(it, true);
    for (int k = 0; k < d; ++k) {
      const float ES = expf(sgn * heps * c2.S[k]), EQ = expf(eps * c2.Q[k]);
      const float cc = heps * (c2.T[k] - EQ * g2[k]);
      v[k] = fwd ? vh[k] * ES + cc : (vh[k] - cc) * ES;
      ld += sgn * heps * c2.S[k];
      x[k] = xo[k];
    }
  }
  // ---- accept probability, loss term and the adjoint seeds ------------------------------------------
  float K1 = 0.f, sq = 0.f;
  gradU(x, g2);
  const float U1 = energyU(x);
  for (int k = 0; k < d; ++k) { K1 += 0.5f * v[k] * v[k]; sq += (x0[k] - x[k]) * (x0[k] - x[k]); }
  const float val = (U0 + K0) - (U1 + K1) + ld;
  const float p = accept_prob(val);
  const float v1 = sq * p + 1e-4f;
  if (live) {
    for (int k = 0; k < d; ++k) A.Lx[n * d + k] = x[k];
    A.p[n] = p;
    A.v1[n] = v1;
  }
  const float dv1 = live ? (A.scale * (-1.f / (v1 * v1)) - 1.f / A.scale) * A.inv_n : 0.f;
  const bool pfin = (val == val) && p > 0.f;   // finite branch of dynamics.py:309 actually taken
  const float dval = (pfin && val < 0.f) ? dv1 * sq * p : 0.f;
  // (a diverged chain -- non-finite end point -- has p = 0 and contributes no gradient, instead of
  //  the reference's 0 * NaN)
  const bool okc = sq < 3.0e38f;
  float lx[DM], lv[DM], deps = 0.f;
  for (int k = 0; k < d; ++k) {
    lx[k] = okc ? dv1 * p * 2.f * (x[k] - x0[k]) - dval * g2[k] : 0.f;
    lv[k] = okc ? -dval * v[k] : 0.f;
  }
  const float lam = dval;

  // ---- reverse sweep ------------------------------------------------------------------------------------
  float dS[DM], dT[DM], dQ[DM], da[DM], db[DM], dvh[DM], dz[DM], dg[DM];
  for (int it = T - 1; it >= 0; --it) {
    const float* ck = ckpt(it);
    for (int k = 0; k < d; ++k) { x[k] = live ? ck[k] : 0.f; v[k] = live ? ck[d + k] : 0.f; }
    step_fwd(it, false);
    const int s = fwd ? it : (T - 1 - it);
    const float t0 = Trg[2 * s], t1 = Trg[2 * s + 1];
    // v' = v_half(vh, g2, V(x', g2))
    for (int k = 0; k < d; ++k) {
      const float ES = expf(sgn * heps * c2.S[k]), EQ = expf(eps * c2.Q[k]);
      const float cc = heps * (c2.T[k] - EQ * g2[k]);
      const float dout = lv[k];
      dvh[k] = dout * ES;
      const float dES = fwd ? dout * vh[k] : dout * (vh[k] - cc);
      const float dcc = fwd ? dout : -dout * ES;
      const float ds = dES * ES + lam;
      dS[k] = ds * sgn * heps;
      deps += ds * sgn * 0.5f * c2.S[k] + dcc * 0.5f * (c2.T[k] - EQ * g2[k]);
      dT[k] = dcc * heps;
      const float dq = -dcc * heps * g2[k] * EQ;
      dg[k] = -dcc * heps * EQ;
      deps += dq * c2.Q[k];
      dQ[k] = dq * eps;
    }
    net_bwd<DM, HM>(Wv, Gv, o, d, H, xo, g2, t0, t1, c2, dS, dT, dQ, da, db, lane);
    for (int k = 0; k < d; ++k) tmp[k] = dg[k] + db[k];
    hessvec(xo, tmp, dz);                                   // Hessian-vector product at x'
    for (int k = 0; k < d; ++k) lx[k] = lx[k] + da[k] + dz[k];   // = d xo
    // x' = x_half(y, k2, vh, X(vh, k2 y)),  k2 = 1 - k1
    for (int k = 0; k < d; ++k) {
      const float kp = 1.f - kin[k], up = kin[k];
      const float ES = expf(sgn * eps * cb.S[k]), EQ = expf(eps * cb.Q[k]);
      const float tr = eps * (EQ * vh[k] + cb.T[k]);
      const float dnw = up * lx[k];
      dz[k] = kp * lx[k] + dnw * ES;                        // d y (direct part)
      const float dES = fwd ? dnw * y[k] : dnw * (y[k] - tr);
      const float dtr = fwd ? dnw : -dnw * ES;
      const float dsx = dES * ES + up * lam;
      dS[k] = dsx * sgn * eps;
      deps += dsx * sgn * cb.S[k] + dtr * (EQ * vh[k] + cb.T[k]);
      dvh[k] += dtr * eps * EQ;
      dT[k] = dtr * eps;
      const float dq = dtr * eps * vh[k] * EQ;
      deps += dq * cb.Q[k];
      dQ[k] = dq * eps;
    }
    for (int k = 0; k < d; ++k) tmp[k] = (1.f - kin[k]) * y[k];
    net_bwd<DM, HM>(Wx, Gx, o, d, H, vh, tmp, t0, t1, cb, dS, dT, dQ, da, db, lane);
    for (int k = 0; k < d; ++k) { dvh[k] += da[k]; dz[k] += (1.f - kin[k]) * db[k]; }    // dz = d y
    // y = x_half(x, k1, vh, X(vh, k1 x))
    for (int k = 0; k < d; ++k) {
      const float kp = kin[k], up = 1.f - kin[k];
      const float ES = expf(sgn * eps * ca.S[k]), EQ = expf(eps * ca.Q[k]);
      const float tr = eps * (EQ * vh[k] + ca.T[k]);
      const float dnw = up * dz[k];
      lx[k] = kp * dz[k] + dnw * ES;                        // d x (direct part)
      const float dES = fwd ? dnw * x[k] : dnw * (x[k] - tr);
      const float dtr = fwd ? dnw : -dnw * ES;
      const float dsx = dES * ES + up * lam;
      dS[k] = dsx * sgn * eps;
      deps += dsx * sgn * ca.S[k] + dtr * (EQ * vh[k] + ca.T[k]);
      dvh[k] += dtr * eps * EQ;
      dT[k] = dtr * eps;
      const float dq = dtr * eps * vh[k] * EQ;
      deps += dq * ca.Q[k];
      dQ[k] = dq * eps;
    }
    for (int k = 0; k < d; ++k) tmp[k] = kin[k] * x[k];
    net_bwd<DM, HM>(Wx, Gx, o, d, H, vh, tmp, t0, t1, ca, dS, dT, dQ, da, db, lane);
    for (int k = 0; k < d; ++k) { dvh[k] += da[k]; lx[k] += kin[k] * db[k]; }
    // vh = v_half(v, g1, V(x, g1))
    for (int k = 0; k < d; ++k) {
      const float ES = expf(sgn * heps * c1.S[k]), EQ = expf(eps * c1.Q[k]);
      const float cc = heps * (c1.T[k] - EQ * g1[k]);
      const float dout = dvh[k];
      lv[k] = dout * ES;
      const float dES = fwd ? dout * v[k] : dout * (v[k] - cc);
      const float dcc = fwd ? dout : -dout * ES;
      const float ds = dES * ES + lam;
      dS[k] = ds * sgn * heps;
      deps += ds * sgn * 0.5f * c1.S[k] + dcc * 0.5f * (c1.T[k] - EQ * g1[k]);
      dT[k] = dcc * heps;
      const float dq = -dcc * heps * g1[k] * EQ;
      dg[k] = -dcc * heps * EQ;
      deps += dq * c1.Q[k];
      dQ[k] = dq * eps;
    }
    net_bwd<DM, HM>(Wv, Gv, o, d, H, x, g1, t0, t1, c1, dS, dT, dQ, da, db, lane);
    for (int k = 0; k < d; ++k) tmp[k] = dg[k] + db[k];
    hessvec(x, tmp, dz);
    for (int k = 0; k < d; ++k) lx[k] = lx[k] + da[k] + dz[k];
  }
  {
    const float s = wave_sum(deps);
    if (lane == 0) atomicAdd(Ge, s);
  }
  __syncthreads();
  // flat gradient: [xnet (P) | vnet (P) | eps], accumulated (+=) into the caller's buffer
  for (int i = tid; i < 2 * P + 1; i += 256) atomicAdd(&A.grad[i], i < 2 * P ? Gx[i] : Ge[0]);
}

}  // namespace l2hmc

using namespace l2hmc;

extern "C" {

int64_t l2hmc_train_workspace_floats(int64_t n_chains, int32_t d, int32_t T) {
  if (n_chains < 0 || d < 1 || T < 1) return fail(L2HMC_ERR_ARG, "l2hmc_train_workspace_floats: bad argument%s");
  return (int64_t)T * n_chains * 2 * d;
}

int64_t l2hmc_train_grad_floats(int32_t d, int32_t H) {
  if (d < 1 || H < 1) return fail(L2HMC_ERR_ARG, "l2hmc_train_grad_floats: bad argument%s");
  return 2LL * net_params(d, H) + 1;
}

int l2hmc_train_propose_grad(const L2hmcTrainArgs* a, void* stream) {
  if (!a) return fail(L2HMC_ERR_ARG, "args is NULL%s");
  if (a->n_chains < 0 || a->d < 1 || a->T < 1 || a->H < 1) return fail(L2HMC_ERR_ARG, "bad n_chains / d / H / T%s");
  if (a->n_chains == 0) return L2HMC_OK;
  if (a->d > 64 || a->H > 16) return fail(L2HMC_ERR_UNSUPPORTED, "training kernel supports d <= 64, H <= 16 (got d = %s%lld, H = %lld)", "", a->d, a->H);
  if (!a->xnet || !a->vnet || !a->masks || !a->trig || !a->x || !a->v || !a->Lx || !a->p || !a->v1 ||
      !a->grad || !a->workspace)
    return fail(L2HMC_ERR_ARG, "l2hmc_train_propose_grad: NULL pointer%s");
  const int ek = a->energy.kind;
  if (ek != L2HMC_ENERGY_GAUSS_DIAG && ek != L2HMC_ENERGY_GAUSS_DENSE && ek != L2HMC_ENERGY_GMM &&
      ek != L2HMC_ENERGY_ROUGHWELL)
    return fail(L2HMC_ERR_UNSUPPORTED, "training supports the Gaussian, GMM and Rough-Well targets (analytic Hessian-vector products)%s");
  if (ek != L2HMC_ENERGY_ROUGHWELL && (!a->energy.mu || !a->energy.prec))
    return fail(L2HMC_ERR_ARG, "energy needs mu and prec (RAW (d,d) precisions for the dense / GMM kinds)%s");
  if (ek == L2HMC_ENERGY_GMM && (!a->energy.logc || a->energy.n_comp < 1 || a->energy.n_comp > 8))
    return fail(L2HMC_ERR_ARG, "GMM training needs logc and 1 <= n_comp <= 8%s");
  if (ek == L2HMC_ENERGY_ROUGHWELL && !(a->energy.eta > 0.f)) return fail(L2HMC_ERR_ARG, "roughwell needs eta > 0%s");
  if (!(a->energy.temperature == 1.f)) return fail(L2HMC_ERR_UNSUPPORTED, "training kernel: temperature must be 1%s");
  if (!a->alpha && !(a->eps_host > 0.f)) return fail(L2HMC_ERR_ARG, "eps must be > 0%s");
  if (!(a->scale > 0.f) || !(a->inv_n > 0.f)) return fail(L2HMC_ERR_ARG, "scale and inv_n must be > 0%s");
  TArgs k;
  k.xnet = *a->xnet; k.vnet = *a->vnet;
  k.masks = a->masks; k.trig = a->trig; k.alpha = a->alpha; k.eps_host = a->eps_host;
  k.N = a->n_chains; k.d = a->d; k.H = a->H; k.T = a->T; k.x = a->x; k.v = a->v;
  k.dir = a->direction; k.dir_all = a->direction_all; k.ekind = a->energy.kind;
  k.mu = a->energy.mu; k.prec = a->energy.prec; k.logc = a->energy.logc; k.eta = a->energy.eta;
  k.ncomp = ek == L2HMC_ENERGY_GMM ? a->energy.n_comp : 1; k.easy = a->energy.easy;
  k.scale = a->scale; k.inv_n = a->inv_n;
  k.Lx = a->Lx; k.p = a->p; k.v1 = a->v1; k.grad = a->grad; k.ws = a->workspace;
  const int P = net_params(a->d, a->H);
  const int ncs = k.ncomp;
  const long long npr = ek == L2HMC_ENERGY_GAUSS_DIAG ? a->d : (ek == L2HMC_ENERGY_ROUGHWELL ? 0 : (long long)ncs * a->d * a->d);
  const long long lds = 4LL * (4 * P + 4 + (long long)a->T * a->d + 2 * a->T + (long long)ncs * a->d + npr + ncs + 4);
  if (lds > 160 * 1024) return fail(L2HMC_ERR_UNSUPPORTED, "training kernel needs %s%lld bytes of LDS", "", lds);
  const unsigned blocks = (unsigned)((a->n_chains + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TRAIN(DMv)                                                                          \
  {                                                                                                \
    auto kern = train_kernel<DMv, 16>;                                                             \
    if (lds > 48 * 1024) {                                                                         \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)); \
    }                                                                                              \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), (size_t)lds, s, k);                          \
  }
  if (a->d <= 8) LAUNCH_TRAIN(8) else LAUNCH_TRAIN(64)
#undef LAUNCH_TRAIN
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

}  // extern "C"
