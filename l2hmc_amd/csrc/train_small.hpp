// train_small.hpp -- the training gradient for SMALL targets, d <= 4 (included by train.hip inside namespace l2hmc, after
// train_fast.hpp).  The notebook's own training configuration (SCGExperiment.ipynb raw 156-181: SCG-2D, 200 chains) has
// d = 2: in the 16-wide dimension tiles of train_fast_kernel 14 of 16 MFMA rows and 3 of 4 values per lane are padding.
//
// Same mathematics as train_fast_kernel / train_kernel (hand-derived reverse mode of one direction-mixed proposal and its
// loss term incl. the Hessian-vector path through grad U; derivation = oracle/l2hmc_train_oracle.py), in the layout of
// traj_small.hpp: 16 chains per workgroup, lane l = (c = l & 15, q = l >> 4) holds chain c, DIMENSION q -- every state and
// adjoint vector is a scalar per lane; hidden vectors stay float4 {h[unit(q, r)][chain c]}.  TWO waves: wave 0 runs the
// trajectory and the adjoint recursion, wave 1 accumulates the weight-gradient tiles from the operands wave 0 leaves in LDS
// (see the kernel head).  Per net evaluation:
//   * layer 1: the d <= 4 dimensions are ONE k-step (k = q): 1 MFMA per input instead of 4;
//   * heads: ONE 16-row block whose row 4 q' + h is (dimension q', head h = S, T, Q): KH MFMAs instead of 3 KH, and lane
//     (c, q) finds z_S, z_T, z_Q of its own dimension in acc[0..2];
//   * reverse: (dz_S, dz_T, dz_Q, 0) of a lane IS the B operand of the transposed head product (k-step h <-> head h);
//     the input adjoints come back on rows 4 j (dimension j) so that lane q reads its own in acc[0];
//   * weight gradients: the same transposes through an LDS scratch and 16 x 16 register tiles as train_fast, but ONE head
//     tile per net (rows = (dimension, head)) instead of three -- and on the second wave, off the step's dependency chain.
// 48 MFMAs per net evaluation + back-propagation instead of 69 + 20 recomputed ... per leapfrog step: 192 instead of 356
// MFMAs, and every elementwise update is scalar instead of float4.  One barrier per back-propagation (the operand hand-over),
// no atomics: the workgroup's flat gradient goes to its workspace slot, train_reduce_kernel adds the slots in block order.
// Targets: diagonal and dense Gaussians, Rough Well, mixtures of Gaussians (<= 8 components); the funnel stays on train_fast.
#pragma once

struct TSLayout { int grp, tb, msk, trg, tr, total, ng; };
__host__ __device__ inline TSLayout ts_layout(int T) {
  TSLayout L;
  L.ng = 6;                                   // layer 2 fwd / transposed, heads fwd / transposed, layer 1 transposed (a, b)
  int p = 0;
  L.grp = p; p += 2 * L.ng * 256;
  L.tb = p; p += 2 * T * 16;
  L.msk = p; p += (T * 4 + 3) / 4 * 4;
  L.trg = p; p += (2 * T + 3) / 4 * 4;
  L.tr = p; p += 2 * 7 * 320;                 // two buffers of seven 16 x 20 transpose scratches (one per operand of a
                                              // back-propagation): wave 0 writes them, wave 1 reads them one barrier later
  L.total = p;
  return L;
}
// checkpointed scalars per lane and step: x, v, v_half, y, x' and, per net evaluation e = 0..3 of the step (slots 5 + 11 e ...),
// everything its back-propagation needs -- h1 (4), h2 (4), tanh(zs), T, tanh(zq): the reverse sweep reads them back one evaluation
// ahead instead of re-evaluating the net (11 KB per step and tile: the tiles of a d <= 4 problem are few).  49 x 64 floats per step
// stay inside the TF_CK x 256 the workspace reserves per tile and step.
constexpr int TS_CK = 49;
static_assert(TS_CK * 64 <= TF_CK * 256, "the d <= 4 trainer's checkpoints must fit the per-tile workspace stride");

template <int EK, int KH>
__global__ __launch_bounds__(128, 1) void train_small_kernel(const TArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  TS_DECL;
  // Two waves per 16-chain tile.  Wave 0 runs the trajectory and the adjoint recursion (the dependency chain of the step);
  // wave 1 only accumulates the weight-gradient tiles: after every back-propagation wave 0 leaves the seven operands in LDS
  // (double-buffered) and both waves meet at one barrier -- the operand transposes and the 20 MFMAs of the chain-contractions
  // (40 % of a single-wave launch, tools/train_phase_timing.py) are off the critical path.  Same arithmetic, same order.
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int d = A.d, H = A.H, T = A.T;
  const TSLayout L = ts_layout(T);
  const int ng = L.ng;
  const NetOff o = net_off(d, H);
  const int P = net_params(d, H);
  const long long n = (long long)blockIdx.x * 16 + c;
  const bool alive = n < A.N, livedim = q < d;
  const bool isf = A.dir != nullptr ? (alive ? A.dir[n] != 0 : true) : (A.dir_all != 0);
  const float sg = isf ? 1.f : -1.f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const float rw_den = A.den;
  const f4 Z = splat(0.f);
  const float live1 = livedim ? 1.f : 0.f;

  // unit carried by MFMA row i / by k index (kq, r): rows 4 q' + r' with r' < KH are live, unit = q' KH + r'
  auto unit_row = [&](int i) { return ((i & 3) < KH) ? (i >> 2) * KH + (i & 3) : -1; };

  // ---- stage: weight fragments (A operands; element (lane (i, kq), r): row i, k-step r, k = kq), tables ------------------
  // One wave stages everything, so the loads must not wait for one another: every lane's (i, kq) is fixed, the 6 groups x 4
  // k-steps of a net are straight-line code with clamped indices (every load is issued, invalid ones are discarded by a
  // select) -- 28 independent loads per net in flight instead of 48 dependent loop trips (the staging was 17 % of a launch).
  {
    const int li = lane & 15, lkq = lane >> 4, ui = unit_row(li);
    const bool uiH = ui >= 0 && ui < H;
    const int uic = uiH ? ui : 0;
    const int hd = li & 3, dm = li >> 2;                      // heads forward: row li = (dimension dm, head hd)
    const bool okd = hd < 3 && dm < d;
    const int dmc = dm < d ? dm : 0, kqc = lkq < d ? lkq : 0;
    auto stage_net = [&](float* dst, float* tbd, const float* W1, const float* W2, const float* W3, const float* W4, const float* b1,
                         const float* b2, const float* b3, const float* b4, const float* Ws, const float* Wt, const float* Wq,
                         const float* bs, const float* bt, const float* bq) {
      const float* Wh = hd == 0 ? Ws : (hd == 1 ? Wt : Wq);
      const float* bh = hd == 0 ? bs : (hd == 1 ? bt : bq);
      f4 G[6];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uk = r < KH ? lkq * KH + r : -1;
        const bool ukH = uk >= 0 && uk < H, ukB = uk == H;
        const int ukc = ukH ? uk : 0;
        const float* Wr = r == 0 ? Ws : (r == 1 ? Wt : Wq);
        // (all loads first: clamped addresses are always valid)
        const float v0 = ukB ? b4[uic] : W4[ukc * H + uic];      // layer 2 forward: rows u' = ui, k = u = uk (+ b4, 1 -> 1)
        const float v1 = W4[uic * H + ukc];                      // layer 2 transposed: rows u = ui, k = u' = uk
        const float v2 = ukB ? bh[dmc] : Wh[ukc * d + dmc];      // heads forward: row (dimension, head), k = unit uk
        const float v3 = Wr[uic * d + kqc];                      // heads transposed: rows = units ui, k = (dimension kq, head r)
        const float v4 = W1[dmc * H + ukc], v5 = W2[dmc * H + ukc];   // layer 1 transposed: row 4 j = dimension j, k = unit uk
        G[0][r] = ((ukH || ukB) && uiH) ? v0 : ((ukB && ui == H) ? 1.f : 0.f);
        G[1][r] = (ukH && uiH) ? v1 : 0.f;
        G[2][r] = (okd && (ukH || ukB)) ? v2 : 0.f;
        G[3][r] = (r < 3 && lkq < d && uiH) ? v3 : 0.f;
        G[4][r] = (hd == 0 && dm < d && ukH) ? v4 : 0.f;
        G[5][r] = (hd == 0 && dm < d && ukH) ? v5 : 0.f;
      }
#pragma unroll
      for (int g = 0; g < 6; ++g) *reinterpret_cast<f4*>(dst + (g * 64 + lane) * 4) = G[g];
      // time / bias table of this net: row srow, unit row li
      const float w3c = W3[uic], w3s = W3[H + uic], bsum = (b1[uic] + b2[uic]) + b3[uic];
      for (int s0 = 0; s0 < T; s0 += 4) {
        const int srow = s0 + lkq;
        if (srow < T) {
          const float v = fmaf(w3c, A.trig[2 * srow], fmaf(w3s, A.trig[2 * srow + 1], bsum));
          tbd[srow * 16 + li] = ui == H ? 1.f : (uiH ? v : 0.f);
        }
      }
    };
    if (wv == 0)
      stage_net(smem + L.grp, smem + L.tb, A.xnet.W1, A.xnet.W2, A.xnet.W3, A.xnet.W4, A.xnet.b1, A.xnet.b2, A.xnet.b3, A.xnet.b4,
                A.xnet.Ws, A.xnet.Wt, A.xnet.Wq, A.xnet.bs, A.xnet.bt, A.xnet.bq);
    else
      stage_net(smem + L.grp + ng * 256, smem + L.tb + T * 16, A.vnet.W1, A.vnet.W2, A.vnet.W3, A.vnet.W4, A.vnet.b1, A.vnet.b2,
              A.vnet.b3, A.vnet.b4, A.vnet.Ws, A.vnet.Wt, A.vnet.Wq, A.vnet.bs, A.vnet.bt, A.vnet.bq);
  }
  for (int i = threadIdx.x; i < 2 * T; i += 128) smem[L.trg + i] = A.trig[i];
  for (int i = threadIdx.x; i < T * 4; i += 128) smem[L.msk + i] = (i & 3) < d ? A.masks[(i >> 2) * d + (i & 3)] : 0.f;

  // per-lane constants: layer-1 forward operands (row c <-> unit, k = q <-> dimension), exp(lam), energy parameters
  float l1xa = 0.f, l1xb = 0.f, l1va = 0.f, l1vb = 0.f, esx = 0.f, eqx = 0.f, esv = 0.f, eqv = 0.f, emu = 0.f, epr = 0.f, Gf = 0.f;
  {                                                 // (clamped addresses, all loads issued, selects afterwards: see above)
    const int ui = unit_row(c);
    const bool uok = livedim && ui >= 0 && ui < H;
    const int qc = livedim ? q : 0, o1 = qc * H + (uok ? ui : 0);
    const float a0 = A.xnet.W1[o1], a1 = A.xnet.W2[o1], a2 = A.vnet.W1[o1], a3 = A.vnet.W2[o1];
    const float s0 = A.xnet.lam_s[qc], s1 = A.xnet.lam_q[qc], s2 = A.vnet.lam_s[qc], s3 = A.vnet.lam_q[qc];
    float m0 = 0.f, p0 = 0.f, ga = 0.f, gb = 0.f;
    const int j = c >> 2, jc = j < d ? j : 0;
    if (EK != L2HMC_ENERGY_ROUGHWELL && EK != L2HMC_ENERGY_GMM) m0 = A.mu[qc];
    if (EK == L2HMC_ENERGY_GAUSS_DIAG) p0 = A.prec[qc];
    if (EK == L2HMC_ENERGY_GAUSS_DENSE) { ga = A.prec[jc * d + qc]; gb = A.prec[qc * d + jc]; }
    l1xa = uok ? a0 : 0.f; l1xb = uok ? a1 : 0.f; l1va = uok ? a2 : 0.f; l1vb = uok ? a3 : 0.f;
    esx = livedim ? expf(s0) : 0.f; eqx = livedim ? expf(s1) : 0.f;
    esv = livedim ? expf(s2) : 0.f; eqv = livedim ? expf(s3) : 0.f;
    emu = livedim ? m0 : 0.f;
    epr = livedim ? p0 : 0.f;
    // dense Gaussian: A operand of y = G dx: row 4 j <- G[j][q] (symmetrised), else 0
    if (EK == L2HMC_ENERGY_GAUSS_DENSE) Gf = (livedim && (c & 3) == 0 && j < d) ? 0.5f * (ga + gb) : 0.f;
  }
  // mixture of Gaussians (distributions.py:104-134): per component the mean of this lane's dimension, the A operand of
  // y_k = G_k (z - mu_k) (as Gf above) and the log-weight constant; KC = 8 components at most, all in registers
  float gmu[KC], gGf[KC], glc[KC];
  if (EK == L2HMC_ENERGY_GMM) {
    const int qc = livedim ? q : 0, j = c >> 2, jc = j < d ? j : 0;
    float lc[KC], mu_[KC], pa[KC], pb[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {                  // all loads first (component index clamped), selects afterwards
      const int kc = k < A.ncomp ? k : 0;
      lc[k] = A.logc[kc];
      mu_[k] = A.mu[kc * d + qc];
      pa[k] = A.prec[(kc * d + jc) * d + qc];
      pb[k] = A.prec[(kc * d + qc) * d + jc];
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const bool on = k < A.ncomp;
      glc[k] = on ? lc[k] : 0.f;
      gmu[k] = (on && livedim) ? mu_[k] : 0.f;
      gGf[k] = (on && livedim && (c & 3) == 0 && j < d) ? 0.5f * (pa[k] + pb[k]) : 0.f;
    }
  }
  __syncthreads();
  TS_MARK(0);      // (profiling builds only, tools/train_phase_timing.py) staging

  const float* grpx = smem + L.grp;
  const float* grpv = grpx + ng * 256;
  auto frag = [&](const float* gb, int g) { return lds4(gb + (g * 64 + lane) * 4); };
  auto chain4 = [&](f4 Wf, f4 in, f4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = MFMA16(Wf[r], in[r], acc);
    return acc;
  };
  auto chainK = [&](f4 Wf, f4 in, f4 acc) {
#pragma unroll
    for (int r = 0; r < KH; ++r) acc = MFMA16(Wf[r], in[r], acc);
    return acc;
  };
  // in: lane (c, q) holds val[row 4 q + r][chain c];  out: lane (i, kq) holds val[row i][chain 4 kq + r]
  float* scr = smem + L.tr;                      // operand buffer of back-propagation n: scr + (n & 1) * 7 * 320
  auto transp_put = [&](float* sb, int slot, f4 val) { *reinterpret_cast<f4*>(sb + slot * 320 + c * 20 + 4 * q) = val; };
  auto transp_get = [&](const float* sb, int slot) {
    f4 ov;
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = sb[slot * 320 + (4 * q + r) * 20 + c];
    return ov;
  };
  // gradient tiles of one net (registers of wave 1, whole reverse sweep)
  struct Acc { f4 hd, w1, w2, w4, tau; };
  float* const slot = A.ws + (long long)gridDim.x * T * TF_CK * 256 + (long long)blockIdx.x * (2 * P + 1);

  if (wv == 1) {
    // ---- wave 1: weight gradients = contractions over the 16 chains of the operands wave 0 left in LDS -------------------------
    Acc GX, GV;
    GX.hd = GX.w1 = GX.w2 = GX.w4 = GX.tau = Z;
    GV = GX;
    for (int nb = 0; nb < 4 * T; ++nb) {            // per step: V, X, X, V (the order of the reverse sweep)
      __syncthreads();
      const float* sb = scr + (nb & 1) * (7 * 320);
      const f4 th2 = transp_get(sb, 0), tdz = transp_get(sb, 1), tda1 = transp_get(sb, 2), t3 = transp_get(sb, 3),
               tb_ = transp_get(sb, 4), th1 = transp_get(sb, 5), tda2 = transp_get(sb, 6);
      // slot 3 carries (a, cos, sin, 0) of dimension 0: columns 1, 2 are the chain's (cos, sin) of its schedule row
      const f4 ta = (c & 3) == 0 ? t3 : Z;
      const f4 tt = c == 0 ? splat(1.f) : ((c == 1 || c == 2) ? t3 : Z);      // rows: 0 -> 1, 1 -> cos, 2 -> sin
      const int k = nb & 3;
      Acc& G = (k == 0 || k == 3) ? GV : GX;
      G.hd = chain4(tdz, th2, G.hd);
      G.w1 = chain4(ta, tda1, G.w1);
      G.w2 = chain4(tb_, tda1, G.w2);
      G.w4 = chain4(th1, tda2, G.w4);
      G.tau = chain4(tt, tda1, G.tau);
    }
    const int ui = unit_row(c);                    // unit on the COLUMN (lane & 15) of the weight-gradient tiles
    auto flush_tiles = [&](const Acc& G, float* Gn) {
      const int hs = H * d + d;
      if (livedim && ui >= 0 && ui <= H) {         // tile rows 4 q + r: (dimension q, head r) / row 4 q: dimension q
        if (ui < H) {
#pragma unroll
          for (int r = 0; r < 3; ++r) Gn[o.Ws + r * hs + ui * d + q] = G.hd[r];
          Gn[o.W1 + q * H + ui] = G.w1[0];
          Gn[o.W1 + (d * H + H) + q * H + ui] = G.w2[0];
        } else {                                   // the constant-1 unit: head biases
#pragma unroll
          for (int r = 0; r < 3; ++r) Gn[o.bs + r * hs + q] = G.hd[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                // dW4[u(row 4 q + r)][u'(c)]; row u = H: b4
        const int u = r < KH ? q * KH + r : -1;
        if (u >= 0 && u <= H && ui >= 0 && ui < H) {
          if (u < H) Gn[o.W4 + u * H + ui] = G.w4[r];
          else Gn[o.b4 + ui] = G.w4[r];
        }
      }
      if (q == 0 && ui >= 0 && ui < H) {           // rows 0, 1, 2 of the (1, cos, sin) product
        Gn[o.b1 + ui] = G.tau[0]; Gn[o.b2 + ui] = G.tau[0]; Gn[o.b3 + ui] = G.tau[0];
        Gn[o.W3 + ui] = G.tau[1];
        Gn[o.W3 + H + ui] = G.tau[2];
      }
    };
    // (slots are not pre-zeroed: the tiles + wave 0's log-scale sums cover every entry of [xnet | vnet] for d <= 4)
    flush_tiles(GX, slot);
    flush_tiles(GV, slot + P);
    return;
  }
  auto relu4i = [&](f4 a) {
    f4 o_ = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) o_[r] = relu_f(a[r]);
    return o_;
  };

  // ---- energies (this lane's dimension) -----------------------------------------------------------------------------------
  auto qsum = [&](float a) {                        // sum over the chain's four dimension lanes
    a = chain4_sum(a);
    return a;
  };
  // GMM: responsibilities r_k, y_k = G_k (z - mu_k) (this lane's dimension), g = sum_k r_k y_k and log sum_k e^{V_k} of the
  // point gradU was last called on -- hessvec / energy_part of the same point reuse them (oracle/l2hmc_train_oracle.py GMMTarget)
  struct GmmParts { float r[KC], y[KC], g, lse; };
  GmmParts GP;
  auto gmm_parts = [&](float z) {
    float V[KC], mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      GP.y[k] = 0.f; GP.r[k] = 0.f; V[k] = -3.0e38f;
      if (k < A.ncomp) {
        const float dx = (z - gmu[k]) * live1;
        const f4 y = MFMA16(gGf[k], dx, Z);
        GP.y[k] = y.x * live1;
        V[k] = glc[k] - 0.5f * qsum(dx * GP.y[k]);
        mx = fmaxf(mx, V[k]);
      }
    }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k < A.ncomp) { GP.r[k] = expf(V[k] - mx); se += GP.r[k]; }
    const float inv = 1.f / se;
    GP.g = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k < A.ncomp) { GP.r[k] *= inv; GP.g += GP.r[k] * GP.y[k]; }
    GP.lse = logf(se) + mx;
  };
  auto gradU = [&](float z) {
    if (EK == L2HMC_ENERGY_GAUSS_DIAG) return epr * (z - emu);
    if (EK == L2HMC_ENERGY_GAUSS_DENSE) { const f4 y = MFMA16(Gf, z - emu, Z); return y.x * live1; }
    if (EK == L2HMC_ENERGY_GMM) { gmm_parts(z); return GP.g; }
    return live1 * (z - (A.eta / rw_den) * sinf(z / rw_den));
  };
  auto hessvec = [&](float z, float vec) {           // GMM: at the point of the last gradU call
    if (EK == L2HMC_ENERGY_GAUSS_DIAG) return epr * vec;
    if (EK == L2HMC_ENERGY_GAUSS_DENSE) { const f4 y = MFMA16(Gf, vec, Z); return y.x * live1; }
    if (EK == L2HMC_ENERGY_GMM) {
      const float u = vec * live1;
      float out = GP.g * qsum(GP.g * u);
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k < A.ncomp) {
          const f4 Gu = MFMA16(gGf[k], u, Z);
          out += GP.r[k] * (Gu.x * live1 - GP.y[k] * qsum(GP.y[k] * u));
        }
      return out;
    }
    return live1 * (1.f - (A.eta / (rw_den * rw_den)) * cosf(z / rw_den)) * vec;
  };
  auto energy_part = [&](float z, float g) {        // this lane's share of U(z)  (GMM: of the last gradU point)
    if (EK == L2HMC_ENERGY_ROUGHWELL) return live1 * (0.5f * z * z + A.eta * cosf(z / rw_den));
    if (EK == L2HMC_ENERGY_GMM) return q == 0 ? -GP.lse : 0.f;
    return 0.5f * (z - emu) * g;
  };

  // ---- one net evaluation (forward): caches h1, h2, ts = tanh(zs), T, tq = tanh(zq) ------------------------------------------
  struct Cache { f4 h1, h2; float ts, Tt, tq; };
  auto net_fwd = [&](int net, float a, float b, f4 tbrow, Cache& C) {
    const float* gb = net ? grpv : grpx;
    const f4 p = MFMA16(net ? l1vb : l1xb, b, MFMA16(net ? l1va : l1xa, a, Z));
    C.h1 = relu4i(p + tbrow);
    C.h2 = relu4i(chainK(frag(gb, 0), C.h1, Z));
    const f4 z = chainK(frag(gb, 2), C.h2, Z);
    C.ts = ftanh(z.x);
    C.Tt = z.y;
    C.tq = ftanh(z.z);
  };

  // log-scale gradients of one net (this wave's share of the gradient; the weight tiles live in wave 1)
  struct Lam { float lamS, lamQ; };
  Lam GX = {0.f, 0.f}, GV = {0.f, 0.f};
  int nbp = 0;                                     // back-propagations so far: selects the operand buffer
  float cs_c = 0.f, sn_c = 0.f;                    // (cos, sin) of this chain's schedule row at the current step

  // ---- back-propagation through one net: consumes dzs, dzt, dzq (+ dA, dB for the log-scales), returns da, db -------------
  auto net_bwd = [&](int net, const Cache& C, float a, float b, float dzs, float dzt, float dzq, float dA, float dB, Lam& G,
                     float& da, float& db) {
    const float* gb = net ? grpv : grpx;
    G.lamS += dA;
    G.lamQ += dB;
    const f4 dz = f4{dzs, dzt, dzq, 0.f};
    const f4 hT = frag(gb, 3);
    f4 dh2 = MFMA16(hT[0], dz[0], Z);
    dh2 = MFMA16(hT[1], dz[1], dh2);
    dh2 = MFMA16(hT[2], dz[2], dh2);
    f4 da2 = Z, da1 = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) da2[r] = C.h2[r] > 0.f ? dh2[r] : 0.f;
    const f4 dh1 = chainK(frag(gb, 1), da2, Z);
#pragma unroll
    for (int r = 0; r < KH; ++r) da1[r] = C.h1[r] > 0.f ? dh1[r] : 0.f;
    da = chainK(frag(gb, 4), da1, Z).x;
    db = chainK(frag(gb, 5), da1, Z).x;
    TS_MARK(5);    // back-propagation, critical path (adjoints of the hidden layers and of the inputs)
    // weight gradients: contractions over the 16 chains, done by wave 1 -- the seven operands go to the LDS buffer of this
    // back-propagation (slot 3 also carries the chain's (cos, sin) for the time-embedding product), then one barrier
    float* sb = scr + (nbp & 1) * (7 * 320);
    transp_put(sb, 0, C.h2); transp_put(sb, 1, dz); transp_put(sb, 2, da1);
    transp_put(sb, 3, f4{a, q == 0 ? cs_c : 0.f, q == 0 ? sn_c : 0.f, 0.f});
    transp_put(sb, 4, f4{b, 0.f, 0.f, 0.f}); transp_put(sb, 5, C.h1); transp_put(sb, 6, da2);
    ++nbp;
    __syncthreads();
    TS_MARK(6);    // operands to LDS + hand-over barrier
  };

  // ---- per-step schedule ---------------------------------------------------------------------------------------------------
  int s_me = 0;
  float k1 = 0.f;
  f4 tbx = Z, tbv = Z;
  auto set_step = [&](int it) {
    s_me = isf ? it : (T - 1 - it);
    const float m = smem[L.msk + s_me * 4 + q];
    k1 = isf ? m : (1.f - m);
    tbx = lds4(smem + L.tb + s_me * 16 + 4 * q);
    tbv = lds4(smem + L.tb + (T + s_me) * 16 + 4 * q);
    cs_c = smem[L.trg + 2 * s_me];
    sn_c = smem[L.trg + 2 * s_me + 1];
  };

  // ---- half updates (forward) ------------------------------------------------------------------------------------------------
  float ldv = 0.f;
  auto v_half_f = [&](const Cache& C, float vin, float g) {
    const float S = esv * C.ts, Q = eqv * C.tq;
    const float ES = fexp(S * (sg * heps)), EQ = fexp(Q * eps);
    const float cc = (C.Tt - EQ * g) * heps;
    ldv += S * (sg * heps);
    return isf ? vin * ES + cc : (vin - cc) * ES;
  };
  auto x_half_f = [&](const Cache& C, float zin, float kp, float vh) {
    const float up = 1.f - kp;
    const float S = esx * C.ts, Q = eqx * C.tq;
    const float ES = fexp(S * (sg * eps)), EQ = fexp(Q * eps);
    const float tr = (EQ * vh + C.Tt) * eps;
    const float nw = isf ? zin * ES + tr : ES * (zin - tr);
    ldv += up * S * (sg * eps);
    return kp * zin + up * nw;
  };

  // ---- load the start state ---------------------------------------------------------------------------------------------------
  const float xs = (alive && livedim) ? x_row0(A, n)[n * d + q] : 0.f;
  float x = xs, v = (alive && livedim) ? A.v[n * d + q] : 0.f;
  float g = gradU(x);
  float red[6];
  red[0] = energy_part(x, g);                    // U0
  red[1] = 0.5f * v * v;                         // K0
  float* ck = A.ws + ((long long)blockIdx.x * T * TS_CK) * 64 + lane;
  auto ckp = [&](int it, int slot) -> float& { return ck[((long long)it * TS_CK + slot) * 64]; };
  auto put_cache = [&](int it, int e, const Cache& Cc) {
    const int b = 5 + 11 * e;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ckp(it, b + r) = Cc.h1[r]; ckp(it, b + 4 + r) = Cc.h2[r]; }
    ckp(it, b + 8) = Cc.ts; ckp(it, b + 9) = Cc.Tt; ckp(it, b + 10) = Cc.tq;
  };
  auto get_cache = [&](int seq) {                   // seq = 0, 1, ...: the evaluations in the order the reverse sweep meets them
    const int it = T - 1 - (seq >> 2), b = 5 + 11 * (3 - (seq & 3));
    Cache Cc;
#pragma unroll
    for (int r = 0; r < 4; ++r) { Cc.h1[r] = ckp(it, b + r); Cc.h2[r] = ckp(it, b + 4 + r); }
    Cc.ts = ckp(it, b + 8); Cc.Tt = ckp(it, b + 9); Cc.tq = ckp(it, b + 10);
    return Cc;
  };

  // ---- forward trajectory with checkpoints ------------------------------------------------------------------------------------------
  Cache C;
  for (int it = 0; it < T; ++it) {
    set_step(it);
    net_fwd(1, x, g, tbv, C);
    put_cache(it, 0, C);
    const float vh = v_half_f(C, v, g);
    net_fwd(0, vh, k1 * x, tbx, C);
    put_cache(it, 1, C);
    const float y = x_half_f(C, x, k1, vh);
    net_fwd(0, vh, (1.f - k1) * y, tbx, C);
    put_cache(it, 2, C);
    const float xo = x_half_f(C, y, 1.f - k1, vh);
    ckp(it, 0) = x; ckp(it, 1) = v; ckp(it, 2) = vh; ckp(it, 3) = y; ckp(it, 4) = xo;
    g = gradU(xo);
    net_fwd(1, xo, g, tbv, C);
    put_cache(it, 3, C);
    v = v_half_f(C, vh, g);
    x = xo;
  }

  TS_MARK(1);      // forward trajectory with checkpoints
  // ---- accept probability, loss term, adjoint seeds ----------------------------------------------------------------------------------
  red[2] = energy_part(x, g);                    // U1
  red[3] = 0.5f * v * v;                         // K1
  red[4] = (xs - x) * (xs - x);                  // |x0 - Lx|^2
  red[5] = ldv * live1;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    red[i] += __shfl_xor(red[i], 16);
    red[i] += __shfl_xor(red[i], 32);
  }
  const float val = (red[0] + red[1]) - (red[2] + red[3]) + red[5];
  const float p = accept_prob(val);
  const float sq = red[4];
  const float v1 = sq * p + 1e-4f;
  if (alive && lane < 16) { A.p[n] = p; A.v1[n] = v1; }
  const float dv1 = alive ? (A.scale * (-1.f / (v1 * v1)) - 1.f / A.scale) * A.inv_n : 0.f;
  const bool pfin = (val == val) && p > 0.f;      // finite branch of dynamics.py:309 actually taken
  const float lam = (pfin && val < 0.f) ? dv1 * sq * p : 0.f;
  const float dv1p = dv1 * p * 2.f;
  const bool okc = sq < 3.0e38f;
  if (alive && livedim) A.Lx[n * d + q] = x;
  float lx = okc ? (x - xs) * dv1p - g * lam : 0.f;
  float lv = okc ? v * (-lam) : 0.f;
  float deps = 0.f;

  // ---- adjoints of the half updates (the formulas of train_fast.hpp, scalar) ---------------------------------------------------------
  auto v_half_b = [&](const Cache& C, float dout, float vin, float gq, float& dvin, float& dg, float& dzs, float& dzt,
                      float& dzq, float& dA, float& dB) {
    const float ts = C.ts, tq = C.tq, Tt = C.Tt;
    const float S = esv * ts, Q = eqv * tq;
    const float ES = fexp(sg * heps * S), EQ = fexp(eps * Q);
    const float cc = heps * (Tt - EQ * gq);
    const float dES = isf ? dout * vin : dout * (vin - cc);
    const float dcc = isf ? dout : -dout * ES;
    const float ds = dES * ES + lam * live1;
    const float dSr = ds * sg * heps;
    const float dq = -dcc * heps * gq * EQ;
    const float dQr = dq * eps;
    dvin = dout * ES;
    dg = -dcc * heps * EQ;
    dA = dSr * S;
    dB = dQr * Q;
    dzs = dSr * esv * (1.f - ts * ts);
    dzt = dcc * heps;
    dzq = dQr * eqv * (1.f - tq * tq);
    deps += ds * sg * 0.5f * S + dcc * 0.5f * (Tt - EQ * gq) + dq * Q;
  };
  auto x_half_b = [&](const Cache& C, float dout, float zin, float kp, float vhq, float& dzin, float& dvh, float& dzs,
                      float& dzt, float& dzq, float& dA, float& dB) {
    const float up = 1.f - kp;
    const float ts = C.ts, tq = C.tq, Tt = C.Tt;
    const float S = esx * ts, Q = eqx * tq;
    const float ES = fexp(sg * eps * S), EQ = fexp(eps * Q);
    const float tr = eps * (EQ * vhq + Tt);
    const float dnw = up * dout;
    const float dES = isf ? dnw * zin : dnw * (zin - tr);
    const float dtr = isf ? dnw : -dnw * ES;
    const float dsx = dES * ES + up * lam * live1;
    const float dSr = dsx * sg * eps;
    const float dq = dtr * eps * vhq * EQ;
    const float dQr = dq * eps;
    dzin = kp * dout + dnw * ES;
    dvh += dtr * eps * EQ;
    dA = dSr * S;
    dB = dQr * Q;
    dzs = dSr * esx * (1.f - ts * ts);
    dzt = dtr * eps;
    dzq = dQr * eqx * (1.f - tq * tq);
    deps += dsx * sg * S + dtr * (EQ * vhq + Tt) + dq * Q;
  };

  TS_MARK(2);      // reductions, accept probability, adjoint seeds
  // ---- reverse sweep ---------------------------------------------------------------------------------------------------------------
  int seq = 0;
  Cache Cn = get_cache(0);                          // the next evaluation's activations, requested one evaluation ahead
  auto next_cache = [&]() {
    C = Cn;
    ++seq;
    if (seq < 4 * T) Cn = get_cache(seq);
  };
  float nx = ckp(T - 1, 0), nv = ckp(T - 1, 1), nvh = ckp(T - 1, 2), ny = ckp(T - 1, 3), nxo = ckp(T - 1, 4);
  for (int it = T - 1; it >= 0; --it) {
    set_step(it);
    const float cx = nx, cv = nv, cvh = nvh, cy = ny, cxo = nxo;
    if (it > 0) {                                // the next iteration's checkpoints: an L2 round trip taken a whole step early
      nx = ckp(it - 1, 0); nv = ckp(it - 1, 1); nvh = ckp(it - 1, 2); ny = ckp(it - 1, 3); nxo = ckp(it - 1, 4);
    }
    float dvh = 0.f, dg, dzs, dzt, dzq, dA, dB, da, db, dz;
    TS_MARK(8);    // step head: schedule record, checkpoint hand-over
    // (1) v' = v_half(vh; g(x'), V(x', g(x')))
    float gq = gradU(cxo);
    next_cache();
    TS_MARK(3);    // grad U + hand-over of the stored activations
    v_half_b(C, lv, cvh, gq, dvh, dg, dzs, dzt, dzq, dA, dB);
    TS_MARK(4);    // adjoint of the half update
    net_bwd(1, C, cxo, gq, dzs, dzt, dzq, dA, dB, GV, da, db);
    lx = lx + da + hessvec(cxo, dg + db);                         // d x'
    // (2) x' = x_half(y, k2; vh, X(vh, k2 y)),  k2 = 1 - k1
    const float k2 = 1.f - k1;
    TS_MARK(8);
    next_cache();
    TS_MARK(3);
    x_half_b(C, lx, cy, k2, cvh, dz, dvh, dzs, dzt, dzq, dA, dB); // dz = d y (direct part)
    TS_MARK(4);
    net_bwd(0, C, cvh, k2 * cy, dzs, dzt, dzq, dA, dB, GX, da, db);
    dvh += da;
    dz += k2 * db;
    // (3) y = x_half(x, k1; vh, X(vh, k1 x))
    TS_MARK(8);
    next_cache();
    TS_MARK(3);
    x_half_b(C, dz, cx, k1, cvh, lx, dvh, dzs, dzt, dzq, dA, dB); // lx = d x (direct part)
    TS_MARK(4);
    net_bwd(0, C, cvh, k1 * cx, dzs, dzt, dzq, dA, dB, GX, da, db);
    dvh += da;
    lx += k1 * db;
    // (4) vh = v_half(v; g(x), V(x, g(x)))
    TS_MARK(8);
    gq = gradU(cx);
    next_cache();
    TS_MARK(3);
    v_half_b(C, dvh, cv, gq, lv, dg, dzs, dzt, dzq, dA, dB);
    TS_MARK(4);
    net_bwd(1, C, cx, gq, dzs, dzt, dzq, dA, dB, GV, da, db);
    lx = lx + da + hessvec(cx, dg + db);
  }
  TS_MARK(8);

  // ---- this wave's part of the workgroup's flat gradient [xnet (P) | vnet (P) | eps]: the step size and the log-scales ---------------
  {
    const float s_ = wave_sum(deps * live1);
    if (lane == 0) slot[2 * P] = s_;
  }
  auto flush_lam = [&](const Lam& G, float* Gn) {  // sums over the 16 chains of the tile
    float ls = G.lamS, lq = G.lamQ;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { ls += __shfl_xor(ls, off); lq += __shfl_xor(lq, off); }
    if (c == 0 && livedim) { Gn[o.ls + q] = ls; Gn[o.lq + q] = lq; }
  };
  flush_lam(GX, slot);
  flush_lam(GV, slot + P);
  TS_FLUSH();
}
