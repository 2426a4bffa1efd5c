// train_fast.hpp -- the training gradient in the register-resident layout of the forward kernel
// (included by train.hip inside namespace l2hmc, after TArgs / NetOff / train_reduce_kernel).
//
// Same mathematics as train_kernel (hand-derived reverse mode of one direction-mixed proposal and its loss
// term, SCGExperiment.ipynb raw 156-169, incl. the Hessian-vector path through grad U; derivation =
// oracle/l2hmc_train_oracle.py), re-cast in the S-layout of traj_kernel:
//
//   * a workgroup of NW waves owns 16 chains, wave w the dimensions 16 w .. 16 w + 15; lane (c, q) holds the
//     float4 {z[16 w + 4 q + r][chain c]} of every state / adjoint vector, and {h[unit(q, r)][chain c]} of every
//     hidden vector -- the B operand and the C/D layout of v_mfma_f32_16x16x4_f32 at once;
//   * forward layers, input adjoints (transposed weights) and the hidden-layer adjoints are MFMA chains on
//     weight fragments staged ONCE per workgroup in both orientations (8 NT + 2 groups of 256 floats per net);
//     the only cross-wave traffic is ONE exchange per net evaluation and one per net back-propagation
//     (the K-split sums over dimensions), exactly as in the forward kernel;
//   * weight gradients dW(k, i) += sum_c in(c, k) dout(c, i) contract over the 16 CHAINS of the tile: both
//     operands are transposed through a per-wave LDS scratch (one ds_write_b128 + four ds_read_b32, no barrier:
//     a wave's LDS operations execute in order) and the 16x16 gradient tiles ACCUMULATE IN REGISTERS over the
//     whole reverse sweep (7 float4 per net); biases ride on the constant-1 hidden unit, so their gradients
//     are rows of those tiles; the time-embedding gradients are one more K = chains product with (1, cos, sin);
//   * no atomics: every workgroup writes its tiles to its slot of the workspace once, train_reduce_kernel adds
//     the slots in block order (bitwise reproducible, as before);
//   * round 3: straight-line staging (a wave stages its tile's fragment groups as independent clamped loads), fragment
//     requests ahead of the exchange barriers, the layer-2 tile and the time-embedding rows formed by one wave each, and
//     the forward evaluations' hidden activations checkpointed so that the reverse sweep redoes only the heads
//     (profiles/r03_train_timing.txt: 0.40 -> 0.28 ms per gradient call at ICG-50 / 4096 chains).
// Covers the elementwise targets (diagonal Gaussian, Rough Well) for d <= 64, dense Gaussians and the funnel
// (analytic Hessian-vector product through wave shuffles) for d <= 16; GMM / larger shapes stay on train_kernel.
#pragma once

struct TFLayout {
  int grp, tb, msk, trg, P, tr, red, total, NT, ng;
};
__host__ __device__ inline TFLayout tf_layout(int T, int NW) {
  TFLayout L;
  L.NT = NW;
  L.ng = 8 * L.NT + 2;
  int p = 0;
  L.grp = p; p += 2 * L.ng * 256;
  L.tb = p; p += 2 * T * 16;
  L.msk = p; p += T * 16 * L.NT;
  L.trg = p; p += (2 * T + 3) / 4 * 4;
  L.P = p; p += 2 * NW * 256;
  L.tr = p; p += NW * (320 + 32);          // per wave: 16 x 20 transpose scratch + (cos, sin) of its 16 chains
  L.red = p; p += NW * 16 * 8 + 16;
  L.total = p;
  return L;
}
// checkpointed float4 per lane and step: x, v, v_half, y, x' and, per net evaluation e = 0..3 of the step (slots 5 + 2 e, + 1; the
// first 64 lanes of the slot: the hidden vectors are the same in every wave, wave 0 writes them), the hidden activations h1, h2:
// the reverse sweep reads them back one evaluation ahead and redoes only the heads -- no layer-1 products, no cross-wave sum, no
// layer 2 in the re-evaluation (storing the head outputs too was measured: at 4096 chains its 1 GB of traffic eats the gain)
constexpr int TF_CK = 13;

template <int EK, int NW, int KH>
__global__ __launch_bounds__(64 * NW, 1) void train_fast_kernel(const TArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_poison(smem);
  TS_DECL;
  const int tid = threadIdx.x, lane = tid & 63, nthr = 64 * NW;
  const int w = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
  const int c = lane & 15, q = lane >> 4;
  const int d = A.d, H = A.H, T = A.T;
  const TFLayout L = tf_layout(T, NW);
  const int NT = L.NT, ng = L.ng, DP = 16 * NT;
  const NetOff o = net_off(d, H);
  const int P = net_params(d, H);
  const long long n = (long long)blockIdx.x * 16 + c;
  const bool alive = n < A.N;
  const bool isf = A.dir != nullptr ? (alive ? A.dir[n] != 0 : true) : (A.dir_all != 0);
  const float sg = isf ? 1.f : -1.f;
  const float eps = A.alpha != nullptr ? expf(*A.alpha) : A.eps_host;
  const float heps = 0.5f * eps;
  const float rw_den = A.den;
  const f4 Z = splat(0.f);
  constexpr int W_TAU = NW > 1 ? 1 : 0;          // the wave that accumulates the time-embedding gradient rows

  // unit carried by MFMA row i / by k index (q, r): rows 4 q' + r' with r' < KH are live, unit = q' KH + r'
  auto unit_row = [&](int i) { return ((i & 3) < KH) ? (i >> 2) * KH + (i & 3) : -1; };

  // ---- stage: weight fragments in both orientations, time/bias tables, masks -----------------------------------
  // Straight-line staging (the loop form -- one element per trip, its load waited for before the next trip -- was 20 % of
  // a launch): a lane's fragment position (row i, k-group kq) is fixed, wave w stages the groups of ITS dimension tile
  // (heads forward / transposed, layer 1 transposed) of both nets plus, for waves 0 and 1, one net's layer-2 pair; every
  // address is clamped into range, all ~40 loads per net are issued back to back and the invalid ones dropped by selects.
  {
    const int li = lane & 15, lkq = lane >> 4, ui = unit_row(li);
    const bool uiH = ui >= 0 && ui < H;
    const int uic = uiH ? ui : 0;
    const int dimr = 16 * w + li;                               // the dimension on this lane's fragment ROW
    const bool drok = dimr < d;
    const int dimrc = drok ? dimr : 0;
    auto stage_tile = [&](float* gb, bool do_l2, const float* W1, const float* W2, const float* W4, const float* b4,
                          const float* Ws, const float* Wt, const float* Wq, const float* bs, const float* bt, const float* bq) {
      f4 G0 = Z, G1 = Z, HF[3], HT[3], LT[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int uk = r < KH ? lkq * KH + r : -1;
        const bool ukH = uk >= 0 && uk < H, ukB = uk == H;
        const int ukc = ukH ? uk : 0;
        const int dimk = 16 * w + 4 * lkq + r;                  // the dimension on this element's K index
        const bool dkok = dimk < d;
        const int dimkc = dkok ? dimk : 0;
        float v0 = 0.f, v1 = 0.f;
        if (do_l2) {                                            // (wave-uniform)
          v0 = ukB ? b4[uic] : W4[ukc * H + uic];               // layer 2 forward: rows u' = ui, k = u = uk  (+ b4, + 1 -> 1)
          v1 = W4[uic * H + ukc];                               // layer 2 transposed: rows u = ui, k = u' = uk
        }
        const float f0 = ukB ? bs[dimrc] : Ws[ukc * d + dimrc]; // heads forward: rows = dims, k = units (bias on unit H)
        const float f1 = ukB ? bt[dimrc] : Wt[ukc * d + dimrc];
        const float f2 = ukB ? bq[dimrc] : Wq[ukc * d + dimrc];
        const float t0 = Ws[uic * d + dimkc], t1 = Wt[uic * d + dimkc], t2 = Wq[uic * d + dimkc];   // heads transposed
        const float l0 = W1[dimrc * H + ukc], l1 = W2[dimrc * H + ukc];                             // layer 1 transposed
        G0[r] = ((ukH || ukB) && uiH) ? v0 : ((ukB && ui == H) ? 1.f : 0.f);
        G1[r] = (ukH && uiH) ? v1 : 0.f;
        const bool hf = drok && (ukH || ukB), ht = dkok && uiH, lt = drok && ukH;
        HF[0][r] = hf ? f0 : 0.f; HF[1][r] = hf ? f1 : 0.f; HF[2][r] = hf ? f2 : 0.f;
        HT[0][r] = ht ? t0 : 0.f; HT[1][r] = ht ? t1 : 0.f; HT[2][r] = ht ? t2 : 0.f;
        LT[0][r] = lt ? l0 : 0.f; LT[1][r] = lt ? l1 : 0.f;
      }
      auto put = [&](int g, f4 v) { *reinterpret_cast<f4*>(gb + (g * 64 + lane) * 4) = v; };
      if (do_l2) { put(0, G0); put(1, G1); }
#pragma unroll
      for (int h = 0; h < 3; ++h) { put(2 + 3 * w + h, HF[h]); put(2 + 3 * NT + 3 * w + h, HT[h]); }
      put(2 + 6 * NT + 2 * w + 0, LT[0]);
      put(2 + 6 * NT + 2 * w + 1, LT[1]);
    };
    stage_tile(smem + L.grp, w == 0, A.xnet.W1, A.xnet.W2, A.xnet.W4, A.xnet.b4, A.xnet.Ws, A.xnet.Wt, A.xnet.Wq, A.xnet.bs,
               A.xnet.bt, A.xnet.bq);
    stage_tile(smem + L.grp + ng * 256, w == (NW > 1 ? 1 : 0), A.vnet.W1, A.vnet.W2, A.vnet.W4, A.vnet.b4, A.vnet.Ws, A.vnet.Wt,
               A.vnet.Wq, A.vnet.bs, A.vnet.bt, A.vnet.bq);
    // time / bias tables: thread (srow = tid / 16, unit row i = tid % 16) of each net
    const int ti = tid & 15, tui = unit_row(ti);
    const bool tuH = tui >= 0 && tui < H;
    const int tuc = tuH ? tui : 0;
    const float xw3c = A.xnet.W3[tuc], xw3s = A.xnet.W3[H + tuc], xbs = (A.xnet.b1[tuc] + A.xnet.b2[tuc]) + A.xnet.b3[tuc];
    const float vw3c = A.vnet.W3[tuc], vw3s = A.vnet.W3[H + tuc], vbs = (A.vnet.b1[tuc] + A.vnet.b2[tuc]) + A.vnet.b3[tuc];
    for (int srow = tid >> 4; srow < T; srow += nthr >> 4) {
      const float ct = A.trig[2 * srow], st = A.trig[2 * srow + 1];
      const float vx = fmaf(xw3c, ct, fmaf(xw3s, st, xbs)), vv = fmaf(vw3c, ct, fmaf(vw3s, st, vbs));
      smem[L.tb + srow * 16 + ti] = tui == H ? 1.f : (tuH ? vx : 0.f);
      smem[L.tb + (T + srow) * 16 + ti] = tui == H ? 1.f : (tuH ? vv : 0.f);
    }
  }
  for (int i = tid; i < 2 * T; i += nthr) smem[L.trg + i] = A.trig[i];
  for (int i = tid; i < T * DP; i += nthr) {
    const int row = i / DP, dim = i % DP;
    smem[L.msk + i] = dim < d ? A.masks[row * d + dim] : 0.f;
  }

  // register-resident per-lane constants: layer-1 forward fragments of this wave's tile, exp(lam), energy parameters
  const int dim0 = 16 * w + 4 * q;
  f4 l1xa, l1xb, l1va, l1vb, esx, eqx, esv, eqv, emu = Z, epr = Z, Gf = Z;
  {                                                 // (clamped addresses, all loads issued, selects afterwards)
    const int ui = unit_row(c);
    const bool uok = ui >= 0 && ui < H;
    const int uc = uok ? ui : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dim = dim0 + r;
      const bool dok = dim < d;
      const int dc = dok ? dim : 0;
      const float a0 = A.xnet.W1[dc * H + uc], a1 = A.xnet.W2[dc * H + uc], a2 = A.vnet.W1[dc * H + uc], a3 = A.vnet.W2[dc * H + uc];
      const float s0 = A.xnet.lam_s[dc], s1 = A.xnet.lam_q[dc], s2 = A.vnet.lam_s[dc], s3 = A.vnet.lam_q[dc];
      float m0 = 0.f, p0 = 0.f, ga = 0.f, gb = 0.f;
      if (EK != L2HMC_ENERGY_ROUGHWELL && EK != L2HMC_ENERGY_FUNNEL) m0 = A.mu[dc];
      if (EK == L2HMC_ENERGY_GAUSS_DIAG) p0 = A.prec[dc];
      const int cc = c < d ? c : 0;
      if (EK == L2HMC_ENERGY_GAUSS_DENSE) { ga = A.prec[cc * d + dc]; gb = A.prec[dc * d + cc]; }
      const bool ok = dok && uok;
      l1xa[r] = ok ? a0 : 0.f; l1xb[r] = ok ? a1 : 0.f; l1va[r] = ok ? a2 : 0.f; l1vb[r] = ok ? a3 : 0.f;
      esx[r] = dok ? expf(s0) : 0.f; eqx[r] = dok ? expf(s1) : 0.f;
      esv[r] = dok ? expf(s2) : 0.f; eqv[r] = dok ? expf(s3) : 0.f;
      if (EK != L2HMC_ENERGY_ROUGHWELL && EK != L2HMC_ENERGY_FUNNEL) emu[r] = dok ? m0 : 0.f;
      if (EK == L2HMC_ENERGY_GAUSS_DIAG) epr[r] = dok ? p0 : 0.f;
      if (EK == L2HMC_ENERGY_GAUSS_DENSE)       // NW == 1: A operand of y = G dx, rows = out dims c, k = 4 q + r
        Gf[r] = (c < d && dok) ? 0.5f * (ga + gb) : 0.f;
    }
  }
  const f4 live4 = f4{dim0 < d ? 1.f : 0.f, dim0 + 1 < d ? 1.f : 0.f, dim0 + 2 < d ? 1.f : 0.f, dim0 + 3 < d ? 1.f : 0.f};
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
  int pb = 0;
  auto exch = [&](f4 p) {
    if (NW == 1) return p;
    float* Pb = smem + L.P + pb * (NW * 256);
    *reinterpret_cast<f4*>(Pb + (w * 64 + lane) * 4) = p;
    __syncthreads();
    f4 s = lds4(Pb + lane * 4);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) s += lds4(Pb + (ww * 64 + lane) * 4);
    pb ^= 1;
    return s;
  };
  // in: lane (c, q) holds val[row 4 q + r][chain c];  out: lane (i, kq) holds val[row i][chain 4 kq + r]
  float* scr = smem + L.tr + w * (320 + 32);
  // (round 6: four scalar writes + ONE ds_read_b128 -- before: one ds_write_b128 + four ds_read_b32, i.e. four data returns on the
  //  critical path of every MFMA that consumes a transposed operand, nine times per net evaluation: 259.5 -> 252.5 us per
  //  gradient call at ICG-50 / 4096 chains, profiles/r06_train_timing.txt.  Banks: the writes of a wave hit (16 q + c + 20 r) mod 64,
  //  all distinct; results are the same bits.)
  auto transp = [&](f4 val) {
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[(4 * q + r) * 20 + c] = val[r];
    return *reinterpret_cast<const f4*>(scr + c * 20 + 4 * q);
  };
  auto relu4i = [&](f4 a) {
    f4 o = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) o[r] = relu_f(a[r]);
    return o;
  };

  // ---- energies ------------------------------------------------------------------------------------------------
  // funnel (distributions.py:155-180; NW == 1): v = z_0 lives in lane (c, 0) component 0; the chain's 4 lanes share
  // v, q = sum_{k >= 1} z_k^2 and the branch (free / clipped at +- 4 sigma) through wave shuffles
  struct Fun { float v, qsum, inv, s; bool clipped; };
  auto fun_parts = [&](f4 z) {
    Fun F;
    F.v = __shfl(z[0], c);
    float part = hsum(z * z) - (q == 0 ? z[0] * z[0] : 0.f);
    part = chain4_sum(part);
    F.qsum = part;
    const float clip = 4.f * A.eta;
    const bool hi = F.v > clip, lo = -clip > F.v;
    F.clipped = hi || lo;
    F.s = hi ? expf(clip) : (lo ? expf(-clip) : expf(F.v));
    F.inv = 1.f / F.s;
    return F;
  };
  auto gradU = [&](f4 z) {
    f4 g;
    if (EK == L2HMC_ENERGY_FUNNEL) {
      const Fun F = fun_parts(z);
      g = z * F.inv;
      if (q == 0) g[0] = F.v / (A.eta * A.eta) + (F.clipped ? 0.f : 0.5f * ((float)(d - 1) - F.qsum * F.inv));
      return g;
    }
    if (EK == L2HMC_ENERGY_GAUSS_DIAG) g = epr * (z - emu);
    else if (EK == L2HMC_ENERGY_GAUSS_DENSE) g = chain4(Gf, z - emu, Z);
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = z[r] - (A.eta / rw_den) * sinf(z[r] / rw_den);
      g = g * live4;
    }
    return g;
  };
  auto hessvec = [&](f4 z, f4 vec) {
    f4 o;
    if (EK == L2HMC_ENERGY_FUNNEL) {
      const Fun F = fun_parts(z);
      const float u0 = __shfl(vec[0], c);
      float dot = hsum(z * vec) - (q == 0 ? z[0] * vec[0] : 0.f);
      dot = chain4_sum(dot);
      const float fr = F.clipped ? 0.f : 1.f;
      o = vec * F.inv - z * (fr * u0 * F.inv);
      if (q == 0) o[0] = u0 * (1.f / (A.eta * A.eta) + fr * 0.5f * F.qsum * F.inv) - fr * dot * F.inv;
      return o;
    }
    if (EK == L2HMC_ENERGY_GAUSS_DIAG) o = epr * vec;
    else if (EK == L2HMC_ENERGY_GAUSS_DENSE) o = chain4(Gf, vec, Z);
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (1.f - (A.eta / (rw_den * rw_den)) * cosf(z[r] / rw_den)) * vec[r];
      o = o * live4;
    }
    return o;
  };
  auto energy_part = [&](f4 z, f4 g) {          // this lane's share of U(z)
    float u = 0.f;
    if (EK == L2HMC_ENERGY_FUNNEL) {
      const Fun F = fun_parts(z);
      const float lp = (F.v / A.eta) * (F.v / A.eta);
      return lane < 16 ? 0.5f * (lp + F.qsum * F.inv + (float)(d - 1) * logf(6.283185307179586f * F.s)) : 0.f;
    }
    if (EK == L2HMC_ENERGY_ROUGHWELL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) u += live4[r] * (0.5f * z[r] * z[r] + A.eta * cosf(z[r] / rw_den));
    } else {
      u = 0.5f * hsum((z - emu) * g);
    }
    return u;
  };

  // ---- one net evaluation (forward): caches h1, h2, ts = tanh(zs), T, tq = tanh(zq) ------------------------------
  struct Cache { f4 h1, h2, ts, Tt, tq; };
  // Layer 1 (this wave's K-split partial, then the cross-wave sum) and the rest of the evaluation are separate (round 6): the
  // forward trajectory re-uses what the inference kernel re-uses (traj_fast.hpp) -- VNet's layer-1 sum at (x', grad U(x')) is
  // the same at the end of step t and at the start of step t + 1 (only the time row differs), and both XNet evaluations of a
  // step see the same v_h -- so a step costs 3 exchanges and 20 layer-1 MFMAs instead of 4 and 32.  The re-used values are
  // the SAME MFMA chains on the same operands: results are bit-identical to evaluating them again.
  struct TailF { f4 f2, fs, ft, fq; };
  auto tail_frags = [&](int net) {
    const float* gb = net ? grpv : grpx;
    // (the fragments of the layers behind the cross-wave sum are requested BEFORE its barrier: LDS loads may not be moved
    //  across a barrier by the compiler, and their latency would otherwise sit on the chain after every exchange)
    return TailF{frag(gb, 0), frag(gb, 2 + 3 * w + 0), frag(gb, 2 + 3 * w + 1), frag(gb, 2 + 3 * w + 2)};
  };
  auto net_tail = [&](const TailF& F, f4 psum, f4 tbrow, Cache& C) {
    C.h1 = relu4i(psum + tbrow);
    C.h2 = relu4i(chainK(F.f2, C.h1, Z));
    const f4 zs = chainK(F.fs, C.h2, Z);
    const f4 zt = chainK(F.ft, C.h2, Z);
    const f4 zq = chainK(F.fq, C.h2, Z);
    C.ts = tanh4(zs);
    C.Tt = zt;
    C.tq = tanh4(zq);
  };
  auto net_fwd_sum = [&](int net, f4 pa_part, f4 pb_part, f4 tbrow, Cache& C) {      // partials -> exchange -> tail
    const TailF F = tail_frags(net);
    net_tail(F, exch(pa_part + pb_part), tbrow, C);
  };

  // gradient tiles of one net (registers, whole reverse sweep)
  struct Acc { f4 hS, hT, hQ, w1, w2, w4, tau, lamS, lamQ; };
  Acc GX, GV;
  GX.hS = GX.hT = GX.hQ = GX.w1 = GX.w2 = GX.w4 = GX.tau = GX.lamS = GX.lamQ = Z;
  GV = GX;

  // ---- back-propagation through one net: consumes dzs, dzt, dzq (+ dA, dB for the log-scales), returns the input
  //      adjoints da, db; `cs`: this wave's (cos, sin) of the 16 chains at the current step ------------------------
  auto net_bwd = [&](int net, const Cache& C, f4 a, f4 b, f4 dzs, f4 dzt, f4 dzq, f4 dA, f4 dB, Acc& G, f4& da, f4& db) {
    const float* gb = net ? grpv : grpx;
    G.lamS += dA;
    G.lamQ += dB;
    // d h2 (partial over this wave's dims, summed over the waves), d a2, d h1, d a1
    f4 dh2 = chain4(frag(gb, 2 + 3 * NT + 3 * w + 0), dzs, Z);
    dh2 = chain4(frag(gb, 2 + 3 * NT + 3 * w + 1), dzt, dh2);
    dh2 = chain4(frag(gb, 2 + 3 * NT + 3 * w + 2), dzq, dh2);
    const f4 f1 = frag(gb, 1), fa = frag(gb, 2 + 6 * NT + 2 * w + 0), fb = frag(gb, 2 + 6 * NT + 2 * w + 1);   // (as in net_fwd)
    dh2 = exch(dh2);
    f4 da2 = Z, da1 = Z;
#pragma unroll
    for (int r = 0; r < KH; ++r) da2[r] = C.h2[r] > 0.f ? dh2[r] : 0.f;
    const f4 dh1 = chainK(f1, da2, Z);
#pragma unroll
    for (int r = 0; r < KH; ++r) da1[r] = C.h1[r] > 0.f ? dh1[r] : 0.f;
    da = chainK(fa, da1, Z);
    db = chainK(fb, da1, Z);
    TS_MARK(6);    // back-propagation: adjoints of the hidden layers and of the inputs (incl. the cross-wave sum)
    // weight gradients: contractions over the 16 chains, operands transposed through the wave's scratch
    const f4 th2 = transp(C.h2);
    G.hS = chain4(transp(dzs), th2, G.hS);
    G.hT = chain4(transp(dzt), th2, G.hT);
    G.hQ = chain4(transp(dzq), th2, G.hQ);
    const f4 tda1 = transp(da1);
    G.w1 = chain4(transp(a), tda1, G.w1);
    G.w2 = chain4(transp(b), tda1, G.w2);
    // the hidden vectors are the same in every wave: ONE wave forms the layer-2 tile, ANOTHER the time-embedding rows
    // (every wave used to compute both and three of four threw them away -- the slowest wave sets the pace)
    if (w == 0) G.w4 = chain4(transp(C.h1), transp(da2), G.w4);
    if (w == W_TAU) {
      // rows: 0 -> 1, 1 -> cos, 2 -> sin of chain 4 q + r.  Branch-free (round 6): the (cos, sin) pairs of the lane's four chains
      // are two unconditional ds_read_b128 and two fmas per component -- as `c == 1 ? scr[..] : ...` the compiler built
      // twelve exec-masked blocks with a ds_read_b32 each, on the ONE wave that forms these rows (the slowest wave sets the pace)
      const f4 cs01 = lds4(scr + 320 + 8 * q), cs23 = lds4(scr + 320 + 8 * q + 4);
      const float cosv[4] = {cs01.x, cs01.z, cs23.x, cs23.z}, sinv[4] = {cs01.y, cs01.w, cs23.y, cs23.w};
      // (0/1 lane weights instead of selects on c: the compiler lowers the ternary chain to a switch with exec-masked blocks)
      const float l0 = c == 0 ? 1.f : 0.f, l1 = c == 1 ? 1.f : 0.f, l2 = c == 2 ? 1.f : 0.f;
      f4 tt;
#pragma unroll
      for (int r = 0; r < 4; ++r) tt[r] = fmaf(l2, sinv[r], fmaf(l1, cosv[r], l0));
      G.tau = chain4(tt, tda1, G.tau);
    }
    TS_MARK(7);    // weight-gradient products (operand transposes + chain contractions)
  };

  // ---- per-step schedule ------------------------------------------------------------------------------------------
  int s_me = 0;
  f4 k1 = Z, tbx = Z, tbv = Z;
  auto set_step = [&](int it) {
    s_me = isf ? it : (T - 1 - it);
    const f4 m = lds4(smem + L.msk + s_me * DP + dim0);
    k1 = isf ? m : (splat(1.f) - m);
    tbx = lds4(smem + L.tb + s_me * 16 + 4 * q);
    tbv = lds4(smem + L.tb + (T + s_me) * 16 + 4 * q);
    if (q == 0) { scr[320 + 2 * c] = smem[L.trg + 2 * s_me]; scr[320 + 2 * c + 1] = smem[L.trg + 2 * s_me + 1]; }
  };

  // ---- half updates (forward) --------------------------------------------------------------------------------------
  f4 ldv = Z;
  auto v_half_f = [&](const Cache& C, f4 vin, f4 g) {
    const f4 S = esv * C.ts, Q = eqv * C.tq;
    const f4 ES = exp4(S * (sg * heps)), EQ = exp4(Q * eps);
    const f4 cc = (C.Tt - EQ * g) * heps;
    ldv += S * (sg * heps);
    return isf ? vin * ES + cc : (vin - cc) * ES;
  };
  auto x_half_f = [&](const Cache& C, f4 zin, f4 kp, f4 vh) {
    const f4 up = splat(1.f) - kp;
    const f4 S = esx * C.ts, Q = eqx * C.tq;
    const f4 ES = exp4(S * (sg * eps)), EQ = exp4(Q * eps);
    const f4 tr = (EQ * vh + C.Tt) * eps;
    const f4 nw = isf ? zin * ES + tr : ES * (zin - tr);
    ldv += up * S * (sg * eps);
    return kp * zin + up * nw;
  };

  // ---- load the start state ----------------------------------------------------------------------------------------
  auto gload = [&](const float* p) {
    f4 r = Z;
    if (alive) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (dim0 + j < d) r[j] = p[n * d + dim0 + j];
    }
    return r;
  };
  const f4 xs = gload(x_row0(A, n));
  f4 x = xs, v = gload(A.v);
  f4 g = gradU(x);
  float red[6];
  red[0] = energy_part(x, g);                    // U0
  red[1] = 0.5f * hsum(v * v);                   // K0
  f4* ck = reinterpret_cast<f4*>(A.ws) + ((long long)blockIdx.x * T * TF_CK) * (NW * 64) + w * 64 + lane;
  auto ckp = [&](int it, int slot) -> f4& { return ck[((long long)it * TF_CK + slot) * (NW * 64)]; };

  f4* ck0 = reinterpret_cast<f4*>(A.ws) + ((long long)blockIdx.x * T * TF_CK) * (NW * 64) + lane;     // wave 0's lanes of a slot
  auto ckh = [&](int it, int slot) -> f4& { return ck0[((long long)it * TF_CK + slot) * (NW * 64)]; };
  auto put_cache = [&](int it, int e, const Cache& Cc) {
    if (w == 0) { ckh(it, 5 + 2 * e) = Cc.h1; ckh(it, 6 + 2 * e) = Cc.h2; }
  };
  struct Hid { f4 h1, h2; };
  auto get_cache = [&](int seq) {                 // seq = 0, 1, ...: the evaluations in the order the reverse sweep meets them
    const int it = T - 1 - (seq >> 2), e = 3 - (seq & 3);
    Hid h;
    h.h1 = ckh(it, 5 + 2 * e); h.h2 = ckh(it, 6 + 2 * e);
    return h;
  };
  // ---- forward trajectory with checkpoints -----------------------------------------------------------------------------
  Cache C;
  f4 pvs = exch(chain4(l1va, x, Z) + chain4(l1vb, g, Z));         // VNet layer-1 sum at the start point
  for (int it = 0; it < T; ++it) {
    set_step(it);
    net_tail(tail_frags(1), pvs, tbv, C);                         // VNet at (x, grad U(x)): the sum the previous step ended on
    put_cache(it, 0, C);
    const f4 vh = v_half_f(C, v, g);
    const f4 pa = chain4(l1xa, vh, Z);                            // shared by the step's two XNet evaluations
    net_fwd_sum(0, pa, chain4(l1xb, k1 * x, Z), tbx, C);
    put_cache(it, 1, C);
    const f4 y = x_half_f(C, x, k1, vh);
    net_fwd_sum(0, pa, chain4(l1xb, (splat(1.f) - k1) * y, Z), tbx, C);
    put_cache(it, 2, C);
    const f4 xo = x_half_f(C, y, splat(1.f) - k1, vh);
    ckp(it, 0) = x; ckp(it, 1) = v; ckp(it, 2) = vh; ckp(it, 3) = y; ckp(it, 4) = xo;
    g = gradU(xo);
    {
      const TailF F = tail_frags(1);
      pvs = exch(chain4(l1va, xo, Z) + chain4(l1vb, g, Z));
      net_tail(F, pvs, tbv, C);
    }
    put_cache(it, 3, C);
    v = v_half_f(C, vh, g);
    x = xo;
  }

  TS_MARK(1);      // forward trajectory with checkpoints
  // ---- accept probability, loss term, adjoint seeds ----------------------------------------------------------------------
  red[2] = energy_part(x, g);                    // U1
  red[3] = 0.5f * hsum(v * v);                   // K1
  red[4] = hsum((xs - x) * (xs - x));            // |x0 - Lx|^2
  red[5] = hsum(ldv);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    red[i] += __shfl_xor(red[i], 16);
    red[i] += __shfl_xor(red[i], 32);
  }
  if (NW > 1) {
    float* R = smem + L.red;
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 6; ++i) R[(w * 16 + lane) * 8 + i] = red[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) s += R[(ww * 16 + c) * 8 + i];
      red[i] = s;
    }
    __syncthreads();
  }
  const float val = (red[0] + red[1]) - (red[2] + red[3]) + red[5];
  const float p = accept_prob(val);
  const float sq = red[4];
  const float v1 = sq * p + 1e-4f;
  if (alive && w == 0 && lane < 16) { A.p[n] = p; A.v1[n] = v1; }
  const float dv1 = alive ? (A.scale * (-1.f / (v1 * v1)) - 1.f / A.scale) * A.inv_n : 0.f;
  const bool pfin = (val == val) && p > 0.f;      // finite branch of dynamics.py:309 actually taken
  const float lam = (pfin && val < 0.f) ? dv1 * sq * p : 0.f;
  const float dv1p = dv1 * p * 2.f;
  const bool okc = sq < 3.0e38f;
  if (alive) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (dim0 + j < d) A.Lx[n * d + dim0 + j] = x[j];
#ifdef L2HMC_DBG_EPILOGUE_SELECT   // NOT a feature: the six lines whose presence made roc-7.2.0's register allocator place a copy before an
                                  // exec restore in train_fast_kernel<2,1,3> (tools/check_exec_prologue.py
                                  // finds it in `hipcc -S -DL2HMC_DBG_EPILOGUE_SELECT train.hip`).  Kept as the reproducer.
    if (A.x_next != nullptr && n < A.n_head) {                 // Metropolis select of the continuing chains (sampler.py:53-55)
      const bool acc = p - A.u[n] >= 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (dim0 + j < d) A.x_next[n * d + dim0 + j] = acc ? x[j] : xs[j];
    }
#endif
  }
  f4 lx = okc ? (x - xs) * dv1p - g * lam : Z;
  f4 lv = okc ? v * (-lam) : Z;
  float deps = 0.f;

  // ---- adjoints of the half updates -----------------------------------------------------------------------------------------
  // v_half: given dout -> d vin, dg, and (dzs, dzt, dzq, dA, dB) for net_bwd
  auto v_half_b = [&](const Cache& C, f4 dout, f4 vin, f4 gq, f4& dvin, f4& dg, f4& dzs, f4& dzt, f4& dzq, f4& dA, f4& dB) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ts = C.ts[r], tq = C.tq[r], Tt = C.Tt[r];
      const float S = esv[r] * ts, Q = eqv[r] * tq;
      const float ES = fexp(sg * heps * S), EQ = fexp(eps * Q);
      const float cc = heps * (Tt - EQ * gq[r]);
      const float dES = isf ? dout[r] * vin[r] : dout[r] * (vin[r] - cc);
      const float dcc = isf ? dout[r] : -dout[r] * ES;
      const float ds = dES * ES + lam * live4[r];
      const float dSr = ds * sg * heps;
      const float dq = -dcc * heps * gq[r] * EQ;
      const float dQr = dq * eps;
      dvin[r] = dout[r] * ES;
      dg[r] = -dcc * heps * EQ;
      dA[r] = dSr * S;
      dB[r] = dQr * Q;
      dzs[r] = dSr * esv[r] * (1.f - ts * ts);
      dzt[r] = dcc * heps;
      dzq[r] = dQr * eqv[r] * (1.f - tq * tq);
      deps += ds * sg * 0.5f * S + dcc * 0.5f * (Tt - EQ * gq[r]) + dq * Q;
    }
  };
  // x_half: given dout -> d zin (direct part), dvh +=, and the net adjoints
  auto x_half_b = [&](const Cache& C, f4 dout, f4 zin, f4 kp, f4 vhq, f4& dzin, f4& dvh, f4& dzs, f4& dzt, f4& dzq, f4& dA, f4& dB) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float up = 1.f - kp[r];
      const float ts = C.ts[r], tq = C.tq[r], Tt = C.Tt[r];
      const float S = esx[r] * ts, Q = eqx[r] * tq;
      const float ES = fexp(sg * eps * S), EQ = fexp(eps * Q);
      const float tr = eps * (EQ * vhq[r] + Tt);
      const float dnw = up * dout[r];
      const float dES = isf ? dnw * zin[r] : dnw * (zin[r] - tr);
      const float dtr = isf ? dnw : -dnw * ES;
      const float dsx = dES * ES + up * lam * live4[r];
      const float dSr = dsx * sg * eps;
      const float dq = dtr * eps * vhq[r] * EQ;
      const float dQr = dq * eps;
      dzin[r] = kp[r] * dout[r] + dnw * ES;
      dvh[r] += dtr * eps * EQ;
      dA[r] = dSr * S;
      dB[r] = dQr * Q;
      dzs[r] = dSr * esx[r] * (1.f - ts * ts);
      dzt[r] = dtr * eps;
      dzq[r] = dQr * eqx[r] * (1.f - tq * tq);
      deps += dsx * sg * S + dtr * (EQ * vhq[r] + Tt) + dq * Q;
    }
  };

  TS_MARK(2);      // reductions, seeds
  // ---- reverse sweep ---------------------------------------------------------------------------------------------------------
  int seq = 0;
  Hid Hn = get_cache(0);                          // the next evaluation's hidden activations, requested one evaluation ahead
  auto next_cache = [&](int net) {                // C of the evaluation the sweep is at: stored h1, h2 + the heads redone
    const float* gb = net ? grpv : grpx;
    C.h1 = Hn.h1; C.h2 = Hn.h2;
    ++seq;
    if (seq < 4 * T) Hn = get_cache(seq);
    const f4 zs = chainK(frag(gb, 2 + 3 * w + 0), C.h2, Z);
    const f4 zt = chainK(frag(gb, 2 + 3 * w + 1), C.h2, Z);
    const f4 zq = chainK(frag(gb, 2 + 3 * w + 2), C.h2, Z);
    C.ts = tanh4(zs);
    C.Tt = zt;
    C.tq = tanh4(zq);
  };
  for (int it = T - 1; it >= 0; --it) {
    set_step(it);
    const f4 cx = ckp(it, 0), cv = ckp(it, 1), cvh = ckp(it, 2), cy = ckp(it, 3), cxo = ckp(it, 4);
    const f4 one = splat(1.f);
    f4 dvh = Z, dg, dzs, dzt, dzq, dA, dB, da, db, dz;
    // (1) v' = v_half(vh; g(x'), V(x', g(x')))
    f4 gq = gradU(cxo);
    TS_MARK(3);
    next_cache(1);
    TS_MARK(5);    // stored hidden activations + the heads redone
    v_half_b(C, lv, cvh, gq, dvh, dg, dzs, dzt, dzq, dA, dB);
    TS_MARK(8);    // adjoint of the half update
    net_bwd(1, C, cxo, gq, dzs, dzt, dzq, dA, dB, GV, da, db);
    lx = lx + da + hessvec(cxo, dg + db);                         // d x'
    // (2) x' = x_half(y, k2; vh, X(vh, k2 y)),  k2 = 1 - k1
    const f4 k2 = one - k1;
    TS_MARK(3);
    next_cache(0);
    TS_MARK(5);
    x_half_b(C, lx, cy, k2, cvh, dz, dvh, dzs, dzt, dzq, dA, dB); // dz = d y (direct part)
    TS_MARK(8);
    net_bwd(0, C, cvh, k2 * cy, dzs, dzt, dzq, dA, dB, GX, da, db);
    dvh += da;
    dz += k2 * db;
    // (3) y = x_half(x, k1; vh, X(vh, k1 x))
    TS_MARK(3);
    next_cache(0);
    TS_MARK(5);
    x_half_b(C, dz, cx, k1, cvh, lx, dvh, dzs, dzt, dzq, dA, dB); // lx = d x (direct part)
    TS_MARK(8);
    net_bwd(0, C, cvh, k1 * cx, dzs, dzt, dzq, dA, dB, GX, da, db);
    dvh += da;
    lx += k1 * db;
    // (4) vh = v_half(v; g(x), V(x, g(x)))
    gq = gradU(cx);
    TS_MARK(3);
    next_cache(1);
    TS_MARK(5);
    v_half_b(C, dvh, cv, gq, lv, dg, dzs, dzt, dzq, dA, dB);
    TS_MARK(8);
    net_bwd(1, C, cx, gq, dzs, dzt, dzq, dA, dB, GV, da, db);
    lx = lx + da + hessvec(cx, dg + db);
  }

  TS_MARK(3);      // reverse sweep
  // ---- this workgroup's flat gradient [xnet (P) | vnet (P) | eps] -> its slot of the workspace ------------------------------------
  float* slot = A.ws + (long long)gridDim.x * T * TF_CK * (NW * 256) + (long long)blockIdx.x * (2 * P + 1);
  {
    float s = wave_sum(deps);
    float* R = smem + L.red + NW * 16 * 8;
    if (lane == 0) R[w] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int ww = 0; ww < NW; ++ww) t += R[ww];
      slot[2 * P] = t;
    }
  }
  const int ui = unit_row(c);                    // unit on the COLUMN (lane & 15) of the weight-gradient tiles
  auto flush = [&](const Acc& G, float* Gn) {
    const int hs = H * d + d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dim = dim0 + r;
      if (dim < d && ui >= 0 && ui <= H) {
        if (ui < H) {
          Gn[o.Ws + 0 * hs + ui * d + dim] = G.hS[r];
          Gn[o.Ws + 1 * hs + ui * d + dim] = G.hT[r];
          Gn[o.Ws + 2 * hs + ui * d + dim] = G.hQ[r];
          Gn[o.W1 + dim * H + ui] = G.w1[r];
          Gn[o.W1 + (d * H + H) + dim * H + ui] = G.w2[r];
        } else {                                 // the constant-1 unit: head biases
          Gn[o.bs + 0 * hs + dim] = G.hS[r];
          Gn[o.bs + 1 * hs + dim] = G.hT[r];
          Gn[o.bs + 2 * hs + dim] = G.hQ[r];
        }
      }
    }
    if (w == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {              // dW4[u(row 4 q + r)][u'(c)]; row u = H: b4
        const int u = r < KH ? q * KH + r : -1;
        if (u >= 0 && u <= H && ui >= 0 && ui < H) {
          if (u < H) Gn[o.W4 + u * H + ui] = G.w4[r];
          else Gn[o.b4 + ui] = G.w4[r];
        }
      }
    }
    if (w == W_TAU) {
      if (q == 0 && ui >= 0 && ui < H) {         // rows 0, 1, 2 of the (1, cos, sin) product
        Gn[o.b1 + ui] = G.tau[0]; Gn[o.b2 + ui] = G.tau[0]; Gn[o.b3 + ui] = G.tau[0];
        Gn[o.W3 + ui] = G.tau[1];
        Gn[o.W3 + H + ui] = G.tau[2];
      }
    }
    // log-scales: sums over the 16 chains of the tile
    f4 ls = G.lamS, lq = G.lamQ;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ls[r] += __shfl_xor(ls[r], off); lq[r] += __shfl_xor(lq[r], off); }
    }
    if (c == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (dim0 + r < d) { Gn[o.ls + dim0 + r] = ls[r]; Gn[o.lq + dim0 + r] = lq[r]; }
    }
  };
  flush(GX, slot);
  flush(GV, slot + P);
  TS_MARK(4);      // flush
  TS_FLUSH();
}
