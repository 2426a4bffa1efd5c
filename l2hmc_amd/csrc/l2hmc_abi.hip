// libl2hmc_hip.so -- C ABI (include/l2hmc.h) of the L2HMC leapfrog hot path for MI355X:
// argument validation, LDS planning, geometry selection, weight packing kernels.
// The fused kernels live in l2hmc_kernels.hpp, instantiated per energy kind in traj_ek*.hip.
#include <cstdlib>
#include "traj_tile.hpp"
#include "traj_lane.hpp"

namespace l2hmc {

thread_local char g_err[512] = "";
thread_local char g_kernel[96] = "";      // the kernel the last trajectory / training call of this thread launched
void note_kernel(const char* fmt, long long a, long long b, long long c, long long d) { snprintf(g_kernel, sizeof(g_kernel), fmt, a, b, c, d); }

int fail(int code, const char* fmt, const char* a, long long b, long long c) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

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

// lane layout of one net from its reference-layout weights (l2hmc_pack_nets): zero padded to the compiled dimension count
// and hidden width (traj_lane.hpp)
__global__ void pack_lane_kernel(L2hmcNet net, int d, int H, float* out) {
  const LaneLayout L = lane_layout(d, H);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L.total) return;
  float val = 0.f;
  if (idx < L.tb) {
    const int which = idx >= L.l1b, r = (idx - (which ? L.l1b : 0)) / L.RS, j = (idx - (which ? L.l1b : 0)) % L.RS;
    if (r < d && j < H) val = (which ? net.W2 : net.W1)[r * H + j];
  } else if (idx < L.l2) {
    const int r = (idx - L.tb) / L.RS, j = (idx - L.tb) % L.RS;
    if (j < H) val = r == 0 ? net.W3[j] : (r == 1 ? net.W3[H + j] : (net.b1[j] + net.b2[j]) + net.b3[j]);
  } else if (idx < L.b4) {
    const int r = (idx - L.l2) / L.RS, j = (idx - L.l2) % L.RS;
    if (r < H && j < H) val = net.W4[r * H + j];
  } else if (idx < L.hd) {
    const int j = idx - L.b4;
    if (j < H) val = net.b4[j];
  } else {
    const int p = (idx - L.hd) / L.HB, o = (idx - L.hd) % L.HB;
    if (p < L.DP / 2) {
      if (o < 6 * L.HU) {
        const int head = o / (2 * L.HU), j = (o % (2 * L.HU)) / 2, k = 2 * p + (o & 1);
        const float* Wh = head == 0 ? net.Ws : (head == 1 ? net.Wt : net.Wq);
        if (k < d && j < H) val = Wh[j * d + k];
      } else {
        const int c = (o - 6 * L.HU) / 2, k = 2 * p + ((o - 6 * L.HU) & 1);
        if (k < d) {
          if (c == 0) val = net.bs[k];
          else if (c == 1) val = net.bt[k];
          else if (c == 2) val = net.bq[k];
          else if (c == 3) val = expf(net.lam_s[k]);
          else if (c == 4) val = expf(net.lam_q[k]);
        }
      }
    }
  }
  out[idx] = val;
}

// f16x2 fragments of one packed net (all its groups): thread = (group, lane)
__global__ void pack_f16_kernel(const float* src, int groups, float* dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= groups * 64) return;
  const WF16 f = wsplit16(reinterpret_cast<const f4*>(src)[idx]);
  reinterpret_cast<h8v*>(dst)[idx] = f.a1;
  reinterpret_cast<h8v*>(dst + (size_t)groups * 256)[idx] = f.a2;
}

__global__ void step_prep_kernel(float* trig, float c, float s) {
  if (threadIdx.x == 0) { trig[0] = c; trig[1] = s; }
}
__global__ void add_inplace_kernel(float* acc, const float* inc, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] += inc[i];
}

__global__ void pack_gauss_kernel(const float* S, int d, int NT, float* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gauss_floats(NT)) return;
  const int g = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3;
  const int to = g / NT, ti = g % NT;
  const int a = 16 * to + (lane & 15), b = 16 * ti + 4 * (lane >> 4) + r;
  out[idx] = (a < d && b < d) ? 0.5f * (S[a * d + b] + S[b * d + a]) : 0.f;
}

// sum 1 / v and sum v in double: one workgroup, fixed order (thread t takes elements t, t + 256, ...; butterfly inside each
// wave, the four wave sums added in wave order: one barrier)
__global__ __launch_bounds__(256) void loss_terms_kernel(const float* v1, long long n, float scale, double inv_n, double* out3) {
  __shared__ double sa[4], sb[4];
  double a = 0.0, b = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const float v = v1[i];
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
    out3[0] = A_; out3[1] = B_;
    out3[2] = inv_n * ((double)scale * A_ - B_ / (double)scale);
  }
}

__global__ void mh_select_kernel(const float* x, const float* Lx, const float* px, const float* u,
                                 long long N, int d, float* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * d) return;
  const long long n = i / d;
  out[i] = (px[n] - u[n] >= 0.f) ? Lx[i] : x[i];
}

// Adam as tf.train.AdamOptimizer applies it (SCGExperiment.ipynb raw 178-181), over the flat parameter vector
// [XNet | VNet | alpha]; `grad` is the buffer l2hmc_train_propose_grad accumulates into.
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr_t, float b1,
                            float b2, float eps, int last_is_log_eps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  if (last_is_log_eps && i == n - 1) gi *= expf(p[i]);      // d/d alpha = eps d/d eps (dynamics.py:50-58)
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

// AIS bookkeeping around one annealed HMC transition (utils/ais.py:44-66); thread = chain.
__global__ void ais_begin_kernel(const float* x, const float* U1, const float* z, float refreshment, float dbeta,
                                 float* w, float* v, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float keep = refreshment < 0.f ? 0.f : sqrtf(1.f - refreshment);
  const float mix = refreshment < 0.f ? 1.f : sqrtf(refreshment);
  float q = 0.f;
  for (int k = 0; k < d; ++k) {
    const float xv = x[n * d + k];
    q += xv * xv;
    v[n * d + k] = refreshment < 0.f ? z[n * d + k] : v[n * d + k] * keep + z[n * d + k] * mix;
  }
  w[n] = w[n] + dbeta * (-U1[n] + 0.5f * q);
}
__global__ void ais_end_kernel(const float* Lx, const float* Lv, const float* p, const float* u, float* x,
                               float* v, float* alpha_sum, long long N, int d) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const bool acc = p[n] - u[n] >= 0.f;
  for (int k = 0; k < d; ++k) {
    const float lv = Lv[n * d + k];
    if (acc) x[n * d + k] = Lx[n * d + k];
    v[n * d + k] = acc ? lv : -lv;                // (rejected: the NEGATED PROPOSED momentum, ais.py:63)
  }
  if (alpha_sum != nullptr) alpha_sum[n] += p[n];
}

// The sampler's Philox draws written out (l2hmc_rng_fill): thread = (proposal, chain, 4-dim block).
__global__ void rng_fill_kernel(unsigned long long seed, unsigned long long prop0, long long chain_off,
                                long long N, int d, int M, float* v, unsigned char* dir, float* u) {
  const int nblk = (d + 3) / 4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * N * nblk) return;
  const int blk = (int)(i % nblk);
  const long long n = (i / nblk) % N, m = i / ((long long)nblk * N);
  if (v != nullptr) {
    const f4 z = philox_normal4(seed, chain_off + n, (unsigned)blk, prop0 + m);
    float* row = v + (m * N + n) * d + 4 * blk;
    if (4 * blk + 0 < d) row[0] = z.x;
    if (4 * blk + 1 < d) row[1] = z.y;
    if (4 * blk + 2 < d) row[2] = z.z;
    if (4 * blk + 3 < d) row[3] = z.w;
  }
  if (blk == 0 && (dir != nullptr || u != nullptr)) {
    bool f;
    float uu;
    philox_dir_u(seed, chain_off + n, prop0 + m, f, uu);
    if (dir != nullptr) dir[m * N + n] = f ? 1 : 0;
    if (u != nullptr) u[m * N + n] = uu;
  }
}

// K7: raw autocovariance sums.  Thread = one series j = (chain, dim) of the (steps, J) history;
// block = 256 consecutive series (coalesced rows) x 32 lags; per-lag block reduction in LDS, one
// double atomicAdd per (block, lag).
const int kLagTile = 32;
__global__ __launch_bounds__(256) void autocov_kernel(const float* X, long long steps, long long J,
                                                      double* S, double* part_out) {
  __shared__ double part[4][kLagTile];
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tau0 = (long long)blockIdx.y * kLagTile;
  float acc[kLagTile];
#pragma unroll
  for (int l = 0; l < kLagTile; ++l) acc[l] = 0.f;
  double dacc[kLagTile];
#pragma unroll
  for (int l = 0; l < kLagTile; ++l) dacc[l] = 0.0;
  if (j < J) {
    const float* col = X + j;
    long long t = 0;
    // fp32 inner chunks of 64 steps, folded into fp64 (the reference accumulates in float64)
    for (long long tb = 0; tb < steps - tau0; tb += 64) {
      const long long te = (tb + 64 < steps - tau0) ? tb + 64 : steps - tau0;
      for (t = tb; t < te; ++t) {
        const float x0 = col[t * J];
#pragma unroll
        for (int l = 0; l < kLagTile; ++l) {
          const long long t2 = t + tau0 + l;
          const float x1 = t2 < steps ? col[t2 * J] : 0.f;
          acc[l] = fmaf(x0, x1, acc[l]);
        }
      }
#pragma unroll
      for (int l = 0; l < kLagTile; ++l) { dacc[l] += (double)acc[l]; acc[l] = 0.f; }
    }
  }
  // wave reduction, then across the 4 waves through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < kLagTile; ++l) {
    double v = dacc[l];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) part[wv][l] = v;
  }
  __syncthreads();
  if (threadIdx.x < kLagTile) {
    const long long tau = tau0 + threadIdx.x;
    if (tau < steps - 1) {
      const double v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
      // with a partial buffer the blocks' contributions are added later in block order (bitwise reproducible);
      // without one (caller gave no workspace) they are accumulated atomically
      if (part_out != nullptr) part_out[(long long)blockIdx.x * (steps - 1) + tau] = v;
      else atomicAdd(&S[tau], v);
    }
  }
}

// S[tau] = sum over blocks, in block order
__global__ void autocov_reduce_kernel(const double* part, long long nblk, long long nlag, double* S) {
  const long long tau = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tau >= nlag) return;
  double a = 0.0;
  for (long long b = 0; b < nblk; ++b) a += part[b * nlag + tau];
  S[tau] = a;
}

__global__ void autocov_finish_kernel(const double* S, long long steps, double inv, double* A) {
  const long long tau = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tau < steps - 1) A[tau] = S[tau] * inv / (double)(steps - tau);
}

// ------------------------------------------------------------------------------------------
// Host side: LDS planning and dispatch
// ------------------------------------------------------------------------------------------
int round4(int v) { return (v + 3) & ~3; }

// Fills the LDS offsets of `k`; returns the dynamic LDS size in bytes.
long long plan_lds(KArgs& k, bool with_nets, bool with_schedule, int NW, int DT) {
  const bool wg = weights_in_global(DT);
  with_nets = with_nets && !wg;
  const int NT = k.NT, DP = 16 * NT;
  long long o = 0;
  if (with_nets) o += 2LL * (net_floats(NT) - (DT <= 2 ? 2 * NT * 256 : 0));   // layer-1 groups live in registers
  k.o_mask = (int)o;
  if (with_schedule) o += (long long)k.T * DP;
  k.o_trig = (int)o;
  if (with_schedule) o += round4(2 * k.T);
  k.o_tb = (int)o;
  if (with_schedule) o += 2LL * k.T * 16;
  k.o_P = (int)o;
  if (NW > 1) o += 2LL * NW * 256;       // 2 buffers x NW waves x 1 partial vector
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
  if ((k.ekind == L2HMC_ENERGY_GAUSS_DENSE || k.ekind == L2HMC_ENERGY_GMM) && !wg)
    o += (long long)nc * gauss_floats(NT);
  k.o_logc = (int)o;
  o += round4(nc);
  return o * 4;
}

// LDS plan of traj_fast_kernel (DT <= 2): scaled tail fragments, constant tables, schedule records, then
// the exchange / reduction / energy areas as in plan_lds.
long long plan_lds_fast(KArgs& k, int NW, int DT, bool f16) {
  const int NT = k.NT, DP = 16 * NT, NTp = NW * DT;
  long long o = 0;
  k.o_fw = (int)o;
  o += 2LL * fast_fw_net(NTp, f16);
  k.o_fc = (int)o;
  o += 2LL * fast_fc_net(NTp);
  k.o_rec = (int)o;
  o += 2LL * fast_rec_dir(NTp, k.T);
  k.o_P = (int)o;
  if (NW > 1) o += 2LL * NW * 256;
  k.xb_stride = DP + 4;
  k.o_XB = (int)o;
  if (NW > 1 && (k.ekind == L2HMC_ENERGY_GAUSS_DENSE || k.ekind == L2HMC_ENERGY_GMM)) o += 16LL * k.xb_stride;
  k.o_red = (int)o;
  o += (long long)NW * 16 * 8;
  const int nc = k.ekind == L2HMC_ENERGY_GMM ? k.ncomp : 1;
  k.o_mu = (int)o;
  o += (long long)nc * DP;
  k.o_prec = (int)o;
  if (k.ekind == L2HMC_ENERGY_GAUSS_DIAG) o += DP;
  if (k.ekind == L2HMC_ENERGY_GAUSS_DENSE || k.ekind == L2HMC_ENERGY_GMM) o += (long long)nc * gauss_floats(NT);
  k.o_logc = (int)o;
  o += round4(nc);
  return o * 4;
}

// LDS plan of traj_tile_kernel: scaled tail fragments, layer-1 fragments (o_state), constants, records, energy vectors
long long plan_lds_tile(KArgs& k, int DT) {
  long long o = 0;
  k.o_fw = (int)o;
  o += 2LL * tile_fw_net(DT);
  k.o_state = (int)o;
  o += tile_l1_floats(DT);
  k.o_fc = (int)o;
  o += 2LL * fast_fc_net(DT);
  k.o_rec = (int)o;
  o += 2LL * fast_rec_dir(DT, k.T);
  k.o_mu = (int)o;
  o += 16LL * DT;
  k.o_prec = (int)o;
  o += 16LL * DT;
  k.o_logc = (int)o;
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
  if (!(e->anneal_beta >= 0.f && e->anneal_beta <= 1.f)) return fail(L2HMC_ERR_ARG, "anneal_beta must be in [0, 1] (0 = off)%s");
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
  k.den = roughwell_den(e);
  k.temperature = e->temperature;
  k.beta = e->anneal_beta > 0.f ? e->anneal_beta : 1.f;
}

// (DT, NW) geometry for d dimensions.  NW = 4 spreads a 16-chain tile over the 4 SIMDs of
// a CU (more parallelism per chain: right when there are few chains); NW = 1 keeps a tile
// in one wave (no LDS exchange, fewer MFMAs: right when chains are plentiful).
// Compute units of the current device (256 on an MI355X in SPX mode; fewer in a CPX / NPS partition or on a cut-down part).
// The dispatcher's chain-count thresholds were MEASURED on 256 CUs and are statements about tiles per CU, so they scale
// with this number: 32 / 64 / 256 / 512 chains per CU.  Queried per call (a host-side attribute read, no device work);
// nothing is cached, the library keeps no state.
int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  return cus;
}

bool pick_geometry(int d, long long N, int variant, int& DT, int& NW, int cus = 0) {
  if (cus <= 0) cus = device_cus();        // (l2hmc_trajectory queries once and passes it down)
  const int NT = tiles_of(d);
  if (NT <= 1) { DT = 1; NW = 1; return variant == 0 || variant == 1; }
  if (variant == 2 && NT >= 3 && NT <= 4) { DT = 2; NW = 2; return true; }   // two waves x two tiles (fast kernel only)
  if (NT <= 4) {
    // measured (tools/bench_configs.py, 16-chain tiles on 256 CUs): with 3-4 dim-tiles the 4-wave tile wins at
    // every chain count (1.7e9 vs 1.1e9 steps/s at d = 50..64); with 2 dim-tiles half of its waves idle, so it
    // only pays while there are fewer tiles than wave slots (N < 8192; 2x slower than one wave per tile above).
    // (Two waves x two tiles was tried for 3-4 dim-tiles: never faster than four waves x one tile.)
    const bool want4 = variant == 4 || (variant == 0 && (NT >= 3 || N < 32LL * cus));
    if (want4) { DT = 1; NW = 4; } else { DT = NT <= 2 ? 2 : 4; NW = 1; }
    return variant == 0 || variant == 1 || variant == 4;
  }
  if (variant == 1) return false;
  NW = 4;
  DT = NT <= 8 ? 2 : (NT <= 16 ? 4 : 8);
  return NT <= 32;
}

int dispatch(int op, const KArgs& k, int DT, int NW, int KH, long long lds, hipStream_t s) {
  switch (k.ekind) {
    case L2HMC_ENERGY_GAUSS_DIAG: return launch_ek<1>(op, k, DT, NW, KH, lds, s);
    case L2HMC_ENERGY_GAUSS_DENSE: return launch_ek<2>(op, k, DT, NW, KH, lds, s);
    case L2HMC_ENERGY_GMM: return launch_ek<3>(op, k, DT, NW, KH, lds, s);
    case L2HMC_ENERGY_ROUGHWELL: return launch_ek<4>(op, k, DT, NW, KH, lds, s);
    case L2HMC_ENERGY_FUNNEL: return launch_ek<5>(op, k, DT, NW, KH, lds, s);
  }
  return fail(L2HMC_ERR_ARG, "unknown energy kind%s");
}

}  // namespace l2hmc

using namespace l2hmc;

#ifdef L2HMC_PHASE_TIMING
static unsigned long long* g_dbg = nullptr;
extern "C" void l2hmc_set_debug_buffer(void* p) { g_dbg = (unsigned long long*)p; }
#define L2HMC_DBG_PTR g_dbg
#else
#define L2HMC_DBG_PTR nullptr
#endif

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int l2hmc_abi_version(void) { return L2HMC_ABI_VERSION; }

int32_t l2hmc_last_kernel(char* buf, int32_t n) {
  if (!buf || n < 1) return fail(L2HMC_ERR_ARG, "l2hmc_last_kernel: bad argument%s");
  snprintf(buf, (size_t)n, "%s", g_kernel);
  return (int32_t)strlen(g_kernel);
}

const char* l2hmc_last_error(void) { return g_err; }

int64_t l2hmc_struct_bytes(int32_t which) {
  switch (which) {
    case L2HMC_STRUCT_NET: return sizeof(L2hmcNet);
    case L2HMC_STRUCT_ENERGY: return sizeof(L2hmcEnergy);
    case L2HMC_STRUCT_TRAJECTORY_ARGS: return sizeof(L2hmcTrajectoryArgs);
    case L2HMC_STRUCT_MLP3: return sizeof(L2hmcMlp3);
    case L2HMC_STRUCT_SPLIT_ARGS: return sizeof(L2hmcSplitArgs);
    case L2HMC_STRUCT_TRAIN_ARGS: return sizeof(L2hmcTrainArgs);
    case L2HMC_STRUCT_TRAIN_SPLIT_ARGS: return sizeof(L2hmcTrainSplitArgs);
    case L2HMC_STRUCT_TRAIN_STEP: return sizeof(L2hmcTrainStep);
  }
  return fail(L2HMC_ERR_ARG, "l2hmc_struct_bytes: unknown struct id%s");
}

int64_t l2hmc_packed_nets_floats(int32_t d, int32_t H) {
  if (d < 1 || H < 1) return fail(L2HMC_ERR_ARG, "d and H must be >= 1%s");
  if (H > 15) return fail(L2HMC_ERR_UNSUPPORTED, "fused nets support H <= 15 (got %s%lld)", "", H);
  if (d > 512) return fail(L2HMC_ERR_UNSUPPORTED, "fused nets support d <= 512 (got %s%lld)", "", d);
  // MFMA fragments, then the lane layout, then (round 6) the fragments once more as f16x2 pairs
  return 2LL * net_floats(tiles_of(d)) + 2LL * lane_layout(d, H).total + 2LL * net_f16_floats(tiles_of(d));
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
    const int LT = lane_layout(d, H).total;                     // traj_lane.hpp: wave-uniform rows for the scalar loads
    hipLaunchKernelGGL(pack_lane_kernel, dim3((LT + 255) / 256), dim3(256), 0, (hipStream_t)stream, *nets[i], d, H,
                       packed + 2 * (size_t)NF + (size_t)i * LT);
    const int NG = net_groups(NT);
    hipLaunchKernelGGL(pack_f16_kernel, dim3((NG * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, packed + (size_t)i * NF, NG,
                       packed + 2 * (size_t)NF + 2 * (size_t)LT + (size_t)i * net_f16_floats(NT));
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

// dense Gaussian on the LDS-resident-state kernel from the same width as the elementwise targets (measured, tools/probe_dense_wide.py:
// x1.2-1.7 over the register-resident kernel at d = 160 ... 256, x6-8 at d = 384 / 512 where that one spills)
#ifndef WIDE_DENSE_MIN_NT
#define WIDE_DENSE_MIN_NT 8
#endif

// L2HMC_F32_MFMA=1 in the environment: every trajectory on the f32-input MFMA (as variant 200 + v does per call)
static bool env_f32_mfma() {
  static const int v = [] { const char* e = getenv("L2HMC_F32_MFMA"); return (e && e[0] && e[0] != '0') ? 1 : 0; }();
  return v != 0;
}

int l2hmc_trajectory(const L2hmcTrajectoryArgs* a_in, void* stream) {
  if (!a_in) return fail(L2HMC_ERR_ARG, "args is NULL%s");
  // variant 200 + v: geometry choice v with the f32-input MFMA forced (no f16x2 contraction anywhere)
  L2hmcTrajectoryArgs a_loc = *a_in;
  const bool force_f32 = a_loc.variant >= 200 || env_f32_mfma();
  if (a_loc.variant >= 200) a_loc.variant -= 200;
  const L2hmcTrajectoryArgs* a = &a_loc;
  if (a->n_chains < 0 || a->d < 1 || a->T < 1) return fail(L2HMC_ERR_ARG, "bad n_chains / d / T%s");
  if (a->n_chains == 0) return L2HMC_OK;
  if (!a->x || !a->masks || !a->trig) return fail(L2HMC_ERR_ARG, "x, masks, trig are required%s");
  if (!a->v && !(a->rng_flags & L2HMC_RNG_V)) return fail(L2HMC_ERR_ARG, "v is required unless L2HMC_RNG_V is set%s");
  if (a->step_begin < 0 || a->n_steps < 0 || a->step_begin + a->n_steps > a->T)
    return fail(L2HMC_ERR_ARG, "steps [%s%lld, +%lld) outside the T-step schedule", "", a->step_begin, a->n_steps);
  const bool has_u = a->u != nullptr || (a->rng_flags & L2HMC_RNG_U);
  if (a->x_next && !has_u) return fail(L2HMC_ERR_ARG, "x_next needs u (or L2HMC_RNG_U)%s");
  if (a->n_proposals > 1 && !has_u) return fail(L2HMC_ERR_ARG, "n_proposals > 1 needs u (the MH step links the proposals)%s");
  if (a->n_proposals < 0) return fail(L2HMC_ERR_ARG, "n_proposals must be >= 0%s");
  if (a->x_hist == a->x && a->x) return fail(L2HMC_ERR_ARG, "x_hist must not alias x%s");
  if (a->x_out == a->x || a->x_next == a->x) return fail(L2HMC_ERR_ARG, "x_out / x_next must not alias x%s");
  if (!a->alpha && !(a->eps_host > 0.f)) return fail(L2HMC_ERR_ARG, "eps must be > 0%s");
  int rc = check_energy(&a->energy, a->d);
  if (rc) return rc;
  int KH = 3;
  if (a->packed_nets) {
    if (l2hmc_packed_nets_floats(a->d, a->H) < 0) return L2HMC_ERR_UNSUPPORTED;
    KH = khid_of(a->H);
  }
  KArgs k;
  memset(&k, 0, sizeof(k));
  k.packed = a->packed_nets;
  k.packed16 = (a->packed_nets && a->H <= 15) ? a->packed_nets + 2 * (size_t)net_floats(tiles_of(a->d)) + 2 * (size_t)lane_layout(a->d, a->H).total : nullptr;
  k.masks = a->masks; k.trig = a->trig; k.alpha = a->alpha;
  k.eps_host = a->eps_host; k.N = a->n_chains; k.d = a->d; k.H = a->H; k.T = a->T;
  k.step_begin = a->step_begin; k.n_steps = a->n_steps; k.NT = tiles_of(a->d);
  k.x = a->x; k.v = a->v; k.dir = a->direction; k.dir_all = a->direction_all; k.u = a->u;
  k.x_out = a->x_out; k.v_out = a->v_out; k.logjac_out = a->logjac_out; k.p_out = a->p_out;
  k.x_next = a->x_next;
  k.x_hist = a->x_hist;
  k.M = a->n_proposals > 1 ? a->n_proposals : 1;
  k.rng_flags = a->rng_flags; k.rng_seed = a->rng_seed; k.rng_prop0 = a->rng_proposal0;
  k.chain_off = a->chain_offset;
  k.dbg = L2HMC_DBG_PTR;
  if (a->ais_beta != nullptr) {
    if (a->packed_nets != nullptr) return fail(L2HMC_ERR_UNSUPPORTED, "AIS mode runs HMC transitions (packed_nets must be NULL, utils/ais.py:60)%s");
    if (!has_u || !a->ais_w) return fail(L2HMC_ERR_ARG, "AIS mode needs u (or L2HMC_RNG_U) and ais_w%s");
    if (a->n_steps < 1 || a->ais_refreshment > 1.f) return fail(L2HMC_ERR_ARG, "AIS mode: bad n_steps / ais_refreshment%s");
    if (a->ais_refreshment >= 0.f && !a->ais_v0 && !((a->rng_flags & L2HMC_RNG_V) && a->rng_proposal0 >= 1))
      return fail(L2HMC_ERR_ARG, "AIS refresh needs ais_v0, or in-kernel momenta with rng_proposal0 >= 1%s");
    k.ais_beta = a->ais_beta; k.ais_v0 = a->ais_v0; k.ais_dbeta = a->ais_dbeta; k.ais_refresh = a->ais_refreshment;
    k.ais_w = a->ais_w; k.ais_alpha = a->ais_alpha;
  }
  fill_energy(k, &a->energy);
  hipStream_t s = (hipStream_t)stream;
  // Wide targets (more than 8 dim-tiles, i.e. d > 128; `variant` 8 forces it from 4 tiles up): the
  // register-resident kernels carry 4 or 8 tiles of state per wave there and spill (d = 512: ~1.6k VGPRs);
  // the LDS-resident-state kernel covers the elementwise energies that exist at this width.
  // One chain per lane (traj_lane.hpp): when the chains alone fill the chip -- a wave is 64 of them -- the padding-free
  // VALU form beats the MFMA tiles (variant 32 forces it: tests).  Measured (tools/bench_lane.py): d <= 2 from 65 536
  // chains, d <= 4 from 131 072; wider states lose to the scalar-load latency of their larger nets.
  const int cus = device_cus();                // once per call: every threshold below is a statement about tiles per CU
  {
    const bool has_u_ = a->u != nullptr || (a->rng_flags & L2HMC_RNG_U);
    const bool lane_able = a->packed_nets != nullptr && a->ais_beta == nullptr && k.beta == 1.f && k.temperature == 1.f &&
                           k.n_steps >= 1 &&
                           lane_supported(k.ekind, a->d, a->H, k.ncomp) && (a->d <= 16 || a->x_next != nullptr || !has_u_);
    if (a->variant == 32 && !lane_able)
      return fail(L2HMC_ERR_UNSUPPORTED, "variant 32 (one chain per lane) needs S/T/Q nets and a Gaussian / mixture / Rough-Well target with d <= 4%s");
    const bool lane_auto = ((a->d <= 2 && a->n_chains >= 256 * cus) || (a->d <= 4 && a->n_chains >= 512 * cus));
    if (lane_able && (a->variant == 32 || (a->variant == 0 && lane_auto)))
    {
      note_kernel("traj_lane_kernel");
      return launch_lane(k, s);
    }
  }
  const bool wide_dense = k.ekind == L2HMC_ENERGY_GAUSS_DENSE || k.ekind == L2HMC_ENERGY_GMM;      // (its precision fragments stream from L2: k.prec is the packed buffer)
  const bool wide_kind = k.ekind == L2HMC_ENERGY_GAUSS_DIAG || k.ekind == L2HMC_ENERGY_ROUGHWELL || wide_dense;   // (mixtures: <= 4 tiles per wave)
  const bool wide_able = a->packed_nets != nullptr && wide_kind && !(k.M > 1 && a->x_next == nullptr) && k.NT <= 32;
  if (a->variant == 8 && !(wide_able && k.NT >= 4))
    return fail(L2HMC_ERR_UNSUPPORTED, "variant 8 (LDS-resident state) needs S/T/Q nets, a Gaussian or Rough-Well target, 64 <= d <= 512 and x_next when n_proposals > 1%s");
  if (wide_able && (a->variant == 8 || (a->variant == 0 && k.NT > (wide_dense ? WIDE_DENSE_MIN_NT : 8)))) {
    KArgs kw = k;
    // f16x2 contractions for the elementwise targets without a tempered / annealed energy, unless the f32-input MFMA is asked for
    if (force_f32 || wide_dense || k.beta != 1.f || k.temperature != 1.f) kw.packed16 = nullptr;
    const long long ldsw = plan_lds_wide(kw);
    if (ldsw <= 160 * 1024) {
      note_kernel(kw.packed16 != nullptr ? "traj_wide_kernel<f16x2>" : "traj_wide_kernel");
      return launch_wide(kw, KH, ldsw, s);
    }
    if (a->variant == 8) return fail(L2HMC_ERR_UNSUPPORTED, "variant 8: %s%lld bytes of LDS needed (T x d too large)", "", ldsw);
  }
  int DT, NW;
  const int geom_variant = a->variant >= 100 ? a->variant - 100 : ((a->variant == 16 || a->variant == 33) ? 0 : a->variant);   // 100 + v: the round-1 kernel
  if (!pick_geometry(a->d, a->n_chains, geom_variant, DT, NW, cus))
    return fail(L2HMC_ERR_UNSUPPORTED, "d = %s%lld not supported with variant %lld", "", a->d, a->variant);
  // The instruction-lean kernel (traj_fast.hpp) covers S/T/Q nets on register-resident geometries; the
  // tempered / annealed energies (HMC-mode AIS) and zero-step calls stay on the general kernel.
  const bool fast = a->packed_nets != nullptr && DT <= 2 && k.n_steps >= 1 && a->variant < 100 &&
                    k.beta == 1.f && k.temperature == 1.f;
  // d <= 4 (SCG-2D, MoG-2D): one dimension per lane, S/T/Q in one MFMA row block (traj_small.hpp)
  const bool small = fast && a->d <= 4 && geom_variant == 0 && k.ekind != L2HMC_ENERGY_FUNNEL &&
                     (k.ekind != L2HMC_ENERGY_GMM || k.ncomp <= 8);
  if (small) {
    // (f16x2 for the hidden layer and the head block unless the f32-input MFMA is asked for: every target this kernel serves --
    //  Gaussians incl. dense, mixtures, Rough Well -- has a grad U that is linear or bounded in the state the range guard watches)
    const long long ldss = plan_lds_fast(k, 1, 1, !force_f32);
    // (the per-step schedule records grow with T: past 160 KiB fall through to the fast / general kernel)
    if (ldss <= 160 * 1024) {
      note_kernel(force_f32 ? "traj_small_kernel<%lld, %lld>" : "traj_small_kernel<%lld, %lld, 1>", k.ekind, KH <= 3 ? 3 : 4);
      return dispatch(force_f32 ? OP_TRAJ_SMALL : OP_TRAJ_SMALL16, k, 1, 1, KH, ldss, s);
    }
  }
  // many chains (>= 2 tiles per SIMD), 3-4 dimension slices, elementwise target: one wave per tile (traj_tile.hpp);
  // variant 16 forces it
  const bool tile_kind = k.ekind == L2HMC_ENERGY_GAUSS_DIAG || k.ekind == L2HMC_ENERGY_ROUGHWELL;
  // (a rejected chain of this kernel resumes from the copy of its start point parked in x_next: u without x_next -- accept
  //  decisions with nowhere to put the selected state -- stays on the four-wave kernel)
  const bool tileable = a->packed_nets != nullptr && tile_kind && k.NT >= 3 && k.NT <= 4 && k.n_steps >= 1 &&
                        k.beta == 1.f && k.temperature == 1.f && !(has_u && a->x_next == nullptr) &&
                        !force_f32;        // (its contractions are f16x2 throughout: traj_tile.hpp)
  if (a->variant == 16 && !tileable)
    return fail(L2HMC_ERR_UNSUPPORTED, "variant 16 (one wave per tile) needs S/T/Q nets, a diagonal-Gaussian or Rough-Well target, 33 <= d <= 64 and x_next whenever u is given%s");
  if (tileable && (a->variant == 16 || (a->variant == 0 && a->n_chains >= 64LL * cus))) {
    const long long ldst = plan_lds_tile(k, k.NT);
    if (ldst <= 160 * 1024) {
      // tiles (waves) per workgroup: 4, two workgroups per CU -- unless the staged tables (the split head fragments are 51 KB,
      // the schedule records grow with T) leave room for ONE workgroup only: then 8 tiles share it, from the chain count
      // (128 per CU) at which 8-tile workgroups still cover every CU
      const int tpw = (2 * ldst > 160 * 1024 && a->n_chains >= 128LL * cus) ? 8 : 4;
      note_kernel(a->d - 16 * (k.NT - 1) <= 2 ? "traj_tile_kernel<%lld, %lld, %lld, %lld, true>" : "traj_tile_kernel<%lld, %lld, %lld, %lld, false>",
                  k.ekind, k.NT, KH <= 3 ? 3 : 4, tpw);
      if (k.ekind == L2HMC_ENERGY_GAUSS_DIAG) return launch_tile_ek<L2HMC_ENERGY_GAUSS_DIAG>(k, k.NT, KH, tpw, ldst, s);
      return launch_tile_ek<L2HMC_ENERGY_ROUGHWELL>(k, k.NT, KH, tpw, ldst, s);
    }
    if (a->variant == 16)
      return fail(L2HMC_ERR_UNSUPPORTED, "variant 16: %s%lld bytes of LDS needed (T too large for the one-wave-per-tile kernel)", "", ldst);
  }
  if (fast) {
    // f16x2 (traj_fast.hpp): every contraction of the step loop as two f16 MFMAs on an exact hi / lo split of both operands --
    // fp32-accurate while |states|, |activations|, |grad U| < 65504 (beyond: inf - inf = NaN, which the accept rule treats as a
    // rejection).  The elementwise targets take it unless the caller asks for the f32-input MFMA (variant 200 + v, or
    // L2HMC_F32_MFMA=1 in the environment); the funnel (grad U ~ e^{-x_0}) stays on f32 everywhere, the mixtures and dense
    // Gaussians on the tile kernels (their d <= 4 kernel takes f16x2 for its nets: above).
    const bool f16 = !force_f32 && (k.ekind == L2HMC_ENERGY_GAUSS_DIAG || k.ekind == L2HMC_ENERGY_ROUGHWELL);
    if (f16) {
      const long long lds16 = plan_lds_fast(k, NW, DT, true);
      if (lds16 <= 160 * 1024) {
        note_kernel("traj_fast_kernel<%lld, %lld, %lld, %lld, 1>", k.ekind, DT, NW, KH <= 3 ? 3 : 4);
        if (k.ekind == L2HMC_ENERGY_GAUSS_DIAG) return launch_fast16_ek<L2HMC_ENERGY_GAUSS_DIAG>(k, DT, NW, KH, lds16, s);
        return launch_fast16_ek<L2HMC_ENERGY_ROUGHWELL>(k, DT, NW, KH, lds16, s);
      }
    }
    const long long ldsf = plan_lds_fast(k, NW, DT);
    if (ldsf <= 160 * 1024) {
      note_kernel("traj_fast_kernel<%lld, %lld, %lld, %lld>", k.ekind, DT, NW, KH <= 3 ? 3 : 4);
      return dispatch(OP_TRAJ_FAST, k, DT, NW, KH, ldsf, s);
    }
  }
  const long long lds = plan_lds(k, a->packed_nets != nullptr, true, NW, DT);
  note_kernel("traj_kernel<%lld, %lld, %lld, %lld>", k.ekind, DT, NW, KH <= 3 ? 3 : 4);
  return dispatch(OP_TRAJ, k, DT, NW, KH, lds, s);
}

int64_t l2hmc_workspace_bytes(int64_t n_chains, int32_t d, int32_t H) {
  const int64_t pf = l2hmc_packed_nets_floats(d, H);
  if (pf < 0) return pf;
  if (n_chains < 0) return fail(L2HMC_ERR_ARG, "l2hmc_workspace_bytes: bad argument%s");
  return 4 * (pf + round4(d) + 4 + ((n_chains + 3) & ~3LL));
}

int l2hmc_step(const L2hmcNet* xnet, const L2hmcNet* vnet, const L2hmcEnergy* energy, const float* x, const float* v,
               float* x_out, float* v_out, float* logjac_inout, const float* mask_row, float cos_t, float sin_t,
               float eps, const uint8_t* dir_or_null, int32_t dir_all, int64_t n_chains, int32_t d, int32_t H,
               void* workspace, void* stream) {
  if (!energy || !x || !v || !mask_row || !workspace || n_chains < 0 || d < 1)
    return fail(L2HMC_ERR_ARG, "l2hmc_step: bad argument%s");
  if ((xnet == nullptr) != (vnet == nullptr)) return fail(L2HMC_ERR_ARG, "l2hmc_step: give both nets or neither (HMC mode)%s");
  if (n_chains == 0) return L2HMC_OK;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int64_t pf = xnet ? l2hmc_packed_nets_floats(d, H) : 0;
  if (pf < 0) return (int)pf;
  float* packed = ws;
  float* mrow = ws + pf;
  float* trig = mrow + round4(d);
  float* lj = trig + 4;
  if (xnet) {
    int rc = l2hmc_pack_nets(xnet, vnet, d, H, packed, stream);
    if (rc) return rc;
  }
  hipError_t e = hipMemcpyAsync(mrow, mask_row, sizeof(float) * (size_t)d, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(step_prep_kernel, dim3(1), dim3(64), 0, s, trig, cos_t, sin_t);
  L2hmcTrajectoryArgs a;
  memset(&a, 0, sizeof(a));
  a.packed_nets = xnet ? packed : nullptr;
  a.energy = *energy;
  a.masks = mrow; a.trig = trig; a.alpha = nullptr; a.eps_host = eps;
  a.n_chains = n_chains; a.d = d; a.H = H; a.T = 1; a.step_begin = 0; a.n_steps = 1;
  a.x = x; a.v = v; a.direction = dir_or_null; a.direction_all = dir_all;
  a.x_out = x_out; a.v_out = v_out; a.logjac_out = logjac_inout ? lj : nullptr;
  int rc = l2hmc_trajectory(&a, stream);
  if (rc) return rc;
  if (logjac_inout)
    hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n_chains + 255) / 256)), dim3(256), 0, s, logjac_inout, lj,
                       (long long)n_chains);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
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
  const long long lds = plan_lds(k, false, false, NW, DT);
  return dispatch(OP_ENERGY, k, DT, NW, 3, lds, (hipStream_t)stream);
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
  const long long lds = plan_lds(k, false, false, NW, DT);
  return dispatch(OP_PACCEPT, k, DT, NW, 3, lds, (hipStream_t)stream);
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

int l2hmc_loss_terms(const float* v1, int64_t n, float scale, double inv_n, double* out3, void* stream) {
  if (!v1 || !out3 || n < 0 || !(scale > 0.f)) return fail(L2HMC_ERR_ARG, "l2hmc_loss_terms: bad argument%s");
  hipLaunchKernelGGL(loss_terms_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v1, (long long)n, scale, inv_n, out3);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_adam_step(float* params, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float epsilon, int64_t step, int32_t last_is_log_eps, void* stream) {
  if (!params || !grad || !m || !v || n < 0 || step < 1 || !(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) ||
      !(beta2 >= 0.f && beta2 < 1.f) || !(epsilon > 0.f))
    return fail(L2HMC_ERR_ARG, "l2hmc_adam_step: bad argument%s");
  if (n == 0) return L2HMC_OK;
  // lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t)   (TF1 Adam: epsilon is added to sqrt(v), uncorrected)
  const double t = (double)step;
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad,
                     m, v, (long long)n, lr_t, beta1, beta2, epsilon, last_is_log_eps);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_ais_begin_step(const float* x, const float* U_final, const float* normals, float refreshment,
                         float dbeta, float* w, float* v, int64_t n_chains, int32_t d, void* stream) {
  if (n_chains < 0 || d < 1 || !x || !U_final || !normals || !w || !v || refreshment > 1.f)
    return fail(L2HMC_ERR_ARG, "l2hmc_ais_begin_step: bad argument%s");
  if (n_chains == 0) return L2HMC_OK;
  hipLaunchKernelGGL(ais_begin_kernel, dim3((unsigned)((n_chains + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, U_final, normals, refreshment, dbeta, w, v, (long long)n_chains, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_ais_end_step(const float* Lx, const float* Lv, const float* p, const float* u, float* x, float* v,
                       float* alpha_sum, int64_t n_chains, int32_t d, void* stream) {
  if (n_chains < 0 || d < 1 || !Lx || !Lv || !p || !u || !x || !v)
    return fail(L2HMC_ERR_ARG, "l2hmc_ais_end_step: bad argument%s");
  if (n_chains == 0) return L2HMC_OK;
  hipLaunchKernelGGL(ais_end_kernel, dim3((unsigned)((n_chains + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     Lx, Lv, p, u, x, v, alpha_sum, (long long)n_chains, d);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int l2hmc_rng_fill(uint64_t seed, uint64_t proposal0, int64_t chain_offset, int64_t n_chains,
                   int32_t d, int32_t n_proposals, float* v_out, uint8_t* dir_out, float* u_out,
                   void* stream) {
  if (n_chains < 0 || d < 1 || n_proposals < 1) return fail(L2HMC_ERR_ARG, "l2hmc_rng_fill: bad argument%s");
  const long long total = (long long)n_proposals * n_chains * ((d + 3) / 4);
  if (total == 0) return L2HMC_OK;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)seed, (unsigned long long)proposal0, (long long)chain_offset,
                     (long long)n_chains, d, n_proposals, v_out, dir_out, u_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

int64_t l2hmc_autocov_workspace_doubles(int64_t steps, int64_t n_chains, int32_t d) {
  if (steps < 2 || n_chains < 0 || d < 1) return fail(L2HMC_ERR_ARG, "l2hmc_autocov_workspace_doubles: bad argument%s");
  return ((n_chains * (int64_t)d + 255) / 256) * (steps - 1);
}

int l2hmc_autocov(const float* X, int64_t steps, int64_t n_chains, int32_t d, double scale,
                  int64_t n_total, double* sums_out, double* A_out, double* workspace, void* stream) {
  if (!X || !sums_out || steps < 2 || n_chains < 0 || d < 1 || !(scale > 0.0) || n_total < 1)
    return fail(L2HMC_ERR_ARG, "l2hmc_autocov: bad argument%s");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(sums_out, 0, sizeof(double) * (size_t)(steps - 1), s);
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
  const long long J = (long long)n_chains * d;
  if (J > 0) {
    const long long gx = (J + 255) / 256, gy = (steps - 1 + kLagTile - 1) / kLagTile;
    if (gx > 0x7fffffffLL || gy > 65535) return fail(L2HMC_ERR_UNSUPPORTED, "history too large%s");
    hipLaunchKernelGGL(autocov_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, X,
                       (long long)steps, J, sums_out, workspace);
    if (workspace != nullptr)
      hipLaunchKernelGGL(autocov_reduce_kernel, dim3((unsigned)((steps - 1 + 255) / 256)), dim3(256), 0, s, workspace,
                         gx, (long long)(steps - 1), sums_out);
  }
  if (A_out) {
    const double inv = 1.0 / ((double)n_total * scale * scale);
    hipLaunchKernelGGL(autocov_finish_kernel, dim3((unsigned)((steps - 1 + 255) / 256)), dim3(256), 0, s,
                       sums_out, (long long)steps, inv, A_out);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

}  // extern "C"
