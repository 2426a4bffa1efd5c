/*
 * l2hmc.h -- C ABI of libl2hmc_hip.so: the MI355X (gfx950) L2HMC leapfrog hot path.
 *
 * The reference (brain-research/l2hmc) has no native/FFI layer: its hot path is the
 * Python API of utils/dynamics.py + utils/sampler.py.  This header is the boundary a
 * drop-in replacement binds instead (ctypes from Python -- see INTEGRATION.md): every
 * entry point names the reference function(s) it replaces.  All paths below are relative
 * to the reference tree (/root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in `_host`;
 *     the caller (torch) owns all buffers; the library allocates nothing and keeps no
 *     pointer after a call returns;
 *   - all arrays are float32, row-major, C-contiguous; chains are rows: x is (N, d);
 *   - calls are asynchronous on `stream` (a hipStream_t; NULL = default stream), never
 *     synchronise, and may run concurrently on distinct streams/devices;
 *   - return value: 0 on success, a negative L2HMC_ERR_* code otherwise; the message is
 *     available from l2hmc_last_error() (thread-local).  Nothing throws or aborts.
 */
#ifndef L2HMC_H_
#define L2HMC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2HMC_ABI_VERSION 6

enum {
  L2HMC_OK = 0,
  L2HMC_ERR_ARG = -1,          /* null / inconsistent argument                       */
  L2HMC_ERR_UNSUPPORTED = -2,  /* shape outside what the fused kernels cover          */
  L2HMC_ERR_HIP = -3           /* HIP runtime error (launch, attribute, no device)    */
};

/* Target energies of utils/distributions.py that are fused into the kernels. */
enum {
  L2HMC_ENERGY_GAUSS_DIAG = 1,  /* Gaussian, diagonal precision   distributions.py:41-57,31-32 */
  L2HMC_ENERGY_GAUSS_DENSE = 2, /* Gaussian, dense precision      distributions.py:41-57,31-32 */
  L2HMC_ENERGY_GMM = 3,         /* mixture of Gaussians           distributions.py:104-134     */
  L2HMC_ENERGY_ROUGHWELL = 4,   /* rough well                     distributions.py:84-97       */
  L2HMC_ENERGY_FUNNEL = 5       /* Gaussian funnel                distributions.py:155-180     */
};

/* One S/T/Q network in the reference's own parameter layout: `Linear` keeps W as
 * (in, out) row-major and b as (out,) (utils/layers.py:29-37); `ScaleTanh` keeps a (1, d)
 * log-scale (utils/layers.py:81-86).  Architecture = SCGExperiment.ipynb `network`:
 *   h1 = relu(a W1 + b1 + b W2 + b2 + tau W3 + b3);  h2 = relu(h1 W4 + b4);
 *   S = exp(lam_s) * tanh(h2 Ws + bs);  T = h2 Wt + bt;  Q = exp(lam_q) * tanh(h2 Wq + bq)
 * Shapes: W1, W2 (d,H); W3 (2,H); W4 (H,H); Ws, Wt, Wq (H,d); b1,b2,b3,b4 (H); bs,bt,bq (d);
 * lam_s, lam_q (d). */
typedef struct L2hmcNet {
  const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4;
  const float *Ws, *bs, *Wt, *bt, *Wq, *bq, *lam_s, *lam_q;
} L2hmcNet;

/* Target energy U(x) (dynamics.py:203-212 `energy`, :217-218 `grad_energy`).
 *   GAUSS_DIAG : mu (d), prec (d) = diagonal of the fp32 precision matrix
 *   GAUSS_DENSE: mu (d), prec = buffer written by l2hmc_pack_gaussian (n_comp = 1)
 *   GMM        : mu (n_comp, d), prec = n_comp packed precisions back to back
 *                (stride l2hmc_packed_gaussian_floats(d)), logc (n_comp) = log(pi_i / sqrt((2 pi)^d det Sigma_i))
 *   ROUGHWELL  : eta, easy (distributions.py:84-97: U = |x|^2/2 + eta sum cos(x / eta^2), or x / eta if easy);
 *                den = the divisor of the cosine argument as the reference rounds it: the Python-DOUBLE product
 *                eps * eps (or eps if easy) converted to float32 ONCE (`x / (self.eps * self.eps)`, :93).  A binding
 *                that holds the caller's double passes float(eps * eps); 0 = derive from the float `eta`
 *                (identical whenever eps is exactly a float; last-bit different for e.g. eps = 0.1)
 *   FUNNEL     : eta = sigma (distributions.py:155-180; clip = 4 sigma)
 * temperature divides U and grad U (dynamics.py:204-212); 1.0 when unused.
 * anneal_beta in (0, 1): the AIS bridge of utils/ais.py:46-47 with the standard-normal initial
 * energy its caller uses (eval_vae.py:55-56):  U := (1 - beta) |x|^2 / 2 + beta U(x);  0 (or 1) = off. */
typedef struct L2hmcEnergy {
  int32_t kind;
  int32_t n_comp;
  const float* mu;
  const float* prec;
  const float* logc;
  float eta;
  int32_t easy;
  float temperature;
  float anneal_beta;
  float den;               /* ROUGHWELL only (ABI 5); 0 = derive from eta */
  int32_t reserved_;       /* keeps the struct a multiple of 8 bytes; must be 0 */
} L2hmcEnergy;

/* Arguments of l2hmc_trajectory (passed by pointer; a HOST struct of device pointers). */
typedef struct L2hmcTrajectoryArgs {
  /* ---- model ------------------------------------------------------------------------- */
  const float* packed_nets; /* from l2hmc_pack_nets; NULL = HMC mode (nets == 0, dynamics.py:73-76) */
  L2hmcEnergy energy;
  const float* masks;       /* (T, d) 0/1 rows, `Dynamics.mask` (dynamics.py:84-97)          */
  const float* trig;        /* (T, 2) [cos(2 pi t/T), sin(2 pi t/T)] (dynamics.py:99-105)    */
  const float* alpha;       /* device scalar log(eps) (dynamics.py:50-58) or NULL ...        */
  float eps_host;           /* ... then this step size is used                               */
  /* ---- state -------------------------------------------------------------------------- */
  int64_t n_chains;
  int32_t d, H, T;
  int32_t step_begin;       /* leapfrog iterations [step_begin, step_begin + n_steps) of the  */
  int32_t n_steps;          /* T-step schedule; a full trajectory is (0, T)                   */
  const float* x;           /* (N, d) start positions                                         */
  const float* v;           /* (N, d) start momenta (the injected N(0,I) draw)                */
  const uint8_t* direction; /* (N) 1 = forward, 0 = backward (sampler.py:34) or NULL ...      */
  int32_t direction_all;    /* ... then every chain runs forward (1) or backward (0)          */
  const float* u;           /* (N) accept uniforms (sampler.py:54) or NULL (no MH step)       */
  /* ---- outputs (any may be NULL) ------------------------------------------------------- */
  float* x_out;             /* (N, d) proposal Lx                                             */
  float* v_out;             /* (N, d) proposal Lv                                             */
  float* logjac_out;        /* (N) summed log|det J| of the executed steps                    */
  float* p_out;             /* (N) accept prob, dynamics.py:302-309                           */
  float* x_next;            /* (N, d) MH-selected state, sampler.py:53-55 (needs u and p)     */
  /* ---- tuning ---------------------------------------------------------------------------- */
  int32_t variant;          /* kernel choice.  0 = automatic (measured rules): d <= 4 the
                             *   one-dimension-per-lane kernel; d <= 64 the instruction-lean tile kernel on 1 or 4
                             *   waves per 16-chain tile, from 16 384 chains with 33 <= d <= 64 and an elementwise
                             *   target one wave per tile; d > 128 the LDS-resident-state kernel (Gaussians incl. dense, mixtures, Rough Well); d <= 2 from 65 536
                             *   chains (d <= 4 from 131 072) ONE CHAIN PER LANE: the nets on packed VALU FMAs with
                             *   wave-uniform weights, no MFMA padding.  32 = force one chain per lane (d <= 4);
                             *   33 = the automatic choice among the MFMA kernels only.  1 / 4 = force
                             *   that many waves per tile; 16 = one wave per tile, 4 or 8 tiles per workgroup (every
                             *   contraction as f16x2; it parks the rejected chains' start point in x_next, so x_next must come
                             *   with u -- the automatic choice falls back to the 4-wave tile otherwise); 8 = the
                             *   LDS-resident-state kernel; 100 + v = geometry v on the general kernel (which also
                             *   serves HMC mode, AIS mode and tempered energies); 200 + v = choice v with the
                             *   f32-input MFMA forced: the tile kernels of the elementwise targets (diagonal Gaussian,
                             *   Rough Well: four-wave, one-wave and LDS-resident-state forms) and the d <= 4 kernel of
                             *   every target it serves otherwise run their contractions as f16x2 -- two f16 MFMAs on an
                             *   two-term split of both operands (22 significand bits worst case: fp32-level, measured equal
                             *   to the f32-input MFMA against float64) for |state|, |grad U| < 2.5e5 (activations < 4.2e6)
                             *   and |weight| < 1023; a proposal whose end points leave that range comes back as NaN
                             *   with accept probability 0 (csrc/traj_fast.hpp; L2HMC_F32_MFMA=1 in the environment =
                             *   200 + v for every call)                                                            */
  /* ---- persistent sampler loop (the notebook's per-MH-step sess.run loop, nb raw 288-298) -- */
  int32_t n_proposals;      /* M >= 1 proposals per launch (0 = 1).  With M > 1: v is (M,N,d),  */
                            /* direction (M,N), u (M,N) [required], p_out / logjac_out (M,N);   */
                            /* each proposal starts from the previous MH-selected state;        */
                            /* x_out / v_out hold the LAST proposal, x_next the final state     */
  float* x_hist;            /* (M, N, d) MH-selected state after every proposal, or NULL        */
  /* ---- in-kernel randomness (replaces tf.random_normal / random_uniform of dynamics.py:247-250, */
  /*      sampler.py:34,54): counter-based Philox4x32-10, keyed by (seed), counter = (global chain, */
  /*      dim/4, proposal index, stream) => independent of kernel geometry and of chain sharding.   */
  uint32_t rng_flags;       /* L2HMC_RNG_V | L2HMC_RNG_DIR | L2HMC_RNG_U: draw that input in-kernel  */
                            /* (the corresponding pointer v / direction / u is then ignored)          */
  uint64_t rng_seed;
  uint64_t rng_proposal0;   /* stream index of this launch's first proposal                          */
  int64_t chain_offset;     /* global index of row 0 (rank r owning rows [lo, hi) passes lo)          */
  /* ---- AIS mode of the persistent loop (utils/ais.py:43-66; HMC mode only: packed_nets = NULL) ------------- */
  /* ais_beta != NULL: proposal m is anneal step m.  Before its transition the log-weight takes                 */
  /*   w += ais_dbeta (|x|^2 / 2 - U(x))  (ais.py:58-59), the momentum is the step's draw z (ais_refreshment    */
  /*   < 0, ais.py:57) or  v sqrt(1 - r) + z sqrt(r)  (ais.py:55), the transition uses the bridge energy        */
  /*   (1 - beta) |x|^2 / 2 + beta U  with beta = ais_beta[m] (energy.anneal_beta is ignored), and after the MH */
  /*   step a rejected chain keeps x with the NEGATED proposed momentum (ais.py:62-65).  One launch replaces     */
  /*   the five per-step launches of l2hmc_ais_begin_step / l2hmc_energy / l2hmc_trajectory / l2hmc_ais_end_step.*/
  const float* ais_beta;    /* (M) bridge schedule, float32(linspace(0, 1, K + 1)[1:])                          */
  const float* ais_v0;      /* (N, d) momentum before step 0 (read only when refreshing), or NULL: the Philox    */
                            /*   draw of proposal index rng_proposal0 - 1                                        */
  float ais_dbeta;          /* beta[1] - beta[0]                                                                 */
  float ais_refreshment;    /* r in [0, 1], or < 0 for fresh momenta                                             */
  float* ais_w;             /* (N) log-weights, accumulated (+=)                                                 */
  float* ais_alpha;         /* (N) summed accept probabilities, accumulated (+=), or NULL                        */
} L2hmcTrajectoryArgs;

enum { L2HMC_RNG_V = 1, L2HMC_RNG_DIR = 2, L2HMC_RNG_U = 4 };

int l2hmc_abi_version(void);
const char* l2hmc_last_error(void);
/* Name of the kernel the last l2hmc_trajectory / l2hmc_train_propose_grad / l2hmc_train_step call of THIS thread chose
 * (e.g. "traj_fast_kernel<1, 1, 4, 3>", spelt as rocprofv3 prints the instantiation; "" before the first call): what a
 * profile of the call has to be matched against.  l2hmc_trajectory_split reports the template of its dominant kernel
 * without arguments: "gemm_xlp_kernel" (decoder products on pre-split bf16 planes, from 3072 chains at config 5's widths
 * in gemm_mode 1), "gemm_nt_kernel", or "net_eval_kernel" for a built-in / caller-supplied energy.  Copies at most n - 1 characters, returns the full length. */
int32_t l2hmc_last_kernel(char* buf, int32_t n);

/* Number of floats of the fragment-ordered weight buffer for both nets, or a negative
 * L2HMC_ERR_* if (d, H) is outside the fused kernels' range. */
int64_t l2hmc_packed_nets_floats(int32_t d, int32_t H);

/* Re-orders XNet and VNet (utils/layers.py parameter layout, built by the `net_factory`
 * of dynamics.py:78-79) into MFMA A-operand fragment order.  Run once per weight update. */
int l2hmc_pack_nets(const L2hmcNet* xnet, const L2hmcNet* vnet, int32_t d, int32_t H,
                    float* packed, void* stream);

int64_t l2hmc_packed_gaussian_floats(int32_t d);
/* i_sigma: (d, d) fp32 precision matrix `Gaussian.i_sigma.astype('float32')`
 * (distributions.py:48,52); packs (S + S^T)/2 in fragment order. */
int l2hmc_pack_gaussian(const float* i_sigma, int32_t d, float* packed, void* stream);

/* The fused generalised-leapfrog trajectory:
 *   Dynamics.forward / .backward           dynamics.py:246-300  (n_steps = T)
 *   Dynamics._forward_step/_backward_step  dynamics.py:115-201  (n_steps = 1; a backward
 *       step at schedule index s is step_begin = T-1-s with direction 0)
 *   Dynamics.p_accept                      dynamics.py:302-309  (p_out)
 *   propose + tf_accept                    sampler.py:28-55     (direction, u, x_next)
 * Each chain runs only in its drawn direction (the reference runs both and discards one). */
int l2hmc_trajectory(const L2hmcTrajectoryArgs* args, void* stream);

/* ONE generalised leapfrog step straight from the reference-layout weights -- the per-step entry point
 * SURVEY.md 8(b) names: Dynamics._forward_step (dynamics.py:115-157; dir = 1) / ._backward_step (:159-201;
 * dir = 0) at the schedule row whose mask (dynamics.py:95-97) and time encoding (:99-105) are passed in.
 * xnet = vnet = NULL is HMC mode (dynamics.py:73-76).  logjac_inout (N) is ACCUMULATED (+=) or NULL.
 * `workspace`: l2hmc_workspace_bytes(n_chains, d, H) bytes of device memory (packed fragments, the schedule
 * row, a log-det temporary); the library allocates nothing.  Callers that take many steps should pack once
 * (l2hmc_pack_nets) and use l2hmc_trajectory, which fuses the whole trajectory. */
int64_t l2hmc_workspace_bytes(int64_t n_chains, int32_t d, int32_t H);
int l2hmc_step(const L2hmcNet* xnet, const L2hmcNet* vnet, const L2hmcEnergy* energy, const float* x, const float* v,
               float* x_out, float* v_out, float* logjac_inout, const float* mask_row, float cos_t, float sin_t,
               float eps, const uint8_t* dir_or_null, int32_t dir_all, int64_t n_chains, int32_t d, int32_t H,
               void* workspace, void* stream);

/* Dynamics.energy / Dynamics.grad_energy (dynamics.py:203-218).  U_out (N) and/or
 * grad_out (N, d) may be NULL. */
int l2hmc_energy(const L2hmcEnergy* energy, const float* x, int64_t n_chains, int32_t d,
                 float* U_out, float* grad_out, void* stream);

/* Dynamics.p_accept (dynamics.py:302-309) on arbitrary end points. */
int l2hmc_p_accept(const L2hmcEnergy* energy, const float* x0, const float* v0,
                   const float* x1, const float* v1, const float* logjac,
                   int64_t n_chains, int32_t d, float* p_out, void* stream);

/* tf_accept (sampler.py:53-55): x_next[n,:] = (px[n] - u[n] >= 0) ? Lx[n,:] : x[n,:]. */
int l2hmc_mh_select(const float* x, const float* Lx, const float* px, const float* u,
                    int64_t n_chains, int32_t d, float* x_next, void* stream);

/* The notebook loss from the per-chain arguments v1 the training entry points leave behind (SCGExperiment.ipynb raw 156-169:
 * loss = scale * mean(1 / v1) - mean(v1) / scale over x- and z-proposals): out3 = { sum 1 / v1, sum v1,
 * inv_n * (scale * out3[0] - out3[1] / scale) } in double, one fixed-order reduction (bitwise reproducible) -- ONE launch
 * instead of a dozen elementwise framework kernels per optimiser step.  Sharded runs all-reduce out3[0..1] themselves. */
int l2hmc_loss_terms(const float* v1, int64_t n, float scale, double inv_n, double* out3, void* stream);

/* ---- split engine: wide / image-conditioned nets + VAE latent-posterior energy (config 5) ------ */
/* Linear-softplus-Linear-softplus-Linear with reference-layout weights W (in, out), b (out):
 * the VAE decoder (mnist_vae.py:104-111) and the sampler's image branch `encoder_sampler`
 * (mnist_vae.py:134-140). */
typedef struct L2hmcMlp3 {
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  int32_t n_in, n_h1, n_h2, n_out;
} L2hmcMlp3;

/* l2hmc_trajectory_split: same contract as l2hmc_trajectory (steps [step_begin, +n_steps) of the
 * T-step schedule, per-chain direction, accept probability, MH select) for
 *   energy  U(z; aux) = sum_pix BCE_with_logits(aux, decoder(z)) + |z|^2 / 2   (mnist_vae.py:122-126)
 *   nets    the S/T/Q architecture with ANY hidden width H and, if aux_encoder != NULL, the 4th Zip
 *           branch aux_encoder(aux) added into the first hidden layer (mnist_vae.py:142-167);
 *           RAW reference-layout weights (not the packed buffer).
 * The dense products run in this library's own fp32 MFMA GEMM (csrc/gemm_f32.hpp) with the bias / softplus /
 * sigmoid / relu / BCE-gradient / chain-rule work fused into its epilogues; no BLAS library is used. */
/* A caller-supplied target energy: the `energy_function` protocol of utils/dynamics.py:203-218 (`self._fn(x[, aux])`
 * and tf.gradients of it) for energies OUTSIDE the fused set -- e.g. a closure like mnist_vae.py:122-126 or any density
 * the caller can differentiate.  The library calls it on the HOST, between kernel launches, whenever the trajectory
 * needs grad U (once per leapfrog step + once at the start) and U (at the two end points):
 *   x        (n_chains, d) current positions, row stride ldx floats (device memory inside the workspace)
 *   U_out    NULL, or (n_chains) DOUBLES to fill with U(x) (|U| can be ~1e3: the accept probability takes a
 *            difference of two of them)
 *   grad_out (n_chains, d) floats, row stride ldg, to fill with grad U(x)
 * All work must be enqueued on `stream` (no synchronisation needed); return 0, or non-zero to abort the trajectory
 * (l2hmc_trajectory_split then returns L2HMC_ERR_ARG).  Tempering / annealing of a user energy is the callback's own
 * business.  This is the SLOW path by construction (a host round trip per gradient); the leapfrog half-updates and the
 * S/T/Q nets stay on the library's kernels. */
typedef int (*L2hmcEnergyCallback)(void* user, const float* x, int64_t ldx, int64_t n_chains, int32_t d,
                                   double* U_out, float* grad_out, int64_t ldg, void* stream);

/* Hessian-vector product of a caller-supplied energy, for TRAINING on it (l2hmc_train_split_grad): the loss
 * back-propagates through grad U (utils/dynamics.py:218 inside the differentiated graph), so the reverse sweep needs
 *   hv_out (n_chains, d; row stride ldhv) = (d^2 U / dx^2)(x) u      for x (row stride ldx) and u (row stride ldu)
 * once per leapfrog step -- e.g. torch.autograd.grad of sum(grad U * u) in a binding.  Same rules as
 * L2hmcEnergyCallback (enqueue on `stream`, return 0 or non-zero to abort). */
typedef int (*L2hmcHvpCallback)(void* user, const float* x, int64_t ldx, const float* u, int64_t ldu, int64_t n_chains,
                                int32_t d, float* hv_out, int64_t ldhv, void* stream);

/* A caller-supplied S/T/Q net (the reference's `net_factory` may return ANY callable [a, b, tau, aux] -> [S, T, Q],
 * utils/dynamics.py:69-79; only the notebook's architecture is fused into kernels).  Called on the host between launches, like
 * the energy callback: `ab` is the (n_chains, 2 d) block [a | b] of first-layer inputs on the device (row stride ldab floats):
 * a = ab[:, :d], b = ab[:, d:]; net = 0 (XNet: a = v_h, b = kept * x) or 1 (VNet: a = x, b = grad U); chain n sits at schedule
 * row `it` if it runs forward (direction[n] != 0, or direction == NULL and direction_all != 0), else T - 1 - it: its time input
 * is tau = (cos, sin)(2 pi row / T) (dynamics.py:99-105).  The callback enqueues, on `stream`, the FINAL S, T, Q (dynamics.py's
 * `scale, translation, transformed`) as the three (n_chains, d) column blocks of stq_out (row stride 3 d).  Nonzero return
 * aborts the trajectory.  Training such nets: L2hmcNetVjpCallback below (ABI 6). */
typedef int (*L2hmcNetCallback)(void* user, int32_t net, const float* ab, int64_t ldab, int64_t n_chains, int32_t d, int32_t it,
                                const uint8_t* direction, int32_t direction_all, float* stq_out, void* stream);
/* (ABI 6) The reverse of ONE evaluation of a caller-supplied net, for l2hmc_train_split_grad: the reference minimises its loss
 * over whatever variables `net_factory` created (utils/dynamics.py:78-79; SCGExperiment.ipynb raw 178-181 `minimize(loss)`;
 * mnist_vae.py:254-262), so the adjoint of an opaque net is the caller's to form.  `ab`, `it`, `direction` identify the
 * evaluation exactly as in L2hmcNetCallback (the library hands back the SAME inputs it kept from the forward pass);
 * d_stq (n_chains, 3 d, row stride 3 d) holds the cotangents of the net's outputs (d S | d T | d Q) -- already weighted by
 * inv_n and the loss.  The callback enqueues on `stream`: (i) the cotangents of the inputs, (d a | d b), into d_ab
 * (n_chains, 2 d, row stride ld_dab); (ii) the accumulation of its OWN parameters' gradients wherever it keeps them (with
 * torch: one `torch.autograd.backward` of the re-evaluated net) -- the library never sees those parameters.  Nonzero return
 * aborts the call. */
typedef int (*L2hmcNetVjpCallback)(void* user, int32_t net, const float* ab, int64_t ldab, int64_t n_chains, int32_t d, int32_t it,
                                   const uint8_t* direction, int32_t direction_all, const float* d_stq, float* d_ab,
                                   int64_t ld_dab, void* stream);

typedef struct L2hmcSplitArgs {
  const L2hmcNet* xnet;
  const L2hmcNet* vnet;
  int32_t H;
  const L2hmcMlp3* aux_encoder;  /* (n_pix -> H) or NULL                                    */
  const L2hmcMlp3* decoder;      /* (d -> n_pix)                                            */
  const float* aux;              /* (N, n_pix) conditioning images                          */
  const float* masks;            /* (T, d) */
  const float* trig;             /* (T, 2) */
  const float* alpha;
  float eps_host;
  int64_t n_chains;
  int32_t d, T, step_begin, n_steps;
  const float* x;
  const float* v;
  const uint8_t* direction;
  int32_t direction_all;
  const float* u;
  float *x_out, *v_out, *logjac_out, *p_out, *x_next;
  float* workspace;              /* l2hmc_split_workspace_floats(...) floats                */
  int64_t workspace_floats;
  int32_t hmc;                   /* 1: nets identically zero (dynamics.py:73-76) = plain leapfrog, forward
                                  *    only; xnet / vnet / masks / trig / aux_encoder are ignored         */
  float bce_scale;               /* AIS bridge from N(0, I) (ais.py:46-47, eval_vae.py:55-62): the BCE term
                                  *    of the energy is scaled by beta in (0, 1); 0 (or 1) = off          */
  const L2hmcEnergy* energy;     /* NULL: the decoder posterior above.  Else one of the built-in targets of
                                  *    utils/distributions.py (decoder / aux / aux_encoder = NULL): the S/T/Q
                                  *    nets of ANY width H on the GEMM engine, grad U from l2hmc_energy's kernels
                                  *    (SCGExperiment.ipynb `network` with H != 10)                       */
  int32_t reuse;                 /* what the SAME workspace still holds from the previous call with the same shapes
                                  *    (the library keeps no state of its own; the caller vouches):
                                  *    1: the prepared weights (transposed copies, time table) -- no weight changed;
                                  *    2: the image branch aux_encoder(aux) -- aux and aux_encoder unchanged.
                                  *    A sampling loop passes 3 from its second proposal on (mnist_vae.py:185-224:
                                  *    weights and images are fixed while the chain runs)                  */
  L2hmcEnergyCallback energy_cb; /* non-NULL: the caller's energy (see L2hmcEnergyCallback); decoder = energy = NULL.
                                  *    aux_encoder + aux may still be given (image-conditioned nets,
                                  *    mnist_vae.py:134-150: aux is (N, aux_encoder->n_in)); HMC mode is allowed  */
  void* energy_cb_user;          /* passed back as the callback's first argument                              */
  int32_t gemm_mode;             /* arithmetic of the decoder-sized dense products (M x 1024 x 1024 ...):
                                  *    0: f32-input MFMA (v_mfma_f32_16x16x4_f32), bit-exact fp32 FMA chains;
                                  *    1: "bf16x3" -- every fp32 operand split EXACTLY into three bf16 terms, the six
                                  *       significant cross products on the bf16 MFMA (16x the f32 MFMA rate), fp32
                                  *       accumulation: dropped terms <= 3 x 2^-24 |x y| per product, i.e. fp32-level
                                  *       accuracy (measured against float64 in profiles/ and the config-5 parity tests);
                                  *       from 3072 chains at config 5's widths on operands PRE-SPLIT into bf16 planes
                                  *    2: the same six products with the split inside the k loop at every size -- bit-identical
                                  *       results to mode 1 (the planes only move where the split happens); kept for tests and A/B
                                  *    3: "f16x2 planes" (round 6; what l2hmc_amd.Dynamics asks for): x = X1 + X2 / 64 with X1 =
                                  *       f16(x), X2 = f16(64 (x - X1)), stored as the three f16 planes X1 | X1 / 64 | X2, so that
                                  *       x y = X1 Y1 + (X1 / 64) Y2 + X2 (Y1 / 64) is THREE f16 MFMAs on one accumulator: half of
                                  *       mode 1's matrix-pipe work at fp32-level accuracy (22 significand bits per operand in the worst case)
                                  *       for operand entries in [4e-3, 65504) -- smaller entries keep an absolute error of
                                  *       2^-30, larger ones overflow to inf.  The sampler's activations, logits and BCE
                                  *       gradients live there; the TRAINER's tangent / adjoint planes (entries scaled by 1 / chains)
                                  *       do not: under L2hmcTrainSplitArgs.gemm_mode 3 its forward evaluations run on f16x2
                                  *       planes, its reverse sweep as mode 1 (csrc/gemm_f32.hpp, gemm_xl.hpp, train_split.hpp)   */
  L2hmcNetCallback net_cb;       /* (ABI 5) non-NULL: the caller's nets (see L2hmcNetCallback); xnet = vnet = aux_encoder = NULL,
                                  *    H is ignored, hmc = 0.  With any target: energy_cb, a built-in energy, or (round 6) the
                                  *    decoder posterior -- the callback's nets then read the images on their own           */
  void* net_cb_user;             /* passed back as the callback's first argument                                         */
} L2hmcSplitArgs;

int64_t l2hmc_split_workspace_floats(int64_t n_chains, int32_t d, int32_t H, int32_t T,
                                     const L2hmcMlp3* aux_encoder, const L2hmcMlp3* decoder);
int l2hmc_trajectory_split(const L2hmcSplitArgs* args, void* stream);
/* Dynamics.energy / grad_energy for the VAE posterior; workspace as for the split trajectory;
 * bce_scale as in L2hmcSplitArgs (0 = off). */
int l2hmc_vae_energy(const L2hmcMlp3* decoder, const float* aux, const float* x, int64_t n_chains,
                     int32_t d, float* U_out, float* grad_out, float* workspace, float bce_scale, void* stream);

/* The operand form of gemm_mode 1 at decoder sizes, exposed for inspection and tests: the exact three-way bf16 split
 * x = h + m + l (round-to-nearest-even at each level) of the fp32 matrix W (rows x K, row stride ld) as three planes of
 * rows_pad x ld_planes bf16 bit patterns (uint16), plane p at planes + p * rows_pad * ld_planes, zero beyond the matrix
 * (rows_pad >= rows; ld_planes >= K, a multiple of 4).  This is what l2hmc_trajectory_split converts the decoder weights
 * to once per call (csrc/gemm_xl.hpp); oracle/bf16x3_oracle.py restates it in numpy. */
int l2hmc_bf16_planes(const float* W, int32_t ld, int64_t rows, int32_t K, uint16_t* planes, int64_t rows_pad,
                      int32_t ld_planes, void* stream);

/* Dynamics.p_accept (dynamics.py:302-309) for energies evaluated separately (l2hmc_vae_energy):
 * p = exp(min(U0 + |v0|^2/2 - U1 - |v1|^2/2 + log_jac, 0)), non-finite -> 0. */
int l2hmc_p_accept_energies(const float* U0, const float* v0, const float* U1, const float* v1, const float* log_jac,
                            int64_t n_chains, int32_t d, float* p_out, void* stream);

/* ---- training (next-row f1): one proposal + the gradient of its loss term ------------------- */
/* Loss of SCGExperiment.ipynb raw lines 156-169 for ONE of its two proposals:
 *   v1_n = |x_n - Lx_n|^2 p_n + 1e-4;   term = scale * mean_n(1 / v1_n) - mean_n(v1_n) / scale
 * (the notebook adds the term of `propose(x)` and of `propose(z), z ~ N(0, I)`; call twice).
 * `grad` (l2hmc_train_grad_floats(d, H) floats) is ACCUMULATED (+=) without atomics -- every workgroup
 * writes its partial gradient to the workspace and a second tiny kernel adds them in block order, so the
 * result is bitwise reproducible --: the flat layout is
 * [XNet | VNet | d/d eps], each net in the field order of L2hmcNet (W1, b1, ..., lam_q);
 * d/d alpha = eps * d/d eps (dynamics.py:50-58).  inv_n = 1 / (chains over ALL ranks) so that
 * per-rank gradients simply all-reduce(sum).  The nets are the RAW reference-layout weights
 * (not the packed buffer); for the dense Gaussian `energy.prec` is the raw (d, d) precision.
 * Targets with analytic Hessian-vector products: Gaussian (diag / dense), GMM (prec = RAW (k,d,d)
 * precisions, logc, n_comp <= 8), Rough Well; any d, H whose 16-chain tile fits the 160 KiB LDS
 * (d = 50, H = 10 uses 122 KiB; larger shapes return L2HMC_ERR_UNSUPPORTED).  Every chain runs in
 * its own direction. */
typedef struct L2hmcTrainArgs {
  const L2hmcNet* xnet;
  const L2hmcNet* vnet;
  L2hmcEnergy energy;
  const float* masks;       /* (T, d)   */
  const float* trig;        /* (T, 2)   */
  const float* alpha;       /* device log(eps) or NULL -> eps_host */
  float eps_host;
  int64_t n_chains;
  int32_t d, H, T;
  const float* x;           /* (N, d) start points                                   */
  const float* v;           /* (N, d) momenta of each chain's own direction          */
  const uint8_t* direction; /* (N) or NULL -> direction_all                          */
  int32_t direction_all;
  float scale;              /* 0.1 in the notebook                                   */
  float inv_n;
  float* Lx;                /* (N, d) proposal                                       */
  float* p;                 /* (N) accept probability                                */
  float* v1;                /* (N) per-chain loss argument                           */
  float* grad;              /* flat gradient, accumulated                            */
  float* workspace;         /* l2hmc_train_workspace_floats(N, d, H, T) floats       */
  int32_t variant;          /* 0 = auto: d <= 4 (Gaussians, GMM, Rough Well) the one-dimension-per-lane kernel; else the
                             *   register-resident kernel (elementwise targets d <= 64, dense Gaussians d <= 16,
                             *   H <= 15); else the general LDS-matrix tile kernel.  1 = as 0 without the d <= 4
                             *   form; 100 = always the general kernel                                      */
} L2hmcTrainArgs;

int64_t l2hmc_train_workspace_floats(int64_t n_chains, int32_t d, int32_t H, int32_t T);
int64_t l2hmc_train_grad_floats(int32_t d, int32_t H);
/* LDS bytes of the fused training kernel l2hmc_train_propose_grad would launch for this shape (> 0), or
 * L2HMC_ERR_UNSUPPORTED when no fused kernel holds it (d / H beyond the 16-chain tile's 160 KiB plan, the funnel beyond
 * d = 16): the host then trains on the GEMM engine (l2hmc_train_split_grad below), which takes any d and H. */
int64_t l2hmc_train_fused_lds_bytes(int32_t energy_kind, int32_t n_comp, int32_t d, int32_t H, int32_t T);
int l2hmc_train_propose_grad(const L2hmcTrainArgs* args, void* stream);

/* One optimiser step's device work in as few launches as the data flow allows (ABI 4; SCGExperiment.ipynb raw 156-181,
 * 254-271: propose(x) and propose(z), the loss of both, Adam, the MH-selected continuation of the x chains):
 *   launch 1  l2hmc_train_propose_grad's kernel over `args` -- with the start points of chains [0, n_head) read from
 *             `x_head` (no staging copy of the caller's state next to z);
 *   launch 2  the fixed-order slot reduction, which OVERWRITES args->grad (no zero fill), with
 *             - the Metropolis select of chains [0, n_head) (sampler.py:53-55) in extra workgroups of the same launch
 *               when u / x_next are given;
 *             - the loss terms of args->v1 (as l2hmc_loss_terms) -> `loss` and, split into float (hi, lo) pairs,
 *               -> `terms` = {sum 1/v1, sum v1, n_head}: the tail of the ONE buffer a sharded step all-reduces;
 *             - Adam on theta / m / v (as l2hmc_adam_step) in the same launch when `theta` is given (a single-process
 *               step; a sharded one all-reduces first and calls l2hmc_adam_step_terms).
 * `terms` may point right behind the gradient (args->grad + l2hmc_train_grad_floats). */
typedef struct L2hmcTrainStep {
  const float* x_head;      /* (n_head, d) or NULL: start points of chains [0, n_head); the others stay in args->x  */
  int64_t n_head;
  const float* u;           /* (n_head) uniforms and ...                                                           */
  float* x_next;            /* ... (n_head, d) selected states; both or neither                                     */
  float* terms;             /* 6 floats or NULL                                                                     */
  double* loss;             /* 3 doubles {sum 1/v1, sum v1, inv_n (scale sum 1/v1 - sum v1 / scale)} or NULL; inv_n =
                             * 1 / k in double when args.inv_n is the float rounding of 1 / k (= l2hmc_loss_terms)   */
  float* theta;             /* flat parameters [XNet | VNet | alpha] or NULL (no optimiser in this call)            */
  float* m;
  float* v;
  float lr, beta1, beta2, epsilon;
  int64_t step;             /* Adam's t (>= 1)                                                                      */
  int32_t train_alpha;      /* 1: the last gradient entry is d/d eps, the parameter is log eps                      */
} L2hmcTrainStep;
int l2hmc_train_step(const L2hmcTrainArgs* args, const L2hmcTrainStep* step, void* stream);
/* Adam after a sharded step's all-reduce: grad = the reduced gradient, terms6 = the reduced tail written by
 * l2hmc_train_step; the loss of the GLOBAL batch, (scale sum 1/v1 - sum v1 / scale) / count, lands in loss_out[2]
 * ([0], [1]: the two sums). */
int l2hmc_adam_step_terms(float* params, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                          float beta2, float epsilon, int64_t step, int32_t last_is_log_eps, const float* terms6,
                          float scale, double* loss_out, void* stream);

/* ---- training on the GEMM engine (next-row f1 for config 5 and for wide nets) ----------------------------------
 * l2hmc_train_propose_grad's contract -- ONE proposal per chain in the chain's own direction, its accept
 * probability, the loss argument and the gradient of the loss term (accumulated into `grad`) -- for the samplers the
 * register-resident kernels cannot hold: S/T/Q nets of any width H, the shared image branch aux_encoder(aux) in their
 * first hidden layer, and the VAE latent-posterior energy (mnist_vae.py:104-226, the "trained sampler" of BASELINE.json
 * config 5); or, with `energy` set, any built-in target of utils/distributions.py with wide nets.
 *   v1_n = sum_k w_nk (Lx_nk - x_nk)^2 p_n + 1e-4;      term = scale * mean_n(1 / v1_n) - mean_n(v1_n) / scale
 *   w = dist_weight (N, d) or NULL (= 1).  SCGExperiment.ipynb raw 164-169: w = 1, scale = 0.1;
 *   mnist_vae.py:207-214 (energy_scale = 0): w = 1 / (exp(2 log_sigma) + 1e-4), scale = 1.
 * grad layout: [XNet | VNet | d/d eps | aux_encoder (W1, b1, W2, b2, W3, b3)], each net in NET_FIELDS order;
 * l2hmc_train_split_grad_floats(d, H, aux_encoder) floats (l2hmc_adam_step applies it, last_is_log_eps = 0 when the
 * image branch follows -- multiply the eps entry by eps first).  The decoder is NOT trained here (the reference trains
 * it by a separate optimiser on the ELBO, mnist_vae.py:229-262).
 * dLx_in (N, d) or NULL: a cotangent added to d loss / d Lx -- what a LATER proposal of the same step sends back when
 * proposals are chained without stop_gradient (mnist_vae.py:185-224); dx0_out (N, d) or NULL receives d loss / d x.
 * inv_n = 0 with dLx_in: the proposal has no loss term of its own (an earlier link of such a chain).
 * Every sum over chains is chunked and added in chunk order: the gradient is bitwise reproducible. */
typedef struct L2hmcTrainSplitArgs {
  const L2hmcNet* xnet;
  const L2hmcNet* vnet;
  int32_t H;
  const L2hmcMlp3* aux_encoder;  /* (n_pix -> H) or NULL                                                   */
  const L2hmcMlp3* decoder;      /* (d -> n_pix); NULL with `energy`                                       */
  const float* aux;              /* (N, n_pix)                                                             */
  const L2hmcEnergy* energy;     /* NULL: the decoder posterior.  Else a built-in target (as l2hmc_energy takes it)  */
  const float* hess;             /* GAUSS_DENSE / GMM: the RAW (n_comp, d, d) precisions (Hessian-vector products)   */
  const float* masks;            /* (T, d) */
  const float* trig;             /* (T, 2) */
  const float* alpha;            /* device log(eps) or NULL -> eps_host */
  float eps_host;
  int64_t n_chains;
  int32_t d, T;
  const float* x;                /* (N, d) start points                          */
  const float* v;                /* (N, d) momenta of each chain's own direction */
  const uint8_t* direction;      /* (N) or NULL -> direction_all                 */
  int32_t direction_all;
  const float* dist_weight;      /* (N, d) or NULL */
  float scale, inv_n;
  const float* dLx_in;           /* (N, d) or NULL */
  float *Lx, *p, *v1;            /* (N, d), (N), (N) */
  float* dx0_out;                /* (N, d) or NULL */
  float* grad;                   /* accumulated */
  float* workspace;              /* l2hmc_train_split_workspace_floats(...) floats */
  int64_t workspace_floats;
  /* ---- the rest of mnist_vae.py:185-226's sampler objective ---------------------------------------------------- */
  float energy_scale;            /* es >= 0 (0 = off): + es inv_n sum_n (1 / ed_n - ed_n) with
                                  *    ed = (U(Lx) - U(x))^2 p + 1e-4   (mnist_vae.py:214,218,224)               */
  float* ediff_out;              /* (N) ed_n, or NULL                                                          */
  /* links of chain_operator (sampler.py:57-85; mnist_vae.py:193-196 `random_lf_composition`): the composed proposals
   * run with log_jac = True and have no accept probability of their own; the ONE accept probability of the
   * composition (and the loss) is formed by the caller from the links' outputs, its cotangents come back in here */
  int32_t no_accept;             /* 1: no p / v1 / loss term; seeds = dLx_in (required), dLv_in, dlogjac_in     */
  const float* dLv_in;           /* (N, d) or NULL: cotangent on the proposal's momentum Lv                     */
  const float* dlogjac_in;       /* (N) or NULL: cotangent on the proposal's summed log-Jacobian                */
  float* Lv_out;                 /* (N, d) or NULL                                                              */
  float* logjac_out;             /* (N) or NULL                                                                 */
  int32_t gemm_mode;             /* as L2hmcSplitArgs.gemm_mode                                                 */
  int32_t net_mode;              /* 0: one launch per S/T/Q net evaluation and one per its reverse wherever the
                                  *    shapes allow (H % 4 == 0, d even, widths <= 256); 1: three GEMM launches each
                                  *    (the form every other shape takes; here for A/B measurements and tests)     */
  /* ---- training on a caller-supplied energy (energy = decoder = NULL; aux only feeds aux_encoder, if any) ------- */
  L2hmcEnergyCallback energy_cb; /* U / grad U at a trajectory point (as in L2hmcSplitArgs), or NULL              */
  L2hmcHvpCallback hvp_cb;       /* its Hessian-vector product; required with energy_cb                          */
  void* energy_cb_user;          /* first argument of both callbacks                                              */
  /* ---- (ABI 6) training caller-supplied nets: xnet = vnet = aux_encoder = NULL, H ignored; with any target (a built-in
   *      `energy`, energy_cb + hvp_cb, or the decoder posterior).  Every net evaluation of the forward pass is net_cb (final S | T | Q into the library's
   *      stash), every one of the reverse sweep net_vjp_cb; `grad` then holds ONE float, d loss / d eps (accumulated) --
   *      the nets' parameter gradients are accumulated by the callback on the caller's side ---------------------------- */
  L2hmcNetCallback net_cb;
  L2hmcNetVjpCallback net_vjp_cb;
  void* net_cb_user;             /* first argument of both                                                        */
} L2hmcTrainSplitArgs;

int64_t l2hmc_train_split_grad_floats(int32_t d, int32_t H, const L2hmcMlp3* aux_encoder);
int64_t l2hmc_train_split_workspace_floats(int64_t n_chains, int32_t d, int32_t H, int32_t T,
                                           const L2hmcMlp3* aux_encoder, const L2hmcMlp3* decoder);
int l2hmc_train_split_grad(const L2hmcTrainSplitArgs* args, void* stream);

/* One Adam update as tf.train.AdamOptimizer applies it (SCGExperiment.ipynb raw 178-181) over the flat parameter
 * vector laid out like the gradient of l2hmc_train_propose_grad ([XNet | VNet | alpha]):
 *   lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t m / (sqrt(v) + eps)
 * step = t >= 1.  last_is_log_eps: the last parameter is alpha = log eps while grad holds d/d eps there, so its
 * gradient is multiplied by exp(alpha) first (dynamics.py:50-58). */
int l2hmc_adam_step(float* params, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float epsilon, int64_t step, int32_t last_is_log_eps, void* stream);

/* AIS bookkeeping around one annealed HMC transition (utils/ais.py:44-66), initial energy N(0, I):
 *   begin: w += dbeta (|x|^2 / 2 - U_final(x))                                   (ais.py:58-59)
 *          v  = normals                          if refreshment < 0               (ais.py:57)
 *             = v sqrt(1 - r) + normals sqrt(r)  otherwise                        (ais.py:55)
 *   end:   accept = p - u >= 0;  x = accept ? Lx : x;  v = accept ? Lv : -Lv;  alpha_sum += p   (ais.py:62-66)
 * The transition itself is l2hmc_trajectory in HMC mode with energy.anneal_beta = beta. */
int l2hmc_ais_begin_step(const float* x, const float* U_final, const float* normals, float refreshment,
                         float dbeta, float* w, float* v, int64_t n_chains, int32_t d, void* stream);
int l2hmc_ais_end_step(const float* Lx, const float* Lv, const float* p, const float* u, float* x, float* v,
                       float* alpha_sum, int64_t n_chains, int32_t d, void* stream);

/* The draws the sampler loop would use, written out ((M,N,d) normals, (M,N) direction bits,
 * (M,N) uniforms; any output may be NULL): for tests, and for reproducing a run's randomness. */
int l2hmc_rng_fill(uint64_t seed, uint64_t proposal0, int64_t chain_offset, int64_t n_chains,
                   int32_t d, int32_t n_proposals, float* v_out, uint8_t* dir_out, float* u_out,
                   void* stream);

/* autocovariance / acl_spectrum (utils/func_utils.py:45-54,114-116) of a recorded chain history
 * X (steps, N, d) kept on the device:
 *   A_out[tau] = mean_t [ sum_{n,k} X[t,n,k] X[t+tau,n,k] / N ] / scale^2,  tau = 0 .. steps-2
 * (no mean subtraction, like the reference).  sums_out (steps-1 doubles) receives the raw
 * sums S(tau) = sum_t sum_{n,k} X[t] X[t+tau] -- the quantity ranks all-reduce when the chains
 * are sharded; pass n_total = chains over ALL ranks for the normalisation of A_out (A_out may be
 * NULL when only the partial sums are wanted).
 * `workspace`: l2hmc_autocov_workspace_doubles(steps, n_chains, d) doubles -- every block writes its partial
 * sums there and a second kernel adds them in block order, so S (and with it the thresholded ESS) is bitwise
 * reproducible; NULL = accumulate with double atomics instead (order-dependent in the last bits). */
int64_t l2hmc_autocov_workspace_doubles(int64_t steps, int64_t n_chains, int32_t d);
int l2hmc_autocov(const float* X, int64_t steps, int64_t n_chains, int32_t d, double scale,
                  int64_t n_total, double* sums_out, double* A_out, double* workspace, void* stream);

/* Binding check.  The argument structs grow by trailing fields from one ABI version to the next (and
 * l2hmc_pack_nets' buffer by the lane layout: always size it with l2hmc_packed_nets_floats).  A binding built
 * against an older header would pass shorter structs, so besides comparing l2hmc_abi_version() with the
 * L2HMC_ABI_VERSION it was written for, a binding compares sizeof() of its own mirror of every struct with
 * what the library was compiled with (l2hmc_amd/_ffi.py does both when it loads the library).
 * which: one of L2HMC_STRUCT_*; returns sizeof in bytes, or L2HMC_ERR_ARG. */
enum { L2HMC_STRUCT_NET = 0, L2HMC_STRUCT_ENERGY = 1, L2HMC_STRUCT_TRAJECTORY_ARGS = 2, L2HMC_STRUCT_MLP3 = 3,
       L2HMC_STRUCT_SPLIT_ARGS = 4, L2HMC_STRUCT_TRAIN_ARGS = 5, L2HMC_STRUCT_TRAIN_SPLIT_ARGS = 6, L2HMC_STRUCT_TRAIN_STEP = 7 };
int64_t l2hmc_struct_bytes(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* L2HMC_H_ */
