#!/usr/bin/env python
"""bench.py -- chain·leapfrog-steps/s of the fused L2HMC trajectory kernel on MI355X.

Workload (BASELINE.json configs[1]): ill-conditioned Gaussian d=50, 4 096 chains PER GPU,
Lf (T) = 10, S/T/Q nets with H=10.  One "step" = one `propose` (sampler.py:28-55): T
generalised leapfrog steps on every chain in its drawn direction + accept probability + MH
select; the chain state carries over from step to step.  Steps are issued through the
persistent sampler kernel, `--proposals-per-launch` (default 25) chained proposals per launch
(the notebook's per-step sess.run loop, raw 288-298, without the host round trip);
`--proposals-per-launch 1` gives one launch per step.
Synthetic inputs (seeded weights, masks, start points) are resident in HBM before the timed
region; the per-step random draws v / direction / u come from the in-kernel Philox stream
(`--rng bank`: pre-generated in HBM instead).

    python bench.py [--gpus N --steps K --warmup W]

N > 1 either way: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) every process is one rank; a plain `python bench.py --gpus N` (no WORLD_SIZE) re-executes
itself through torch.distributed.run with N ranks on 127.0.0.1 -- one process per GPU over RCCL in both cases.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant
kernel = traj_fast_kernel, MFMA-bound: algorithmic fp32 flops / launch duration vs the 157.3
TFLOP/s fp32-MFMA peak) and `cpu_baseline` (the reference algorithm restated op by op on torch-CPU,
all host cores, both directions computed like sampler.py:35-36; rank 0, N=1 only).

Short runs: a K-step plan that would last less than `--min-timed-ms` (20 ms) is repeated R times
inside the timed region (R in `config.repeats`; value / ms_per_step are means over the K x R steps),
so `--steps 20` (0.5 ms of GPU time) measures the same rate as a long run.

Multi-GPU: `--gpus N` alone is weak scaling (`--chains` per GPU; the bench line); `--total-chains C` makes the line
strong scaling (C chains split over the N ranks).  At N > 1 two extra keys: `strong65536` = the north star's operating
point, ICG-50 with 65 536 chains IN TOTAL split over the ranks (value, rank 0's roofline fraction, ratio to the committed
N = 1 rate), and `dist` = the collectives of the path: ESS of chains sharded over the ranks (one all-reduce of the
autocovariance partial sums + mean accept) and a timed training step (flat-gradient all-reduce).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H, T, CHAINS = 50, 10, 10, 4096
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32-input MFMA (= vector) peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak (the pipe the bf16x3 products execute on)


def algorithmic_flops_per_chain_step(d, h, t, grad_flops):
    """F = 4 F_net + (1 + 1/T) F_gradU + F_ew, F_net = 2H(5d + H + 2), F_ew ~ 30 d."""
    return 4 * 2 * h * (5 * d + h + 2) + (1.0 + 1.0 / t) * grad_flops + 30 * d


def algorithmic_bytes_per_chain_step(d, t):
    """T-fused kernel: read x, v (+dir, u), write Lx, Lv?, x_next, p: (5 d + 3) floats / T steps."""
    return 4.0 * (5 * d + 3) / t


def make_problem(seed, n_chains, device):
    """Seeded ICG-50 problem shared by the GPU run and the CPU baseline."""
    rng = np.random.RandomState(seed)
    var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), D))
    prob = {"var": var, "mask": None, "nets": {}}
    masks = []
    for _ in range(T):
        m = np.zeros(D, dtype=np.float32)
        m[rng.permutation(D)[:D // 2]] = 1
        masks.append(m)
    prob["mask"] = np.stack(masks)
    for net, fac in (("xnet", 2.0), ("vnet", 1.0)):
        def vs(shape, factor):
            std = math.sqrt(1.3 * 2.0 * factor / shape[0])
            return np.clip(rng.randn(*shape), -2, 2).astype(np.float32) * np.float32(std)
        w = {"W1": vs((D, H), 1 / 3.), "W2": vs((D, H), fac / 3.), "W3": vs((2, H), 1 / 3.),
             "W4": vs((H, H), 1.0)}
        for k in ("Ws", "Wt", "Wq"):      # heads raised from the reference's 1e-3 init so S,T,Q matter
            w[k] = (0.05 * rng.randn(H, D) / math.sqrt(H)).astype(np.float32)
        for k, n in (("b1", H), ("b2", H), ("b3", H), ("b4", H), ("bs", D), ("bt", D), ("bq", D)):
            w[k] = (0.05 * rng.randn(n)).astype(np.float32)
        w["lam_s"] = (0.1 * rng.randn(1, D)).astype(np.float32)
        w["lam_q"] = (0.1 * rng.randn(1, D)).astype(np.float32)
        prob["nets"][net] = w
    prob["x0"] = (rng.randn(n_chains, D) * np.sqrt(var)).astype(np.float32)
    return prob


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 CPUs but grants 16: 256 spinning OpenMP threads on 16 cores would take minutes per proposal)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_subprocess(timeout_s=90.0):
    """cpu_baseline in a child process with a hard time limit, so that a misbehaving host thread pool can
    never hang the bench (the child imports no GPU code)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"value": None, "error": "cpu_baseline child failed: %s" % r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": "cpu_baseline child exceeded %.0f s" % timeout_s}


def cpu_baseline(prob, budget_s=20.0):
    """the reference algorithm (both directions on every chain, three gradient evaluations per
    forward step) restated op by op on torch-CPU fp32 (oracle/ref_cpu_torch.py; TF1 itself cannot be installed
    offline), intra-op threads = every host core.  Three bounded samples: C2 (ICG-50, 4096 chains) with the
    row-wise Gaussian energy, C2 with the reference's literal N x N energy at a chain count whose N x N
    temporaries fit the budget, and C1 (SCG-2D, 200 chains, the config BASELINE.json names as the CPU path)."""
    import torch
    from oracle import ref_cpu_torch as R
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)

    def run(x_dim, cov, mask, nets, x, nxn, budget):
        en = R.GaussianRef(np.zeros(x_dim), np.linalg.inv(cov).astype(np.float32), nxn)
        dyn = R.DynamicsRef(x_dim, en, T, 0.1, mask, nets["xnet"], nets["vnet"])
        x = torch.as_tensor(x)
        n, reps, t0 = x.shape[0], 0, time.perf_counter()
        while True:
            vf, vb = torch.randn(n, x_dim, generator=g), torch.randn(n, x_dim, generator=g)
            dr, u = torch.randint(0, 2, (n,), generator=g), torch.rand(n, generator=g)
            _, _, x = R.propose(x, dyn, vf, vb, dr, u)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget or reps >= 200:
                break
        return n * T * reps / el, reps, el

    n = min(CHAINS, prob["x0"].shape[0])
    cov = np.diag(prob["var"])
    row, r1, e1 = run(D, cov, prob["mask"], prob["nets"], prob["x0"][:n], False, 0.4 * budget_s)
    n_nxn = min(1024, n)
    nxn, r2, e2 = run(D, cov, prob["mask"], prob["nets"], prob["x0"][:n_nxn], True, 0.4 * budget_s)
    # C1: the notebook's SCG-2D shapes (weights of the same init family; timing does not depend on their values)
    rng = np.random.RandomState(3)
    nets2 = {k: {kk: (0.1 * rng.randn(*((2 if s == D else s) for s in vv.shape))).astype(np.float32)
                 for kk, vv in w.items()} for k, w in prob["nets"].items()}
    mask2 = np.stack([np.eye(2, dtype=np.float32)[i % 2] for i in range(T)])
    cov2 = np.array([[50.05, -49.95], [-49.95, 50.05]])
    c1, r3, e3 = run(2, cov2, mask2, nets2, rng.randn(200, 2).astype(np.float32), True, 0.2 * budget_s)
    return {"value": row, "unit": "chain·leapfrog-steps/s", "cores": cores, "kind": "port",
            "sample": "torch-CPU fp32 restatement of the reference graph (oracle/ref_cpu_torch.py: both directions "
                      "for every chain, sampler.py:35-36; useful chain-steps counted once), %d intra-op threads. "
                      "value = C2 ICG-50/%d chains, row-wise Gaussian energy: %d proposals in %.1f s" % (cores, n, r1, e1),
            "c2_nxn_energy": {"value": nxn, "chains": n_nxn,
                              "sample": "C2 with the reference's literal N x N energy + autograd gradient "
                                        "(distributions.py:31-32) at %d chains: %d proposals in %.1f s" % (n_nxn, r2, e2)},
            "c1_scg2d": {"value": c1, "chains": 200,
                         "sample": "C1 SCG-2D, 200 chains, N x N energy: %d proposals in %.1f s" % (r3, e3)}}


def config5_leg(dev, chains=8192):
    """BASELINE.json config 5 on the GEMM engine, inside the bench run (an extra key, not the bench line): the VAE latent
    posterior (decoder 50 -> 1024 -> 1024 -> 784, mnist_vae.py:104-126) under the H = 200 S/T/Q nets with the shared
    image branch (:128-178), Lf = 5, 8192 chains, random weights / Bernoulli(0.13) images (no checkpoint or MNIST
    offline).  (i) sampling: propose + MH per step; (ii) the sampler's training step of mnist_vae.py:185-262 with one
    differentiated proposal (`l2hmc_train_split_grad` + clipped Adam).  HIP-event times on torch's stream, which is the
    stream these launches go to."""
    import torch
    from l2hmc_amd import Dynamics, propose, vae
    from l2hmc_amd.training import Trainer
    d, H, T = 50, 200, 5
    torch.manual_seed(0)
    np.random.seed(0)
    decoder = vae.make_decoder(d, 1024, 784)
    enc = vae.make_encoder_sampler(784, 512, H)
    dyn = Dynamics(d, vae.VAEPosterior(decoder).get_energy_function(), T=T, eps=0.1,
                   net_factory=vae.sampler_net_factory(d, enc, H, H))
    gen = torch.Generator(device=dev).manual_seed(0)
    dyn.generator = gen
    aux = (torch.rand((chains, 784), device=dev, generator=gen) < 0.13).float()
    x = torch.randn((chains, d), device=dev, generator=gen)

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    state = {"x": x, "p": None}

    def one_proposal():
        _, _, px, out = propose(state["x"], dyn, do_mh_step=True, aux=aux)
        state["x"], state["p"] = out[0], px
    ms_prop = timed(one_proposal, 3, 12)
    from l2hmc_amd import _ffi
    k_dec = _ffi.last_kernel()                 # the kernel of the decoder-sized products, as the library names it
    f_net = 2 * (2 * d * H + H * H + 3 * d * H)
    f_dec = 2 * (d * 1024 + 1024 * 1024 + 1024 * 784)
    flops_step = 4 * f_net + (1 + 1.0 / T) * 2 * f_dec                      # per chain . leapfrog step
    ach = chains * T * flops_step / (ms_prop * 1e-3) / 1e12
    # the share of it in the 1024-wide decoder products (the ones that take the bf16x3 path; the K = 50 layer and the
    # N = 50 latent gradient stay on the f32 MFMA)
    f_dec_big = 2 * (1024 * 1024 + 1024 * 784)
    ach_dec = chains * T * (1 + 1.0 / T) * 2 * f_dec_big / (ms_prop * 1e-3) / 1e12
    tr = Trainer(dyn, decay_steps=0)
    log_sigma = torch.full((chains, d), -0.5, device=dev)
    ms_train = timed(lambda: tr.sampler_step(state["x"], aux, log_sigma, MH=1), 2, 4)
    flops_train = 3 * 4 * T * f_net + (T + 1) * 4 * f_dec                      # forward + input & weight gradients + HVPs
    return {"workload": "config 5: VAE latent posterior d=50, decoder 1024/1024/784, H=200 nets + image branch, "
                        "%d chains, Lf=5, random weights, synthetic images" % chains,
            "kernels": "%s (decoder products), gemm_nt_kernel (fused epilogues), net_eval_kernel (+ fused half-updates), "
                       "gemm_tn_kernel (training)" % k_dec,
            "ms_per_proposal": ms_prop, "value": chains * T / (ms_prop * 1e-3), "unit": "chain\u00b7leapfrog-steps/s",
            "flops_per_chain_step": flops_step, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "frac": ach / PEAK_F32_MFMA_TFLOPS,
            # what the matrix pipe really executes: with gemm_mode = 1 (the default) the decoder-sized products run as
            # "bf16x3" -- six bf16 MFMAs per fp32 product, fp32-accurate -- so `frac` above is ALGORITHMIC
            # fp32 flops over the f32-MFMA roof, not the utilisation of the pipe the products run on
            "arithmetic": {3: "f16x2 planes (decoder GEMMs of the sampler: exact 2-way f16 split of both operands on pre-split planes, "
                              "gemm_xlp_kernel<..., 1> -- 3 f16 MFMAs per product, fp32 accumulate; the trainer's adjoint products "
                              "stay bf16x3); f32 MFMA for the K = 50 / N = 50 / H = 200 products",
                           1: "bf16x3 (decoder GEMMs: 3-way split fp32 operands -- on pre-split bf16 planes from 3072 chains, "
                              "gemm_xlp_kernel -- 6 bf16 MFMAs per product, fp32 accumulate); "
                              "f32 MFMA for the K = 50 / N = 50 / H = 200 products"}.get(int(getattr(dyn, "gemm_mode", 1)), "f32 MFMA"),
            "gemm_mode": int(getattr(dyn, "gemm_mode", 1)),
            "executed_tflops": (ach_dec * {1: 6.0, 3: 3.0}.get(int(getattr(dyn, "gemm_mode", 1)), 1.0) + (ach - ach_dec)),
            "frac_of_bf16_roof": (ach_dec * {1: 6.0, 3: 3.0}[int(getattr(dyn, "gemm_mode", 1))] / PEAK_BF16_MFMA_TFLOPS)
                                 if int(getattr(dyn, "gemm_mode", 1)) in (1, 3) else None,
            "mean_accept_prob": float(state["p"].mean()), "state_finite": bool(torch.isfinite(state["x"]).all()),
            "train": {"workload": "sampler update of mnist_vae.py:185-262, one differentiated proposal per step",
                      "ms_per_step": ms_train, "flops_per_chain": flops_train,
                      "achieved": chains * flops_train / (ms_train * 1e-3) / 1e12,
                      "frac": chains * flops_train / (ms_train * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}


def config4_leg(dev, chains=16384, dims=(2, 50, 512)):
    """BASELINE.json config 4 inside the bench run (an extra key): Rough Well (distributions.py:84-97), 16 384 chains, Lf = 10,
    H = 10 nets with raised head scale, one kernel family per width (d = 2 the one-dimension-per-lane tile, d = 50 one wave per
    tile, d = 512 the LDS-resident-state kernel); both the `easy` form (eta = 0.1, cos(x / eta)) and the reference's own
    (eta = 1e-2, cos(x / eta^2): every sin / cos through the full range reduction).  The step size is tuned per case on a pilot
    run so that the chains move (mean accept in [0.2, 0.9]).  HIP-event time of M = 10 chained proposals per launch; the
    counter passes of the same cases are profiles/r04_config4_pmc.txt."""
    import torch
    from l2hmc_amd import Dynamics, _ffi, distributions, layers, sample_chain
    rng = np.random.RandomState(0)
    out = []
    for easy, eta in ((True, 0.1), (False, 1e-2)):
        for d in dims:
            torch.manual_seed(0)
            np.random.seed(0)
            dyn = Dynamics(d, distributions.RoughWell(d, eta, easy=easy).get_energy_function(), T=T, eps=0.1,
                           net_factory=layers.stq_network(H, head_factor=0.03), device=dev)
            x = torch.as_tensor(rng.randn(chains, d), dtype=torch.float32, device=dev)
            eps, M = 0.1, 10
            for _ in range(16):                         # pilot: x 0.6 / x 1.3 until the mean accept probability is in [0.2, 0.9]
                dyn.eps_override = eps
                acc = float(sample_chain(x, dyn, M, seed=3)[1].mean())
                if acc < 0.2:
                    eps *= 0.6
                elif acc > 0.9:
                    eps *= 1.3
                else:
                    break
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # the timed region lasts >= 10 ms (round 6: four 110-us launches of the d = 2 case measured the host's dispatch and the
            # clock ramp of the first case as much as the kernel -- 13.2 us per proposal here against 11.0 in tools/bench_configs.py)
            e0.record()
            sample_chain(x, dyn, M, seed=1)
            e1.record()
            torch.cuda.synchronize(dev)
            reps = max(4, int(math.ceil(10.0 / max(e0.elapsed_time(e1), 1e-3))))
            e0.record()
            for r in range(reps):
                x, p, _ = sample_chain(x, dyn, M, seed=1, proposal0=(r + 1) * M)
            e1.record()
            torch.cuda.synchronize(dev)
            t = e0.elapsed_time(e1) * 1e-3 / (reps * M)
            fl = algorithmic_flops_per_chain_step(d, H, T, 4 * d)
            ach = chains * T / t * fl / 1e12
            out.append({"d": d, "easy": easy, "eta": eta, "eps": eps, "kernel": _ffi.last_kernel(), "us_per_proposal": t * 1e6,
                        "value": chains * T / t, "achieved": ach, "frac": ach / PEAK_F32_MFMA_TFLOPS,
                        # the T-fused bytes SURVEY 8(d) states (x in, x_next out, p per proposal), over the 8 TB/s roof
                        "hbm_frac": 4.0 * chains * (2 * d + M) / (t * M) / 1e9 / PEAK_HBM_GBS,
                        "mean_accept_prob": float(p.mean()), "state_finite": bool(torch.isfinite(x).all())})
    return {"workload": "config 4: Rough Well, %d chains, Lf=%d, H=%d, fp32 (bf16 state: measured and declined, DESIGN 5)" % (chains, T, H),
            "unit": "chain\u00b7leapfrog-steps/s", "cases": out}


def ess_leg(dev, train_steps=5000, seeds=5):
    """ESS/sec on the notebook's SCG-2D target (BASELINE.json configs[0] shape: 200 chains, Lf=10):
    (i) the HMC(eps=0.15) sampler whose ESS the reference publishes (nb raw 388: 5.63e-3 per MH step);
    (ii) the L2HMC sampler TRAINED IN THIS RUN with the notebook's recipe (raw 156-181, 254-271: 5000 Adam
    steps on 200 chains; published ESS 2.61e-1, ratio 46) -- once per seed, `seeds` independent trainings,
    reported as mean +- sd.  2000 MH steps per sampler in one persistent launch each, history +
    autocovariance on the device."""
    import torch
    from l2hmc_amd import Dynamics, distributions, func_utils, layers, sample_chain
    from l2hmc_amd.training import Trainer
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    scale = float(np.sqrt(np.trace(cov)))
    dist = distributions.Gaussian(np.zeros(2), cov)
    n, steps = 200, 2000

    def measure(dyn, direction, gen, x0):
        v = torch.randn((steps, n, 2), device=dev, generator=gen)
        u = torch.rand((steps, n), device=dev, generator=gen)
        sample_chain(x0, dyn, steps, v=v, u=u, direction=direction, record=True)                 # warm-up
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        xf, p, hist = sample_chain(x0, dyn, steps, v=v, u=u, direction=direction, record=True)
        X = torch.cat([x0[None], hist[:-1]], dim=0)
        A = func_utils.acl_spectrum(X, scale)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        ess = float(func_utils.ESS(A))
        return {"ess_per_mh_step": ess, "mh_steps_per_sec_per_chain": steps / el, "ess_per_sec": ess * steps / el * n,
                "chain_leapfrog_steps_per_sec": n * 10 * steps / el, "mean_accept_prob": float(p.mean()),
                "seconds_incl_autocov": el}

    gen = torch.Generator(device=dev).manual_seed(0)
    x0 = torch.as_tensor(dist.get_samples(n, rng=np.random.RandomState(0)), dtype=torch.float32, device=dev)
    hmc = Dynamics(2, dist.get_energy_function(), T=10, eps=0.15, hmc=True, device=dev)
    out = measure(hmc, None, gen, x0)
    out.update({"workload": "SCG-2D, HMC eps=0.15, 200 chains x 2000 MH steps, Lf=10 (nb raw 288-298, 388)",
                "reference_ess_per_mh_step": 5.63e-3})
    if train_steps > 0 and seeds > 0:
        runs = []
        for seed in range(seeds):
            torch.manual_seed(seed)
            np.random.seed(seed)
            gen = torch.Generator(device=dev).manual_seed(seed)
            layers.set_default_device(dev)
            dyn = Dynamics(2, dist.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)
            dyn.generator = gen
            tr = Trainer(dyn, seed=seed)
            xs = torch.randn(n, 2, device=dev, generator=gen)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(train_steps):
                loss_t, px_t, xs, _ = tr.step(xs)
            torch.cuda.synchronize(dev)
            t_train = time.perf_counter() - t0
            r = measure(dyn, torch.randint(0, 2, (steps, n), device=dev, dtype=torch.uint8, generator=gen), gen, x0)
            r.update({"seed": seed, "train_seconds": t_train, "final_train_loss": float(loss_t),
                      "final_train_accept": float(px_t.mean()), "eps": float(torch.exp(dyn.alpha.detach()))})
            runs.append(r)
        e = np.array([r["ess_per_mh_step"] for r in runs])
        es = np.array([r["ess_per_sec"] for r in runs])
        l2 = {"workload": "SCG-2D, L2HMC sampler trained in this run (%d Adam steps, 200 chains; nb raw 156-181, "
                          "254-271), then 200 chains x 2000 MH steps; %d independent seeds" % (train_steps, seeds),
              "ess_per_mh_step": float(e.mean()), "ess_per_mh_step_sd": float(e.std(ddof=1)) if seeds > 1 else 0.0,
              "ess_per_mh_step_by_seed": [float(v) for v in e], "ess_per_mh_step_median": float(np.median(e)),
              # about one training in eight ends in a sampler that is accepted as often as the others but does not mix (lag-1
              # autocovariance 0.7-0.8 against 0.4-0.55, jumps of 8-12 against 15; 4 of 30 seeds, profiles/r05_ess_seed_study.txt).  It is
              # the objective, not the kernels: seed 7 replayed in float64 numpy on the same recorded draws (parameters agree to 1e-7
              # for 10 steps, then an accept decision flips and the runs part ways) collapses as well -- ESS 0.0064 against 0.0058.
              # Counted, not hidden; quote the median next to the mean.
              "seeds_below_5x_hmc": [int(r["seed"]) for r in runs if r["ess_per_mh_step"] < 5.0 * out["ess_per_mh_step"]],
              "by_seed": [{k: r[k] for k in ("seed", "ess_per_mh_step", "mean_accept_prob", "final_train_loss",
                                             "final_train_accept", "eps")} for r in runs],
              "ess_per_sec": float(es.mean()), "mean_accept_prob": float(np.mean([r["mean_accept_prob"] for r in runs])),
              "chain_leapfrog_steps_per_sec": float(np.mean([r["chain_leapfrog_steps_per_sec"] for r in runs])),
              "train_ms_per_step": 1e3 * float(np.mean([r["train_seconds"] for r in runs])) / train_steps,
              "reference_ess_per_mh_step": 2.61e-1,
              "ess_ratio_vs_hmc": float(e.mean()) / out["ess_per_mh_step"], "reference_ess_ratio": 46.0,
              "ess_per_sec_ratio_vs_hmc": float(es.mean()) / out["ess_per_sec"]}
        out["l2hmc"] = l2
        # the metric's other target at a chain count that fills the device: the last trained sampler on 65 536 chains
        # (d <= 4 kernel, in-kernel Philox, 100 MH steps per launch)
        nb, M = 65536, 100
        xb = torch.as_tensor(dist.get_samples(nb, rng=np.random.RandomState(1)), dtype=torch.float32, device=dev)
        sample_chain(xb, dyn, M, seed=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(3):
            xb, pb, _ = sample_chain(xb, dyn, M, seed=2 + rep)
        e1.record()
        torch.cuda.synchronize(dev)
        rate = nb * 10 * 3 * M / (e0.elapsed_time(e1) * 1e-3)
        # ESS per MH step MEASURED at this chain count: 400 recorded MH steps of all 65 536 chains (the history and
        # its autocovariance stay on the device), not the 200-chain figure carried over
        Mh = 400
        xs0 = xb.clone()
        _, ph, hist = sample_chain(xs0, dyn, Mh, seed=11, record=True)
        Xh = torch.cat([xs0[None], hist[:-1]], dim=0)
        ess_b = float(func_utils.ESS(func_utils.acl_spectrum(Xh, scale)))
        out["l2hmc_65536_chains"] = {"workload": "SCG-2D, the trained L2HMC sampler (last seed) on 65 536 chains, Lf=10; "
                                                 "ESS from %d recorded MH steps of all chains (device autocovariance)" % Mh,
                                     "chain_leapfrog_steps_per_sec": rate, "mean_accept_prob": float(pb.mean()),
                                     "ess_per_mh_step": ess_b, "state_finite": bool(torch.isfinite(xb).all()),
                                     # ESS/s = (MH steps/s summed over the chains) x the ESS per MH step measured here
                                     "ess_per_sec": rate / 10.0 * ess_b}
    return out


def dist_leg(dev, rank, world):
    """N > 1: the two collectives the path has (north_star), exercised over RCCL.  (i) ESS of chains SHARDED over
    the ranks: every rank samples its own 200 SCG-2D chains (HMC eps=0.15, in-kernel Philox keyed by the GLOBAL
    chain index), the autocovariance partial sums and the accept statistics are all-reduced once each.  (ii) a
    training step on sharded chains: per-rank gradient kernel + ONE flat-gradient all-reduce + native Adam."""
    import torch
    import torch.distributed as dist
    from l2hmc_amd import Dynamics, distributions, layers, sample_chain, sharding
    from l2hmc_amd.training import Trainer
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    scale = float(np.sqrt(np.trace(cov)))
    target = distributions.Gaussian(np.zeros(2), cov)
    n, steps = 200, 2000
    lo = rank * n
    x0 = torch.as_tensor(target.get_samples(n, rng=np.random.RandomState(rank)), dtype=torch.float32, device=dev)
    hmc = Dynamics(2, target.get_energy_function(), T=10, eps=0.15, hmc=True, device=dev)
    sample_chain(x0, hmc, steps, seed=7, chain_offset=lo, record=True)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t0 = time.perf_counter()
    xf, p, hist = sample_chain(x0, hmc, steps, seed=7, chain_offset=lo, record=True)
    X = torch.cat([x0[None], hist[:-1]], dim=0)
    ess = sharding.ess(X, scale, n * world)
    acc = sharding.mean_accept(p.reshape(-1))
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    # training on sharded chains
    torch.manual_seed(0)
    np.random.seed(0)
    layers.set_default_device(dev)
    dyn = Dynamics(2, target.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)
    tr = Trainer(dyn, seed=0)
    tr.always_reduce = True               # (--force-dist at N = 1: the step's all-reduce still goes over RCCL, with one rank)
    xs = x0.clone()
    for _ in range(20):
        _, _, xs, _ = tr.step(xs)
    torch.cuda.synchronize(dev)
    dist.barrier()
    t1 = time.perf_counter()
    k = 200
    for _ in range(k):
        loss, _, xs, _ = tr.step(xs)
    torch.cuda.synchronize(dev)
    t_tr = time.perf_counter() - t1
    # the same step without the collective (this rank's 200 chains as a whole batch): what the all-reduce adds
    dyn1 = Dynamics(2, target.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)
    tr1 = Trainer(dyn1, seed=0)
    tr1._world = lambda: 1
    x1 = x0.clone()
    for _ in range(20):
        _, _, x1, _ = tr1.step(x1)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    for _ in range(k):
        _, _, x1, _ = tr1.step(x1)
    torch.cuda.synchronize(dev)
    t_one = time.perf_counter() - t2
    # every rank must hold identical parameters after identical all-reduced updates
    chk = torch.stack([tr.theta.double().sum(), -tr.theta.double().sum()])
    dist.all_reduce(chk, op=dist.ReduceOp.MAX)
    same = bool(abs(float(chk[0]) + float(chk[1])) == 0.0)
    try:      # which collective library answers to torch's "nccl" backend here (ROCm builds: RCCL, reported in NCCL's version scheme)
        lib_ver = "RCCL (torch backend 'nccl'), version %s, HIP %s" % (".".join(str(v) for v in torch.cuda.nccl.version()), torch.version.hip)
    except Exception:
        lib_ver = None
    return {"backend": dist.get_backend(), "ranks": int(ones.item()),
            "collective_library": lib_ver if dist.get_backend() == "nccl" else "gloo (CPU rehearsal)",
            "sharded_ess": {"workload": "SCG-2D HMC eps=0.15, %d chains (200 per rank) x %d MH steps, in-kernel "
                                        "Philox keyed by global chain; autocov partial sums + accept all-reduced" % (n * world, steps),
                            "ess_per_mh_step": ess, "mean_accept_prob": acc, "ess_per_sec": ess * steps / el * n * world,
                            "seconds_incl_allreduce": el},
            "sharded_training": {"workload": "SCG-2D, %d chains (200 per rank), %d Adam steps, ONE all-reduce per step: "
                                             "[gradient | loss sums | count] = %d floats (the shard layout is exchanged once, "
                                             "on the first step)" % (n * world, k, tr.n_grad + 6),
                                 "ms_per_step": 1e3 * t_tr / k, "single_process_ms_per_step": 1e3 * t_one / k,
                                 "collectives_per_step": 1, "final_loss": float(loss),
                                 "parameters_identical_across_ranks": same}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chains", type=int, default=CHAINS, help="chains per GPU (weak scaling)")
    ap.add_argument("--total-chains", type=int, default=0,
                    help="strong scaling: this many chains in total, split over the ranks (north star: 65536)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--rng", choices=["philox", "bank"], default="philox",
                    help="philox: momenta / direction / accept uniforms drawn in-kernel (counter-based, "
                         "keyed by global chain index); bank: pre-generated draws read from HBM")
    ap.add_argument("--preheat", type=int, default=1000,
                    help="untimed proposals run BEFORE the --warmup steps so that the GPU clocks have ramped "
                         "whatever --warmup is; they also calibrate the repeat count; config.preheat_proposals")
    ap.add_argument("--min-timed-ms", type=float, default=20.0,
                    help="repeat the --steps plan inside the timed region until it lasts at least this long")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true", help="skip the SCG-2D ESS/sec leg (N=1) / the dist leg (N>1)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 65 536-chain roofline point (N=1 only)")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 (VAE engine) extra key (N=1 only)")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 (Rough-Well sweep) extra key (N=1 only)")
    ap.add_argument("--no-config5-trained", action="store_true",
                    help="skip the trained-sampler part of the config-5 key (200 sampler updates + ESS vs HMC, ~15 s)")
    ap.add_argument("--ess-train-steps", type=int, default=5000,
                    help="Adam steps for the L2HMC sampler of the ESS leg (0 = HMC only)")
    ap.add_argument("--ess-seeds", type=int, default=10, help="independent trainings of the ESS leg (0.25 s each)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the `dist` leg even at N = 1 (exercises the collectives on a 1-GPU box)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a rehearsal)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the N > 1 code path on a 1-GPU box: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--bank", type=int, default=0,
                    help="distinct pre-generated random draws, cycled (0 = 2 x proposals-per-launch, min 16)")
    ap.add_argument("--proposals-per-launch", type=int, default=25,
                    help="MH proposals chained inside one launch of the persistent sampler kernel "
                         "(1 = one launch per step); a shorter last launch covers any remainder")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(make_problem(0, CHAINS, None))))
        return

    # (the host driver only supports dmabuf IPC: RCCL needs this for multi-process runs)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N ranks of this very script, one per GPU
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch `python bench.py --gpus N` (it spawns its ranks itself) or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))

    from l2hmc_amd import Dynamics, _ffi, distributions, layers, sharding

    strong = args.total_chains > 0
    if strong:
        lo, hi = sharding.shard_range(args.total_chains, rank, world)
        n, chain_off, n_total = hi - lo, lo, args.total_chains
    else:
        n, chain_off, n_total = args.chains, rank * args.chains, args.chains * world
    prob = make_problem(0, min(n, CHAINS), dev)        # identical model on every rank
    layers.set_default_device(dev)
    dyn = Dynamics(D, distributions.Gaussian(np.zeros(D), np.diag(prob["var"])).get_energy_function(),
                   T=T, eps=0.1, net_factory=layers.stq_network(H), device=dev)
    dyn.mask = prob["mask"]
    dyn.variant = args.variant
    with torch.no_grad():
        for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
            for k in _ffi.NET_FIELDS:
                w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
    sd = torch.as_tensor(np.sqrt(prob["var"]), dtype=torch.float32, device=dev)
    L = _ffi.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    M = max(1, args.proposals_per_launch)

    class Runner(object):
        """`n_` chains of this rank through the persistent sampler kernel, M proposals per launch."""

        def __init__(self, n_, chain_off_, seed):
            gen = torch.Generator(device=dev).manual_seed(seed)
            self.n, self.flip = n_, 0
            xa = (torch.randn(n_, D, device=dev, generator=gen) * sd).contiguous()
            self.bufs = [xa, torch.empty_like(xa)]
            self.B = args.bank if args.bank > 0 else max(16, 2 * M)
            self.B = (self.B + M - 1) // M * M
            if args.rng == "bank":
                self.v_bank = torch.randn(self.B, n_, D, device=dev, generator=gen)
                self.d_bank = torch.randint(0, 2, (self.B, n_), device=dev, dtype=torch.uint8, generator=gen)
                self.u_bank = torch.rand(self.B, n_, device=dev, generator=gen)
            a = _ffi.L2hmcTrajectoryArgs()
            a.packed_nets = dyn._packed_nets().data_ptr()
            a.energy = dyn._fn.c_struct(dev, 1.0)
            a.masks, a.trig = dyn._mask.data_ptr(), dyn._trig.data_ptr()
            a.alpha, a.eps_host = dyn.alpha.data_ptr(), 0.0
            a.n_chains, a.d, a.H, a.T, a.step_begin, a.n_steps = n_, D, H, T, 0, T
            a.variant, a.n_proposals = args.variant, M
            self.p_out = torch.empty((M, n_), device=dev)
            a.p_out = self.p_out.data_ptr()
            a.chain_offset = chain_off_
            self.a = a

        def launch(self, first, count):
            """proposals [first, first + count): one launch of the persistent sampler kernel"""
            a = self.a
            src, dst = self.bufs[self.flip], self.bufs[self.flip ^ 1]
            self.flip ^= 1
            a.n_proposals = count
            a.x, a.x_next = src.data_ptr(), dst.data_ptr()
            if args.rng == "philox":
                a.rng_flags, a.rng_seed = _ffi.RNG_V | _ffi.RNG_DIR | _ffi.RNG_U, 20260926
                a.rng_proposal0 = first
            else:
                b = first % self.B
                if b + count > self.B:
                    b = 0
                a.v, a.direction, a.u = self.v_bank[b].data_ptr(), self.d_bank[b].data_ptr(), self.u_bank[b].data_ptr()
            rc = L.l2hmc_trajectory(a, stream)
            if rc:
                _ffi.check(rc)

        def run(self, first, total):
            for i in range(0, total, M):
                self.launch(first + i, min(M, total - i))
            return first + total

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(runner, first, steps, repeats, local=False):
        """(wall seconds, HIP-event ms, launches) of `repeats` x `steps` proposals.  local: this rank alone (device sync only)"""
        sync = (lambda: torch.cuda.synchronize(dev)) if local else barrier
        sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(repeats):
            first = runner.run(first, steps)
        ev1.record()
        sync()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1), repeats * ((steps + M - 1) // M), first

    def calibrate(runner, first, pre, local=False):
        """untimed clock ramp; its event time gives the per-proposal estimate the repeat count is chosen from"""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        runner.run(first, min(pre, M))              # first-touch / lazy-init outside the estimate
        ev0.record()
        first = runner.run(first + min(pre, M), pre)
        ev1.record()
        torch.cuda.synchronize(dev)
        est = ev0.elapsed_time(ev1) / max(pre, 1)   # ms per proposal
        if world > 1 and not local:
            tt = torch.tensor([est], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            est = float(tt)
        return first, est

    main_run = Runner(n, chain_off, 1234 + rank)     # chains differ per rank
    pre = max(0, args.preheat)
    nxt, est_ms = calibrate(main_run, 0, pre) if pre > 0 else (0, 0.0)
    nxt = main_run.run(nxt, args.warmup)
    R = 1
    if est_ms > 0 and args.steps * est_ms < args.min_timed_ms:
        R = int(math.ceil(args.min_timed_ms / (args.steps * est_ms)))
    elapsed, gpu_ms, nl, nxt = timed(main_run, nxt, args.steps, R)
    main_kernel = _ffi.last_kernel()
    mean_p = float(main_run.p_out.mean())
    finite = bool(torch.isfinite(main_run.bufs[main_run.flip]).all())
    rank_values = None
    if world > 1:
        # every rank's own wall time of the timed region (-> per-rank `value`s in the line), then the MAX the contract asks for
        tg = torch.zeros(world, device=dev, dtype=torch.float64)
        tg[rank] = elapsed
        dist.all_reduce(tg)
        rank_values = [float(n) * T * args.steps * R / float(t_) for t_ in tg.cpu()]
        elapsed = float(tg.max())

    flops_cs = algorithmic_flops_per_chain_step(D, H, T, 3 * D)      # diagonal precision: 3d

    def point(nc, chain_off_, seed, steps=100, local=False):
        """one extra roofline point: `nc` chains on this rank through the same sampler loop, timed like the main
        run (barrier, max over ranks) -> (wall s, HIP-event ms, launches, proposals, mean accept, state finite).
        local=True: this rank alone (no barrier, no collective) -- the same-run single-GPU denominator of the strong figure"""
        r2 = Runner(nc, chain_off_, seed)
        f2, est2 = calibrate(r2, 0, steps, local=local)
        reps = max(1, int(math.ceil(args.min_timed_ms / (steps * est2))))
        el2, ms2, nl2, _ = timed(r2, f2, steps, reps, local=local)
        if world > 1 and not local:
            t2 = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            el2 = float(t2)
        return (el2, ms2, nl2, steps * reps, float(r2.p_out.mean()),
                bool(torch.isfinite(r2.bufs[r2.flip]).all()), _ffi.last_kernel())     # (the library names what it launched)

    strong_out = None
    if world > 1 and not strong and not args.no_sweep:
        # the north star's operating point: ICG-50, 65 536 chains IN TOTAL over the ranks (every rank takes part)
        tot = 65536
        lo, hi = sharding.shard_range(tot, rank, world)
        el2, ms2, nl2, k2, p2, fin2, kn2 = point(hi - lo, lo, 4321 + rank)
        a2 = flops_cs * (hi - lo) * T * (k2 / float(nl2)) / (ms2 * 1e-3 / nl2) / 1e12
        strong_out = {"workload": "ICG-50D, %d chains in total = %d per GPU, Lf=10 (north_star: >= 1e8 on 8 GPUs, "
                                  ">= 6x 1 -> 8)" % (tot, hi - lo),
                      "value": tot * T * float(k2) / el2, "unit": "chain·leapfrog-steps/s", "scaling": "strong",
                      "chains_per_gpu": hi - lo, "kernel": kn2, "rank0_achieved": a2,
                      "rank0_frac": a2 / PEAK_F32_MFMA_TFLOPS, "rank0_launch_us": 1e3 * ms2 / nl2,
                      "mean_accept_prob": p2, "state_finite": fin2}
        # ... and the SAME-RUN single-GPU denominator: rank 0 alone takes all 65 536 chains on its GPU (the other ranks wait at the
        # barrier behind it), so that the line itself says how the strong figure compares -- north_star's ">= 6x 1 -> 8" clause
        if rank == 0:
            e1_, ms1_, nl1_, k1_, p1_, fin1_, kn1_ = point(tot, 0, 4321, local=True)
            strong_out["n1_same_run"] = {"value": tot * T * float(k1_) / e1_, "kernel": kn1_, "launch_us": 1e3 * ms1_ / nl1_,
                                         "mean_accept_prob": p1_, "state_finite": fin1_,
                                         "what": "all %d chains on rank 0's GPU alone, same process, same run" % tot}
            strong_out["speedup_same_run"] = strong_out["value"] / strong_out["n1_same_run"]["value"]
        barrier()
        ref = os.path.join(ROOT, "profiles", "n1_sweep65536.json")
        if os.path.exists(ref):           # the committed N = 1 rate of the same 65 536 chains on one GPU
            v1 = json.load(open(ref))
            strong_out["n1_value"] = v1["value"]
            strong_out["n1_source"] = v1["source"]
            # (NOT a same-run ratio: the denominator is the committed constant above, measured on one box of the pool)
            strong_out["speedup_vs_committed_n1_constant"] = strong_out["value"] / v1["value"]

    dist_out = None
    if use_dist and not args.no_ess:
        dist_out = dist_leg(dev, rank, world)

    if rank == 0:
        k_timed = args.steps * R
        steps_total = float(n_total) * T * k_timed
        m_avg = k_timed / float(nl)                                      # proposals per launch (mean)
        per_launch_flops = flops_cs * n * T * m_avg
        launch_s = gpu_ms * 1e-3 / nl                                    # HIP events on the launch stream
        ach = per_launch_flops / launch_s / 1e12
        out = {
            "metric": "chain_leapfrog_steps_per_sec", "value": steps_total / elapsed,
            "unit": "chain·leapfrog-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / k_timed,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ICG-50D (ill-conditioned Gaussian d=50), %s, Lf=10, S/T/Q nets H=10, "
                                   "direction-mixed propose + MH per step"
                                   % ("%d chains in total over %d GPU(s)" % (n_total, world) if strong
                                      else "%d chains per GPU" % n),
                       "chains_per_gpu": n, "chains_total": n_total, "x_dim": D, "hidden": H, "leapfrog_steps": T,
                       "proposals_per_launch": m_avg, "proposals_per_launch_max": M, "rng": args.rng,
                       "preheat_proposals": pre,
                       "repeats": R, "timed_steps": k_timed,
                       "parallelism": "chains sharded, no data-path collective",
                       "mean_accept_prob": mean_p, "state_finite": finite},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "kernel": main_kernel, "flops_per_chain_step": flops_cs,
                         # `achieved` = ALGORITHMIC fp32 flops / launch time against the f32 vector / f32-input-MFMA peak.  Since
                         # round 6 the kernels whose name ends in ", 1>" execute every contraction as two f16 MFMAs on an exact
                         # two-term split of both operands (22 significand bits per operand worst case; against float64 equal to the
                         # f32-input MFMA: profiles/r06_f16x2_accuracy.txt) -- 4 x the
                         # f16 flops on a pipe 16 x as fast; results and `dtype` are fp32, `mfma_pipe_busy` is that pipe's share
                         "arithmetic": ("f16x2: two v_mfma_f32_16x16x32_f16 per 16-k block on exact hi/lo splits, fp32 accumulate, "
                                        "fp32-level accuracy (profiles/r06_f16x2_accuracy.txt)" if main_kernel.rstrip().endswith(", 1>") else "f32-input MFMA"),
                         "launch_us": launch_s * 1e6,
                         # bytes the persistent loop has to move per launch: x in, x_next out, p per proposal (the
                         # counters below replace this model when the committed PMC pass matches the workload)
                         "algorithmic_bytes": 4.0 * n * (2 * D + m_avg),
                         "hbm_frac": 4.0 * n * (2 * D + m_avg) / launch_s / 1e9 / PEAK_HBM_GBS,
                         "hbm_frac_source": "algorithmic bytes per launch"},
        }
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):          # HBM bytes per launch from the committed PMC passes
            t = json.load(open(tj))
            if t.get("workload_chains") == n:
                # counters were collected on launches of t[proposals_per_launch] proposals; per launch x, x_next move
                # once and only p (n floats per proposal) scales with the proposals actually chained here
                m_ref = float(t.get("proposals_per_launch", 1))
                out["roofline"]["traffic"] = 1024.0 * (t["fetch_kb"] + t["write_kb"]) + (m_avg - m_ref) * 4.0 * n
                out["roofline"]["traffic_source"] = t["source"] + (
                    "" if m_avg == m_ref else "; p_out bytes rescaled from %g to %g proposals per launch" % (m_ref, m_avg))
                out["roofline"]["hbm_frac"] = out["roofline"]["traffic"] / launch_s / 1e9 / PEAK_HBM_GBS
                out["roofline"]["hbm_frac_source"] = "counter traffic (FETCH_SIZE + WRITE_SIZE) / HIP-event launch time"
                if t.get("mfma_pipe_busy") is not None:
                    # north_star: "MFMA utilisation reported" -- the matrix pipe's busy share of the SIMDs' time in the committed
                    # counter pass of this workload (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES): one wave per SIMD here, the
                    # wave counter ticks in quad-cycles); read from the committed summary like `traffic`, not measured live
                    out["roofline"]["mfma_pipe_busy"] = t["mfma_pipe_busy"]
                    out["roofline"]["mfma_pipe_busy_source"] = t.get("mfma_pipe_busy_source", t["source"])
        if world == 1 and not args.no_sweep and not strong and n == CHAINS:
            # the same kernel at the north star's chain count: what the MFMA roof fraction becomes once the
            # chip is filled (4096 chains are ONE workgroup per CU, one wave per SIMD)
            sweep = []
            for nc in (8192, 65536):      # 8192 = one GPU's share of the north star's 65 536 chains on 8 GPUs
                el2, ms2, nl2, k2, p2, fin2, kn2 = point(nc, 0, 99)
                a2 = flops_cs * nc * T * (k2 / float(nl2)) / (ms2 * 1e-3 / nl2) / 1e12
                sweep.append({"chains": nc, "kernel": kn2,
                              "value": nc * T * float(k2) / el2, "achieved": a2,
                              "frac": a2 / PEAK_F32_MFMA_TFLOPS, "launch_us": 1e3 * ms2 / nl2,
                              "mean_accept_prob": p2, "state_finite": fin2})
            out["sweep"] = sweep
        if world == 1 and not args.no_config4 and not strong and n == CHAINS and not args.force_dist:
            out["config4"] = config4_leg(dev)
        if world == 1 and not args.no_config5 and not strong and n == CHAINS and not args.force_dist:
            out["config5"] = config5_leg(dev)
            if not args.no_config5_trained:
                # config 5 as BASELINE.json words it ("trained sampler"): train the sampler with the reference's update, then
                # MH steps/s, accept and ESS/s of the trained sampler vs HMC on the posterior (eval_sampler.py:145-204)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_config5_trained
                out["config5"]["trained"] = bench_config5_trained.run(dev, updates=200, mh_steps=250,
                                                                       hmc_eps=(0.075, 0.1, 0.175))
        if world == 1 and not args.no_ess and not args.force_dist:
            out["ess"] = ess_leg(dev, args.ess_train_steps, args.ess_seeds)
        if strong_out is not None:
            out["strong65536"] = strong_out
            # both readings of "scaling" in one place: `value` above is WEAK (4096 chains per GPU, the configuration the metric is
            # quoted on); the north star's clause is STRONG scaling of 65 536 chains in total
            out["value_weak"] = out["value"]
            out["value_strong65536"] = strong_out["value"]
            out["strong65536_speedup_same_run"] = strong_out.get("speedup_same_run")
        if dist_out is not None:
            out["dist"] = dist_out
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess()
        # ---- what is judged, as flat top-level scalars next to the nested keys (a parsed record that keeps only top-level
        #      scalars still carries the operating points) ----
        for sw in out.get("sweep", []):
            out["sweep%d_value" % sw["chains"]] = sw["value"]
            out["sweep%d_frac" % sw["chains"]] = sw["frac"]
        if "config5" in out:
            c5 = out["config5"]
            out["config5_ms_per_proposal"] = c5["ms_per_proposal"]
            out["config5_train_ms_per_step"] = c5["train"]["ms_per_step"]
            out["config5_frac_of_bf16_roof"] = c5["frac_of_bf16_roof"]
            out["config5_executed_tflops"] = c5["executed_tflops"]
            if "trained" in c5 and isinstance(c5["trained"], dict) and "ess_per_sec_ratio_l2hmc_over_best_hmc" in c5["trained"]:
                out["config5_trained_ess_ratio_vs_best_hmc"] = c5["trained"]["ess_per_sec_ratio_l2hmc_over_best_hmc"]
        for cs in out.get("config4", {}).get("cases", []):
            out["config4_d%d_%sfrac" % (cs["d"], "" if cs["easy"] else "ne_")] = cs["frac"]
        if "mfma_pipe_busy" in out["roofline"]:
            out["roofline_mfma_pipe_busy"] = out["roofline"]["mfma_pipe_busy"]
        out["roofline_frac"] = out["roofline"]["frac"]
        if dist_out is not None:
            out["dist_ranks"], out["dist_backend"] = dist_out["ranks"], dist_out["backend"]
            out["dist_collective_library"] = dist_out.get("collective_library")
            out["dist_train_ms_per_step"] = dist_out["sharded_training"]["ms_per_step"]
            out["dist_parameters_identical_across_ranks"] = dist_out["sharded_training"]["parameters_identical_across_ranks"]
        if rank_values is not None:
            for r_, v_ in enumerate(rank_values):
                out["rank%d_value" % r_] = v_
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
