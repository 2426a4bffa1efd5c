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

    python bench.py [--gpus N --steps K --warmup W]            (N>1: launched by torchrun)

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant
kernel = traj_kernel, MFMA-bound: algorithmic fp32 flops / launch duration vs the 157.3
TFLOP/s fp32-MFMA peak) and `cpu_baseline` (the numpy oracle -- reference algorithm, both
directions computed like sampler.py:35-36 -- timed on the host cores, rank 0, N=1 only).
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


def cpu_baseline(prob, budget_s=12.0):
    """Reference algorithm (numpy oracle, fp32, both directions for all chains) on the host.
    numpy's elementwise ops run on ONE thread; only the small matmuls go to the BLAS pool, which is
    capped at 8 threads here -- `cores` reports that cap (the threads the run could actually use)."""
    from oracle import l2hmc_oracle as O
    threads = min(8, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter, threads = None, 1
    n = min(CHAINS, prob["x0"].shape[0])
    en = O.Gaussian(np.zeros(D), np.diag(1.0 / prob["var"]))
    dyn = O.Dynamics(D, en, T, 0.1, prob["mask"], prob["nets"]["xnet"], prob["nets"]["vnet"])
    rng = np.random.RandomState(1)
    x = prob["x0"][:n]
    reps, t0 = 0, time.perf_counter()
    with np.errstate(all="ignore"):
        while True:
            vf, vb = rng.randn(n, D).astype(np.float32), rng.randn(n, D).astype(np.float32)
            dr, u = rng.randint(0, 2, n), rng.rand(n).astype(np.float32)
            _, _, _, x = O.propose(x, dyn, vf, vb, dr, u, both_directions=True)
            reps += 1
            el = time.perf_counter() - t0
            if el > budget_s or reps >= 200:
                break
    if limiter is not None:
        limiter.restore_original_limits()
    return {"value": n * T * reps / el, "unit": "chain·leapfrog-steps/s", "cores": threads,
            "kind": "port",
            "sample": "%d proposals of %d chains (ICG d=50, T=10), numpy fp32 oracle (reference algorithm: both "
                      "directions computed for every chain, sampler.py:35-36; row-wise Gaussian energy instead of "
                      "the reference's N x N product); useful chain-steps counted once; %.1f s" % (reps, n, el)}


def ess_leg(dev, train_steps=5000):
    """ESS/sec on the notebook's SCG-2D target (BASELINE.json configs[0] shape: 200 chains, Lf=10):
    (i) the HMC(eps=0.15) sampler whose ESS the reference publishes (nb raw 388: 5.63e-3 per MH step);
    (ii) the L2HMC sampler TRAINED IN THIS RUN with the notebook's recipe (raw 156-181, 254-271: 5000 Adam
    steps on 200 chains; published ESS 2.61e-1, ratio 46).  2000 MH steps per sampler in one persistent
    launch each, history + autocovariance on the device."""
    import torch
    from l2hmc_amd import Dynamics, distributions, func_utils, layers, sample_chain
    from l2hmc_amd.training import Trainer
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    scale = float(np.sqrt(np.trace(cov)))
    dist = distributions.Gaussian(np.zeros(2), cov)
    n, steps = 200, 2000
    gen = torch.Generator(device=dev).manual_seed(0)
    x0 = torch.as_tensor(dist.get_samples(n, rng=np.random.RandomState(0)), dtype=torch.float32, device=dev)

    def measure(dyn, direction):
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

    hmc = Dynamics(2, dist.get_energy_function(), T=10, eps=0.15, hmc=True, device=dev)
    out = measure(hmc, None)
    out.update({"workload": "SCG-2D, HMC eps=0.15, 200 chains x 2000 MH steps, Lf=10 (nb raw 288-298, 388)",
                "reference_ess_per_mh_step": 5.63e-3})
    if train_steps > 0:
        torch.manual_seed(0)
        np.random.seed(0)
        layers.set_default_device(dev)
        dyn = Dynamics(2, dist.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)
        dyn.generator = gen
        tr = Trainer(dyn)
        xs = torch.randn(n, 2, device=dev, generator=gen)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(train_steps):
            _, _, xs, _ = tr.step(xs)
        torch.cuda.synchronize(dev)
        t_train = time.perf_counter() - t0
        l2 = measure(dyn, torch.randint(0, 2, (steps, n), device=dev, dtype=torch.uint8, generator=gen))
        l2.update({"workload": "SCG-2D, L2HMC sampler trained in this run (%d Adam steps, 200 chains; nb raw 156-181, "
                               "254-271), then 200 chains x 2000 MH steps" % train_steps,
                   "train_seconds": t_train, "train_ms_per_step": 1e3 * t_train / train_steps,
                   "reference_ess_per_mh_step": 2.61e-1,
                   "ess_ratio_vs_hmc": l2["ess_per_mh_step"] / out["ess_per_mh_step"], "reference_ess_ratio": 46.0,
                   "ess_per_sec_ratio_vs_hmc": l2["ess_per_sec"] / out["ess_per_sec"]})
        out["l2hmc"] = l2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chains", type=int, default=CHAINS, help="chains per GPU")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--rng", choices=["philox", "bank"], default="philox",
                    help="philox: momenta / direction / accept uniforms drawn in-kernel (counter-based, "
                         "keyed by global chain index); bank: pre-generated draws read from HBM")
    ap.add_argument("--preheat", type=int, default=1000,
                    help="untimed proposals run BEFORE the --warmup steps so that the GPU clocks have ramped "
                         "whatever --warmup is (a 20-proposal warm-up lasts 0.6 ms: the timed steps would then "
                         "run 7 %% slower on cold clocks); reported in config.preheat_proposals")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true", help="skip the SCG-2D ESS/sec leg (N=1 only)")
    ap.add_argument("--ess-train-steps", type=int, default=5000,
                    help="Adam steps for the L2HMC sampler of the ESS leg (0 = HMC only)")
    ap.add_argument("--bank", type=int, default=0,
                    help="distinct pre-generated random draws, cycled (0 = 2 x proposals-per-launch, min 16)")
    ap.add_argument("--proposals-per-launch", type=int, default=25,
                    help="MH proposals chained inside one launch of the persistent sampler kernel "
                         "(1 = one launch per step); a shorter last launch covers any remainder")
    args = ap.parse_args()

    # (the host driver only supports dmabuf IPC: RCCL needs this for multi-process runs)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert args.gpus == world, "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"

    from l2hmc_amd import Dynamics, _ffi, distributions, layers
    from oracle import l2hmc_oracle as O   # only for NET_KEYS naming + the cpu_baseline leg

    n = args.chains
    prob = make_problem(0, n, dev)                     # identical model on every rank
    layers.set_default_device(dev)
    dyn = Dynamics(D, distributions.Gaussian(np.zeros(D), np.diag(prob["var"])).get_energy_function(),
                   T=T, eps=0.1, net_factory=layers.stq_network(H), device=dev)
    dyn.mask = prob["mask"]
    dyn.variant = args.variant
    with torch.no_grad():
        for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
            for k in O.NET_KEYS:
                w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)   # chains differ per rank
    xa = (torch.randn(n, D, device=dev, generator=gen) *
          torch.as_tensor(np.sqrt(prob["var"]), dtype=torch.float32, device=dev)).contiguous()
    xb = torch.empty_like(xa)
    M = max(1, args.proposals_per_launch)
    B = args.bank if args.bank > 0 else max(16, 2 * M)
    B = (B + M - 1) // M * M
    if args.rng == "bank":
        v_bank = torch.randn(B, n, D, device=dev, generator=gen)
        d_bank = torch.randint(0, 2, (B, n), device=dev, dtype=torch.uint8, generator=gen)
        u_bank = torch.rand(B, n, device=dev, generator=gen)

    L = _ffi.lib()
    a = _ffi.L2hmcTrajectoryArgs()
    a.packed_nets = dyn._packed_nets().data_ptr()
    a.energy = dyn._fn.c_struct(dev, 1.0)
    a.masks, a.trig = dyn._mask.data_ptr(), dyn._trig.data_ptr()
    a.alpha, a.eps_host = dyn.alpha.data_ptr(), 0.0
    a.n_chains, a.d, a.H, a.T, a.step_begin, a.n_steps = n, D, H, T, 0, T
    a.variant, a.n_proposals = args.variant, M
    p_out = torch.empty((M, n), device=dev)
    a.p_out = p_out.data_ptr()
    stream = torch.cuda.current_stream(dev).cuda_stream
    bufs = [xa, xb]

    state = {"flip": 0}

    def launch(first, count):
        """proposals [first, first + count): one launch of the persistent sampler kernel"""
        src, dst = bufs[state["flip"]], bufs[state["flip"] ^ 1]
        state["flip"] ^= 1
        b = first % B
        if b + count > B:
            b = 0
        a.n_proposals = count
        a.x, a.x_next = src.data_ptr(), dst.data_ptr()
        if args.rng == "philox":
            a.rng_flags, a.rng_seed = _ffi.RNG_V | _ffi.RNG_DIR | _ffi.RNG_U, 20260926
            a.rng_proposal0, a.chain_offset = first, rank * n
        else:
            a.v = v_bank[b].data_ptr()
            a.direction = d_bank[b].data_ptr()
            a.u = u_bank[b].data_ptr()
        rc = L.l2hmc_trajectory(a, stream)
        if rc:
            _ffi.check(rc)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def plan(first, total):
        return [(first + i, min(M, total - i)) for i in range(0, total, M)]

    pre = max(0, args.preheat)
    for f, c in plan(0, pre):                    # clock ramp (untimed, not part of --warmup)
        launch(f, c)
    for f, c in plan(pre, args.warmup):
        launch(f, c)
    timed = plan(pre + args.warmup, args.steps)
    nl = len(timed)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for f, c in timed:
        launch(f, c)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    mean_p = float(p_out.mean())
    finite = bool(torch.isfinite(bufs[state["flip"]]).all())
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    if rank == 0:
        steps_total = float(n) * world * T * args.steps
        flops_cs = algorithmic_flops_per_chain_step(D, H, T, 3 * D)      # diagonal precision: 3d
        m_avg = args.steps / float(nl)                                   # proposals per launch (mean)
        per_launch_flops = flops_cs * n * T * m_avg
        launch_s = gpu_ms * 1e-3 / nl                                    # HIP events on the launch stream
        ach = per_launch_flops / launch_s / 1e12
        out = {
            "metric": "chain_leapfrog_steps_per_sec", "value": steps_total / elapsed,
            "unit": "chain·leapfrog-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ICG-50D (ill-conditioned Gaussian d=50), %d chains per GPU, Lf=10, "
                                   "S/T/Q nets H=10, direction-mixed propose + MH per step" % n,
                       "chains_per_gpu": n, "x_dim": D, "hidden": H, "leapfrog_steps": T,
                       "proposals_per_launch": M, "rng": args.rng, "preheat_proposals": pre,
                       "parallelism": "chains sharded, no data-path collective",
                       "mean_accept_prob": mean_p, "state_finite": finite},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "kernel": "traj_kernel", "flops_per_chain_step": flops_cs,
                         "launch_us": launch_s * 1e6,
                         "hbm_frac": algorithmic_bytes_per_chain_step(D, T) * n * T * m_avg / launch_s / 1e9 / PEAK_HBM_GBS},
        }
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):          # HBM bytes per launch from the committed PMC passes
            t = json.load(open(tj))
            if t.get("workload_chains") == n and t.get("proposals_per_launch", 1) == M:
                out["roofline"]["traffic"] = 1024.0 * (t["fetch_kb"] + t["write_kb"])
                out["roofline"]["traffic_source"] = t["source"]
        if world == 1 and not args.no_ess:
            out["ess"] = ess_leg(dev, args.ess_train_steps)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
