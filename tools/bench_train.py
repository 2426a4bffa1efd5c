#!/usr/bin/env python
"""Training-step timing (GPU box): the notebook's training configuration (SCGExperiment.ipynb raw
156-181, 254-271: SCG 2-d, 200 chains, T=10, H=10) and the 50-d ICG at 200 / 4096 chains.

    python tools/bench_train.py [--no-cpu]

Prints, per configuration: the `l2hmc_train_propose_grad` call rate (HIP events around back-to-back calls, one x- plus one
z-proposal = the device work of a training step; for the d <= 4 kernel this loop is HOST-bound -- the kernel itself is
46 us on the notebook config by `rocprofv3 --kernel-trace`, profiles/r03_train_timing.txt), the whole `Trainer.step` wall time (kernel + RNG
+ Adam + MH select), and -- as the CPU reference point -- the numpy restatement of the same
loss-and-gradient (oracle/l2hmc_train_oracle.py, float32) on the same inputs."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l2hmc_amd import _ffi
if os.environ.get("L2HMC_LIB"):          # kernel experiments: an alternative build of the library
    _ffi.LIB_PATH = os.path.abspath(os.environ["L2HMC_LIB"])
from l2hmc_amd import Dynamics, distributions as D, layers
from l2hmc_amd.training import Trainer


def make(case, n, dev):
    rng = np.random.RandomState(0)
    if case == "mog2d":                           # config 3's target: two components at (+-2, 0), variance 0.1
        d, cov = 2, np.diag([4.1, 0.1])
        dist = D.GMM([np.array([2.0, 0.0]), np.array([-2.0, 0.0])], [0.1 * np.eye(2), 0.1 * np.eye(2)], [0.5, 0.5])
        dyn = Dynamics(d, dist.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)
        dyn.generator = torch.Generator(device=dev).manual_seed(0)
        return dyn, torch.as_tensor(dist.get_samples(n, rng).astype(np.float32), device=dev), None
    if case.startswith("rough"):                   # config 4's target at a dimension beyond the fused trainers: GEMM engine
        d = int(case[5:])
        dist = D.RoughWell(d, 0.1, easy=True)
        dyn = Dynamics(d, dist.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)
        dyn.generator = torch.Generator(device=dev).manual_seed(0)
        return dyn, torch.as_tensor(rng.randn(n, d).astype(np.float32), device=dev), None
    if case == "scg2d":
        d, cov = 2, np.array([[50.05, -49.95], [-49.95, 50.05]])
    else:
        d = 50
        cov = np.diag(np.logspace(-2, 2, d))
    dyn = Dynamics(d, D.Gaussian(np.zeros(d), cov).get_energy_function(), T=10, eps=0.1,
                   net_factory=layers.stq_network(10), device=dev)
    dyn.generator = torch.Generator(device=dev).manual_seed(0)
    x = torch.as_tensor(rng.randn(n, d).astype(np.float32) * np.sqrt(np.diag(cov)).astype(np.float32), device=dev)
    return dyn, x, cov


def cpu_reference(dyn, x, cov, reps):
    from oracle import l2hmc_oracle as O
    from oracle import l2hmc_train_oracle as TO
    d = x.shape[1]
    tgt = TO.GaussianTarget(np.zeros(d, np.float32), np.linalg.inv(cov).astype(np.float32), np.float32)
    nets = {n: {k: w[k].detach().cpu().numpy() for k in O.NET_KEYS} for n, w in (("x", dyn._xw), ("v", dyn._vw))}
    xn = x.cpu().numpy()
    rng = np.random.RandomState(1)
    v = rng.randn(*xn.shape).astype(np.float32)
    dr = rng.randint(0, 2, xn.shape[0]).astype(np.uint8)
    mask = dyn._mask.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(reps):
        for start in (xn, v):     # x-proposal and z-proposal
            TO.propose_loss_and_grad(start, v, dr, tgt, nets["x"], nets["v"], 0.1, mask, 10, dtype=np.float32)
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda", 0)
    for case, n, steps in (("scg2d", 200, 200), ("mog2d", 200, 200), ("icg50", 200, 50), ("icg50", 4096, 20),
                           ("rough128", 4096, 10), ("rough512", 4096, 5)):
        dyn, x, cov = make(case, n, dev)
        tr = Trainer(dyn)
        tr.variant = int(os.environ.get("L2HMC_TRAIN_VARIANT", "0"))      # kernel-choice experiments (include/l2hmc.h)
        v = torch.randn_like(x)
        dr = torch.randint(0, 2, (n,), device=dev, dtype=torch.uint8)
        for _ in range(3):
            tr._propose_grad(x, v, dr, n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        x2, v2, d2 = torch.cat([x, v]), torch.cat([v, v]), torch.cat([dr, dr])
        for _ in range(steps):
            tr._propose_grad(x2, v2, d2, n)          # x- and z-proposal of a training step: one launch
        e1.record()
        torch.cuda.synchronize()
        kern = e0.elapsed_time(e1) * 1e-3 / steps
        xs = x
        for _ in range(3):
            _, _, xs, _ = tr.step(xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, _, xs, _ = tr.step(xs)
        torch.cuda.synchronize()
        full = (time.perf_counter() - t0) / steps
        line = "%s chains %5d: propose+grad kernel %9.1f us / training step, Trainer.step %9.1f us" % (case, n, kern * 1e6, full * 1e6)
        if "--no-cpu" not in sys.argv and cov is not None:
            try:
                cpu = cpu_reference(dyn, x, cov, 2 if n <= 200 else 1)
                line += ", numpy oracle %9.1f ms (x%.0f)" % (cpu * 1e3, cpu / kern)
            except Exception as e:      # the oracle signature is test infrastructure; report, don't die
                line += ", numpy oracle failed: %r" % (e,)
        print(line, flush=True)


if __name__ == "__main__":
    main()
