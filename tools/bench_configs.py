#!/usr/bin/env python
"""Throughput of the fused sampler kernel on every BASELINE.json configuration (GPU box):

    python tools/bench_configs.py

C1 SCG d=2 / 200 chains, C2 ICG d=50 / 4096 chains (the bench.py workload), C3 MoG d=2 / 65 536
chains / Lf=25, C4 Rough Well d in {2 .. 512} / 16 384 chains, all with H=10 S/T/Q nets (random,
head std raised so S, T, Q are exercised), direction-mixed propose + MH, M chained proposals per
launch with the in-kernel Philox draws.  The step size is tuned PER CONFIGURATION on a short pilot run (x 0.6 / x 1.3
until the mean accept probability lies in [0.2, 0.9]; printed) -- round 3 ran every configuration at eps = 0.1, where the
wide targets accept nothing (rates of a sampler that never moves).  Reports useful chain.leapfrog-steps/s and the fraction of
the fp32-MFMA roof with the algorithmic FLOPs of SURVEY.md 8(d).  (C5, the VAE posterior on the
split engine, is tools/bench_vae.py.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l2hmc_amd import _ffi
if os.environ.get("L2HMC_DBG_LIB"):                       # time an alternative build of the library
    _ffi.LIB_PATH = os.path.abspath(os.environ["L2HMC_DBG_LIB"])
import bench
from l2hmc_amd import Dynamics, distributions as D, layers, sample_chain


def tune_eps(dyn, x, M, lo=0.2, hi=0.9):
    """step size with a mean accept probability inside [lo, hi] (pilot: M proposals from x per trial)"""
    eps = 0.1
    for _ in range(16):
        dyn.eps_override = eps
        _, p, _ = sample_chain(x, dyn, M, seed=3)
        a = float(p.mean())
        if a < lo:
            eps *= 0.6
        elif a > hi:
            eps *= 1.3
        else:
            break
    return eps


def run(name, dist, d, n, T, grad_flops, x0, M, reps):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    dyn = Dynamics(d, dist.get_energy_function(), T=T, eps=0.1,
                   net_factory=layers.stq_network(10, head_factor=0.03), device=dev)
    dyn.variant = int(os.environ.get("L2HMC_VARIANT", "0"))      # kernel geometry override (see l2hmc.h)
    x = torch.as_tensor(x0, dtype=torch.float32, device=dev)
    eps = tune_eps(dyn, x, M)
    for _ in range(2):
        sample_chain(x, dyn, M, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    p = None
    for r in range(reps):
        x, p, _ = sample_chain(x, dyn, M, seed=1, proposal0=(r + 1) * M)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / (reps * M)
    steps = n * T / t
    fl = bench.algorithmic_flops_per_chain_step(d, 10, T, grad_flops)
    from l2hmc_amd import _ffi as ffi_
    print("%-30s chains %6d d %3d Lf %2d M %3d eps %.4f: %8.2f us / proposal  %.3e steps/s  mfma-frac %.3f  accept %.2f  %s"
          % (name, n, d, T, M, eps, t * 1e6, steps, steps * fl / 1e12 / bench.PEAK_F32_MFMA_TFLOPS, float(p.mean()),
             ffi_.last_kernel()), flush=True)


def main():
    rng = np.random.RandomState(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wide":        # only the wide Rough-Well cases
        for d in (64, 128, 256, 512):
            run("C4 RoughWell (easy, eta=0.1)", D.RoughWell(d, 0.1, easy=True), d, 16384, 10, 4 * d,
                rng.randn(16384, d), 10, 4)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "one":         # one Rough-Well case (the counter passes of tools/collect_r04.sh)
        d = int(sys.argv[2])
        if len(sys.argv) > 3 and sys.argv[3] == "noneasy":
            run("C4 RoughWell (eta=1e-2)", D.RoughWell(d, 1e-2, easy=False), d, 16384, 10, 4 * d, rng.randn(16384, d), 10, 4)
        else:
            run("C4 RoughWell (easy, eta=0.1)", D.RoughWell(d, 0.1, easy=True), d, 16384, 10, 4 * d, rng.randn(16384, d), 10, 4)
        return
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    g = D.Gaussian(np.zeros(2), cov)
    run("C1 SCG-2D", g, 2, 200, 10, 2 * 2 * 2, g.get_samples(200, rng=rng), 25, 20)
    var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), 50))
    run("C2 ICG-50D", D.Gaussian(np.zeros(50), np.diag(var)), 50, 4096, 10, 3 * 50,
        rng.randn(4096, 50) * np.sqrt(var), 25, 8)
    mog = D.GMM([np.array([2.0, 0.0]), np.array([-2.0, 0.0])], [0.1 * np.eye(2), 0.1 * np.eye(2)], [0.5, 0.5])
    run("C3 MoG-2D", mog, 2, 65536, 25, 2 * (2 * 4 + 4), mog.get_samples(65536), 10, 4)
    for d in (2, 8, 32, 50, 128, 512):
        run("C4 RoughWell (easy, eta=0.1)", D.RoughWell(d, 0.1, easy=True), d, 16384, 10, 4 * d,
            rng.randn(16384, d), 10, 4)
    # the reference's own (non-easy) form, distributions.py:84-97 with easy=False: cos(x / eta^2), eta = 1e-2 -- arguments
    # of 1e4 x, i.e. every sin / cos goes through the full range reduction ("run non-easy as a second series")
    for d in (2, 8, 32, 50, 128, 512):
        run("C4 RoughWell (eta=1e-2)", D.RoughWell(d, 1e-2, easy=False), d, 16384, 10, 4 * d,
            rng.randn(16384, d), 10, 4)


if __name__ == "__main__":
    main()
