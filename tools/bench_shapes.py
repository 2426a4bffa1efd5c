#!/usr/bin/env python
"""us per proposal of the sampler loop over the shapes that take the DIFFERENT instantiations of traj_fast_kernel (GPU box):
    python tools/bench_shapes.py [lib.so]        (A/B builds: tools/build_variant_full.sh; profiles/r06_resident_tails.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import _ffi
if len(sys.argv) > 1:
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import torch
from l2hmc_amd import Dynamics, distributions as D, layers, sample_chain

dev = torch.device("cuda", 0)


def rate(name, dist, d, n, T, variant, M=10, reps=5, eps=0.05):
    torch.manual_seed(0)
    np.random.seed(0)
    dyn = Dynamics(d, dist.get_energy_function(), T=T, eps=eps, net_factory=layers.stq_network(10, head_factor=0.03), device=dev)
    dyn.variant = variant
    x = torch.randn((n, d), device=dev)
    for _ in range(3):
        sample_chain(x, dyn, M, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        x, p, _ = sample_chain(x, dyn, M, seed=1, proposal0=(r + 1) * M)
    e1.record()
    torch.cuda.synchronize()
    print("%-22s d %3d chains %6d variant %3d: %8.2f us / proposal  accept %.2f  %s"
          % (name, d, n, variant, e0.elapsed_time(e1) * 1e3 / (reps * M), float(p.mean()), _ffi.last_kernel()), flush=True)


def main():
    rng = np.random.RandomState(0)

    def dense(d):
        A = rng.randn(d, d) / np.sqrt(d)
        return D.Gaussian(np.zeros(d), A @ A.T + 0.5 * np.eye(d))
    var50 = np.exp(np.linspace(np.log(1e-2), np.log(1e2), 50))
    for n in (4096, 8192):
        rate("ICG-50 f16x2", D.Gaussian(np.zeros(50), np.diag(var50)), 50, n, 10, 4, M=25, eps=0.02)
        rate("ICG-50 f32 MFMA", D.Gaussian(np.zeros(50), np.diag(var50)), 50, n, 10, 204, M=25, eps=0.02)
        rate("Rough Well 50 f16x2", D.RoughWell(50, 0.1, easy=True), 50, n, 10, 4, M=25)
    for d in (8, 16):
        rate("Rough Well f16x2", D.RoughWell(d, 0.1, easy=True), d, 16384, 10, 0, eps=0.1)
        rate("Rough Well f32 MFMA", D.RoughWell(d, 0.1, easy=True), d, 16384, 10, 200, eps=0.1)
        rate("diag Gaussian f16x2", D.Gaussian(np.zeros(d), np.diag(np.linspace(0.5, 2.0, d))), d, 16384, 10, 0, eps=0.1)
        rate("dense Gaussian", dense(d), d, 16384, 10, 0, eps=0.1)
    rate("GMM 2 comps", D.GMM([rng.randn(12) for _ in range(2)], [np.eye(12)] * 2, [0.5, 0.5]), 12, 16384, 10, 0, eps=0.1)
    rate("funnel", D.GaussianFunnel(10), 10, 16384, 10, 0, eps=0.05)
    rate("dense Gaussian", dense(50), 50, 4096, 10, 0, eps=0.03)
    rate("GMM 2 comps", D.GMM([rng.randn(50) for _ in range(2)], [np.eye(50)] * 2, [0.5, 0.5]), 50, 4096, 10, 0)
    for d in (32, 128):
        rate("Rough Well f16x2", D.RoughWell(d, 0.1, easy=True), d, 16384, 10, 0, eps=0.03)
        rate("dense Gaussian", dense(d), d, 16384, 10, 0, eps=0.03)


if __name__ == "__main__":
    main()
