#!/usr/bin/env python
"""One chain per lane (variant 32, traj_lane.hpp) against the MFMA-tile kernels the dispatcher would otherwise take
(variant 33) over chain counts and dimensions:   python tools/bench_lane.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l2hmc_amd import Dynamics, distributions as D, layers, sample_chain


def rate(dist, d, n, T, variant, M=10, reps=3):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    dyn = Dynamics(d, dist.get_energy_function(), T=T, eps=0.1, net_factory=layers.stq_network(10, head_factor=0.03), device=dev)
    dyn.variant = variant
    x = torch.randn((n, d), device=dev)
    for _ in range(2):
        sample_chain(x, dyn, M, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        x, p, _ = sample_chain(x, dyn, M, seed=1, proposal0=(r + 1) * M)
    e1.record()
    torch.cuda.synchronize()
    return n * T * reps * M / (e0.elapsed_time(e1) * 1e-3)


def main():
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    targets = [("SCG-2D dense", D.Gaussian(np.zeros(2), cov), 2, 10),
               ("MoG-2D", D.GMM([np.array([2., 0.]), np.array([-2., 0.])], [0.1 * np.eye(2)] * 2, [0.5, 0.5]), 2, 25),
               ("RoughWell d=4", D.RoughWell(4, 0.1, easy=True), 4, 10)]
    for name, dist, d, T in targets:
        for n in (16384, 32768, 65536, 131072, 262144):
            a = rate(dist, d, n, T, 100)      # the general MFMA kernel family's automatic choice is variant 0; 100 = round-1 form
            b = rate(dist, d, n, T, 32)
            c = rate(dist, d, n, T, 33)       # the automatic choice among the MFMA kernels
            print("%-14s chains %7d: tiles (auto, lane off) %.3e   lane %.3e   (x%.2f)   [general kernel %.3e]" % (name, n, c, b, b / c, a), flush=True)


if __name__ == "__main__":
    main()
