#!/usr/bin/env python
"""Register-resident XNet weights in the one-chain-per-lane kernel (traj_lane.hpp, RES = 1) against the scalar-load form
(L2HMC_LANE_RES=0), d = 2 targets:   python tools/bench_lane_resident.py   (GPU box; profiles/r06_lane_resident.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l2hmc_amd import Dynamics, distributions as D, layers, sample_chain, _ffi


def rate(dist, d, n, T, res, M=10, reps=5):
    os.environ["L2HMC_LANE_RES"] = str(int(res))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    dyn = Dynamics(d, dist.get_energy_function(), T=T, eps=0.1, net_factory=layers.stq_network(10, head_factor=0.03), device=dev)
    dyn.variant = 32
    x0 = torch.randn((n, d), device=dev)
    x = x0
    for _ in range(2):
        sample_chain(x, dyn, M, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        x, p, _ = sample_chain(x, dyn, M, seed=1, proposal0=(r + 1) * M)
    e1.record()
    torch.cuda.synchronize()
    return n * T * reps * M / (e0.elapsed_time(e1) * 1e-3), x, p, _ffi.last_kernel()


def main():
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    targets = [("SCG-2D dense", D.Gaussian(np.zeros(2), cov), 2, 10),
               ("MoG-2D", D.GMM([np.array([2., 0.]), np.array([-2., 0.])], [0.1 * np.eye(2)] * 2, [0.5, 0.5]), 2, 25),
               ("RoughWell d=2", D.RoughWell(2, 0.1, easy=True), 2, 10),
               ("diag Gauss d=2", D.Gaussian(np.zeros(2), np.diag([1.0, 0.01])), 2, 10)]
    for name, dist, d, T in targets:
        for n in (16384, 32768, 65536, 131072, 262144):
            a, xa, pa, ka = rate(dist, d, n, T, 0)
            b, xb, pb, kb = rate(dist, d, n, T, 1)
            c, xc, pc, kc = rate(dist, d, n, T, 2)
            same = bool(torch.equal(xa, xb)) and bool(torch.equal(pa, pb)) and bool(torch.equal(xa, xc)) and bool(torch.equal(pa, pc))
            print("%-14s chains %7d: scalar loads %.3e   XNet pairs resident %.3e (x%.2f)   all weights, 4 per VGPR %.3e (x%.2f)   bit-identical %s   [%s]"
                  % (name, n, a, b, b / a, c, c / a, same, kc), flush=True)


if __name__ == "__main__":
    main()
