#!/usr/bin/env python
"""BASELINE.json config 4 names an "fp32 vs bf16" series for the Rough-Well sweep.  Arithmetic here is fp32
throughout (parity with the reference is the first gate), so bf16 could only narrow the (chain, dim) STATE
at the HBM boundary.  This tool measures whether that boundary matters (GPU box):

  * fused trajectory (the product path): state bytes move once per T = 10 steps;
  * per-step launches (`n_steps = 1`, the pattern north_star's "per-step kernel" unit assumes): x, v in and
    x', v' out every step -- the pattern a narrower state would help most.

For each d it reports the measured time per leapfrog step of 16 384 chains, the state bytes that pattern
moves per step in fp32, the time those bytes take at the measured streaming rate of this GPU (6.3 TB/s,
MI355X_MICROARCH.md), and the UPPER BOUND on the speed-up bf16 state could give (halving exactly those
bytes at zero conversion cost)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l2hmc_amd import Dynamics, distributions as D, layers

HBM = 6.3e12
N, T = 16384, 10


def main():
    dev = torch.device("cuda", 0)
    print("Rough Well (easy, eta = 0.1), %d chains, Lf = %d, H = 10; fp32 state; times are HIP-event means" % (N, T))
    print("%5s | %28s | %40s" % ("d", "fused trajectory (product)", "per-step launches (n_steps = 1)"))
    print("%5s | %9s %9s %8s | %9s %9s %9s %10s" % ("", "us/step", "HBM us", "bf16 max", "us/step", "HBM us", "HBM frac", "bf16 max"))
    for d in (2, 8, 32, 50, 128, 256, 512):
        torch.manual_seed(0)
        np.random.seed(0)
        dyn = Dynamics(d, D.RoughWell(d, 0.1, easy=True).get_energy_function(), T=T, eps=0.1,
                       net_factory=layers.stq_network(10, head_factor=0.03), device=dev)
        x = torch.randn(N, d, device=dev)
        v = torch.randn(N, d, device=dev)
        dr = torch.randint(0, 2, (N,), device=dev, dtype=torch.uint8)

        def timed(fn, reps):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps          # us

        t_fused = timed(lambda: dyn.run(x, v, 0, T, direction=dr, want=("x", "v", "logjac")), 20) / T
        t_step = timed(lambda: dyn.run(x, v, 3, 1, direction=dr, want=("x", "v", "logjac")), 50)
        b_step = 4.0 * (4 * d + 2) * N                        # x, v in; x', v' out; logjac; dir      
        b_fused = b_step / T
        h_step, h_fused = b_step / HBM * 1e6, b_fused / HBM * 1e6
        print("%5d | %9.2f %9.3f %7.2f%% | %9.2f %9.3f %8.1f%% %9.2f%%"
              % (d, t_fused, h_fused, 100 * 0.5 * h_fused / t_fused, t_step, h_step, 100 * h_step / t_step,
                 100 * 0.5 * h_step / t_step))
    print("bf16 max = share of the step time that halving the state bytes could remove if their transfer were fully "
          "exposed (it is not: loads are issued one proposal ahead).  Even per-step launches at d = 512 stay below the "
          "HBM roof; the fused product path moves 1/T of those bytes.  A bf16 state would also forfeit the 1e-4 "
          "accept-probability parity (bf16 has 8 bits of mantissa: |dx| ~ 4e-3 |x| per round trip).")


if __name__ == "__main__":
    main()
