#!/usr/bin/env python
"""The reference's only documented example, SCGExperiment.ipynb, end to end on MI355X:
strongly-correlated Gaussian (raw lines 103-108), S/T/Q nets with H=10 (51-78), 5000 Adam steps
on 200 chains (156-181, 254-271), 2000 sampling steps (288-298), HMC baselines (317-319),
autocorrelation spectrum and ESS (330-334, 393).  The notebook's recorded outcome (raw 200-249,
388): acceptance 0.44-0.49 late in training, ESS L2HMC 2.61e-1, ESS HMC(0.15) 5.63e-3, ratio 46.

    python examples/scg_experiment.py [--steps 5000] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import Dynamics, distributions, func_utils, layers, sample_chain  # noqa: E402
from l2hmc_amd.training import Trainer  # noqa: E402


# the notebook's `network` factory (SCGExperiment.ipynb raw 51-78; H = 10, head factor 0.001)
network = layers.stq_network(10)


def ess_of(x0, hist, scale):
    X = torch.cat([x0[None], hist[:-1]], dim=0)        # states BEFORE each step, like raw 291-298
    return float(func_utils.ESS(func_utils.acl_spectrum(X, scale)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--samples", type=int, default=200)
    ap.add_argument("--eval-steps", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    dev = torch.device("cuda", 0)

    x_dim = 2
    mu = np.zeros(2,)
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    distribution = distributions.Gaussian(mu, cov)
    dynamics = Dynamics(x_dim, distribution.get_energy_function(), T=10, eps=0.1, net_factory=network)
    dynamics.generator = torch.Generator(device=dev).manual_seed(args.seed)
    trainer = Trainer(dynamics, seed=args.seed)        # Adam, lr 1e-3 * 0.96 ** floor(step / 1000)

    samples = torch.randn(args.samples, x_dim, device=dev, generator=dynamics.generator)
    t0 = time.perf_counter()
    for t in range(args.steps):
        loss, px, samples, lr = trainer.step(samples)
        if t % 100 == 0:
            print('Step: %d / %d, Loss: %.2e, Acceptance sample: %.2f, LR: %.5f' % (
                t, args.steps, float(loss), float(px.mean()), lr))
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    print('training: %.1f s (%.2f ms per step), eps = %.4f' % (t_train, 1e3 * t_train / max(args.steps, 1),
                                                               float(dynamics.eps)))

    x0 = torch.as_tensor(distribution.get_samples(n=args.samples), dtype=torch.float32, device=dev)
    scale = float(np.sqrt(np.trace(cov)))
    _, p_l2, hist = sample_chain(x0, dynamics, args.eval_steps, record=True)
    ess_l2hmc = ess_of(x0, hist, scale)
    ess_hmc = {}
    for eps in (0.1, 0.15, 0.2):
        hmc = Dynamics(x_dim, distribution.get_energy_function(), T=10, eps=eps, hmc=True)
        hmc.generator = dynamics.generator
        _, p_h, h = sample_chain(hist[-1], hmc, args.eval_steps, record=True)
        ess_hmc[eps] = ess_of(hist[-1], h, scale)
    print('mean accept: L2HMC %.3f' % float(p_l2.mean()))
    print('ESS L2HMC: %.2e -- ESS HMC: %.2e -- Ratio: %d   (notebook: 2.61e-01 -- 5.63e-03 -- 46)' % (
        ess_l2hmc, ess_hmc[0.15], ess_l2hmc / ess_hmc[0.15]))
    print('ESS HMC eps=0.1: %.2e, eps=0.2: %.2e' % (ess_hmc[0.1], ess_hmc[0.2]))
    return ess_l2hmc, ess_hmc


if __name__ == "__main__":
    main()
