#!/usr/bin/env python
"""A target that is NOT in utils/distributions.py -- a banana-shaped density written as a plain torch callable -- sampled
and trained on exactly as the reference's `Dynamics(x_dim, energy_function, ...)` takes any TensorFlow energy
(utils/dynamics.py:37,57,203-218).  U / grad U (and, for training, Hessian-vector products) come from autograd through the
callable between kernel launches -- the slow path by construction --, the S/T/Q nets, the leapfrog half-updates, the
log-determinant, the accept probability and the MH select run on the library's HIP kernels.

    python examples/user_energy.py [--steps 300] [--chains 512]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import Dynamics, func_utils, layers, sample_chain  # noqa: E402
from l2hmc_amd.training import Trainer  # noqa: E402


def banana(x, b=0.3, s=2.0):
    """U(x) = x0^2 / (2 s^2) + 1/2 sum_{k >= 1} (x_k + b x0^2 - s^2 b)^2"""
    t = x[:, 1:] + b * x[:, :1] ** 2 - s * s * b
    return 0.5 * x[:, 0] ** 2 / (s * s) + 0.5 * (t * t).sum(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--chains", type=int, default=512)
    ap.add_argument("--dim", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    dev = torch.device("cuda", 0)
    d, N = args.dim, args.chains

    dyn = Dynamics(d, banana, T=10, eps=0.1, net_factory=layers.stq_network(10), device=dev)   # a plain callable drops in
    dyn.generator = torch.Generator(device=dev).manual_seed(args.seed)
    trainer = Trainer(dyn, seed=args.seed)        # picks the GEMM-engine trainer: a user energy has no fused kernel
    x = torch.randn((N, d), device=dev, generator=dyn.generator)
    t0 = time.perf_counter()
    for step in range(args.steps):
        loss, px, x, lr = trainer.step(x)
        if step % max(1, args.steps // 6) == 0 or step == args.steps - 1:
            print("step %4d  loss %10.4e  accept %.3f  eps %.4f" % (step, float(loss), float(px.mean()),
                                                                   float(torch.exp(dyn.alpha.detach()))))
    torch.cuda.synchronize()
    print("%.1f ms per training step (U, grad U and Hessian-vector products by autograd)" %
          (1e3 * (time.perf_counter() - t0) / max(1, args.steps)))

    # sample with the trained sampler and with HMC at the same step count; ESS from the device autocovariance
    M = 400
    for name, sampler in (("L2HMC (trained)", dyn),
                          ("HMC eps=0.25", Dynamics(d, banana, T=10, eps=0.25, hmc=True, device=dev))):
        xf, p, hist = sample_chain(x, sampler, M, seed=1, record=True)
        X = torch.cat([x[None], hist[:-1]], dim=0)
        Xc = X - X.mean(dim=(0, 1), keepdim=True)
        scale = float(torch.sqrt((Xc * Xc).sum(2).mean()))
        ess = float(func_utils.ESS(func_utils.acl_spectrum(Xc, scale)))
        print("%-16s accept %.3f  ESS per MH step %.3e" % (name, float(p.mean()), ess))


if __name__ == "__main__":
    main()
