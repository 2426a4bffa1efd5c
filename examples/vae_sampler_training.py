#!/usr/bin/env python
"""The sampler's share of the reference's VAE experiment (mnist_vae.py) on MI355X: the decoder-posterior target
(mnist_vae.py:104-126), the H = 200 S/T/Q nets with the shared image branch (:128-178) and the sampler update of the
training loop (:185-262: MH chained proposals from the encoder's sample, sampler_loss, global-norm clipping, Adam) --
every step on the GEMM engine of libl2hmc_hip.so (`Trainer(dynamics).sampler_step`).

There is no MNIST and no checkpoint offline, and the VAE's own encoder / decoder optimisers are ordinary dense-net
training outside the hot path, so this script uses stand-ins for the two things the sampler update only READS: a fixed
decoder (random weights, output layer scaled so that the posterior differs from the prior) and an "encoder" that
returns mu = 0, log_sigma = -0.3 for every image; images are Bernoulli(0.13) rows.

    python examples/vae_sampler_training.py [--steps 200] [--batch 512] [--MH 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import Dynamics, propose, vae  # noqa: E402
from l2hmc_amd.training import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=512)          # hps.batch_size
    ap.add_argument("--MH", type=int, default=5)               # hps.MH
    ap.add_argument("--leapfrogs", type=int, default=5)        # hps.leapfrogs
    ap.add_argument("--latent", type=int, default=50)          # hps.latent_dim
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    dev = torch.device("cuda", 0)
    d, N = args.latent, args.batch

    decoder = vae.make_decoder(d, 1024, 784)                   # mnist_vae.py:104-111
    with torch.no_grad():
        decoder.layers[4].W.mul_(30.0)                         # stand-in for a trained output layer
    encoder_sampler = vae.make_encoder_sampler(784, 512, 200)  # :134-140
    dynamics = Dynamics(d, vae.VAEPosterior(decoder).get_energy_function(), T=args.leapfrogs, eps=0.1,
                        net_factory=vae.sampler_net_factory(d, encoder_sampler, 200, 200))    # :142-178
    dynamics.generator = torch.Generator(device=dev).manual_seed(args.seed)
    trainer = Trainer(dynamics, lr=1e-3, decay_steps=0)        # piecewise-constant rate, first piece (:247)
    gen = dynamics.generator

    def batch():
        inp = (torch.rand((N, 784), device=dev, generator=gen) < 0.13).float()
        log_sigma = torch.full((N, d), -0.3, device=dev)
        latent_q = torch.randn((N, d), device=dev, generator=gen) * torch.exp(log_sigma)      # mu + noise * sigma, :119
        return inp, latent_q, log_sigma

    def jump_and_accept(k=4):
        j = a = 0.0
        for _ in range(k):
            inp, z, _ = batch()
            _, _, px, out = propose(z, dynamics, do_mh_step=True, aux=inp)
            j += float(((out[0] - z) ** 2).sum(1).mean()) / k
            a += float(px.mean()) / k
        return j, a
    j0, a0 = jump_and_accept()
    t0 = time.perf_counter()
    for t in range(args.steps):
        inp, latent_q, log_sigma = batch()
        loss, latent_T, px, lr = trainer.sampler_step(latent_q, inp, log_sigma, MH=args.MH)
        if t % 50 == 0:
            print('Step:%d/%d::Loss sampler: %.3e:: p_accept: %.3f:: eps: %.4f:: Lr: %g' % (
                t, args.steps, float(loss), float(px.mean()), float(torch.exp(dynamics.alpha.detach())), lr))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    j1, a1 = jump_and_accept()
    print("%d sampler updates (batch %d, MH %d, Lf %d) in %.2f s = %.1f ms per update" % (
        args.steps, N, args.MH, args.leapfrogs, el, 1e3 * el / args.steps))
    print("mean squared jump per proposal %.3f -> %.3f, acceptance %.3f -> %.3f" % (j0, j1, a0, a1))
    assert bool(torch.isfinite(trainer.theta).all())


if __name__ == "__main__":
    main()
