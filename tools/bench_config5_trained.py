#!/usr/bin/env python
"""BASELINE.json config 5 AS WORDED: "MNIST-VAE latent posterior (mnist_vae.py) d=50, 8 192 chains, TRAINED sampler".

(i)  trains the sampler's variables (XNet, VNet, the shared image branch, eps) with the reference's sampler update --
     `SplitTrainer.sampler_step`: MH = 5 chained proposals from the encoder's sample, sampler_loss, global-norm clipping
     at 5, Adam at 1e-3 (mnist_vae.py:185-262) -- on batches of 512 chains (hps.batch_size), all on the GEMM engine;
(ii) then measures, on the decoder posterior of 8 192 chains (64 images x 128 chains each, Lf = 5), what
     eval_sampler.py:145-204 measures for its one image x 200 chains: MH steps/s, accept probability, the autocovariance
     of the centred chains and the ESS per MH step (utils/func_utils.py:45-54,114-120) -- for the TRAINED L2HMC sampler
     (plain proposals, and eval_sampler.py:161-162's chain_operator with nb_steps ~ U{1..3}) and for HMC at the reference's
     step-size grid (eval_sampler.py:185: 0.05 ... 0.175) -- history and autocovariance on the device.

No MNIST and no checkpoint exist offline: the decoder is a fixed random one (output layer scaled so the posterior differs
from the prior), the VAE encoder is the stand-in mu = 0, log_sigma = -0.3, images are Bernoulli(0.13) rows -- the sampler
is really trained, the model it samples is synthetic.  Prints one JSON object.

    python tools/bench_config5_trained.py [--updates 300] [--chains 8192] [--mh-steps 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(dev, updates=300, chains=8192, mh_steps=300, burn=100, train_batch=512, seed=0, hmc_eps=(0.05, 0.075, 0.1, 0.125, 0.15, 0.175),
        checkpoints=()):
    from l2hmc_amd import Dynamics, chain_operator, func_utils, propose, vae
    from l2hmc_amd.training import Trainer
    d, H, T = 50, 200, 5
    torch.manual_seed(seed)
    np.random.seed(seed)
    decoder = vae.make_decoder(d, 1024, 784)
    with torch.no_grad():
        decoder.layers[4].W.mul_(30.0)                         # stand-in for a trained output layer
    enc = vae.make_encoder_sampler(784, 512, H)
    energy = vae.VAEPosterior(decoder).get_energy_function()
    dyn = Dynamics(d, energy, T=T, eps=0.1, net_factory=vae.sampler_net_factory(d, enc, H, H), device=dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    dyn.generator = gen
    trainer = Trainer(dyn, lr=1e-3, decay_steps=0, seed=seed)
    log_sigma_v = -0.3

    def batch(n):
        inp = (torch.rand((n, 784), device=dev, generator=gen) < 0.13).float()
        ls = torch.full((n, d), log_sigma_v, device=dev)
        return inp, torch.randn((n, d), device=dev, generator=gen) * torch.exp(ls), ls

    # ---- (i) train (in stages when `checkpoints` asks for the ESS along the way) -----------------------------------------
    trace = []
    state = {"t": 0, "seconds": 0.0}

    def train_to(target):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        while state["t"] < target:
            t = state["t"]
            inp, zq, ls = batch(train_batch)
            loss, _, px, _ = trainer.sampler_step(zq, inp, ls, MH=5)
            if t % max(1, updates // 6) == 0 or t == updates - 1:
                trace.append({"update": t, "loss": float(loss), "accept": float(px.mean()), "eps": float(torch.exp(dyn.alpha.detach()))})
            state["t"] = t + 1
        torch.cuda.synchronize(dev)
        state["seconds"] += time.perf_counter() - t0

    # ---- (ii) measure on 64 images x (chains / 64) chains ----------------------------------------------------------------
    n_img = 64
    per = chains // n_img
    imgs = (torch.rand((n_img, 784), device=dev, generator=gen) < 0.13).float()
    aux = imgs.repeat_interleave(per, dim=0).contiguous()
    z0 = torch.randn((chains, d), device=dev, generator=gen) * float(np.exp(log_sigma_v))

    def measure(step_fn, label):
        hist = torch.empty((mh_steps, chains, d), dtype=torch.float32, device=dev)
        z, acc = z0, 0.0
        for _ in range(3):
            z, _ = step_fn(z)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(mh_steps):
            hist[t] = z                                         # eval_sampler.py:176-181 records the state BEFORE the step
            z, p = step_fn(z)
            acc += float(p.mean()) if t % 25 == 0 else 0.0
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / mh_steps
        X = hist[burn:]
        # centre by the per-image posterior mean (eval_sampler.py:183: the mean over time and chains of the one image)
        mu = X.reshape(mh_steps - burn, n_img, per, d).mean(dim=(0, 2), keepdim=True)
        Xc = (X.reshape(mh_steps - burn, n_img, per, d) - mu).reshape(mh_steps - burn, chains, d).contiguous()
        scale = float(torch.sqrt((Xc * Xc).sum(2).mean()))      # A(0) = 1: sqrt of the mean total variance
        A = func_utils.acl_spectrum(Xc, scale)
        return {"sampler": label, "ms_per_mh_step": ms, "mh_steps_per_sec_per_chain": 1e3 / ms,
                "mean_accept_prob": acc / len(range(0, mh_steps, 25)), "ess_per_mh_step": float(func_utils.ESS(A)),
                "ess_per_sec": float(func_utils.ESS(A)) * chains * 1e3 / ms, "autocov_lag1": float(A[1]),
                "autocov_lag10": float(A[10]), "state_finite": bool(torch.isfinite(z).all())}

    def l2hmc_step(z):
        _, _, px, out = propose(z, dyn, do_mh_step=True, aux=aux)
        return out[0], px

    host_rng = np.random.RandomState(seed)

    def l2hmc_cs_step(z):                                       # eval_sampler.py:161-162
        _, _, p, out = chain_operator(z, dyn, int(host_rng.randint(1, 4)), aux=aux, do_mh_step=True)
        return out[0], p
    along = []
    for cp in sorted(set(int(c) for c in checkpoints if 0 < int(c) < updates)):
        train_to(cp)
        r = measure(l2hmc_step, "L2HMC after %d updates, propose" % cp)
        r.update({"updates": cp, "train_seconds_so_far": state["seconds"]})
        along.append(r)
    train_to(updates)
    t_train = state["seconds"]
    res = [measure(l2hmc_step, "L2HMC trained, propose"), measure(l2hmc_cs_step, "L2HMC trained, chain_operator nb_steps~U{1..3}")]
    for eps in hmc_eps:
        hd = Dynamics(d, energy, T=T, eps=float(eps), hmc=True, device=dev)
        hd.generator = gen

        def hmc_step(z, hd=hd):
            _, _, px, out = propose(z, hd, do_mh_step=True, aux=aux)
            return out[0], px
        res.append(measure(hmc_step, "HMC eps=%.3f" % eps))
    best_hmc = max(res[2:], key=lambda r: r["ess_per_sec"])
    f_net = 2 * (2 * d * H + H * H + 3 * d * H)
    f_dec = 2 * (d * 1024 + 1024 * 1024 + 1024 * 784)
    flops_step = 4 * f_net + (1 + 1.0 / T) * 2 * f_dec
    return {"workload": "config 5 with a TRAINED sampler: VAE latent posterior d=50 (decoder 1024/1024/784, synthetic), H=200 nets + "
                        "image branch, Lf=5; %d sampler updates (batch %d, MH=5, clipped Adam) then %d chains = %d images x %d, "
                        "%d MH steps (%d burn-in)" % (updates, train_batch, chains, n_img, per, mh_steps, burn),
            "train": {"updates": updates, "seconds": t_train, "ms_per_update": 1e3 * t_train / max(1, updates), "trace": trace},
            "samplers": res,
            "along_the_way": along,
            "l2hmc_chain_leapfrog_steps_per_sec": chains * T * 1e3 / res[0]["ms_per_mh_step"],
            "l2hmc_frac_of_fp32_roof": chains * T * flops_step * 1e3 / res[0]["ms_per_mh_step"] / 1e12 / 157.3,
            "best_hmc": best_hmc["sampler"],
            "ess_per_sec_ratio_l2hmc_over_best_hmc": max(res[0]["ess_per_sec"], res[1]["ess_per_sec"]) / max(best_hmc["ess_per_sec"], 1e-30)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--updates", type=int, default=300)
    ap.add_argument("--chains", type=int, default=8192)
    ap.add_argument("--mh-steps", type=int, default=300)
    ap.add_argument("--checkpoints", type=str, default="", help="comma-separated update counts at which the L2HMC ESS is measured too")
    a = ap.parse_args()
    cps = [int(c) for c in a.checkpoints.split(",") if c]
    print(json.dumps(run(torch.device("cuda", 0), updates=a.updates, chains=a.chains, mh_steps=a.mh_steps, checkpoints=cps), indent=1))
