#!/usr/bin/env python
"""Time of one sampler-training step on BASELINE.json config 5 shapes (GPU box): latent 50, H = 200 nets with the shared
image branch, decoder 50 -> 1024 -> 1024 -> 784, Lf = 5; mnist_vae.py:185-262's sampler objective (MH chained proposals,
clipped Adam) on the GEMM engine (`l2hmc_train_split_grad`).  Random weights (no checkpoint / MNIST offline).

  python tools/bench_vae_train.py [chains=8192] [MH=1] [net_mode=0]     (net_mode 1: three products per net evaluation, the round-4 form)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from l2hmc_amd.training import Trainer
from tests.helpers import hip_dynamics, synthetic_vae_case, to_dev

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
MH = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = synthetic_vae_case(N=N, seed=0)
dyn = hip_dynamics(g)
dyn.eps_override = None
dyn.net_mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dyn.generator = torch.Generator(device="cuda").manual_seed(0)
tr = Trainer(dyn, decay_steps=0)
x, aux = to_dev(g["x"]), to_dev(g["aux"])
log_sigma = to_dev((0.3 * np.random.RandomState(1).randn(N, 50) - 0.5).astype(np.float32))
for _ in range(2):
    loss, xT, px, lr = tr.sampler_step(x, aux, log_sigma, MH=MH)
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    loss, xT, px, lr = tr.sampler_step(x, aux, log_sigma, MH=MH)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / K
T = int(g["T"])
d, H = 50, 200
f_net = 2 * (2 * d * H + H * H + 3 * d * H)
f_dec = 2 * (50 * 1024 + 1024 * 1024 + 1024 * 784)
# algorithmic flops per chain of ONE differentiated proposal: forward (4 T net evaluations, T + 1 energy gradients =
# forward + reverse through the decoder), reverse sweep (input + weight gradients of every net evaluation = 2 x forward,
# one Hessian-vector product = tangent forward + tangent reverse per distinct point)
flops = 4 * T * f_net + (T + 1) * 2 * f_dec + 4 * T * 2 * f_net + (T + 1) * 2 * f_dec
n_diff = 1 if MH == 1 else MH                 # proposals differentiated (stop_gradient off)
fwd_only = 0 if MH == 1 else (MH - 1) * (4 * T * f_net + (T + 1) * 2 * f_dec)
tot = N * (n_diff * flops + fwd_only)
print("config 5 sampler training, %d chains, MH = %d, net_mode %d: %.2f ms per step = %.1f TFLOP/s algorithmic (%.3g flop per chain); "
      "loss %.4e, mean accept %.3f, workspace %.2f GB"
      % (N, MH, dyn.net_mode, 1e3 * el, tot / el / 1e12, tot / N, float(loss), float(px.mean()), tr._ws.numel() * 4 / 2 ** 30))
