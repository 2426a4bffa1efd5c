#!/usr/bin/env python
"""Throughput of the split engine on BASELINE.json config 5 shapes (GPU box): latent 50, H=200 nets
with the image branch, decoder 50->1024->1024->784, Lf=5, random weights (no checkpoint offline)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from l2hmc_amd import _ffi, propose
if os.environ.get("L2HMC_LIB"):          # kernel experiments: an alternative build of the library
    _ffi.LIB_PATH = os.path.abspath(os.environ["L2HMC_LIB"])
from tests.helpers import hip_dynamics, synthetic_vae_case, to_dev

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = synthetic_vae_case(N=N, seed=0)
dyn = hip_dynamics(g)
dyn.gemm_mode = int(sys.argv[2]) if len(sys.argv) > 2 else dyn.gemm_mode      # 0 = f32-input MFMA, 1 = bf16x3
dyn.split_streams = int(os.environ.get("L2HMC_SPLIT_STREAMS", dyn.split_streams))   # 2 = two half-batches on two streams (round 6), 1 = off
x, aux = to_dev(g["x"]), to_dev(g["aux"])
gen = torch.Generator(device="cuda").manual_seed(0)
dyn.generator = gen
for _ in range(3):
    _, _, px, out = propose(x, dyn, do_mh_step=True, aux=aux)
    x = out[0]
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    _, _, px, out = propose(x, dyn, do_mh_step=True, aux=aux)
    x = out[0]
torch.cuda.synchronize()
el = time.perf_counter() - t0
T = int(g["T"])
flops = 4 * 2 * (50 * 200 * 2 + 200 * 200 + 200 * 150) + (1 + 1.0 / T) * 2 * 2 * (50 * 1024 + 1024 * 1024 + 1024 * 784)
print("gemm_mode %d streams %d | " % (dyn.gemm_mode, dyn.split_streams), end="")
print("config 5, %d chains: %.2f ms per proposal (Lf=%d) = %.3e chain-leapfrog-steps/s; ~%.1f TFLOP/s of %.2e flop/chain-step; mean accept %.3f"
      % (N, 1e3 * el / K, T, N * T * K / el, N * T * K / el * flops / 1e12, flops, float(px.mean())))
