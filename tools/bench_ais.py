#!/usr/bin/env python
"""AIS wall time (GPU box): 4096 chains x 200 anneal steps x 10 leapfrogs on ICG-50, one persistent launch
(utils/ais.py:43-82 fused into the trajectory kernel's proposal loop; round 1 used five launches per anneal step)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from l2hmc_amd import distributions as D
from l2hmc_amd.ais import ais_estimate

d, N, K, T = 50, 4096, 200, 10
var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), d))
init = D.Gaussian(np.zeros(d), np.eye(d)).get_energy_function()
final = D.Gaussian(np.zeros(d), np.diag(var)).get_energy_function()
x0 = torch.randn(N, d, device="cuda")
for _ in range(2):
    est, alpha = ais_estimate(init, final, K, x0, step_size=0.05, leapfrogs=T, x_dim=d, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
est, alpha = ais_estimate(init, final, K, x0, step_size=0.05, leapfrogs=T, x_dim=d, seed=1)
torch.cuda.synchronize()
el = time.perf_counter() - t0
exact = 0.5 * np.sum(np.log(var))        # log(Z_final / Z_init) of the two Gaussians
print("AIS ICG-50: %d chains x %d anneal steps x %d leapfrogs in %.2f ms (1 kernel launch; %.3e chain-leapfrog-steps/s): "
      "AIS estimate of log Z ratio %.3f (a stochastic lower bound; exact %.3f), mean accept %.3f" % (N, K, T, 1e3 * el, N * K * T / el, float(est), exact, float(alpha)))
