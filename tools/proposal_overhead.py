#!/usr/bin/env python
"""Per-proposal fixed cost of the persistent sampler loop (GPU box): time of an M-proposal launch
for n_steps in {0, 1, 2, 5, 10} leapfrog steps per proposal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from l2hmc_amd import Dynamics, distributions, layers
from oracle import l2hmc_oracle as O

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
prob = bench.make_problem(0, chains, dev)
dyn = Dynamics(bench.D, distributions.Gaussian(np.zeros(bench.D), np.diag(prob["var"])).get_energy_function(),
               T=bench.T, eps=0.1, net_factory=layers.stq_network(bench.H), device=dev)
dyn.mask = prob["mask"]
dyn.variant = 4
with torch.no_grad():
    for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
        for k in O.NET_KEYS:
            w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
x = torch.as_tensor(prob["x0"], device=dev)
for ns in (0, 1, 2, 5, 10):
    def go():
        return dyn.run(x, None, 0, ns, want=("p", "x_next"), n_proposals=M, rng={"seed": 1})
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 10
    print("n_steps %2d: %8.1f us per %d-proposal launch = %.2f us per proposal" % (ns, t, M, t / M))
