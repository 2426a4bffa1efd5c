#!/usr/bin/env python
"""Time of one trajectory launch vs the number of leapfrog steps (GPU box): the intercept is the
per-launch cost (launch + LDS staging + first gradient + epilogue), the slope the per-step cost."""
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
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
prob = bench.make_problem(0, chains, dev)
dyn = Dynamics(bench.D, distributions.Gaussian(np.zeros(bench.D), np.diag(prob["var"])).get_energy_function(),
               T=bench.T, eps=0.1, net_factory=layers.stq_network(bench.H), device=dev)
dyn.mask = prob["mask"]
dyn.variant = variant
with torch.no_grad():
    for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
        for k in O.NET_KEYS:
            w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
x = torch.as_tensor(prob["x0"], device=dev)
v = torch.randn_like(x)
for ns in (0, 1, 2, 5, 10):
    for _ in range(5):
        dyn.run(x, v, 0, ns, direction_all=1, want=("x", "v", "p"))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        dyn.run(x, v, 0, ns, direction_all=1, want=("x", "v", "p"))
    e1.record()
    torch.cuda.synchronize()
    print("n_steps %2d: %.1f us per launch" % (ns, e0.elapsed_time(e1) * 1e3 / 100))
