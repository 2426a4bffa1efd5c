#!/usr/bin/env python
"""Time the bench workload with an alternative build of the library (GPU box):
    python tools/time_lib.py <path/to/lib.so> <chains> <proposals_per_launch>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
import torch
import bench
from l2hmc_amd import Dynamics, distributions, layers
from oracle import l2hmc_oracle as O

n, M = int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
prob = bench.make_problem(0, n, dev)
dyn = Dynamics(bench.D, distributions.Gaussian(np.zeros(bench.D), np.diag(prob["var"])).get_energy_function(),
               T=bench.T, eps=0.1, net_factory=layers.stq_network(bench.H), device=dev)
dyn.mask = prob["mask"]
dyn.variant = int(os.environ.get("L2HMC_VARIANT", "4"))
with torch.no_grad():
    for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
        for k in O.NET_KEYS:
            w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
x = torch.as_tensor(prob["x0"], device=dev)
go = lambda: dyn.run(x, None, 0, bench.T, want=("p", "x_next"), n_proposals=M, rng={"seed": 1})
for _ in range(3):
    go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = max(2, 200 // M)
e0.record()
for _ in range(R):
    go()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / R / M
print("%s: chains %d M %d: %.2f us / proposal = %.3e steps/s" % (os.path.basename(sys.argv[1]), n, M, t * 1e6, n * bench.T / t))
