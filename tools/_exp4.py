import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
from l2hmc_amd import Dynamics, distributions as D, layers, sample_chain
dev = torch.device("cuda", 0)
def rate(name, dist, d, n, T, variant, M=10, reps=10, eps=0.1):
    torch.manual_seed(0); np.random.seed(0)
    dyn = Dynamics(d, dist.get_energy_function(), T=T, eps=eps, net_factory=layers.stq_network(10, head_factor=0.03), device=dev)
    dyn.variant = variant
    x = torch.randn((n, d), device=dev)
    for _ in range(5): sample_chain(x, dyn, M, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): x, p, _ = sample_chain(x, dyn, M, seed=1, proposal0=(r + 1) * M)
    e1.record(); torch.cuda.synchronize()
    print("%-16s d %3d chains %6d: %8.2f us / proposal  accept %.2f  x-sum %.6e  %s" % (name, d, n, e0.elapsed_time(e1) * 1e3 / (reps * M), float(p.mean()), float(x.double().sum()), _ffi.last_kernel()), flush=True)
cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
mog = D.GMM([np.array([2., 0.]), np.array([-2., 0.])], [0.1 * np.eye(2)] * 2, [0.5, 0.5])
rate("SCG-2D", D.Gaussian(np.zeros(2), cov), 2, 200, 10, 0, M=25)
rate("SCG-2D", D.Gaussian(np.zeros(2), cov), 2, 16384, 10, 0)
rate("MoG-2D", mog, 2, 8192, 25, 0)
rate("MoG-2D", mog, 2, 16384, 25, 0)
rate("RoughWell", D.RoughWell(2, 0.1, easy=True), 2, 16384, 10, 0)
rate("RoughWell d=4", D.RoughWell(4, 0.1, easy=True), 4, 16384, 10, 0)
rate("diag d=3", D.Gaussian(np.zeros(3), np.diag([1., 4., 0.25])), 3, 16384, 10, 0)
