import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from l2hmc_amd import Dynamics, distributions as D, layers, propose
dev = torch.device('cuda', 0)
rng = np.random.RandomState(0)
def run(name, dist, d, N, split, variant=0):
    dyn = Dynamics(d, dist.get_energy_function(), T=10, eps=0.05, net_factory=layers.stq_network(10), device=dev)
    dyn.generator = torch.Generator(device=dev).manual_seed(0)
    if split: dyn._split = True
    dyn.variant = variant
    x = torch.as_tensor(rng.randn(N, d).astype(np.float32), device=dev)
    for _ in range(2):
        _, _, px, out = propose(x, dyn, do_mh_step=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 5
    for _ in range(K):
        _, _, px, out = propose(x, dyn, do_mh_step=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%s d=%d N=%d %s: %.3f ms / proposal = %.3e chain-steps/s, accept %.3f" % (name, d, N, "split" if split else "fused variant %d" % variant, dt * 1e3, N * 10 / dt, float(px.mean())), flush=True)
for d in (160, 192, 256, 384, 512):
    R = np.linalg.qr(rng.randn(d, d))[0]
    cov = (R.T * np.exp(rng.uniform(-1, 1, size=d))) @ R
    g = D.Gaussian(np.zeros(d), cov)
    for N in (4096, 16384):
        for split, variant in ((False, 4), (False, 8), (True, 0)):
            try: run("dense", g, d, N, split, variant)
            except Exception as e: print("dense d=%d N=%d split=%s failed: %r" % (d, N, split, e))
