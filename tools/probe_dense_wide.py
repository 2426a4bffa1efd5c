import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from l2hmc_amd import Dynamics, distributions as D, layers, propose
dev = torch.device('cuda', 0)
rng = np.random.RandomState(0)
def run(name, dist, d, N, split, variant=0, x0=None):
    dyn = Dynamics(d, dist.get_energy_function(), T=10, eps=0.05, net_factory=layers.stq_network(10), device=dev)
    dyn.generator = torch.Generator(device=dev).manual_seed(0)
    if split: dyn._split = True
    dyn.variant = variant
    x = torch.as_tensor((rng.randn(N, d) if x0 is None else x0(N)).astype(np.float32), device=dev)
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

for d, K in ((192, 2), (320, 4)):       # ((2 pi)^d overflows a double from d = 386: the reference formula ends there)
    mus = [rng.randn(d) * 0.3 for _ in range(K)]
    covs = []
    for _ in range(K):
        R = np.linalg.qr(rng.randn(d, d))[0]
        # (the reference keeps pi_i / sqrt((2 pi)^d det Sigma_i) in float32: it exists at this d only for variances around 1 / (2 pi))
        covs.append((R.T * np.exp(rng.uniform(np.log(0.1), np.log(0.25), size=d))) @ R)
    pis = [1.0 / K] * K
    pis[0] += 1 - sum(pis)
    mog = D.GMM(mus, covs, pis)
    for N in (4096, 16384):
        for variant in (4, 8):
            try: run("gmm%d" % K, mog, d, N, False, variant, x0=lambda n: np.asarray(mus)[rng.randint(0, K, n)] + 0.4 * rng.randn(n, d))
            except Exception as e: print("gmm d=%d N=%d variant=%s failed: %r" % (d, N, variant, e))
