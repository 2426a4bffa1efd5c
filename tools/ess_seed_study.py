#!/usr/bin/env python
"""SCG-2D training as bench.py's ESS leg runs it (SCGExperiment.ipynb raw 156-181, 254-271: 200 chains, Lf = 10, 5000 Adam steps),
looked at per seed (GPU box) -- the study behind README's ESS paragraph (VERDICT r04 "weak" #3: one training in ten ended in a
sampler that mixes worse than plain HMC).

    python tools/ess_seed_study.py sweep 30            # 30 independent trainings: ESS / MH step, lag-1 autocovariance, accept, eps
    python tools/ess_seed_study.py replay 7 [steps]    # seed 7 again, every random draw recorded, and the SAME training in float64
                                                       # numpy (oracle/l2hmc_train_oracle.py + TF1's Adam) on those draws:
                                                       # parameter distance along the way, ESS of both trained samplers

The oracle is the checker here (test infrastructure); nothing of it runs in the product path."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from l2hmc_amd import Dynamics, _ffi, distributions, func_utils, layers, sample_chain     # noqa: E402
from l2hmc_amd.training import Trainer, _SHAPES                                             # noqa: E402

DEV = torch.device("cuda:0")
COV = np.array([[50.05, -49.95], [-49.95, 50.05]])
SCALE = float(np.sqrt(np.trace(COV)))
N, STEPS = 200, 2000
DIST = distributions.Gaussian(np.zeros(2), COV)
X0 = torch.as_tensor(DIST.get_samples(N, rng=np.random.RandomState(0)), dtype=torch.float32, device=DEV)


def make(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    layers.set_default_device(DEV)
    dyn = Dynamics(2, DIST.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10), device=DEV)
    dyn.generator = gen
    tr = Trainer(dyn, seed=seed)
    xs = torch.randn(N, 2, device=DEV, generator=gen)
    return dyn, tr, xs, gen


def measure(dyn, gen):
    """ESS per MH step like bench.py's leg + what tells a mixing sampler from one that only looks busy"""
    v = torch.randn((STEPS, N, 2), device=DEV, generator=gen)
    u = torch.rand((STEPS, N), device=DEV, generator=gen)
    direction = torch.randint(0, 2, (STEPS, N), device=DEV, dtype=torch.uint8, generator=gen)
    xf, p, hist = sample_chain(X0, dyn, STEPS, v=v, u=u, direction=direction, record=True)
    X = torch.cat([X0[None], hist[:-1]], dim=0)
    A = func_utils.acl_spectrum(X, SCALE)
    ess = float(func_utils.ESS(A))
    Xn = X.double()
    step1 = (Xn[1:] - Xn[:-1]).norm(dim=2)
    step2 = (Xn[2:] - Xn[:-2]).norm(dim=2)
    moved = step1 > 0
    A = np.asarray(A, dtype=np.float64)
    return {"ess_per_mh_step": ess, "mean_accept_prob": float(p.mean()), "acl_lag1": float(A[1]) if len(A) > 1 else None,
            "acl_lag2": float(A[2]) if len(A) > 2 else None,
            "mean_jump": float(step1[moved].mean()) if bool(moved.any()) else 0.0,
            # |x_{t+2} - x_t| / (|x_{t+1} - x_t| + |x_{t+2} - x_{t+1}|): 0 = every second move undoes the first, ~0.7 = independent directions
            "two_step_ratio": float((step2 / (step1[1:] + step1[:-1] + 1e-30))[moved[1:] & moved[:-1]].mean()),
            "var_final": [float(t) for t in np.cov(xf.cpu().numpy().T).ravel()]}


def sweep(n_seeds, train_steps=5000):
    hmc = Dynamics(2, DIST.get_energy_function(), T=10, eps=0.15, hmc=True, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(0)
    v = torch.randn((STEPS, N, 2), device=DEV, generator=gen)
    u = torch.rand((STEPS, N), device=DEV, generator=gen)
    xf, p, hist = sample_chain(X0, hmc, STEPS, v=v, u=u, record=True)
    ess_hmc = float(func_utils.ESS(func_utils.acl_spectrum(torch.cat([X0[None], hist[:-1]], dim=0), SCALE)))
    rows = []
    for seed in range(n_seeds):
        dyn, tr, xs, gen = make(seed)
        t0 = time.perf_counter()
        for _ in range(train_steps):
            loss_t, px_t, xs, _ = tr.step(xs)
        torch.cuda.synchronize()
        r = measure(dyn, gen)
        r.update({"seed": seed, "train_seconds": time.perf_counter() - t0, "final_train_loss": float(loss_t),
                  "final_train_accept": float(px_t.mean()), "eps": float(torch.exp(dyn.alpha.detach()))})
        rows.append(r)
        print("seed %2d  ESS %.4f  lag1 %.3f  accept %.3f  jump %.2f  two-step %.3f  eps %.3f  loss %.1f" % (
            seed, r["ess_per_mh_step"], r["acl_lag1"], r["mean_accept_prob"], r["mean_jump"], r["two_step_ratio"], r["eps"],
            r["final_train_loss"]), flush=True)
    e = np.array([r["ess_per_mh_step"] for r in rows])
    low = [r["seed"] for r in rows if r["ess_per_mh_step"] < 5.0 * ess_hmc]
    out = {"workload": "SCG-2D, %d trainings of %d Adam steps on 200 chains, then 200 chains x 2000 MH steps" % (n_seeds, train_steps),
           "hmc_ess_per_mh_step": ess_hmc, "ess_mean": float(e.mean()), "ess_sd": float(e.std(ddof=1)), "ess_median": float(np.median(e)),
           "seeds_below_5x_hmc": low, "collapse_frequency": len(low) / float(n_seeds), "reference_ess_per_mh_step": 2.61e-1,
           "by_seed": rows}
    print(json.dumps({k: v for k, v in out.items() if k != "by_seed"}))
    return out


# ---- float64 replay of one training --------------------------------------------------------------------------------------------
def flat_of(nets):
    return np.concatenate([np.asarray(nets[n][k], np.float64).ravel() for n in ("xnet", "vnet") for k, _ in _SHAPES])


def replay(seed, train_steps=5000, checkpoints=(1, 10, 50, 200, 1000, 2000, 5000)):
    from oracle import l2hmc_train_oracle as TO
    from oracle.l2hmc_oracle import NET_KEYS
    dyn, tr, xs, gen = make(seed)
    L = _ffi.lib()
    d, T = 2, 10
    nets = {n: {k: w[k].detach().cpu().numpy().astype(np.float64).copy() for k in NET_KEYS}
            for n, w in (("xnet", dyn._xw), ("vnet", dyn._vw))}
    alpha = float(dyn.alpha.detach().cpu())
    mask = dyn._mask.cpu().numpy().astype(np.float64)
    x_or = xs.cpu().numpy().astype(np.float64)
    target = TO.GaussianTarget(np.zeros(2, np.float32), DIST.i_sigma.astype(np.float32), np.float64)
    m_or = {"alpha": 0.0, **{n + k: np.zeros_like(nets[n][k]) for n in nets for k in NET_KEYS}}
    v_or = {"alpha": 0.0, **{n + k: np.zeros_like(nets[n][k]) for n in nets for k in NET_KEYS}}
    W = torch.empty((3, N, d), dtype=torch.float32, device=DEV)
    dr = torch.empty((3, N), dtype=torch.uint8, device=DEV)
    uu = torch.empty((3, N), dtype=torch.float32, device=DEV)
    rows = []
    t0 = time.perf_counter()
    for step in range(train_steps):
        # the draws Trainer.step is about to make: same call, same stream position (seed, 3 * global_step, chain 0)
        _ffi.check(L.l2hmc_rng_fill(tr.seed, 3 * tr.global_step, 0, N, d, 3, W.data_ptr(), dr.data_ptr(), uu.data_ptr(),
                                    _ffi.current_stream(DEV)))
        Wh, dh, uh = W.cpu().numpy().astype(np.float64), dr.cpu().numpy(), uu.cpu().numpy().astype(np.float64)
        lr = tr.lr_at(tr.global_step)
        loss_g, px_g, xs, _ = tr.step(xs)
        # ---- the same step in float64
        eps = np.exp(alpha)
        total, grads, g_alpha = 0.0, None, 0.0
        for start, v0, dd in ((x_or, Wh[1], dh[1]), (Wh[0], Wh[2], dh[2])):
            loss, Lx, p, gr = TO.propose_loss_and_grad(start, v0, dd, target, nets["xnet"], nets["vnet"], eps, mask, T,
                                                       dtype=np.float64, float32_weights=False)
            total += loss
            if grads is None:
                grads, Lx_x, p_x = gr, Lx, p
            else:
                for n in ("xnet", "vnet"):
                    for k in NET_KEYS:
                        grads[n][k] = grads[n][k] + gr[n][k]
            g_alpha += gr["eps"] * eps
        x_or = np.where((p_x - uh[0] >= 0)[:, None], Lx_x, x_or)
        t = step + 1
        lr_t = lr * np.sqrt(1.0 - tr.beta2 ** t) / (1.0 - tr.beta1 ** t)
        for n in ("xnet", "vnet"):
            for k in NET_KEYS:
                gi = grads[n][k].reshape(nets[n][k].shape)
                m_or[n + k] = tr.beta1 * m_or[n + k] + (1 - tr.beta1) * gi
                v_or[n + k] = tr.beta2 * v_or[n + k] + (1 - tr.beta2) * gi * gi
                nets[n][k] = nets[n][k] - lr_t * m_or[n + k] / (np.sqrt(v_or[n + k]) + tr.epsilon)
        if tr.train_alpha:
            m_or["alpha"] = tr.beta1 * m_or["alpha"] + (1 - tr.beta1) * g_alpha
            v_or["alpha"] = tr.beta2 * v_or["alpha"] + (1 - tr.beta2) * g_alpha * g_alpha
            alpha = alpha - lr_t * m_or["alpha"] / (np.sqrt(v_or["alpha"]) + tr.epsilon)
        if t in checkpoints or t == train_steps:
            th_g = tr.theta.detach().cpu().numpy().astype(np.float64)
            th_o = np.concatenate([flat_of(nets), [alpha]])
            rel = float(np.abs(th_g - th_o).max() / max(1e-30, np.abs(th_o).max()))
            rows.append({"step": t, "max_param_diff_rel": rel, "loss_gpu": float(loss_g), "loss_f64": float(total),
                         "x_state_diff": float(np.abs(xs.cpu().numpy() - x_or).max()), "eps_gpu": float(np.exp(th_g[-1])),
                         "eps_f64": float(np.exp(alpha))})
            print("step %5d  max |theta_gpu - theta_f64| / max|theta| %.2e   loss gpu %.4f f64 %.4f   chain-state diff %.2e   eps %.4f / %.4f   (%.0f s)"
                  % (t, rel, float(loss_g), float(total), rows[-1]["x_state_diff"], rows[-1]["eps_gpu"], rows[-1]["eps_f64"],
                     time.perf_counter() - t0), flush=True)
    r_gpu = measure(dyn, torch.Generator(device=DEV).manual_seed(1000 + seed))
    # the float64-trained parameters on the same kernels
    with torch.no_grad():
        for n, w in (("xnet", dyn._xw), ("vnet", dyn._vw)):
            for k in NET_KEYS:
                w[k].copy_(torch.as_tensor(nets[n][k], dtype=torch.float32).reshape(w[k].shape))
        dyn.alpha.fill_(float(alpha))
    dyn._packed_key = None
    r_f64 = measure(dyn, torch.Generator(device=DEV).manual_seed(1000 + seed))
    out = {"seed": seed, "train_steps": train_steps, "along_the_way": rows, "sampler_trained_on_gpu": r_gpu,
           "sampler_trained_in_float64": r_f64}
    print(json.dumps({"seed": seed, "ess_gpu_trained": r_gpu["ess_per_mh_step"], "ess_float64_trained": r_f64["ess_per_mh_step"],
                      "lag1_gpu": r_gpu["acl_lag1"], "lag1_f64": r_f64["acl_lag1"]}))
    return out


if __name__ == "__main__":
    mode = sys.argv[1]
    res = sweep(int(sys.argv[2])) if mode == "sweep" else replay(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 5000)
    os.makedirs("gpurun_out/r05_ess", exist_ok=True)
    with open("gpurun_out/r05_ess/%s_%s.json" % (mode, sys.argv[2]), "w") as f:
        json.dump(res, f, indent=1)
