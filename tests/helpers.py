"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np

from oracle import l2hmc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
               if not f.endswith("p_accept_edge.npz"))


def load(case):
    return np.load(os.path.join(GOLDEN, case + ".npz"))


def oracle_energy(g, dtype=np.float32):
    kind = str(g["energy.kind"])
    if kind == "gaussian":
        return O.Gaussian(g["energy.mu"], g["energy.i_sigma"], dtype)
    if kind == "gmm":
        return O.GMM(g["energy.mus"], g["energy.i_sigmas"], g["energy.constants"], dtype)
    if kind == "roughwell":
        return O.RoughWell(float(g["energy.eta"]), bool(g["energy.easy"]), dtype)
    if kind == "funnel":
        return O.GaussianFunnel(float(g["energy.sigma"]), dtype)
    raise ValueError(kind)


def golden_nets(g):
    if int(g["hmc"]):
        return None, None
    return ({k: g["xnet." + k] for k in O.NET_KEYS}, {k: g["vnet." + k] for k in O.NET_KEYS})


def oracle_dynamics(g, dtype=np.float32):
    xn, vn = golden_nets(g)
    return O.Dynamics(int(g["x_dim"]), oracle_energy(g, dtype), int(g["T"]), g["eps"], g["mask"],
                      xn, vn, dtype=dtype)


def rel_err(a, b):
    """max |a-b| / max(1,|b|) over the entries where the reference value b is finite."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(1.0, np.abs(b[m]))))


def abs_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    return float(np.max(np.abs(a[m] - b[m]))) if m.any() else 0.0


def check_x_next(x_next, x, Lx, px, u, tol):
    """x_next rows must equal the proposal where px-u >= 0 (sampler.py:53-55) and the old
    state elsewhere; rows whose |px-u| < tol are ties that fp32 noise may flip."""
    x_next, x, Lx = np.asarray(x_next), np.asarray(x), np.asarray(Lx)
    margin = np.asarray(px, dtype=np.float64) - np.asarray(u, dtype=np.float64)
    acc, rej = margin >= tol, margin < -tol
    fin = np.all(np.isfinite(Lx), axis=1)
    assert rel_err(x_next[acc & fin], Lx[acc & fin]) <= tol
    assert np.array_equal(x_next[rej], x[rej])


# ---- HIP side ------------------------------------------------------------------------------
def hip_energy(g):
    """l2hmc_amd energy function carrying exactly the golden's fp32 parameters."""
    from l2hmc_amd import distributions as D
    kind = str(g["energy.kind"])
    if kind == "gaussian":
        obj = D.Gaussian.__new__(D.Gaussian)
        obj.mu, obj.sigma, obj.i_sigma = g["energy.mu"], None, g["energy.i_sigma"]
        return obj.get_energy_function()
    if kind == "gmm":
        obj = D.GMM.__new__(D.GMM)
        obj.mus = list(g["energy.mus"])
        obj.i_sigmas = list(g["energy.i_sigmas"])
        obj.constants = list(g["energy.constants"])
        obj.nb_mixtures, obj.k = len(obj.mus), obj.mus[0].shape[0]
        return obj.get_energy_function()
    if kind == "roughwell":
        return D.RoughWell(int(g["x_dim"]), float(g["energy.eta"]), bool(g["energy.easy"])).get_energy_function()
    if kind == "funnel":
        return D.GaussianFunnel(int(g["x_dim"])).get_energy_function()
    raise ValueError(kind)


def hip_dynamics(g, variant=0):
    """l2hmc_amd.Dynamics loaded with the golden's weights, mask and step size."""
    import torch
    from l2hmc_amd import Dynamics, layers
    hmc = bool(int(g["hmc"]))
    dyn = Dynamics(int(g["x_dim"]), hip_energy(g), T=int(g["T"]), eps=float(g["eps"]), hmc=hmc,
                   net_factory=None if hmc else layers.stq_network(int(g["H"])))
    dyn.mask = g["mask"]
    dyn.eps_override = float(g["eps"])
    dyn.variant = variant
    if not hmc:
        with torch.no_grad():
            for w, pre in ((dyn._xw, "xnet."), (dyn._vw, "vnet.")):
                for k in O.NET_KEYS:
                    w[k].copy_(torch.as_tensor(g[pre + k]).reshape(w[k].shape))
    return dyn


def to_dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def to_np(t):
    return t.detach().cpu().numpy()
