"""CPU restatement of the VAE experiment's sampler objective and its gradient (TEST INFRASTRUCTURE).

Only tests/ may import this.  It restates, in torch on the CPU (float64 by default) and op by op,

  * the decoder-posterior energy                      mnist_vae.py:104-111,122-126
  * the S/T/Q nets with the shared image branch       mnist_vae.py:134-167, utils/layers.py:29-37,81-95
  * the generalised leapfrog, both directions         utils/dynamics.py:115-201,246-309
  * propose + MH select                               utils/sampler.py:28-55
  * the sampler loss, chained over MH proposals       mnist_vae.py:185-226 (incl. energy_scale and the
                                                      random_lf_composition branch = sampler.py:57-85)

and differentiates the loss with torch autograd (the energy gradient inside the dynamics is itself an
autograd call with create_graph=True, i.e. the Hessian-vector products the HIP trainer derives by hand are
taken by double back-propagation here).  Pinned against the reference's own graph by
tests/golden/train_vae_small.npz (tests/test_oracle_golden.py); the GPU tests then use it as the checker at
sizes the fixture does not cover.  Fixture-style dict in, dict of numpy arrays out.
"""
import math

import numpy as np
import torch

NET_KEYS = ("W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4", "Ws", "bs", "Wt", "bt", "Wq", "bq", "lam_s", "lam_q")
MLP_KEYS = ("W1", "b1", "W2", "b2", "W3", "b3")


def _mlp3(w, x):                                   # Linear-softplus-Linear-softplus-Linear
    sp = torch.nn.functional.softplus
    h = sp(x @ w["W1"] + w["b1"])
    h = sp(h @ w["W2"] + w["b2"])
    return h @ w["W3"] + w["b3"]


def _net(w, a, b, tau, aux_h):                     # mnist_vae.py:142-167
    h = torch.relu(a @ w["W1"] + w["b1"] + b @ w["W2"] + w["b2"] + tau @ w["W3"] + w["b3"] + aux_h)
    h = torch.relu(h @ w["W4"] + w["b4"])
    S = torch.exp(w["lam_s"].reshape(1, -1)) * torch.tanh(h @ w["Ws"] + w["bs"])
    T = h @ w["Wt"] + w["bt"]
    Q = torch.exp(w["lam_q"].reshape(1, -1)) * torch.tanh(h @ w["Wq"] + w["bq"])
    return S, T, Q


class _Dyn(object):
    def __init__(self, dec, xnet, vnet, enc, aux, eps, mask, T):
        self.dec, self.xn, self.vn, self.aux = dec, xnet, vnet, aux
        self.aux_h = _mlp3(enc, aux)               # one shared encoder_sampler (mnist_vae.py:134-150)
        self.eps, self.mask, self.T = eps, mask, T

    def energy(self, z):                           # mnist_vae.py:122-126, TF's stable BCE
        l = _mlp3(self.dec, z)
        bce = torch.clamp(l, min=0) - l * self.aux + torch.log1p(torch.exp(-torch.abs(l)))
        return bce.sum(1) + 0.5 * (z * z).sum(1)

    def grad(self, z):                             # dynamics.py:217-218
        if not z.requires_grad:
            z = z.detach().requires_grad_(True)
        return torch.autograd.grad(self.energy(z).sum(), z, create_graph=True)[0]

    def tau(self, step, n):                        # dynamics.py:99-105
        t = 2.0 * math.pi * step / self.T
        return torch.tensor([math.cos(t), math.sin(t)], dtype=self.eps.dtype).repeat(n, 1)

    def fstep(self, x, v, step):                   # dynamics.py:115-157
        eps, t = self.eps, self.tau(step, x.shape[0])
        g1 = self.grad(x)
        S, T, Q = _net(self.vn, x, g1, t, self.aux_h)
        sv1 = 0.5 * eps * S
        vh = v * torch.exp(sv1) + 0.5 * eps * (-(torch.exp(eps * Q) * g1) + T)
        m = self.mask[int(step)]
        mb = 1.0 - m
        S, T, Q = _net(self.xn, vh, m * x, t, self.aux_h)
        sx1 = eps * S
        y = m * x + mb * (x * torch.exp(sx1) + eps * (torch.exp(eps * Q) * vh + T))
        S, T, Q = _net(self.xn, vh, mb * y, t, self.aux_h)
        sx2 = eps * S
        xo = mb * y + m * (y * torch.exp(sx2) + eps * (torch.exp(eps * Q) * vh + T))
        g2 = self.grad(xo)
        S, T, Q = _net(self.vn, xo, g2, t, self.aux_h)
        sv2 = 0.5 * eps * S
        vo = vh * torch.exp(sv2) + 0.5 * eps * (-(torch.exp(eps * Q) * g2) + T)
        return xo, vo, (sv1 + sv2 + mb * sx1 + m * sx2).sum(1)

    def bstep(self, xo, vo, step):                 # dynamics.py:159-201
        eps, t = self.eps, self.tau(step, xo.shape[0])
        g1 = self.grad(xo)
        S, T, Q = _net(self.vn, xo, g1, t, self.aux_h)
        sv2 = -0.5 * eps * S
        vh = (vo - 0.5 * eps * (-(torch.exp(eps * Q) * g1) + T)) * torch.exp(sv2)
        m = self.mask[int(step)]
        mb = 1.0 - m
        S, T, Q = _net(self.xn, vh, mb * xo, t, self.aux_h)
        sx2 = -eps * S
        y = mb * xo + m * (torch.exp(sx2) * (xo - eps * (torch.exp(eps * Q) * vh + T)))
        S, T, Q = _net(self.xn, vh, m * y, t, self.aux_h)
        sx1 = -eps * S
        x = m * y + mb * (torch.exp(sx1) * (y - eps * (torch.exp(eps * Q) * vh + T)))
        g2 = self.grad(x)
        S, T, Q = _net(self.vn, x, g2, t, self.aux_h)
        sv1 = -0.5 * eps * S
        v = torch.exp(sv1) * (vh - 0.5 * eps * (-(torch.exp(eps * Q) * g2) + T))
        return x, v, (sv1 + sv2 + mb * sx1 + m * sx2).sum(1)

    def run(self, x, v, forward):                  # dynamics.py:246-300
        X, V, j = x, v, 0.0
        for t in range(self.T):
            X, V, lj = self.fstep(X, V, t) if forward else self.bstep(X, V, self.T - t - 1)
            j = j + lj
        return X, V, j

    def p_accept(self, x0, v0, x1, v1, lj):        # dynamics.py:302-309
        H0 = self.energy(x0) + 0.5 * (v0 * v0).sum(1)
        H1 = self.energy(x1) + 0.5 * (v1 * v1).sum(1)
        p = torch.exp(torch.clamp(H0 - H1 + lj, max=0.0))
        return torch.where(torch.isfinite(p), p, torch.zeros_like(p))


def sampler_loss_and_grad(g, draws, MH=1, stop_gradient=False, R=None, dtype=torch.float64, energy_scale=0.0):
    """g: fixture-style dict (dec.*, enc.*, xnet.*, vnet.*, eps, mask, T, x, aux, log_sigma);
    draws: list of MH dicts {v_fwd, v_bwd, dir, u} (sampler.py:34-36 draws both momenta; each chain uses its
    direction's) -- or, for the `random_lf_composition` branch (mnist_vae.py:193-196 = chain_operator,
    sampler.py:57-85), {nb_steps, init_v, v_fwd: (K, N, d), v_bwd, dir: (K, N), u}.
    R: optional (N, d) -- adds sum(final_x * R) to the loss (the cotangent a later proposal would
    send).  energy_scale: mnist_vae.py:214,218,224.  Returns loss, per-parameter gradients ('xnet.W1', ..., 'enc.b3', 'alpha'), grad of the start point
    ('x0'), and the last proposal's Lx / px / x_next / v."""
    def T_(a, grad=False):
        t = torch.tensor(np.asarray(a, dtype=np.float64), dtype=dtype)
        return t.requires_grad_(True) if grad else t
    par = {}
    nets = {}
    for pre, keys in (("xnet.", NET_KEYS), ("vnet.", NET_KEYS), ("enc.", MLP_KEYS)):
        nets[pre] = {}
        for k in keys:
            par[pre + k] = T_(g[pre + k], True)
            nets[pre][k] = par[pre + k]
    dec = {k: T_(g["dec." + k]) for k in MLP_KEYS}
    alpha = T_(np.log(np.float64(g["eps"])), True)          # dynamics.py:50-54: eps = exp(alpha)
    par["alpha"] = alpha
    eps = torch.exp(alpha)
    aux = T_(g["aux"])
    dyn = _Dyn(dec, nets["xnet."], nets["vnet."], nets["enc."], aux, eps, T_(g["mask"]), int(g["T"]))
    x0 = T_(g["x"], True)
    w = 1.0 / (torch.exp(2.0 * T_(g["log_sigma"])) + 1e-4)  # stop_gradient(exp(2 log_sigma)) + 1e-4, mnist_vae.py:208
    init_x = x0
    for t in range(MH):
        dr = draws[t]
        if stop_gradient:
            init_x = init_x.detach()
        if "nb_steps" in dr:                               # chain_operator (sampler.py:57-85)
            init_v = T_(dr["init_v"])                      # :58-59 -- NOT the momentum any link starts from (:35-36)
            xs, vs, lj = init_x, init_v, 0.0
            for k in range(int(dr["nb_steps"])):
                dbit = T_(np.asarray(dr["dir"][k]).astype(np.float64)).reshape(-1, 1)
                X1, V1, j1 = dyn.run(xs, T_(dr["v_fwd"][k]), True)
                X2, V2, j2 = dyn.run(xs, T_(dr["v_bwd"][k]), False)
                xs = dbit * X1 + (1.0 - dbit) * X2         # sampler.py:38
                vs = dbit * V1 + (1.0 - dbit) * V2         # :40-41 (init_v was passed, so Lv is mixed and threaded on)
                lj = lj + dbit[:, 0] * j1 + (1.0 - dbit[:, 0]) * j2      # :44 with log_jac=True, summed at :66
            final_x = xs
            px = dyn.p_accept(init_x, init_v, xs, vs, lj)  # :79
        else:
            dbit = T_(np.asarray(dr["dir"]).astype(np.float64)).reshape(-1, 1)
            X1, V1, j1 = dyn.run(init_x, T_(dr["v_fwd"]), True)
            X2, V2, j2 = dyn.run(init_x, T_(dr["v_bwd"]), False)
            p1 = dyn.p_accept(init_x, T_(dr["v_fwd"]), X1, V1, j1)
            p2 = dyn.p_accept(init_x, T_(dr["v_bwd"]), X2, V2, j2)
            final_x = dbit * X1 + (1.0 - dbit) * X2            # sampler.py:38
            px = dbit[:, 0] * p1 + (1.0 - dbit[:, 0]) * p2     # sampler.py:44
        v = ((final_x - init_x) ** 2 * w).sum(1) * px + 1e-4
        loss = (1.0 / MH) * ((1.0 / v).mean() - v.mean())  # only the last iteration's terms survive (:187-189)
        ed = None
        if energy_scale != 0.0:                            # :214,218,224
            ed = (dyn.energy(final_x) - dyn.energy(init_x)) ** 2 * px + 1e-4
            loss = loss + energy_scale * (1.0 / MH) * ((1.0 / ed).mean() - ed.mean())
        prev_x = init_x
        acc = (px - T_(dr["u"])) >= 0                       # sampler.py:53-55
        init_x = torch.where(acc.reshape(-1, 1), final_x, init_x)
    if R is not None:
        loss = loss + (final_x * T_(R)).sum()
    names = sorted(par)
    grads = torch.autograd.grad(loss, [par[n] for n in names] + [x0], allow_unused=True)
    out = {"loss": float(loss.detach()), "v": v.detach().numpy(), "Lx": final_x.detach().numpy(), "px": px.detach().numpy(),
           "x_next": init_x.detach().numpy(), "x_last_start": prev_x.detach().numpy(),
           "ediff": None if ed is None else ed.detach().numpy()}
    for n, gr in zip(names + ["x0"], grads):
        out["grad." + n] = (gr if gr is not None else torch.zeros(())).detach().numpy()
    return out
