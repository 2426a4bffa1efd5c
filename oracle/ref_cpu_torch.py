"""Multi-threaded CPU baseline of the hot path -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

SURVEY.md 8(d) asks for the reference algorithm timed on ALL host cores next to the GPU number.
TensorFlow 1.x cannot be installed offline and the reference's sources cannot travel to the GPU
box, and numpy's elementwise ops are single-threaded, so this module restates the Gaussian-target
path of ``oracle/l2hmc_oracle.py`` op by op on torch-CPU fp32 tensors (intra-op thread pool =
``torch.set_num_threads``): the same graph the reference builds --

  * ``utils/dynamics.py:115-157`` `_forward_step`, ``:159-201`` `_backward_step`, ``:246-300``
    `forward` / `backward`, ``:302-309`` `p_accept`  (three / two gradient evaluations per step,
    exactly as the reference computes them: no reuse);
  * ``utils/sampler.py:28-55`` `propose`: BOTH directions on every chain, then mixed;
  * the notebook net (SCGExperiment.ipynb raw 51-78) with ``utils/layers.py`` semantics;
  * ``utils/distributions.py:31-32,41-57`` Gaussian energy either literally as the reference
    evaluates it (``nxn=True``: diag_part of the N x N product, gradient via autograd like
    ``tf.gradients``) or row-wise (``nxn=False``, analytic gradient).

Pinned by ``tests/test_oracle_golden.py::test_torch_cpu_baseline_matches_goldens`` against the same
golden vectors as the numpy oracle.  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` import it.
"""
import math

import torch

NET_KEYS = ('W1', 'b1', 'W2', 'b2', 'W3', 'b3', 'W4', 'b4',
            'Ws', 'bs', 'Wt', 'bt', 'Wq', 'bq', 'lam_s', 'lam_q')


def _net(w, a, b, tau):
    h = torch.relu(((a @ w['W1'] + w['b1']) + (b @ w['W2'] + w['b2'])) + (tau @ w['W3'] + w['b3']))
    h = torch.relu(h @ w['W4'] + w['b4'])
    S = torch.exp(w['lam_s']) * torch.tanh(h @ w['Ws'] + w['bs'])
    T = h @ w['Wt'] + w['bt']
    Q = torch.exp(w['lam_q']) * torch.tanh(h @ w['Wq'] + w['bq'])
    return S, T, Q


class GaussianRef:
    def __init__(self, mu, i_sigma, nxn):
        self.mu = torch.as_tensor(mu, dtype=torch.float32)
        self.S = torch.as_tensor(i_sigma, dtype=torch.float32)
        self.nxn = nxn

    def energy(self, x):
        dx = x - self.mu
        if self.nxn:                                   # distributions.py:31-32, literally
            return torch.diagonal(0.5 * ((dx @ self.S) @ dx.t()))
        return 0.5 * torch.sum((dx @ self.S) * dx, dim=1)

    def grad(self, x):
        if self.nxn:                                   # tf.gradients of the sum over the batch (dynamics.py:217-218)
            xx = x.detach().requires_grad_(True)
            return torch.autograd.grad(self.energy(xx).sum(), xx)[0]
        dx = x - self.mu
        return 0.5 * (dx @ self.S + dx @ self.S.t())


class DynamicsRef:
    def __init__(self, x_dim, energy, T, eps, mask, xnet, vnet):
        self.d, self.T, self.eps, self.en = x_dim, int(T), float(eps), energy
        self.mask = torch.as_tensor(mask, dtype=torch.float32)
        self.xw = {k: torch.as_tensor(xnet[k], dtype=torch.float32) for k in NET_KEYS}
        self.vw = {k: torch.as_tensor(vnet[k], dtype=torch.float32) for k in NET_KEYS}

    def _tau(self, step, n):
        ang = 2.0 * math.pi * float(step) / self.T
        return torch.tensor([[math.cos(ang), math.sin(ang)]], dtype=torch.float32).repeat(n, 1)

    def hamiltonian(self, x, v):
        return self.en.energy(x) + 0.5 * torch.sum(v * v, dim=1)

    def forward_step(self, x, v, step):
        eps, t = self.eps, self._tau(step, x.shape[0])
        g1 = self.en.grad(x)
        S, Tt, Q = _net(self.vw, x, g1, t)
        sv1 = 0.5 * eps * S
        v_h = v * torch.exp(sv1) + 0.5 * eps * (-(torch.exp(eps * Q) * g1) + Tt)
        m = self.mask[int(step)]
        mb = 1.0 - m
        S, Tt, Q = _net(self.xw, v_h, m * x, t)
        sx1 = eps * S
        y = m * x + mb * (x * torch.exp(sx1) + eps * (torch.exp(eps * Q) * v_h + Tt))
        S, Tt, Q = _net(self.xw, v_h, mb * y, t)
        sx2 = eps * S
        x_o = mb * y + m * (y * torch.exp(sx2) + eps * (torch.exp(eps * Q) * v_h + Tt))
        g2 = self.en.grad(x_o)
        _ = self.en.grad(x_o)                          # the reference evaluates grad_energy(x_o) twice (:147,152)
        S, Tt, Q = _net(self.vw, x_o, g2, t)
        sv2 = 0.5 * eps * S
        v_o = v_h * torch.exp(sv2) + 0.5 * eps * (-(torch.exp(eps * Q) * g2) + Tt)
        return x_o, v_o, torch.sum(sv1 + sv2 + mb * sx1 + m * sx2, dim=1)

    def backward_step(self, x_o, v_o, step):
        eps, t = self.eps, self._tau(step, x_o.shape[0])
        g1 = self.en.grad(x_o)
        S, Tt, Q = _net(self.vw, x_o, g1, t)
        sv2 = -0.5 * eps * S
        v_h = (v_o - 0.5 * eps * (-(torch.exp(eps * Q) * g1) + Tt)) * torch.exp(sv2)
        m = self.mask[int(step)]
        mb = 1.0 - m
        S, Tt, Q = _net(self.xw, v_h, mb * x_o, t)
        sx2 = -eps * S
        y = mb * x_o + m * (torch.exp(sx2) * (x_o - eps * (torch.exp(eps * Q) * v_h + Tt)))
        S, Tt, Q = _net(self.xw, v_h, m * y, t)
        sx1 = -eps * S
        x = m * y + mb * (torch.exp(sx1) * (y - eps * (torch.exp(eps * Q) * v_h + Tt)))
        g2 = self.en.grad(x)
        S, Tt, Q = _net(self.vw, x, g2, t)
        sv1 = -0.5 * eps * S
        v = torch.exp(sv1) * (v_h - 0.5 * eps * (-(torch.exp(eps * Q) * g2) + Tt))
        return x, v, torch.sum(sv1 + sv2 + mb * sx1 + m * sx2, dim=1)

    def p_accept(self, x0, v0, x1, v1, lj):
        p = torch.exp(torch.clamp(self.hamiltonian(x0, v0) - self.hamiltonian(x1, v1) + lj, max=0.0))
        return torch.where(torch.isfinite(p), p, torch.zeros_like(p))

    def run(self, x, v, fwd):
        X, V, j = x, v, torch.zeros(x.shape[0])
        for t in range(self.T):
            X, V, lj = self.forward_step(X, V, t) if fwd else self.backward_step(X, V, self.T - t - 1)
            j = j + lj
        return X, V, self.p_accept(x, v, X, V, j)


def propose(x, dyn, v_fwd, v_bwd, direction, u):
    """sampler.py:28-55 with injected draws: both directions on every chain, mix, MH select."""
    Lx1, _, p1 = dyn.run(x, v_fwd, True)
    Lx2, _, p2 = dyn.run(x, v_bwd, False)
    b = direction.to(torch.float32)[:, None]
    Lx = b * Lx1 + (1 - b) * Lx2
    px = b[:, 0] * p1 + (1 - b[:, 0]) * p2
    x_next = torch.where((px - u >= 0)[:, None], Lx, x)
    return Lx, px, x_next
