"""Training step of the sampler -- the notebook's objective and optimiser (SCGExperiment.ipynb raw
lines 156-181 and 254-271; `utils/losses.py:53-59` is the same per-proposal term).

    loss = scale (mean 1/v1 + mean 1/v2) - (mean v1 + mean v2) / scale,      scale = 0.1
    v1 = |x - Lx|^2 px + 1e-4  from propose(x) (whose MH-selected state continues the chains),
    v2 likewise from propose(z), z ~ N(0, I);   Adam, lr = 1e-3 * 0.96 ** floor(step / 1000).

The proposals and the gradient of their loss terms w.r.t. all net parameters and alpha come from ONE
launch of ONE HIP kernel over the 2N chains [x; z] (`l2hmc_train_propose_grad`, hand-derived reverse
mode incl. the Hessian-vector path through grad U).  With chains sharded over ranks the flat gradient is all-reduced ONCE per step
(RCCL on the GPU box / gloo in tests): the loss is a mean over chains, so summing per-rank
gradients computed with inv_n = 1 / (global chain count) is exact.  torch is used for the Adam
update of the parameter tensors and the collective only.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _ffi
from .distributions import as_device_f32

_SHAPES = (("W1", "dH"), ("b1", "H"), ("W2", "dH"), ("b2", "H"), ("W3", "2H"), ("b3", "H"),
           ("W4", "HH"), ("b4", "H"), ("Ws", "Hd"), ("bs", "d"), ("Wt", "Hd"), ("bt", "d"),
           ("Wq", "Hd"), ("bq", "d"), ("lam_s", "d"), ("lam_q", "d"))


def _numel(code, d, H):
    return {"dH": d * H, "H": H, "2H": 2 * H, "HH": H * H, "Hd": H * d, "d": d}[code]


class Trainer(object):
    def __init__(self, dynamics, lr=1e-3, decay_steps=1000, decay_rate=0.96, scale=0.1):
        if dynamics.hmc:
            raise ValueError("an HMC-mode Dynamics has nothing to train")
        self.dyn, self.scale = dynamics, float(scale)
        d, H = dynamics.x_dim, dynamics.H
        L = _ffi.lib()
        self.n_grad = _ffi.check(L.l2hmc_train_grad_floats(d, H))
        self.flat = torch.zeros(self.n_grad, dtype=torch.float32, device=dynamics.device)
        # (tensor, offset, numel) in the flat layout [XNet | VNet | eps]
        self.slots, off = [], 0
        for w in (dynamics._xw, dynamics._vw):
            for name, code in _SHAPES:
                n = _numel(code, d, H)
                self.slots.append((w[name], off, n))
                off += n
        assert off + 1 == self.n_grad
        self.params = [t for t, _, _ in self.slots] + ([dynamics.alpha] if dynamics.alpha.requires_grad else [])
        self.opt = torch.optim.Adam(self.params, lr=lr)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: decay_rate ** (s // decay_steps))
        self.global_step = 0
        self._ws = None

    # ---- one proposal + its gradient (accumulated into self.flat) --------------------------------
    def _propose_grad(self, start, v, direction, n_total):
        dyn = self.dyn
        N, d = start.shape
        L = _ffi.lib()
        need = _ffi.check(L.l2hmc_train_workspace_floats(N, d, dyn.T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.float32, device=dyn.device)
        Lx = torch.empty_like(start)
        p = torch.empty(N, dtype=torch.float32, device=dyn.device)
        v1 = torch.empty(N, dtype=torch.float32, device=dyn.device)
        xs = _ffi.L2hmcNet(*[dyn._xw[k].data_ptr() for k in _ffi.NET_FIELDS])
        vs = _ffi.L2hmcNet(*[dyn._vw[k].data_ptr() for k in _ffi.NET_FIELDS])
        fn = dyn._fn
        buf = fn._buffers(dyn.device)
        if fn.kind == _ffi.ENERGY_GAUSS_DIAG:
            prec = buf["prec"]
        elif fn.kind in (_ffi.ENERGY_GAUSS_DENSE, _ffi.ENERGY_GMM):
            prec = buf["_raw"]                         # RAW (k, d, d) precisions, not the MFMA packing
        elif fn.kind == _ffi.ENERGY_ROUGHWELL:
            prec = None
        else:
            raise NotImplementedError("training supports the Gaussian, GMM and Rough-Well targets")
        a = _ffi.L2hmcTrainArgs()
        a.xnet, a.vnet = C.pointer(xs), C.pointer(vs)
        a.energy = _ffi.L2hmcEnergy(fn.kind, fn.n_comp, _ffi.ptr(buf["mu"]), _ffi.ptr(prec), _ffi.ptr(buf["logc"]),
                                    fn.eta, int(fn.easy), 1.0)
        a.masks, a.trig = dyn._mask.data_ptr(), dyn._trig.data_ptr()
        if dyn.eps_override is None:
            a.alpha, a.eps_host = dyn.alpha.data_ptr(), 0.0
        else:
            a.alpha, a.eps_host = None, float(dyn.eps_override)
        a.n_chains, a.d, a.H, a.T = N, d, dyn.H, dyn.T
        a.x, a.v = start.data_ptr(), v.data_ptr()
        a.direction, a.direction_all = direction.data_ptr(), 1
        a.scale, a.inv_n = self.scale, 1.0 / float(n_total)
        a.Lx, a.p, a.v1 = Lx.data_ptr(), p.data_ptr(), v1.data_ptr()
        a.grad, a.workspace = self.flat.data_ptr(), self._ws.data_ptr()
        _ffi.check(L.l2hmc_train_propose_grad(a, _ffi.current_stream(dyn.device)))
        return Lx, p, v1

    def loss_and_grad(self, x, z=None, draws=None):
        """Loss and gradients (left in `.grad` of every parameter) for chain states `x`.
        draws: optional dict of injected randomness {x_dir, x_v, z, z_dir, z_v} (tests)."""
        dyn = self.dyn
        x = as_device_f32(x, dyn.device)
        N, d = x.shape
        gen, dev = dyn.generator, dyn.device
        draws = draws or {}

        def get(key, make):
            return as_device_f32(draws[key], dev) if key in draws else make()
        z = get("z", lambda: torch.randn((N, d), device=dev, generator=gen)) if z is None else as_device_f32(z, dev)
        xv = get("x_v", lambda: torch.randn((N, d), device=dev, generator=gen))
        zv = get("z_v", lambda: torch.randn((N, d), device=dev, generator=gen))

        def bits(key):
            if key in draws:
                return torch.as_tensor(draws[key], device=dev).to(torch.uint8).contiguous()
            return torch.randint(0, 2, (N,), device=dev, dtype=torch.uint8, generator=gen)
        xd, zd = bits("x_dir"), bits("z_dir")
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        n_total = N * world
        self.flat.zero_()
        # the x- and the z-proposal are independent and their loss terms add: ONE launch over the 2N
        # chains [x; z] (each chain's term still weighted 1 / n_total) instead of two half-empty ones
        Lxz, pxz, v12 = self._propose_grad(torch.cat([x, z]), torch.cat([xv, zv]), torch.cat([xd, zd]), n_total)
        Lx, px, v1, v2 = Lxz[:N], pxz[:N], v12[:N], v12[N:]
        terms = torch.stack([(1.0 / v1).sum(), (1.0 / v2).sum(), v1.sum(), v2.sum()]).double()
        if world > 1:
            dist.all_reduce(self.flat)                  # the ONE collective of a training step
            dist.all_reduce(terms)
        terms = terms / n_total
        loss = self.scale * (terms[0] + terms[1]) - (terms[2] + terms[3]) / self.scale
        for t, off, n in self.slots:
            t.grad = self.flat[off:off + n].view(t.shape)
        if dyn.alpha.requires_grad:
            dyn.alpha.grad = (self.flat[-1] * torch.exp(dyn.alpha.detach())).reshape(dyn.alpha.shape)
        return loss, Lx, px

    def step(self, x, u=None):
        """One optimiser step like nb raw 262-268: returns (loss, px, x_next, lr) where x_next is
        the MH-selected continuation of the chains."""
        from .sampler import tf_accept
        loss, Lx, px = self.loss_and_grad(x)
        lr = self.sched.get_last_lr()[0]
        self.opt.step()
        self.sched.step()
        self.global_step += 1
        x_next = tf_accept(x, Lx, px, u=u, dynamics=self.dyn)
        return loss, px, x_next, lr
