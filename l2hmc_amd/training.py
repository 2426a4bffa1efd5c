"""Training step of the sampler -- the notebook's objective and optimiser (SCGExperiment.ipynb raw
lines 156-181 and 254-271; `utils/losses.py:53-59` is the same per-proposal term).

    loss = scale (mean 1/v1 + mean 1/v2) - (mean v1 + mean v2) / scale,      scale = 0.1
    v1 = |x - Lx|^2 px + 1e-4  from propose(x) (whose MH-selected state continues the chains),
    v2 likewise from propose(z), z ~ N(0, I);   Adam, lr = 1e-3 * 0.96 ** floor(step / 1000).

The proposals and the gradient of their loss terms w.r.t. all net parameters and alpha come from ONE
launch of ONE HIP kernel over the 2N chains [x; z] (`l2hmc_train_propose_grad`, hand-derived reverse
mode incl. the Hessian-vector path through grad U).  With chains sharded over ranks the flat gradient
is all-reduced ONCE per step (RCCL on the GPU box / gloo in tests): the loss is a mean over chains, so
summing per-rank gradients computed with inv_n = 1 / (global chain count) is exact.

A training step is THREE launches and no torch optimiser (round 4; `l2hmc_train_step`): `l2hmc_rng_fill` (z, both
momenta, both direction vectors, the accept uniforms -- one Philox call); the gradient kernel, which reads the chains'
state where the caller keeps it; the fixed-order slot reduction, which overwrites the gradient and carries -- in extra
workgroups of the same launch -- the loss terms, the chains' Metropolis select and TF1's Adam over the flat parameter
vector [XNet | VNet | alpha] (the parameter tensors of the nets are views of it).  Sharded over ranks a step is ONE collective: the slot reduction
leaves the rank's loss sums and chain count behind the gradient, `[gradient | loss sums | count]` is all-reduced as one
buffer, and `l2hmc_adam_step_terms` applies Adam and forms the global loss.  The shard layout (global chain count, this
rank's Philox offset) is exchanged ONCE, on the first step of every rank, or declared with `set_sharding`.  torch provides
the buffers and the collective.
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


RNG_STREAM = "philox4x32-10/hw-boxmuller"       # (l2hmc_kernels.hpp philox_normal4; "…/libm-normals" up to ABI 3)


class Trainer(object):
    def __new__(cls, dynamics, *args, **kwargs):
        if getattr(dynamics, "_user_nets", False):
            # dynamics.py:78-79 + SCGExperiment.ipynb raw 178-181: the reference minimises over whatever variables net_factory
            # created.  Arbitrary callables train on the GEMM-engine trainer, their adjoints by callback (ABI 6 net_vjp_cb).
            if cls is not Trainer and cls is not SplitTrainer:
                raise NotImplementedError("caller-supplied nets train on the GEMM engine: use Trainer(dynamics)")
            return object.__new__(SplitTrainer)
        if getattr(dynamics, "_user", False) and cls is not SplitTrainer and cls is not Trainer:
            raise NotImplementedError("a caller-supplied energy trains on the GEMM engine: use Trainer(dynamics)")
        # samplers that run on the GEMM engine (nets wider than H = 15, the image-conditioned VAE sampler) train there
        if cls is Trainer and getattr(dynamics, "_split", False):
            return object.__new__(SplitTrainer)
        # ... and so do shapes no fused training kernel holds (d beyond the 16-chain tile's LDS plan: Rough Well d >= ~190)
        if cls is Trainer and hasattr(dynamics, "_fn"):
            fn = dynamics._fn
            rc = _ffi.lib().l2hmc_train_fused_lds_bytes(int(fn.kind), int(fn.n_comp or 1), int(dynamics.x_dim),
                                                        int(dynamics.H), int(dynamics.T))
            if rc == -2:                                   # L2HMC_ERR_UNSUPPORTED (include/l2hmc.h)
                return object.__new__(SplitTrainer)
        return object.__new__(cls)

    def __init__(self, dynamics, lr=1e-3, decay_steps=1000, decay_rate=0.96, scale=0.1,
                 beta1=0.9, beta2=0.999, epsilon=1e-8, seed=0):
        if dynamics.hmc:
            raise ValueError("an HMC-mode Dynamics has nothing to train")
        if (dynamics.use_temperature and float(dynamics.temperature) != 1.0) or float(dynamics.anneal_beta or 0.0) != 0.0:
            raise NotImplementedError("the training kernel differentiates the plain energy U: a tempered "
                                      "(temperature != 1) or annealed (anneal_beta) Dynamics is not supported")
        self.dyn, self.scale = dynamics, float(scale)
        self.lr0, self.decay_steps, self.decay_rate = float(lr), int(decay_steps), float(decay_rate)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.seed = int(seed)
        d, H = dynamics.x_dim, dynamics.H
        dev = dynamics.device
        L = _ffi.lib()
        self.n_grad = _ffi.check(L.l2hmc_train_grad_floats(d, H))
        self._alloc_flat(dev)                                                           # self.flat: the gradient
        # flat parameter vector [XNet | VNet | alpha]; every net parameter becomes a VIEW of it, so the
        # native Adam update is seen by the layers, by `Dynamics` and by its packed-weight cache
        self.theta = torch.zeros(self.n_grad, dtype=torch.float32, device=dev)
        self.slots, off = [], 0
        with torch.no_grad():
            for w in (dynamics._xw, dynamics._vw):
                for name, code in _SHAPES:
                    n = _numel(code, d, H)
                    t = w[name]
                    view = self.theta[off:off + n].view(t.shape)
                    view.copy_(t)
                    t.data = view
                    self.slots.append((t, off, n))
                    off += n
            assert off + 1 == self.n_grad
            self.train_alpha = bool(getattr(dynamics.alpha, "requires_grad", False))
            self.theta[-1].copy_(dynamics.alpha.reshape(()))
            dynamics.alpha.data = self.theta[-1].view(dynamics.alpha.shape)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.global_step = 0
        self._ws = None
        self.variant = 0             # 100: force the general tile kernel (l2hmc.h)
        self._io = None              # per-N buffers of step()
        self._layout = None
        self._auto_layout = None

    N_TAIL = 8                       # floats behind the gradient in the all-reduced buffer (6 used)

    def _alloc_flat(self, dev):
        """[gradient (n_grad) | tail]: ONE buffer, so that a sharded step's gradient, loss sums and chain count travel in
        ONE all-reduce; `self.flat` is the gradient view."""
        self._flat_ext = torch.zeros(self.n_grad + self.N_TAIL, dtype=torch.float32, device=dev)
        self.flat = self._flat_ext[:self.n_grad]

    def _allreduce_flat(self, sums64, count):
        """The ONE collective of a sharded step: all-reduce [gradient | (hi, lo) float pairs of the double sums | count];
        returns (reduced sums as a float64 tensor, reduced count as a float64 scalar tensor) -- on the device, no sync.
        Accuracy: each rank's double sum travels as float32 hi + lo (exact to ~2^-48 of the value), but the collective ADDS
        in float32, so the reduced loss sums are float32-accurate: relative 2^-24 of the sum per addition (the gradient next
        to them is float32 anyway; the single-process loss stays a double).  The chain count (two 12-bit-scaled halves) is
        exact up to 2^36 chains."""
        k = int(sums64.numel())
        assert 2 * k + 2 <= self.N_TAIL
        tail = self._flat_ext[self.n_grad:]
        hi = sums64.to(torch.float32)
        tail[0:2 * k:2] = hi
        tail[1:2 * k:2] = (sums64 - hi.double()).to(torch.float32)
        tail[2 * k] = float(count // 4096)
        tail[2 * k + 1] = float(count % 4096)
        dist.all_reduce(self._flat_ext[:self.n_grad + 2 * k + 2])
        t = tail.double()
        return t[0:2 * k:2] + t[1:2 * k:2], t[2 * k] * 4096.0 + t[2 * k + 1]

    # ---- checkpoint (the reference saves variables with tf.train.Saver, mnist_vae.py:290,334, and has to smuggle
    #      the masks around it, eval_sampler.py:52-59,156): everything a run needs to continue bit for bit ----
    def state_dict(self):
        return {"dynamics": self.dyn.state_dict(), "theta": self.theta.detach().cpu().clone(),
                "m": self.m.cpu().clone(), "v": self.v.cpu().clone(), "global_step": int(self.global_step),
                "seed": int(self.seed),
                # which random stream the continuation will draw from: ABI 4 switched the in-kernel normals from libm to the
                # hardware log / sqrt / sin / cos forms (last-bit different draws); a checkpoint resumed on another stream
                # still trains, but not "bit for bit" -- load_state_dict warns
                "rng_stream": RNG_STREAM, "abi": int(_ffi.ABI_VERSION),
                "hyper": {"lr": self.lr0, "decay_steps": self.decay_steps, "decay_rate": self.decay_rate,
                          "scale": self.scale, "beta1": self.beta1, "beta2": self.beta2, "epsilon": self.epsilon}}

    def load_state_dict(self, sd):
        if sd["theta"].numel() != self.theta.numel():
            raise ValueError("checkpoint has %d parameters, this sampler %d" % (sd["theta"].numel(), self.theta.numel()))
        # a checkpoint that STATES its stream (round 5 on) or an ABI <= 3 (libm normals) is compared; one with neither key
        # (rounds 1-4: ABI 4 already drew the hardware normals without saying so) is of unknown stream -- no warning
        stream = sd.get("rng_stream")
        if stream is None and "abi" in sd and int(sd["abi"]) <= 3:
            stream = "philox4x32-10/libm-normals"
        if stream is not None and stream != RNG_STREAM:
            import warnings
            warnings.warn("checkpoint was written on random stream %r, this library draws %r: the run continues, but not bit "
                          "for bit (rebuild with -DL2HMC_LIBM_NORMALS for the old normals)" % (stream, RNG_STREAM))
        self.dyn.mask = sd["dynamics"]["mask"]
        with torch.no_grad():
            self.theta.copy_(sd["theta"].to(self.theta.device))       # the net tensors and alpha are views of theta
            self.m.copy_(sd["m"].to(self.m.device))
            self.v.copy_(sd["v"].to(self.v.device))
        self.global_step, self.seed = int(sd["global_step"]), int(sd["seed"])
        h = sd.get("hyper", {})
        self.lr0, self.decay_steps, self.decay_rate = h.get("lr", self.lr0), h.get("decay_steps", self.decay_steps), h.get("decay_rate", self.decay_rate)
        self.scale, self.beta1, self.beta2, self.epsilon = (h.get("scale", self.scale), h.get("beta1", self.beta1),
                                                            h.get("beta2", self.beta2), h.get("epsilon", self.epsilon))
        self.dyn._packed_key = None

    # ---- learning-rate schedule (nb raw 178-180: exponential_decay(..., staircase=True)) -------------
    def lr_at(self, step):
        return self.lr0 * self.decay_rate ** (step // self.decay_steps)

    # ---- one launch: proposals + their gradient (accumulated into self.flat) ---------------------------
    def _propose_grad(self, start, v, direction, n_total, out=None):
        a, keep, (Lx, p, v1) = self._train_args(start, v, direction, n_total, out)
        _ffi.check(_ffi.lib().l2hmc_train_propose_grad(a, _ffi.current_stream(self.dyn.device)))
        return Lx, p, v1

    def _train_args(self, start, v, direction, n_total, out=None):
        """L2hmcTrainArgs for the chains `start` (and what has to stay alive while the call runs)."""
        dyn = self.dyn
        N, d = start.shape
        L = _ffi.lib()
        need = _ffi.check(L.l2hmc_train_workspace_floats(N, d, dyn.H, dyn.T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.float32, device=dyn.device)
        if out is None:
            out = (torch.empty_like(start), torch.empty(N, dtype=torch.float32, device=dyn.device),
                   torch.empty(N, dtype=torch.float32, device=dyn.device))
        Lx, p, v1 = out
        xs = _ffi.L2hmcNet(*[dyn._xw[k].data_ptr() for k in _ffi.NET_FIELDS])
        vs = _ffi.L2hmcNet(*[dyn._vw[k].data_ptr() for k in _ffi.NET_FIELDS])
        fn = dyn._fn
        buf = fn._buffers(dyn.device)
        if fn.kind == _ffi.ENERGY_GAUSS_DIAG:
            prec = buf["prec"]
        elif fn.kind in (_ffi.ENERGY_GAUSS_DENSE, _ffi.ENERGY_GMM):
            prec = buf["_raw"]                         # RAW (k, d, d) precisions, not the MFMA packing
        elif fn.kind in (_ffi.ENERGY_ROUGHWELL, _ffi.ENERGY_FUNNEL):
            prec = None
        else:
            raise NotImplementedError("training supports the Gaussian, GMM, Rough-Well and funnel targets")
        a = _ffi.L2hmcTrainArgs()
        a.xnet, a.vnet = C.pointer(xs), C.pointer(vs)
        a.energy = _ffi.L2hmcEnergy(fn.kind, fn.n_comp, _ffi.ptr(buf["mu"]), _ffi.ptr(prec), _ffi.ptr(buf["logc"]),
                                    fn.eta, int(fn.easy), 1.0, 0.0, fn.den, 0)
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
        a.variant = int(self.variant)
        return a, (xs, vs, buf), (Lx, p, v1)

    def _world(self):
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    # a one-rank process group normally takes the single-process path (no collective); `always_reduce = True` sends it through
    # the sharded path anyway -- the step's ONE all-reduce over RCCL with a single rank (bench.py --force-dist on a 1-GPU box)
    always_reduce = False

    def _sharded(self, world):
        return world > 1 or (self.always_reduce and dist.is_available() and dist.is_initialized())

    def set_sharding(self, n_total, chain_offset):
        """Declare this rank's place in the global chain batch: `n_total` chains over all ranks, this rank's row 0 is
        global chain `chain_offset` (e.g. from `sharding.shard_range`).  With a declared layout no step ever exchanges
        the layout; `set_sharding(None, None)` returns to the discovered layout (call it on EVERY rank: the next step of
        each then takes part in the exchange)."""
        self._layout = None if n_total is None else (int(n_total), int(chain_offset))
        self._auto_layout = None
        self._stale = self._reduced = None

    def _shard(self, N):
        """(global chain count, global index of this rank's row 0): ranks may hold different numbers of chains
        (sharding.shard_range hands out blocks whose sizes differ by up to one), so the loss normalisation and the
        Philox chain offsets come from the ranks' local counts, not from N * world.  Unless the layout was declared
        (`set_sharding`), the counts are exchanged ONCE -- on every rank's first step, unconditionally, so all ranks
        enter it together -- and kept.  A rank whose local count changes afterwards must not quietly re-exchange (it would
        sit in a collective no other rank enters: the hang the review of round 2 found) and must not raise BEFORE the
        step's all-reduce either (the other ranks would then wait for it until the backend times out -- review of round
        4): it takes this one step on the cached layout, so every rank passes the collective, and raises at the END of
        the step; every other rank learns of it from the reduced chain count of that same all-reduce, which no longer
        matches its cached total, and raises at the start of ITS next step (`_check_reduced_count`: an asynchronous
        two-float copy to pinned memory, read one step late -- no host synchronisation in a healthy run).  Recover with
        `set_sharding(n_total, chain_offset)`, or `set_sharding(None, None)` on every rank."""
        world = self._world()
        if world == 1:
            return N, 0
        self._check_reduced_count()
        if getattr(self, "_layout", None) is not None:
            return self._layout
        auto = getattr(self, "_auto_layout", None)
        if auto is None:
            rank = dist.get_rank()
            counts = torch.zeros(world, dtype=torch.float64, device=self.dyn.device)
            counts[rank] = float(N)
            dist.all_reduce(counts)                     # once per layout, not per step
            counts = counts.cpu()
            auto = self._auto_layout = (int(N), int(counts.sum()), int(counts[:rank].sum()))
        elif auto[0] != int(N):
            self._stale = (auto[0], int(N))             # raised by `_raise_if_stale` once this step's collective is behind us
        return auto[1], auto[2]

    _STALE_MSG = ("this rank's chain count changed from %d to %d under a discovered shard layout: declare the new layout with "
                  "set_sharding(n_total, chain_offset), or call set_sharding(None, None) on EVERY rank so that all of them "
                  "re-enter the layout exchange together")

    def _raise_if_stale(self):
        st = getattr(self, "_stale", None)
        if st is not None:
            self._stale = None
            raise RuntimeError(self._STALE_MSG % st)

    def _note_reduced_count(self, cnt2, n_total, hi_scale=1.0):
        """Keep the chain count the step's all-reduce produced (a 2-float device view: hi_scale * hi + lo) for
        `_check_reduced_count`."""
        if cnt2.is_cuda:
            host = torch.empty(2, dtype=cnt2.dtype, pin_memory=True)
            host.copy_(cnt2, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = cnt2.detach().clone(), None
        self._reduced = (host, ev, int(n_total), float(hi_scale))

    def _check_reduced_count(self):
        red = getattr(self, "_reduced", None)
        if red is None:
            return
        host, ev, expect, hi_scale = red
        self._reduced = None
        if ev is not None:
            ev.synchronize()                            # (recorded a whole step ago: returns at once)
        got = int(round(float(host[0]) * hi_scale + float(host[1])))
        if got != expect:
            raise RuntimeError("the last step's all-reduce counted %d chains over all ranks, this rank's shard layout says %d: "
                               "some rank's chain count changed -- call set_sharding(None, None) on EVERY rank (or declare "
                               "the new layout) before the next step" % (got, expect))

    def _loss_terms(self, v12, n_total):
        """[sum 1/v1, sum v1, the single-process loss] of the rank's per-chain loss arguments: one fixed-order double
        reduction on the device (l2hmc_loss_terms), in a FRESH tensor (an allocation, not a launch)."""
        lt = torch.empty(3, dtype=torch.float64, device=v12.device)
        _ffi.check(_ffi.lib().l2hmc_loss_terms(v12.data_ptr(), v12.numel(), self.scale, 1.0 / float(n_total),
                                               lt.data_ptr(), _ffi.current_stream(v12.device)))
        return lt

    def _reduce_and_loss(self, v12, N, n_total, world):
        """The step's collective (sharded: gradient + loss sums + count in ONE all-reduce) and the loss of the global batch."""
        lt = self._loss_terms(v12, n_total)
        if world == 1:
            return lt[2]
        sums, cnt = self._allreduce_flat(lt[:2], N)
        self._note_reduced_count(self._flat_ext[self.n_grad + 4:self.n_grad + 6], n_total, hi_scale=4096.0)
        return (self.scale * sums[0] - sums[1] / self.scale) / cnt

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
        world = self._world()
        n_total, _ = self._shard(N)
        self.flat.zero_()
        # the x- and the z-proposal are independent and their loss terms add: ONE launch over the 2N
        # chains [x; z] (each chain's term still weighted 1 / n_total) instead of two half-empty ones
        Lxz, pxz, v12 = self._propose_grad(torch.cat([x, z]), torch.cat([xv, zv]), torch.cat([xd, zd]), n_total)
        loss = self._reduce_and_loss(v12, N, n_total, world)      # (sharded: the ONE collective of a training step)
        for t, off, n in self.slots:
            t.grad = self.flat[off:off + n].view(t.shape)
        if self.train_alpha:
            self.dyn.alpha.grad = (self.flat[-1] * torch.exp(dyn.alpha.detach())).reshape(dyn.alpha.shape)
        self._raise_if_stale()
        return loss, Lxz[:N], pxz[:N]

    def _buffers(self, N, d):
        if self._io is None or self._io["N"] != N:
            dev, f32 = self.dyn.device, torch.float32
            self._io = {"N": N,
                        # rows: [x | z | v_x | v_z]: the Philox fill writes rows 1..3 (three "proposals" of N chains)
                        "W": torch.empty((4, N, d), dtype=f32, device=dev),
                        "dir": torch.empty((3, N), dtype=torch.uint8, device=dev),
                        "u": torch.empty((3, N), dtype=f32, device=dev),
                        "Lx": torch.empty((2 * N, d), dtype=f32, device=dev),
                        "p": torch.empty(2 * N, dtype=f32, device=dev),
                        "v1": torch.empty(2 * N, dtype=f32, device=dev)}
        return self._io

    def step(self, x, u=None):
        """One optimiser step like nb raw 262-268: returns (loss, px, x_next, lr) where x_next is
        the MH-selected continuation of the chains.  Three launches (module docstring); sharded: + ONE all-reduce and
        the Adam launch behind it."""
        dyn = self.dyn
        x = as_device_f32(x, dyn.device)
        N, d = x.shape
        L = _ffi.lib()
        s = _ffi.current_stream(dyn.device)
        io = self._buffers(N, d)
        W = io["W"]
        world = self._world()
        n_total, chain_off = self._shard(N)
        # z, v_x, v_z (rows 1..3 of W), the direction bits of both proposals (rows 1, 2 of dir) and the
        # accept uniforms (row 0 of u): one call, stream position = (seed, 3 * global_step, global chain)
        _ffi.check(L.l2hmc_rng_fill(self.seed, 3 * self.global_step, chain_off, N, d, 3, W[1].data_ptr(),
                                    io["dir"].data_ptr(), io["u"].data_ptr(), s))
        # (the accept probabilities, the selected states and the loss go to FRESH tensors -- allocations, not launches)
        p12 = torch.empty(2 * N, dtype=torch.float32, device=dyn.device)
        x_next = torch.empty_like(x)
        lt = torch.empty(3, dtype=torch.float64, device=dyn.device)
        uu = io["u"][0] if u is None else as_device_f32(u, dyn.device)
        # chains [x; z]: rows N .. 2N-1 of W[0:2] are z; the x rows are read from the caller's tensor (x_head)
        a, keep, _ = self._train_args(W[0:2].view(2 * N, d), W[2:4].view(2 * N, d), io["dir"][1:3].view(2 * N), n_total,
                                      out=(io["Lx"], p12, io["v1"]))
        lr = self.lr_at(self.global_step)
        self.global_step += 1
        st = _ffi.L2hmcTrainStep()
        st.x_head, st.n_head = x.data_ptr(), N
        st.u, st.x_next = uu.data_ptr(), x_next.data_ptr()
        st.loss = lt.data_ptr()
        sharded = self._sharded(world)
        if not sharded:
            st.theta, st.m, st.v = self.theta.data_ptr(), self.m.data_ptr(), self.v.data_ptr()
            st.lr, st.beta1, st.beta2, st.epsilon = lr, self.beta1, self.beta2, self.epsilon
            st.step, st.train_alpha = self.global_step, int(self.train_alpha)
        else:
            st.terms = self._flat_ext[self.n_grad:].data_ptr()
        _ffi.check(L.l2hmc_train_step(a, st, s))
        if sharded:
            dist.all_reduce(self._flat_ext[:self.n_grad + 6])           # the ONE collective of a training step
            self._note_reduced_count(self._flat_ext[self.n_grad + 4:self.n_grad + 6], n_total)
            n_par = self.n_grad if self.train_alpha else self.n_grad - 1
            _ffi.check(L.l2hmc_adam_step_terms(self.theta.data_ptr(), self.flat.data_ptr(), self.m.data_ptr(),
                                               self.v.data_ptr(), n_par, lr, self.beta1, self.beta2, self.epsilon,
                                               self.global_step, int(self.train_alpha),
                                               self._flat_ext[self.n_grad:].data_ptr(), self.scale, lt.data_ptr(), s))
        dyn._packed_key = None                          # the weights changed under the packed-fragment cache
        self._raise_if_stale()
        return lt[2], p12[:N], x_next, lr


_MLP_FIELDS = ("W1", "b1", "W2", "b2", "W3", "b3")


class SplitTrainer(Trainer):
    """`Trainer` on the GEMM engine (`l2hmc_train_split_grad`, csrc/train_split.hpp): S/T/Q nets of any width and, for the
    image-conditioned VAE sampler of mnist_vae.py:128-178, the shared `encoder_sampler` branch and the decoder-posterior
    energy.  `Trainer(dynamics)` returns one of these whenever the Dynamics runs on the split engine.

    Built-in targets: same objective, schedule and `step()` as `Trainer` (SCGExperiment.ipynb raw 156-181).
    VAE sampler: `sampler_loss_and_grad` / `sampler_step` are mnist_vae.py:185-262's sampler objective -- MH chained
    proposals from the encoder's sample, the distance term weighted by the approximate posterior's variance, global-norm
    clipping at 5 and Adam with the piecewise-constant rate.  The flat parameter vector is
    [XNet | VNet | alpha | encoder_sampler]; the decoder and the VAE encoder are not this optimiser's variables
    (mnist_vae.py:255-262 trains them with separate optimisers on the ELBO / the likelihood)."""

    def __init__(self, dynamics, lr=1e-3, decay_steps=1000, decay_rate=0.96, scale=None,
                 beta1=0.9, beta2=0.999, epsilon=1e-8, seed=0, clip_norm=None):
        if dynamics.hmc:
            raise ValueError("an HMC-mode Dynamics has nothing to train")
        if (dynamics.use_temperature and float(dynamics.temperature) != 1.0) or float(dynamics.anneal_beta or 0.0) != 0.0:
            raise NotImplementedError("training differentiates the plain energy U: a tempered or annealed Dynamics "
                                      "is not supported")
        from .vae import mlp3_struct
        self._mlp3_struct = mlp3_struct
        self.dyn = dynamics
        self.vae = bool(dynamics._vae)
        self.user = bool(getattr(dynamics, "_user", False))     # U, grad U and Hessian-vector products by callback (slow path)
        # the VAE experiment's sampler (mnist_vae.py:185-262): the built-in decoder posterior, or the same model handed
        # over as a plain closure energy(z, aux) with the image-conditioned nets
        self.image_sampler = self.vae or (self.user and dynamics._xw is not None and dynamics._xw["aux_encoder"] is not None)
        self.scale = float(scale) if scale is not None else (1.0 if self.image_sampler else 0.1)
        if self.image_sampler and self.scale != 1.0:
            # mnist_vae.py:207-226 has no scale (its loss is mean(1/v) - mean(v)); the composed-proposal branch of
            # sampler_loss_and_grad forms its cotangents for exactly that objective
            raise ValueError("the image-conditioned sampler objective (mnist_vae.py:207-226) has no `scale`: leave it at 1")
        self.clip_norm = clip_norm if clip_norm is not None else (5.0 if self.image_sampler else None)   # mnist_vae.py:258
        self.lr0, self.decay_steps, self.decay_rate = float(lr), int(decay_steps), float(decay_rate)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.seed = int(seed)
        d, H = dynamics.x_dim, dynamics.H
        dev = dynamics.device
        L = _ffi.lib()
        self.unets = bool(getattr(dynamics, "_user_nets", False))
        self.enc = None if self.unets else dynamics._xw["aux_encoder"]
        enc_s = mlp3_struct(self.enc) if self.enc is not None else None
        if self.unets:
            # Arbitrary callables (dynamics.py:69-79): the variables are whatever the nets expose -- the layer kit's
            # `parameters()` or a torch Module's named_parameters() (`Dynamics.parameters`).  Flat vector [those, in order |
            # alpha]; every one of them becomes a view of theta (the native Adam update is seen by the caller's nets) and its
            # `.grad` a view of the flat gradient (the caller's autograd accumulates straight into what is all-reduced).
            plist = [(k, t) for k, t in dynamics.parameters() if k != "alpha"]
            seen, uniq = set(), []
            for k, t in plist:                       # (one tensor shared by both nets is one variable)
                if t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    uniq.append((k, t))
            if not uniq:
                raise ValueError("the caller-supplied nets expose no parameters (layer-kit parameters() or torch Module "
                                 "named_parameters()): nothing to train")
            for k, t in uniq:
                if t.dtype != torch.float32 or t.device != dev:
                    raise ValueError("net parameter %s must be float32 on %s" % (k, dev))
            self._uparams = uniq
            self.n_grad = sum(int(t.numel()) for _, t in uniq) + 1
        else:
            self.n_grad = _ffi.check(L.l2hmc_train_split_grad_floats(d, H, C.byref(enc_s) if enc_s is not None else None))
        self._alloc_flat(dev)
        self.theta = torch.zeros(self.n_grad, dtype=torch.float32, device=dev)
        self.slots, off = [], 0
        with torch.no_grad():
            if self.unets:
                for k, t in self._uparams:
                    n = int(t.numel())
                    view = self.theta[off:off + n].view(t.shape)
                    view.copy_(t)
                    t.data = view
                    self.slots.append((t, off, n))
                    off += n
            for w in (() if self.unets else (dynamics._xw, dynamics._vw)):
                for name, code in _SHAPES:
                    n = _numel(code, d, H)
                    t = w[name]
                    view = self.theta[off:off + n].view(t.shape)
                    view.copy_(t)
                    t.data = view
                    self.slots.append((t, off, n))
                    off += n
            self.alpha_index = off
            self.train_alpha = bool(getattr(dynamics.alpha, "requires_grad", False))
            self.theta[off].copy_(dynamics.alpha.reshape(()))
            dynamics.alpha.data = self.theta[off].view(dynamics.alpha.shape)
            off += 1
            if self.enc is not None:
                for name in _MLP_FIELDS:
                    t = self.enc[name]
                    n = t.numel()
                    view = self.theta[off:off + n].view(t.shape)
                    view.copy_(t)
                    t.data = view
                    self.slots.append((t, off, n))
                    off += n
            assert off == self.n_grad, (off, self.n_grad)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.global_step = 0
        self._ws = None
        self.variant = 0
        self._io = None
        self._layout = None

    def lr_at(self, step):
        if self.image_sampler and self.decay_steps <= 0:
            return self.lr0
        return Trainer.lr_at(self, step)

    # ---- one proposal + its gradient (accumulated into self.flat) ---------------------------------------------------
    def _propose_grad(self, start, v, direction, n_total, out=None, aux=None, dist_weight=None, dLx_in=None,
                      dx0_out=None, energy_scale=0.0, ediff_out=None, no_accept=False, dLv_in=None, dlogjac_in=None,
                      Lv_out=None, logjac_out=None):
        dyn = self.dyn
        N, d = start.shape
        L = _ffi.lib()
        enc_s = self._mlp3_struct(self.enc) if self.enc is not None else None
        dec_s = self._mlp3_struct(dyn._fn.decoder) if self.vae else None
        need = _ffi.check(L.l2hmc_train_split_workspace_floats(N, d, 4 if self.unets else dyn.H, dyn.T,
                                                               C.byref(enc_s) if enc_s is not None else None,
                                                               C.byref(dec_s) if dec_s is not None else None))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(int(need), dtype=torch.float32, device=dyn.device)
        if out is None:
            out = (torch.empty_like(start), torch.empty(N, dtype=torch.float32, device=dyn.device),
                   torch.empty(N, dtype=torch.float32, device=dyn.device))
        Lx, p, v1 = out
        a = _ffi.L2hmcTrainSplitArgs()
        cb_error = []
        keep_nets = None
        if self.unets:
            # forward by net_cb, reverse by net_vjp_cb (include/l2hmc.h, ABI 6): the parameters' `.grad` are views of self.flat
            # (set here, before the caller's autograd accumulates into them in place)
            for t, off, n in self.slots:
                t.grad = self.flat[off:off + n].view(t.shape)
            ncb, vcb = dyn._net_callbacks(self._ws, direction, None if aux is None else as_device_f32(aux, dyn.device), cb_error)
            keep_nets = (_ffi.NET_CALLBACK(ncb), _ffi.NET_VJP_CALLBACK(vcb))           # alive for the duration of the call
            a.net_cb, a.net_vjp_cb = C.cast(keep_nets[0], C.c_void_p), C.cast(keep_nets[1], C.c_void_p)
            a.H = 0
        else:
            xs = _ffi.L2hmcNet(*[dyn._xw[k].data_ptr() for k in _ffi.NET_FIELDS])
            vs = _ffi.L2hmcNet(*[dyn._vw[k].data_ptr() for k in _ffi.NET_FIELDS])
            a.xnet, a.vnet, a.H = C.pointer(xs), C.pointer(vs), dyn.H
        a.aux_encoder = C.pointer(enc_s) if enc_s is not None else None
        keep = None
        if self.user:
            fn, ws = dyn._fn, self._ws
            base = ws.data_ptr()
            if self.enc is not None:
                if aux is None:
                    raise ValueError("the image-conditioned sampler needs aux=")
                aux = as_device_f32(aux, dyn.device)
                a.aux = aux.data_ptr()

            def view(ptr, n, dd, ld):             # an (n, dd) block of the workspace the library points at
                return ws.as_strided((n, dd), (ld, 1), (ptr - base) // 4)

            def energy_cb(_user, xp, ldx, n, dd, Up, gp, ldg, _stream):
                try:
                    U, g = fn.evaluate(view(xp, n, dd, ldx), 1.0, want_U=bool(Up), want_grad=True, aux=aux)
                    if tuple(g.shape) != (n, dd):
                        raise ValueError("grad_energy must return shape (N, d), got %s" % (tuple(g.shape),))
                    view(gp, n, dd, ldg).copy_(g)
                    if Up:
                        o = (Up - base) // 4
                        ws[o:o + 2 * n].view(torch.float64).copy_(U)
                    return 0
                except Exception as e:            # never let an exception cross the C frame
                    cb_error.append(e)
                    return 1

            def hvp_cb(_user, xp, ldx, up, ldu, n, dd, hp, ldh, _stream):
                try:
                    view(hp, n, dd, ldh).copy_(fn.hvp(view(xp, n, dd, ldx), view(up, n, dd, ldu), aux=aux))
                    return 0
                except Exception as e:
                    cb_error.append(e)
                    return 1
            keep = (_ffi.ENERGY_CALLBACK(energy_cb), _ffi.HVP_CALLBACK(hvp_cb))     # alive for the duration of the call
            a.energy_cb, a.hvp_cb = C.cast(keep[0], C.c_void_p), C.cast(keep[1], C.c_void_p)
        elif self.vae:
            if aux is None:
                raise ValueError("the image-conditioned sampler needs aux=")
            aux = as_device_f32(aux, dyn.device)
            if aux.shape != (N, dyn._fn.n_pix):
                raise ValueError("aux must be (N, %d)" % dyn._fn.n_pix)
            a.decoder, a.aux = C.pointer(dec_s), aux.data_ptr()
        else:
            fn = dyn._fn
            keep = fn.c_struct(dyn.device, 1.0, 0.0)
            a.energy = C.pointer(keep)
            if fn.kind in (_ffi.ENERGY_GAUSS_DENSE, _ffi.ENERGY_GMM):
                a.hess = fn._buffers(dyn.device)["_raw"].data_ptr()
        a.masks, a.trig = dyn._mask.data_ptr(), dyn._trig.data_ptr()
        if dyn.eps_override is None:
            a.alpha, a.eps_host = dyn.alpha.data_ptr(), 0.0
        else:
            a.alpha, a.eps_host = None, float(dyn.eps_override)
        a.n_chains, a.d, a.T = N, d, dyn.T
        a.x, a.v = start.data_ptr(), v.data_ptr()
        a.direction, a.direction_all = direction.data_ptr(), 1
        a.dist_weight = _ffi.ptr(dist_weight)
        a.scale, a.inv_n = self.scale, 1.0 / float(n_total)
        a.dLx_in, a.dx0_out = _ffi.ptr(dLx_in), _ffi.ptr(dx0_out)
        a.Lx, a.p, a.v1 = Lx.data_ptr(), p.data_ptr(), v1.data_ptr()
        # (caller-supplied nets: the library's share of the gradient is the ONE float d loss / d eps, at alpha's slot)
        a.grad = self.flat[self.alpha_index:].data_ptr() if self.unets else self.flat.data_ptr()
        a.workspace, a.workspace_floats = self._ws.data_ptr(), self._ws.numel()
        a.energy_scale, a.ediff_out, a.no_accept = float(energy_scale), _ffi.ptr(ediff_out), int(bool(no_accept))
        a.dLv_in, a.dlogjac_in = _ffi.ptr(dLv_in), _ffi.ptr(dlogjac_in)
        a.Lv_out, a.logjac_out = _ffi.ptr(Lv_out), _ffi.ptr(logjac_out)
        a.gemm_mode = int(getattr(dyn, "gemm_mode", 0))
        a.net_mode = int(getattr(dyn, "net_mode", 0))          # 1: three products per net evaluation (A/B, tests)
        rc = L.l2hmc_train_split_grad(a, _ffi.current_stream(dyn.device))
        if cb_error:
            raise cb_error[0]
        _ffi.check(rc)
        return Lx, p, v1

    def _publish_grads(self):
        for t, off, n in self.slots:
            t.grad = self.flat[off:off + n].view(t.shape)
        if self.train_alpha:
            self.dyn.alpha.grad = (self.flat[self.alpha_index] * torch.exp(self.dyn.alpha.detach())).reshape(self.dyn.alpha.shape)

    def loss_and_grad(self, x, z=None, draws=None):
        if self.image_sampler:
            raise TypeError("the VAE sampler's objective needs the images: use sampler_loss_and_grad")
        loss, Lx, px = Trainer.loss_and_grad(self, x, z, draws)
        self._publish_grads()
        return loss, Lx, px

    def _adam(self, lr):
        """TF1 Adam over the flat vector; alpha's entry holds d/d eps and is turned into d/d alpha first."""
        L = _ffi.lib()
        s = _ffi.current_stream(self.dyn.device)
        i = self.alpha_index
        if self.train_alpha:
            self.flat[i] *= torch.exp(self.theta[i])
        else:
            self.flat[i] = 0.0
        if self.clip_norm is not None:                    # tf.clip_by_global_norm(gradients, 5.0), mnist_vae.py:258
            gn = torch.linalg.vector_norm(self.flat.double()).float()
            self.flat *= torch.clamp(self.clip_norm / torch.clamp(gn, min=1e-30), max=1.0)
        self.global_step += 1
        if not self.train_alpha:
            keep = (self.theta[i].clone(), self.m[i].clone(), self.v[i].clone())
        _ffi.check(L.l2hmc_adam_step(self.theta.data_ptr(), self.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                     self.n_grad, lr, self.beta1, self.beta2, self.epsilon, self.global_step, 0, s))
        if not self.train_alpha:
            self.theta[i], self.m[i], self.v[i] = keep
        self.dyn._packed_key = None

    def step(self, x, u=None):
        if self.image_sampler:
            raise TypeError("the VAE sampler trains with sampler_step(aux, latent_q, log_sigma)")
        dyn = self.dyn
        x = as_device_f32(x, dyn.device)
        N, d = x.shape
        L = _ffi.lib()
        s = _ffi.current_stream(dyn.device)
        io = self._buffers(N, d)
        W = io["W"]
        world = self._world()
        n_total, chain_off = self._shard(N)
        _ffi.check(L.l2hmc_rng_fill(self.seed, 3 * self.global_step, chain_off, N, d, 3, W[1].data_ptr(),
                                    io["dir"].data_ptr(), io["u"].data_ptr(), s))
        W[0].copy_(x)
        self.flat.zero_()
        self._propose_grad(W[0:2].view(2 * N, d), W[2:4].view(2 * N, d), io["dir"][1:3].view(2 * N), n_total,
                           out=(io["Lx"], io["p"], io["v1"]))
        loss = self._reduce_and_loss(io["v1"], N, n_total, world)      # (sharded: the step's ONE collective)
        lr = self.lr_at(self.global_step)
        self._adam(lr)
        x_next = torch.empty_like(x)
        uu = io["u"][0] if u is None else as_device_f32(u, dyn.device)
        _ffi.check(L.l2hmc_mh_select(x.data_ptr(), io["Lx"].data_ptr(), io["p"].data_ptr(), uu.data_ptr(), N, d,
                                     x_next.data_ptr(), s))
        self._raise_if_stale()
        return loss, io["p"][:N].clone(), x_next, lr

    # ---- the VAE experiment's sampler objective (mnist_vae.py:185-226) ---------------------------------------------------
    def sampler_loss_and_grad(self, latent_q, aux, log_sigma, MH=1, stop_gradient=False, draws=None, energy_scale=0.0,
                              random_lf_composition=0):
        """sampler_loss of mnist_vae.py:185-226 and its gradient (left in `.grad` of every sampler parameter and in
        `self.flat`).  MH proposals are chained from `latent_q` with an MH step after each (:204,220); like the
        reference only the LAST proposal's terms enter the loss (inverse_term / other_term / energy_loss are
        re-initialised inside the loop, :186-188) with weight 1 / MH, and unless `stop_gradient` their gradient flows
        back through the earlier proposals.
          energy_scale (:54,214,218,224): + energy_scale / MH * (mean(1 / ed) - mean(ed)),
              ed = (U(final_x) - U(init_x))^2 px + 1e-4.
          random_lf_composition = R > 0 (:50,193-196): every MH iteration is `chain_operator(init_x, dynamics, nb_steps,
              aux, do_mh_step=True)` with nb_steps ~ U{1, ..., R - 1} -- nb_steps composed `propose(log_jac=True)` links
              and ONE accept probability against (init_x, a fresh init_v) (sampler.py:57-85, its quirks included).
        draws: optional list (one dict per MH iteration) of injected randomness (tests): {v, dir, u} for a plain
        proposal; {nb_steps, init_v, v: [..], dir: [..], u} for a composition.
        Returns (loss, latent_T, px of the last iteration)."""
        if not self.image_sampler:
            raise TypeError("sampler_loss_and_grad is the VAE experiment's objective (an image-conditioned sampler)")
        dyn = self.dyn
        dev, gen = dyn.device, dyn.generator
        x = as_device_f32(latent_q, dev)
        aux = as_device_f32(aux, dev)
        N, d = x.shape
        wgt = (1.0 / (torch.exp(2.0 * as_device_f32(log_sigma, dev)) + 1e-4)).contiguous()
        n_total, _ = self._shard(N)
        world = self._world()
        R = int(random_lf_composition)
        es = float(energy_scale)

        def normal():
            return torch.randn((N, d), device=dev, generator=gen)

        def bits():
            return torch.randint(0, 2, (N,), device=dev, dtype=torch.uint8, generator=gen)

        def dirs_of(a):
            return torch.as_tensor(a, device=dev).to(torch.uint8).contiguous()

        # ---- forward: every MH iteration but the last one runs forward only -------------------------------------------
        its = []
        for t in range(MH):
            dr = draws[t] if draws is not None else {}
            u = as_device_f32(dr["u"], dev) if "u" in dr else torch.rand(N, device=dev, generator=gen)
            it = {"x0": x, "u": u}
            if R > 0:
                if R < 2:
                    raise ValueError("random_lf_composition must be 0 or >= 2 (nb_steps ~ U{1..R-1}, mnist_vae.py:194)")
                # nb_steps ~ U{1..R-1} (:194): ONE draw for the whole batch, from a host stream keyed by (seed, step, t)
                # so that every rank of a sharded run composes the same number of links
                K = int(dr["nb_steps"]) if "nb_steps" in dr else int(np.random.RandomState(
                    (self.seed * 1000003 + self.global_step * 7919 + t) % (2 ** 32)).randint(1, R))
                it["init_v"] = as_device_f32(dr["init_v"], dev) if "init_v" in dr else normal()
                it["links"] = []
                xs, lj = x, torch.zeros(N, dtype=torch.float32, device=dev)
                for k in range(K):
                    vk = as_device_f32(dr["v"][k], dev) if "v" in dr else normal()
                    dk = dirs_of(dr["dir"][k]) if "dir" in dr else bits()
                    it["links"].append((xs, vk, dk))
                    o = dyn.run(xs, vk, 0, dyn.T, direction=dk, want=("x", "v", "logjac"), aux=aux)
                    xs, vK, lj = o["x"], o["v"], lj + o["logjac"]
                it.update(Lx=xs, Lv=vK, logjac=lj)
                # the ONE accept probability of the composition (sampler.py:79) against (init_x, init_v)
                U0, U1 = dyn.energy(x, aux=aux).double(), dyn.energy(xs, aux=aux).double()
                val = (U0 + 0.5 * it["init_v"].double().square().sum(1)) - (U1 + 0.5 * vK.double().square().sum(1)) + lj.double()
                p = torch.exp(torch.clamp(val, max=0.0))
                p = torch.where(torch.isfinite(p), p, torch.zeros_like(p))          # dynamics.py:309 (NaN -> 0)
                it.update(U0=U0, U1=U1, val=val, p=p.float())
            else:
                it["v"] = as_device_f32(dr["v"], dev) if "v" in dr else normal()
                it["dir"] = dirs_of(dr["dir"]) if "dir" in dr else bits()
                if t + 1 < MH:
                    o = dyn.run(x, it["v"], 0, dyn.T, direction=it["dir"], u=u, want=("x", "p"), aux=aux)
                    it.update(Lx=o["x"], p=o["p"])
            if t + 1 < MH or R > 0:
                it["acc"] = ((it["p"] - u) >= 0)[:, None]
                x = torch.where(it["acc"], it["Lx"], it["x0"])                       # sampler.py:53-55
            its.append(it)

        # a composition's links, last to first: cotangents on the composition's end point -> d loss / d init_x
        def links_backward(it, dLx, dLv, dlj, want_dx0):
            links, cot = it["links"], dLx
            for k in range(len(links) - 1, -1, -1):
                xk, vk, dk = links[k]
                need = want_dx0 or k > 0
                dx0 = torch.empty((N, d), dtype=torch.float32, device=dev) if need else None
                # (only the LAST link's momentum reaches p_accept: the earlier links' Lv are dropped by propose, :35-36)
                self._propose_grad(xk, vk, dk, float("inf"), aux=aux, dLx_in=cot.contiguous(), dx0_out=dx0, no_accept=True,
                                   dLv_in=dLv if k == len(links) - 1 else None, dlogjac_in=dlj)
                cot = dx0
            return cot

        self.flat.zero_()
        inv = 1.0 / float(n_total * MH)
        last = its[-1]
        # ---- the last iteration: loss terms and their cotangents ---------------------------------------------------------
        want_dx0 = MH > 1 and not stop_gradient
        if R > 0:
            x0, xe, p64, val = last["x0"], last["Lx"], last["p"].double(), last["val"]
            g0, g1 = dyn.grad_energy(x0, aux=aux), dyn.grad_energy(xe, aux=aux)
            dx = xe - x0
            sq = (wgt * dx * dx).sum(1).double()
            v1 = sq * p64 + 1e-4
            dv1 = (-1.0 / (v1 * v1) - 1.0) * inv
            dU = last["U1"] - last["U0"]
            ed = dU * dU * p64 + 1e-4
            de = (-1.0 / (ed * ed) - 1.0) * inv * es
            live = ((val < 0) & torch.isfinite(val) & (p64 > 0)).double()
            lm = ((dv1 * sq + de * dU * dU) * p64 * live).float()
            eu = (de * 2.0 * dU * p64).float()
            dv1p = (dv1 * p64 * 2.0).float()
            dLx = dv1p[:, None] * wgt * dx + (eu - lm)[:, None] * g1
            dLv = (-lm)[:, None] * last["Lv"]
            cot = links_backward(last, dLx, dLv.contiguous(), lm.contiguous(), want_dx0)
            if want_dx0:
                cot = cot - dv1p[:, None] * wgt * dx + (lm - eu)[:, None] * g0
            terms = torch.stack([(1.0 / v1).sum() - v1.sum(), (1.0 / ed).sum() - ed.sum()])
            p_last = last["p"]
        else:
            dx0 = torch.empty((N, d), dtype=torch.float32, device=dev) if want_dx0 else None
            ed = torch.empty(N, dtype=torch.float32, device=dev) if es > 0 else None
            Lx, p_last, v1 = self._propose_grad(last["x0"], last["v"], last["dir"], n_total * MH, aux=aux, dist_weight=wgt,
                                                dx0_out=dx0, energy_scale=es, ediff_out=ed)
            last.update(Lx=Lx, p=p_last)
            last["acc"] = ((p_last - last["u"]) >= 0)[:, None]
            x = torch.where(last["acc"], Lx, last["x0"])
            cot = dx0
            v1 = v1.double()
            terms = torch.stack([(1.0 / v1).sum() - v1.sum(),
                                 ((1.0 / ed.double()).sum() - ed.double().sum()) if es > 0 else torch.zeros((), dtype=torch.float64, device=dev)])
        # (sharded: the loss sums ride behind the gradient in the ONE all-reduce at the end of this call)
        # ---- the earlier iterations, last to first: x_{t+1} = where(p_t - u_t >= 0, Lx_t, x_t) (sampler.py:53-55): the
        #      accepted rows' cotangent goes into iteration t's proposal, the rejected rows' straight on to x_t
        for t in range(MH - 2, -1, -1):
            if stop_gradient:
                break
            it = its[t]
            acc = it["acc"].float()
            dLx = (cot * acc).contiguous()
            need = t > 0
            if R > 0:
                through = links_backward(it, dLx, None, None, need)
            else:
                through = torch.empty((N, d), dtype=torch.float32, device=dev) if need else None
                self._propose_grad(it["x0"], it["v"], it["dir"], float("inf"), aux=aux, dist_weight=wgt, dLx_in=dLx,
                                   dx0_out=through)
            if not need:
                break
            cot = through + cot * (1.0 - acc)
        if world > 1:
            terms, cnt = self._allreduce_flat(terms, N)               # the ONE collective: [gradient | loss sums | count]
            # (the other ranks learn of a changed chain count from this reduced count, one step late: `_check_reduced_count`)
            self._note_reduced_count(self._flat_ext[self.n_grad + 4:self.n_grad + 6], n_total, hi_scale=4096.0)
            loss = (terms[0] + es * terms[1]) / (cnt * MH)
        else:
            loss = (terms[0] + es * terms[1]) * inv
        self._publish_grads()
        self._raise_if_stale()                                        # this rank's own count changed: raise now that the collective is behind us
        return loss, x, p_last

    def sampler_step(self, latent_q, aux, log_sigma, MH=5, stop_gradient=False, energy_scale=0.0, random_lf_composition=0):
        """One update of the sampler's variables (mnist_vae.py:255-261): clipped Adam on sampler_loss."""
        loss, x_T, px = self.sampler_loss_and_grad(latent_q, aux, log_sigma, MH=MH, stop_gradient=stop_gradient,
                                                   energy_scale=energy_scale, random_lf_composition=random_lf_composition)
        lr = self.lr_at(self.global_step)
        self._adam(lr)
        return loss, x_T, px, lr
