"""`Dynamics` -- API of the reference's utils/dynamics.py, executed by fused HIP kernels.

Constructor and method signatures follow utils/dynamics.py:34-309; tensors are float32
torch tensors on a ROCm device instead of `tf.Tensor`s, and every method runs eagerly (one
kernel launch through the C ABI of include/l2hmc.h) instead of building a TF graph.

Differences that are additions, not removals:
  * randomness can be injected (`init_v`, and `direction` / `u` in sampler.propose) so runs
    are reproducible and comparable with the reference on identical draws;
  * `mask` is an assignable (T, d) tensor that is part of `state_dict()` (the reference keeps
    it as an un-checkpointed graph constant, eval_sampler.py:52-59);
  * the energy must be one of the fused targets of `l2hmc_amd.distributions` and the nets the
    S/T/Q architecture of `l2hmc_amd.layers.stq_network` (anything else raises: there is
    no eager fallback).
"""
import math

import numpy as np
import torch

from . import _ffi
from .distributions import ENERGY_USER, EnergyFunction, UserEnergy, as_device_f32
from .layers import default_device, extract_stq

TF_FLOAT = torch.float32
NP_FLOAT = np.float32


class _ZeroNet(object):
    """HMC mode: S = T = Q = 0 (dynamics.py:73-76)."""

    def __call__(self, inp):
        return [torch.zeros_like(inp[0]) for _ in range(3)]


class Dynamics(object):
    def __init__(self,
                 x_dim,
                 energy_function,
                 T=25,
                 eps=0.1,
                 hmc=False,
                 net_factory=None,
                 eps_trainable=True,
                 use_temperature=False,
                 device=None,
                 grad_energy=None):
        """utils/dynamics.py:35-43 (+ `device`, `grad_energy`).  `energy_function`: an energy of
        l2hmc_amd.distributions / l2hmc_amd.vae (fused into the kernels), or ANY callable `fn(x[, aux=]) -> (N,)` on
        ROCm tensors (a target outside that set, e.g. the closure of mnist_vae.py:122-126): then U and grad U come from
        the caller's torch code between launches (`grad_energy(x[, aux=]) -> (N, d)` if given, else autograd) -- the
        slow path, see distributions.UserEnergy."""
        self.x_dim = int(x_dim)
        self.T = int(T)
        self.hmc = bool(hmc)
        self.use_temperature = use_temperature
        self.temperature = 1.0          # the reference's fed placeholder (dynamics.py:47)
        self.device = torch.device(device) if device is not None else default_device()
        self.generator = None           # optional torch.Generator for the momentum draws
        self.variant = 0                # kernel geometry override (0 = auto), see l2hmc.h
        self.eps_override = None        # float: bypass exp(alpha) (exact step size for parity tests)
        self.anneal_beta = 0.0          # AIS bridge (utils/ais.py:46-47): U := (1-b) |x|^2/2 + b U; 0 = off
        self._user_nets = False         # nets outside the fused architecture: evaluated by the caller's torch code (net_cb)
        self.gemm_mode = 3              # GEMM engine, decoder-sized products: 3 = f16x2 planes (exact 2-way f16 split, three f16
        #                                 MFMAs per product block; the trainer runs mode 1 for it), 1 = bf16x3 (exact 3-way bf16
        #                                 split of every fp32 operand, six bf16 MFMAs), 0 = f32-input MFMA: include/l2hmc.h
        self.net_mode = 0               # GEMM-engine trainer: 0 = one launch per net evaluation / per its reverse; 1 = three products each (A/B)

        if not isinstance(energy_function, EnergyFunction):
            if not callable(energy_function):
                raise TypeError("Dynamics needs an energy function (got %r): one of l2hmc_amd.distributions / "
                                "l2hmc_amd.vae, or a callable fn(x[, aux=]) -> (N,)" % (energy_function,))
            energy_function = UserEnergy(energy_function, grad_energy, self.x_dim)
        elif grad_energy is not None:
            raise TypeError("grad_energy= only goes with a caller-supplied energy callable")
        if energy_function.x_dim is not None and energy_function.x_dim != self.x_dim:
            raise ValueError("energy is %d-dimensional, Dynamics x_dim=%d" % (energy_function.x_dim, x_dim))
        self._fn = energy_function

        # eps = exp(alpha), one scalar shared by all steps and both nets (dynamics.py:50-58)
        alpha = torch.tensor(math.log(eps), dtype=torch.float32, device=self.device)
        if not self.hmc:
            self.alpha = torch.nn.Parameter(alpha, requires_grad=bool(eps_trainable))
        else:
            self.alpha = alpha

        self._init_mask()
        self._trig = self._time_table()

        if self.hmc:
            self.XNet = _ZeroNet()
            self.VNet = _ZeroNet()
            self._xw = self._vw = None
            self.H = 0
        else:
            self.XNet = net_factory(x_dim, scope='XNet', factor=2.0)
            self.VNet = net_factory(x_dim, scope='VNet', factor=1.0)
            self._xw = extract_stq(self.XNet, self.x_dim)
            self._vw = extract_stq(self.VNet, self.x_dim)
            if self._xw is None or self._vw is None or self._xw['H'] != self._vw['H']:
                # dynamics.py:69-79: `net_factory` may return ANY callable [a, b, tau, aux] -> [S, T, Q].  Only the notebook's
                # architecture is fused into kernels; anything else is evaluated by the caller's own torch code between the
                # library's launches (L2hmcSplitArgs.net_cb; training: L2hmcTrainSplitArgs.net_vjp_cb) -- the slow path.
                if not (callable(self.XNet) and callable(self.VNet)):
                    raise TypeError("net_factory must return callables net([a, b, tau, aux]) -> [S, T, Q]")
                self._xw = self._vw = None
                self._user_nets = True
                self.H = 0
            else:
                self.H = self._xw['H']
                # mnist_vae.py:134-150 builds ONE `encoder_sampler` and hands it to both nets; the split engine
                # evaluates that shared image branch once per trajectory.
                ax, av = self._xw['aux_encoder'], self._vw['aux_encoder']
                for w in (self._xw, self._vw):
                    for k in _ffi.NET_FIELDS:
                        if w[k].device != self.device:
                            raise ValueError("net parameters live on %s, Dynamics on %s" % (w[k].device, self.device))
                if (ax is None) != (av is None) or (ax is not None and any(
                        ax[k].data_ptr() != av[k].data_ptr() for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'))):
                    # Two DIFFERENT image branches (or one net with a branch and one without): the fused form evaluates ONE
                    # shared branch per trajectory (mnist_vae.py:134-150 builds one `encoder_sampler` for both nets).  Round 6:
                    # such nets are what they are -- callables -- and take the general path (evaluated by their own torch code
                    # between the launches, each reading the images itself) instead of being refused.
                    self._xw = self._vw = None
                    self._user_nets = True
                    self.H = 0
        self._packed = None
        self._packed_key = None
        # Engine choice: the single fused kernel covers the built-in targets with H <= 15 and no
        # image branch; everything else (VAE posterior, wide nets, encoder_sampler(aux) branch) runs
        # on the split engine (own fp32 MFMA GEMMs with fused epilogues + update kernels) of the same library.
        from .vae import ENERGY_VAE
        # _vae: image-conditioned decoder posterior (needs aux=); _split: trajectories run on the GEMM engine
        # (`l2hmc_trajectory_split`) -- the VAE posterior, and the built-in targets whenever the nets are wider than
        # the fused kernel's H <= 15 (nb:51-78 with H != 10; mnist_vae.py:142-167 uses 200)
        self._vae = energy_function.kind == ENERGY_VAE
        self._user = energy_function.kind == ENERGY_USER          # caller's torch code supplies U, grad U (slow path)
        self._split = self._vae or self._user or self._user_nets or (not self.hmc and self.H > 15)
        self._aux_nets = (not self.hmc) and not self._user_nets and self._xw['aux_encoder'] is not None
        if not (self._vae or self._user) and self._aux_nets:
            raise NotImplementedError("an aux (image) branch is only implemented together with the VAE posterior energy "
                                      "or a caller-supplied energy")
        self._split_ws = None
        self._split_key, self._split_aux, self._last_reuse = None, (None, -1), 0
        self._slot1 = [None, None, (None, -1), 0]      # the same four for the second half-batch (`split_streams`)
        self._side_stream = None
        # GEMM engine, decoder posterior: 2 = two half-batches on two HIP streams from 6144 chains.  OFF (1) by default: measured
        # 3.38 ms against 3.09 ms per proposal at config 5 (profiles/r06_config5_two_streams.txt) -- a half-size plane product
        # takes 75 us alone against 90 us for the full one, two of them side by side 91 + 103 us: the main loop is not bound by
        # the CU it runs on (the chip clocks down as the matrix pipes fill), so there is no idle pipe for the other half's
        # HBM-bound phases to hide under.
        self.split_streams = 1

    # ---- masks / time encoding -----------------------------------------------------------------
    def _init_mask(self):
        """dynamics.py:84-93: T masks with floor(d/2) ones, numpy global RNG."""
        rows = []
        for _ in range(self.T):
            ind = np.random.permutation(np.arange(self.x_dim))[:int(self.x_dim / 2)]
            m = np.zeros((self.x_dim,))
            m[ind] = 1
            rows.append(m)
        self.mask = np.stack(rows)

    # the caches of prepared weights (packed fragments of the fused kernels, transposed copies in the GEMM engine's
    # workspace): whoever changes parameters behind torch's version counters (the native Adam) sets `_packed_key = None`
    @property
    def _packed_key(self):
        return self.__dict__.get('_pk')

    @_packed_key.setter
    def _packed_key(self, value):
        self.__dict__['_pk'] = value
        if value is None:
            self.__dict__['_split_key'] = None
            if '_slot1' in self.__dict__:
                self._slot1[1] = None

    def invalidate_caches(self):
        """Forget every prepared copy of the parameters (the packed MFMA fragments of the fused kernels, the transposed
        weights / time table / `encoder_sampler(aux)` rows in the GEMM engine's workspace).  The caches are keyed on each
        parameter tensor's storage address and torch version counter and on the identity + version of `aux`: writes
        that bypass the version counters -- `p.data.copy_()`, `p.data.add_()`, a DLPack consumer or a foreign kernel
        writing into the weights, `aux.data[...] = ...` -- are NOT seen; call this after them.  (The library's own
        optimiser does; in-place torch ops on the parameters themselves, `p.copy_()`, `p.add_()`, bump the counters and
        need nothing.)"""
        self._packed_key = None
        self._split_aux = (None, -1)
        self._slot1[2] = (None, -1)

    @property
    def mask(self):
        return self._mask

    @mask.setter
    def mask(self, value):
        m = torch.as_tensor(np.asarray(value.detach().cpu() if isinstance(value, torch.Tensor) else value,
                                       dtype=np.float32))
        if tuple(m.shape) != (self.T, self.x_dim):
            raise ValueError("mask must be (T, x_dim) = (%d, %d)" % (self.T, self.x_dim))
        self._mask = m.to(self.device).contiguous()

    def _get_mask(self, step):
        m = self._mask[int(step)]
        return m, 1. - m

    def _time_table(self):
        """dynamics.py:99-105 for t = 0..T-1, in float32 like the reference's graph."""
        t = np.arange(self.T, dtype=np.float32)
        ang = np.float32(2 * np.pi) * t / np.float32(self.T)
        tab = np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
        return torch.as_tensor(tab, device=self.device).contiguous()

    def _format_time(self, t, tile=1):
        return self._trig[int(t)].unsqueeze(0).repeat(tile, 1)

    @property
    def eps(self):
        return torch.exp(self.alpha.detach())

    # ---- parameters ------------------------------------------------------------------------------
    def parameters(self):
        """[(name, tensor)] with the reference's variable names."""
        if self.hmc:
            return []
        if self._user_nets:          # arbitrary callables: the layer kit's `parameters()` ([(name, tensor)]), or a torch
            out = [('alpha', self.alpha)]          # Module's named_parameters() under the net's scope; anything else: nothing
            for scope, net in (('XNet', self.XNet), ('VNet', self.VNet)):
                if isinstance(net, torch.nn.Module):
                    out += [('%s/%s' % (scope, k), v) for k, v in net.named_parameters()]
                elif hasattr(net, 'parameters'):
                    out += [kv for kv in net.parameters()
                            if isinstance(kv, tuple) and len(kv) == 2 and isinstance(kv[0], str) and torch.is_tensor(kv[1])]
            return out
        return [('alpha', self.alpha)] + self.XNet.parameters() + self.VNet.parameters()

    def state_dict(self):
        sd = {k: v.detach().cpu().clone() for k, v in self.parameters()}
        sd['mask'] = self._mask.cpu().clone()
        if self.hmc:
            sd['alpha'] = self.alpha.cpu().clone()
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, v in self.parameters():
                v.copy_(sd[k].to(v.device))
            if self.hmc and 'alpha' in sd:
                self.alpha = sd['alpha'].to(self.device)
        self.mask = sd['mask']
        self._packed_key = None

    def _net_callbacks(self, ws, direction, aux, cb_error):
        """(net_cb, net_vjp_cb): the Python bodies of include/l2hmc.h's L2hmcNetCallback / L2hmcNetVjpCallback for this
        Dynamics' caller-supplied nets (dynamics.py:69-79: any callable [a, b, tau, aux] -> [S, T, Q]), working on views of the
        workspace tensor `ws` the library points into.  Chain n's time input is row `it` of the schedule if it runs forward, row
        T - 1 - it otherwise (dynamics.py:99-105, :285).  net_cb evaluates under no_grad (a net whose parameters require
        grad must not hang a graph on the persistent workspace); net_vjp_cb RE-evaluates the net on the kept inputs with
        autograd on and runs ONE backward: the inputs' cotangents go to the library, the parameters' gradients accumulate
        in their `.grad` (which the trainer aliases to its flat gradient vector)."""
        base, T = ws.data_ptr(), self.T
        fwd_mask = (direction != 0) if direction is not None else None

        def inputs(net, abp, ldab, n, dd, it, dall):
            ab = ws.as_strided((n, 2 * dd), (ldab, 1), (abp - base) // 4)
            if fwd_mask is not None:
                tau = self._trig[torch.where(fwd_mask, it, T - 1 - it)]
            else:
                tau = self._trig[it if dall else T - 1 - it].expand(n, 2)
            return (self.XNet if net == 0 else self.VNet), ab[:, :dd], ab[:, dd:], tau

        def outputs(stq, n, dd):
            if len(stq) != 3:
                raise ValueError("a net must return [S, T, Q], got %d outputs" % len(stq))
            res = []
            for i, t in enumerate(stq):                # (HMC-style nets return plain zeros, dynamics.py:73-76)
                t = torch.as_tensor(t, dtype=torch.float32, device=ws.device)
                if t.dim() != 0 and tuple(t.shape) != (n, dd):
                    raise ValueError("net output %d must be (N, d) = (%d, %d), got %s" % (i, n, dd, tuple(t.shape)))
                res.append(t)
            return res

        @torch.no_grad()
        def net_cb(_user, net, abp, ldab, n, dd, it, _dirp, dall, outp, _stream):
            try:
                fn, a, b, tau = inputs(net, abp, ldab, n, dd, it, dall)
                out = ws.as_strided((n, 3 * dd), (3 * dd, 1), (outp - base) // 4)
                for i, t in enumerate(outputs(fn([a, b, tau, aux]), n, dd)):
                    out[:, i * dd:(i + 1) * dd] = t
                return 0
            except Exception as e:                     # never let an exception cross the C frame
                cb_error.append(e)
                return 1

        def net_vjp_cb(_user, net, abp, ldab, n, dd, it, _dirp, dall, dstqp, dabp, ld_dab, _stream):
            try:
                with torch.no_grad():
                    fn, a, b, tau = inputs(net, abp, ldab, n, dd, it, dall)
                    a, b = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
                    dstq = ws.as_strided((n, 3 * dd), (3 * dd, 1), (dstqp - base) // 4)
                    dab = ws.as_strided((n, 2 * dd), (ld_dab, 1), (dabp - base) // 4)
                with torch.enable_grad():
                    stq = outputs(fn([a, b, tau, aux]), n, dd)
                    live = [(t, dstq[:, i * dd:(i + 1) * dd]) for i, t in enumerate(stq) if t.requires_grad]
                    if live:
                        torch.autograd.backward([t for t, _ in live], [g.clone() for _, g in live])
                with torch.no_grad():
                    for j, t in enumerate((a, b)):
                        if t.grad is None:
                            dab[:, j * dd:(j + 1) * dd] = 0.0
                        else:
                            dab[:, j * dd:(j + 1) * dd] = t.grad
                return 0
            except Exception as e:
                cb_error.append(e)
                return 1
        return net_cb, net_vjp_cb

    def _packed_nets(self):
        """Fragment-ordered weights for the kernels; re-packed when any parameter changed."""
        if self.hmc:
            return None
        key = tuple((w[k].data_ptr(), w[k]._version) for w in (self._xw, self._vw) for k in _ffi.NET_FIELDS)
        if key != self._packed_key:
            L = _ffi.lib()
            n = _ffi.check(L.l2hmc_packed_nets_floats(self.x_dim, self.H))
            if self._packed is None or self._packed.numel() != n:
                self._packed = torch.empty(n, dtype=torch.float32, device=self.device)
            structs = []
            for w in (self._xw, self._vw):
                for k in _ffi.NET_FIELDS:
                    if not w[k].is_contiguous():
                        raise ValueError("net parameter %s must be contiguous" % k)
                structs.append(_ffi.L2hmcNet(*[w[k].data_ptr() for k in _ffi.NET_FIELDS]))
            _ffi.check(L.l2hmc_pack_nets(structs[0], structs[1], self.x_dim, self.H,
                                         self._packed.data_ptr(), _ffi.current_stream(self.device)))
            self._packed_key = key
        return self._packed

    # ---- the fused trajectory ----------------------------------------------------------------------
    @staticmethod
    def _aux_key(aux):
        """What identifies the CONTENT of the conditioning images between two launches: the storage address, the version
        counter torch bumps on every in-place write (shared by all views of the storage -- `as_device_f32` hands a fresh
        `detach()` view to every launch, so object identity would never match) and the shape.  `_split_aux` keeps a
        reference to the tensor, so the address cannot be recycled for other data in between."""
        return (aux.data_ptr(), aux._version, tuple(aux.shape))

    def _check_aux(self, aux):
        if self._vae or (self._user and self._aux_nets):
            if aux is None:
                raise ValueError("this Dynamics is image-conditioned (mnist_vae.py): pass aux=")
        elif self._user or self._user_nets:
            pass                   # forwarded to the caller's energy function / nets when given
        elif aux is not None:
            raise ValueError("aux= is only meaningful for an aux-conditioned model (mnist_vae.py)")

    def _run_split(self, x, v, step_begin, n_steps, direction, direction_all, u, want, aux):
        """Launch `l2hmc_trajectory_split` (include/l2hmc.h)."""
        import ctypes as C
        from .vae import mlp3_struct
        x = as_device_f32(x, self.device)
        v = as_device_f32(v, self.device)
        N, d = x.shape
        if v.shape != x.shape or d != self.x_dim:
            raise ValueError("x, v must be (N, %d)" % self.x_dim)
        if self._vae:
            aux = as_device_f32(aux, self.device)
            if aux.shape != (N, self._fn.n_pix):
                raise ValueError("aux must be (N, %d)" % self._fn.n_pix)
        elif self._user and self._aux_nets:
            aux = as_device_f32(aux, self.device)
            if aux.shape != (N, self._xw['aux_encoder']['dims'][0]):
                raise ValueError("aux must be (N, %d)" % self._xw['aux_encoder']['dims'][0])
        out = {}
        for k in ('x', 'v', 'x_next'):
            if k in want:
                out[k] = torch.empty_like(x)
        if 'x_next' in want:
            out.setdefault('p', None)
        for k in ('logjac', 'p'):
            if k in want or k in out:
                out[k] = torch.empty(N, dtype=torch.float32, device=x.device)
        if direction is not None:
            direction = direction.to(device=x.device, dtype=torch.uint8).contiguous()
        if u is not None:
            u = as_device_f32(u, self.device)
        # Two half-batches on two HIP streams (round 6 experiment, `split_streams = 2`; off by default -- see __init__): chains
        # never interact, so rows [0, h) and [h, N) are two independent trajectories.  Alone, every decoder-sized product is 256
        # tiles on 256 CUs marching in phase -- 70 us of matrix pipe, then 20 us in which all of them write their epilogues at
        # HBM speed with the pipe idle; the idea was that side by side (128 tiles each) one half's HBM-bound phases fall under
        # the other half's MFMA-bound ones.  Only for the built-in decoder posterior (no host callbacks between the launches)
        # and from 2 x 3072 chains, so that each half still takes the pre-split-planes form.
        h = (N // 2 + 255) // 256 * 256
        if int(self.split_streams) >= 2 and self._vae and not self.hmc and h >= 3072 and N - h >= 3072:
            cur = torch.cuda.current_stream(x.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=x.device)
            side = self._side_stream
            side.wait_stream(cur)                       # (the inputs were produced on the caller's stream)

            def rows(t, lo, hi):
                return None if t is None else t[lo:hi]
            self._launch_split(x[:h], v[:h], step_begin, n_steps, rows(direction, 0, h), direction_all, rows(u, 0, h),
                               aux[:h], {k: t[:h] for k, t in out.items()})
            with torch.cuda.stream(side):
                self._swap_slot()
                try:
                    self._launch_split(x[h:], v[h:], step_begin, n_steps, rows(direction, h, N), direction_all, rows(u, h, N),
                                       aux[h:], {k: t[h:] for k, t in out.items()})
                finally:
                    self._swap_slot()
            cur.wait_stream(side)                       # (the results are consumed on the caller's stream)
            return out
        self._launch_split(x, v, step_begin, n_steps, direction, direction_all, u, aux, out)
        return out

    def _swap_slot(self):
        """exchange the GEMM engine's workspace + its reuse keys with those of the second half-batch"""
        cur = [self._split_ws, self.__dict__.get('_split_key'), self._split_aux, self._last_reuse]
        (self._split_ws, self.__dict__['_split_key'], self._split_aux, self._last_reuse), self._slot1 = self._slot1, cur

    def _launch_split(self, x, v, step_begin, n_steps, direction, direction_all, u, aux, out):
        """ONE `l2hmc_trajectory_split` call on torch's current stream: (N, d) row blocks in, `out`'s tensors filled"""
        import ctypes as C
        from .vae import mlp3_struct
        N, d = x.shape
        dec = mlp3_struct(self._fn.decoder) if self._vae else None
        if self.hmc or self._user_nets:               # (HMC, either direction: the inverse leapfrog is the step with -eps)
            xs = vs = enc = None
        else:
            xs = _ffi.L2hmcNet(*[self._xw[k].data_ptr() for k in _ffi.NET_FIELDS])
            vs = _ffi.L2hmcNet(*[self._vw[k].data_ptr() for k in _ffi.NET_FIELDS])
            enc = mlp3_struct(self._xw['aux_encoder']) if self._xw['aux_encoder'] is not None else None
        L = _ffi.lib()
        H_ws = 4 if self._user_nets else max(self.H, 1)    # (caller-supplied nets: the library plans no hidden activations)
        need = _ffi.check(L.l2hmc_split_workspace_floats(N, d, H_ws, self.T,
                                                         C.byref(enc) if enc is not None else None,
                                                         C.byref(dec) if dec is not None else None))
        if self._split_ws is None or self._split_ws.numel() < need:
            self._split_ws = torch.empty(int(need), dtype=torch.float32, device=self.device)
        a = _ffi.L2hmcSplitArgs()
        a.xnet = C.pointer(xs) if xs is not None else None
        a.vnet = C.pointer(vs) if vs is not None else None
        # (a caller-supplied energy anneals itself inside the callback -- evaluate(..., anneal_beta=) below; the library
        #  refuses bce_scale next to energy_cb)
        a.H, a.hmc, a.bce_scale = H_ws, int(self.hmc), (0.0 if self._user else float(self.anneal_beta))
        a.aux_encoder = C.pointer(enc) if enc is not None else None
        cb_error = []
        if self._vae:
            a.decoder, a.aux = C.pointer(dec), aux.data_ptr()
        elif self._user:
            ws, temp, beta = self._split_ws, (float(self.temperature) if self.use_temperature else 1.0), float(self.anneal_beta)
            base = ws.data_ptr()

            def energy_cb(_user, xp, ldx, n, dd, Up, gp, ldg, _stream):
                # (the library hands over addresses inside the workspace tensor: view them, no copies besides the
                #  results; torch enqueues on its current stream, which is the stream the library was given)
                try:
                    xv = ws.as_strided((n, dd), (ldx, 1), (xp - base) // 4)
                    U, g = self._fn.evaluate(xv, temp, want_U=bool(Up), want_grad=True, aux=aux, anneal_beta=beta)
                    if tuple(g.shape) != (n, dd):
                        raise ValueError("grad_energy must return shape (N, d), got %s" % (tuple(g.shape),))
                    ws.as_strided((n, dd), (ldg, 1), (gp - base) // 4).copy_(g)
                    if Up:
                        o = (Up - base) // 4
                        ws[o:o + 2 * n].view(torch.float64).copy_(U)
                    return 0
                except Exception as e:                       # never let an exception cross the C frame
                    cb_error.append(e)
                    return 1
            cb = _ffi.ENERGY_CALLBACK(energy_cb)             # (kept alive by this frame for the duration of the call)
            a.energy_cb = C.cast(cb, C.c_void_p)
            if self._aux_nets:
                a.aux = aux.data_ptr()
        else:                                        # built-in target (utils/distributions.py) under wide nets
            en = self._fn.c_struct(x.device, float(self.temperature) if self.use_temperature else 1.0, self.anneal_beta)
            a.energy = C.pointer(en)
        if self._user_nets:
            # the caller's nets (dynamics.py:69-79: any callable [a, b, tau, aux] -> [S, T, Q]): evaluated here, on views of the
            # workspace, between the library's launches; chain n's time input is row `it` of the schedule if it runs forward,
            # row T - 1 - it otherwise (dynamics.py:99-105, :285)
            net_cb = self._net_callbacks(self._split_ws, direction, aux, cb_error)[0]
            ncb = _ffi.NET_CALLBACK(net_cb)              # (kept alive by this frame for the duration of the call)
            a.net_cb = C.cast(ncb, C.c_void_p)
        a.masks, a.trig = self._mask.data_ptr(), self._trig.data_ptr()
        if self.eps_override is None:
            a.alpha, a.eps_host = self.alpha.data_ptr(), 0.0
        else:
            a.alpha, a.eps_host = None, float(self.eps_override)
        a.n_chains, a.d, a.T = N, d, self.T
        a.step_begin, a.n_steps = int(step_begin), int(n_steps)
        a.x, a.v = x.data_ptr(), v.data_ptr()
        a.direction, a.direction_all, a.u = _ffi.ptr(direction), int(direction_all), _ffi.ptr(u)
        a.x_out, a.v_out = _ffi.ptr(out.get('x')), _ffi.ptr(out.get('v'))
        a.logjac_out, a.p_out, a.x_next = _ffi.ptr(out.get('logjac')), _ffi.ptr(out.get('p')), _ffi.ptr(out.get('x_next'))
        a.workspace, a.workspace_floats = self._split_ws.data_ptr(), self._split_ws.numel()
        # What the workspace still holds from the previous launch (L2hmcSplitArgs.reuse): the prepared weights while no
        # parameter changed (same storage and torch version counters; the native optimiser resets `_packed_key`, which
        # clears this too), the image branch while the same `aux` storage is passed again unmodified (`_aux_key`).
        wkey = None
        img = self._vae or (self._user and self._aux_nets)          # an image branch whose result can be reused
        if not self.hmc and not self._user_nets:
            ws_ = [self._xw[k] for k in _ffi.NET_FIELDS] + [self._vw[k] for k in _ffi.NET_FIELDS]
            for m3 in (self._xw['aux_encoder'], self._fn.decoder if self._vae else None):
                if m3 is not None:
                    ws_ += [m3[k] for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3')]
            # (gemm_mode: the decoder weights sit in the workspace as the planes of THAT arithmetic)
            wkey = (self._split_ws.data_ptr(), N, self.T, int(self.gemm_mode)) + tuple((t.data_ptr(), t._version) for t in ws_)
        reuse = 0
        if wkey is not None and wkey == self._split_key:
            reuse |= 1
            if img and self._split_aux[0] is not None and self._aux_key(aux) == self._split_aux[1]:
                reuse |= 2
        a.reuse = self._last_reuse = reuse
        a.gemm_mode = int(self.gemm_mode)
        rc = L.l2hmc_trajectory_split(a, _ffi.current_stream(x.device))
        if cb_error:
            raise cb_error[0]
        _ffi.check(rc)
        self._split_key = wkey
        self._split_aux = (aux, self._aux_key(aux)) if (img and aux is not None) else (None, -1)

    def _run_split_chain(self, x, v, step_begin, n_steps, direction, direction_all, u, want, M, rng, aux):
        """M chained proposals on the split engine (the sampler loop of mnist_vae.py:185-224 / eval_sampler.py): a
        proposal there is milliseconds of GEMM work, so the loop stays on the host -- per proposal one
        `l2hmc_rng_fill` (the library's Philox stream: same draws as the fused kernels' in-kernel RNG, keyed by
        proposal index and GLOBAL chain index) and one `l2hmc_trajectory_split`; nothing is copied to the host."""
        x = as_device_f32(x, self.device)
        N, d = x.shape
        L = _ffi.lib()
        if 'x_next' not in want:
            raise ValueError("chained proposals need want=('x_next', ...): the MH step links them")
        need_u = u is None
        if rng is None and (v is None or need_u):
            raise ValueError("v and u are required unless rng= is given")
        ps, ljs, hist = [], [], []
        o = None
        for m in range(M):
            vm = None if v is None else (v[m] if M > 1 else v)
            um = None if u is None else (u[m] if M > 1 else u)
            dm = None if direction is None else (direction[m] if M > 1 else direction)
            if rng is not None and (vm is None or um is None or (dm is None and not self.hmc and rng.get('direction', True))):
                fv = torch.empty((N, d), dtype=torch.float32, device=self.device) if vm is None else None
                fd = (torch.empty(N, dtype=torch.uint8, device=self.device)
                      if (dm is None and not self.hmc and rng.get('direction', True)) else None)
                fu = torch.empty(N, dtype=torch.float32, device=self.device) if um is None else None
                _ffi.check(L.l2hmc_rng_fill(int(rng['seed']) & 0xFFFFFFFFFFFFFFFF, int(rng.get('proposal0', 0)) + m,
                                            int(rng.get('chain_offset', 0)), N, d, 1, _ffi.ptr(fv), _ffi.ptr(fd),
                                            _ffi.ptr(fu), _ffi.current_stream(self.device)))
                vm = fv if vm is None else vm
                dm = fd if fd is not None else dm
                um = fu if um is None else um
            o = self._run_split(x, vm, step_begin, n_steps, dm, direction_all, um,
                                tuple(k for k in want if k != 'x_hist'), aux)
            x = o['x_next']
            if 'p' in o:
                ps.append(o['p'])
            if 'logjac' in o:
                ljs.append(o['logjac'])
            if 'x_hist' in want:
                hist.append(x)
        out = dict(o)
        if M > 1:
            if ps:
                out['p'] = torch.stack(ps)
            if ljs:
                out['logjac'] = torch.stack(ljs)
        if 'x_hist' in want:
            out['x_hist'] = torch.stack(hist)
        return out

    def _randn_like(self, x):
        return torch.randn(x.shape, dtype=torch.float32, device=x.device, generator=self.generator)

    def run(self, x, v, step_begin, n_steps, direction=None, direction_all=1, u=None,
            want=('x', 'v', 'logjac'), n_proposals=1, rng=None, aux=None, ais=None):
        """Launch `l2hmc_trajectory` (include/l2hmc.h).  Returns a dict of the requested
        outputs among x, v, logjac, p, x_next, x_hist.  With n_proposals = M > 1 the kernel
        runs M chained proposals (persistent sampler loop): v is (M, N, d), direction and u
        are (M, N), p / logjac come back as (M, N), x_hist as (M, N, d).
        rng = dict(seed=, proposal0=0, chain_offset=0): inputs passed as None among v /
        direction / u are drawn in-kernel from the Philox stream (include/l2hmc.h)."""
        M = int(n_proposals)
        if self._split:
            if M != 1 or rng is not None:
                return self._run_split_chain(x, v, step_begin, n_steps, direction, direction_all, u, want, M, rng, aux)
            return self._run_split(x, v, step_begin, n_steps, direction, direction_all, u, want, aux)
        x = as_device_f32(x, self.device)
        N, d = x.shape
        lead = (M,) if M > 1 else ()
        flags = 0
        if rng is not None:
            flags = (_ffi.RNG_V if v is None else 0) | (_ffi.RNG_U if (u is None and rng.get('u', True)) else 0)
            if direction is None and not self.hmc and rng.get('direction', True):
                flags |= _ffi.RNG_DIR
        if v is not None:
            v = as_device_f32(v, self.device)
            if tuple(v.shape) != lead + (N, d):
                raise ValueError("v must be %s" % (lead + (N, d),))
        elif not flags & _ffi.RNG_V:
            raise ValueError("v is required unless rng= is given")
        if d != self.x_dim:
            raise ValueError("x must be (N, %d)" % self.x_dim)
        out = {}
        if 'x' in want:
            out['x'] = torch.empty_like(x)
        if 'v' in want:
            out['v'] = torch.empty_like(x)
        for k in ('logjac', 'p'):
            if k in want:
                out[k] = torch.empty(lead + (N,), dtype=torch.float32, device=x.device)
        if 'x_next' in want:
            out['x_next'] = torch.empty_like(x)
            out.setdefault('p', torch.empty(lead + (N,), dtype=torch.float32, device=x.device))
        if 'x_hist' in want:
            out['x_hist'] = torch.empty((M, N, d), dtype=torch.float32, device=x.device)
        if direction is not None:
            direction = direction.to(device=x.device, dtype=torch.uint8).contiguous()
            if tuple(direction.shape) != lead + (N,):
                raise ValueError("direction must be %s" % (lead + (N,),))
        if u is not None:
            u = as_device_f32(u, self.device)
            if tuple(u.shape) != lead + (N,):
                raise ValueError("u must be %s" % (lead + (N,),))
        a = _ffi.L2hmcTrajectoryArgs()
        a.packed_nets = _ffi.ptr(self._packed_nets())
        a.energy = self._fn.c_struct(x.device, self.temperature if self.use_temperature else 1.0, self.anneal_beta)
        a.masks, a.trig = self._mask.data_ptr(), self._trig.data_ptr()
        if self.eps_override is None:
            a.alpha, a.eps_host = self.alpha.data_ptr(), 0.0
        else:
            a.alpha, a.eps_host = None, float(self.eps_override)
        a.n_chains, a.d, a.H, a.T = N, d, self.H, self.T
        a.step_begin, a.n_steps = int(step_begin), int(n_steps)
        a.x, a.v = x.data_ptr(), _ffi.ptr(v)
        if rng is not None:
            a.rng_flags, a.rng_seed = flags, int(rng['seed']) & 0xFFFFFFFFFFFFFFFF
            a.rng_proposal0, a.chain_offset = int(rng.get('proposal0', 0)), int(rng.get('chain_offset', 0))
        a.direction, a.direction_all = _ffi.ptr(direction), int(direction_all)
        a.u = _ffi.ptr(u)
        a.x_out, a.v_out = _ffi.ptr(out.get('x')), _ffi.ptr(out.get('v'))
        a.logjac_out, a.p_out = _ffi.ptr(out.get('logjac')), _ffi.ptr(out.get('p'))
        a.x_next = _ffi.ptr(out.get('x_next'))
        if ais is not None:          # AIS mode of the persistent loop (include/l2hmc.h): proposal m = anneal step m
            a.ais_beta, a.ais_dbeta = ais['beta'].data_ptr(), float(ais['dbeta'])
            a.ais_refreshment = float(ais['refreshment'])
            a.ais_v0 = _ffi.ptr(ais.get('v0'))
            a.ais_w, a.ais_alpha = ais['w'].data_ptr(), ais['alpha'].data_ptr()
        a.x_hist = _ffi.ptr(out.get('x_hist'))
        a.variant = int(self.variant)
        a.n_proposals = M
        _ffi.check(_ffi.lib().l2hmc_trajectory(a, _ffi.current_stream(x.device)))
        return out

    # ---- reference API -------------------------------------------------------------------------------
    def kinetic(self, v):
        """dynamics.py:107-108."""
        return 0.5 * torch.sum(torch.square(v), dim=1)

    def energy(self, x, aux=None):
        """dynamics.py:203-212."""
        self._check_aux(aux)
        if self._vae:
            return self._fn.evaluate(x, aux=aux, anneal_beta=self.anneal_beta)[0]
        if self._user:
            return self._fn.evaluate(as_device_f32(x, self.device), self.temperature if self.use_temperature else 1.0,
                                     aux=aux, anneal_beta=self.anneal_beta)[0].to(torch.float32)
        return self._fn.evaluate(x, self.temperature if self.use_temperature else 1.0,
                                 anneal_beta=self.anneal_beta)[0]

    def grad_energy(self, x, aux=None):
        """dynamics.py:217-218 (analytic, computed by the HIP energy kernel)."""
        self._check_aux(aux)
        if self._vae:
            return self._fn.evaluate(x, want_U=False, want_grad=True, aux=aux, anneal_beta=self.anneal_beta)[1]
        if self._user:
            return self._fn.evaluate(as_device_f32(x, self.device), self.temperature if self.use_temperature else 1.0,
                                     want_U=False, want_grad=True, aux=aux, anneal_beta=self.anneal_beta)[1]
        return self._fn.evaluate(x, self.temperature if self.use_temperature else 1.0,
                                 want_U=False, want_grad=True, anneal_beta=self.anneal_beta)[1]

    def hamiltonian(self, x, v, aux=None):
        """dynamics.py:214-215."""
        return self.energy(x, aux=aux) + self.kinetic(v)

    def _forward_step(self, x, v, step, aux=None):
        """dynamics.py:115-157 -> (x_o, v_o, log_jac_contrib)."""
        self._check_aux(aux)
        o = self.run(x, v, int(step), 1, direction_all=1, aux=aux)
        return o['x'], o['v'], o['logjac']

    def _backward_step(self, x_o, v_o, step, aux=None):
        """dynamics.py:159-201 -> (x, v, log_jac_contrib)."""
        self._check_aux(aux)
        o = self.run(x_o, v_o, self.T - 1 - int(step), 1, direction_all=0, aux=aux)
        return o['x'], o['v'], o['logjac']

    def _trajectory(self, x, init_v, aux, log_jac, direction_all):
        self._check_aux(aux)
        x = as_device_f32(x, self.device)
        v = self._randn_like(x) if init_v is None else init_v      # dynamics.py:247-250
        want = ('x', 'v', 'logjac') if log_jac else ('x', 'v', 'p')
        o = self.run(x, v, 0, self.T, direction_all=direction_all, want=want, aux=aux)
        return o['x'], o['v'], (o['logjac'] if log_jac else o['p'])

    def forward(self, x, init_v=None, aux=None, log_path=False, log_jac=False):
        """dynamics.py:246-272 -> (X, V, p_accept) or (X, V, log_jac)."""
        return self._trajectory(x, init_v, aux, log_jac, 1)

    def backward(self, x, init_v=None, aux=None, log_jac=False):
        """dynamics.py:274-300."""
        return self._trajectory(x, init_v, aux, log_jac, 0)

    def p_accept(self, x0, v0, x1, v1, log_jac, aux=None):
        """dynamics.py:302-309."""
        self._check_aux(aux)
        x0, v0, x1, v1 = (as_device_f32(t, self.device) for t in (x0, v0, x1, v1))
        lj = as_device_f32(log_jac, self.device)
        N, d = x0.shape
        p = torch.empty(N, dtype=torch.float32, device=x0.device)
        if self._vae or self._user:        # energies from the decoder GEMMs / the caller's code, then one small kernel
            U0, U1 = self.energy(x0, aux=aux).contiguous(), self.energy(x1, aux=aux).contiguous()
            _ffi.check(_ffi.lib().l2hmc_p_accept_energies(U0.data_ptr(), v0.data_ptr(), U1.data_ptr(), v1.data_ptr(),
                                                          lj.data_ptr(), N, d, p.data_ptr(),
                                                          _ffi.current_stream(x0.device)))
            return p
        e = self._fn.c_struct(x0.device, self.temperature if self.use_temperature else 1.0, self.anneal_beta)
        _ffi.check(_ffi.lib().l2hmc_p_accept(e, x0.data_ptr(), v0.data_ptr(), x1.data_ptr(),
                                             v1.data_ptr(), lj.data_ptr(), N, d, p.data_ptr(),
                                             _ffi.current_stream(x0.device)))
        return p
