"""Sampling operators -- API of the reference's utils/sampler.py:28-85.

`propose` launches ONE fused kernel per call: every chain runs only in its drawn direction
(the reference runs forward AND backward on all chains and discards half, sampler.py:35-44),
and the accept probability and the Metropolis select are part of the same kernel.

Randomness (all optional, keyword-only, for reproducibility / parity on identical draws):
  direction : (N,) 0/1, 1 = forward       -- the draw of sampler.py:34
  v         : (N, d) momenta, or a pair (v_fwd, v_bwd) -- the reference draws one N(0, I)
              per direction (dynamics.py:247-250, 275-278); a chain uses the one of its own
              direction
  u         : (N,) accept uniforms         -- the draw of sampler.py:54
"""
import torch

from . import _ffi
from .distributions import as_device_f32

TF_FLOAT = torch.float32


def _rand(shape, dynamics):
    return torch.rand(shape, dtype=torch.float32, device=dynamics.device, generator=dynamics.generator)


def propose(x, dynamics, init_v=None, aux=None, do_mh_step=False, log_jac=False, *,
            direction=None, v=None, u=None):
    """sampler.py:28-51 -> (Lx, Lv, px, outputs).

    HMC mode (:29-31): one forward trajectory from `init_v` (or a fresh draw); `outputs` always
    holds the MH-selected state.  L2HMC mode (:33-51): direction-mixed proposal; as in the
    reference `init_v` is NOT the start momentum (each proposal draws its own) and only
    controls whether `Lv` is returned; with `log_jac=True` the third return is the
    log-Jacobian, not a probability (used by `chain_operator`)."""
    dynamics._check_aux(aux)
    x = as_device_f32(x, dynamics.device)
    N = x.shape[0]
    if dynamics.hmc:
        v0 = init_v if init_v is not None else (v if v is not None else dynamics._randn_like(x))
        uu = u if u is not None else _rand((N,), dynamics)
        o = dynamics.run(x, v0, 0, dynamics.T, direction_all=1, u=uu, want=('x', 'v', 'p', 'x_next'), aux=aux)
        return o['x'], o['v'], o['p'], [o['x_next']]

    if direction is None:
        direction = torch.randint(0, 2, (N,), device=dynamics.device, dtype=torch.uint8,
                                  generator=dynamics.generator)
    else:
        direction = torch.as_tensor(direction, device=dynamics.device).to(torch.uint8)
    if v is None:
        v0 = dynamics._randn_like(x)
    elif isinstance(v, (tuple, list)):
        v_f, v_b = (as_device_f32(t, dynamics.device) for t in v)
        v0 = torch.where(direction.bool().unsqueeze(1), v_f, v_b)
    else:
        v0 = v
    want = ['x', 'v', 'logjac' if log_jac else 'p']
    uu = None
    if do_mh_step and not log_jac:
        uu = u if u is not None else _rand((N,), dynamics)
        want.append('x_next')
    o = dynamics.run(x, v0, 0, dynamics.T, direction=direction, u=uu, want=tuple(want), aux=aux)
    Lv = o['v'] if init_v is not None else None
    px = o['logjac'] if log_jac else o['p']
    outputs = []
    if do_mh_step:
        if log_jac:     # reference quirk: tf_accept on a log-Jacobian "probability" (:44-49)
            outputs.append(tf_accept(x, o['x'], px, u=u, dynamics=dynamics))
        else:
            outputs.append(o['x_next'])
    return o['x'], Lv, px, outputs


def tf_accept(x, Lx, px, u=None, dynamics=None):
    """sampler.py:53-55: rows with px - u >= 0 take the proposal (HIP kernel l2hmc_mh_select)."""
    x = as_device_f32(x)
    Lx = as_device_f32(Lx, x.device)
    px = as_device_f32(px, x.device)
    N, d = x.shape
    if u is None:
        u = torch.rand((N,), dtype=torch.float32, device=x.device,
                       generator=None if dynamics is None else dynamics.generator)
    u = as_device_f32(u, x.device)
    out = torch.empty_like(x)
    _ffi.check(_ffi.lib().l2hmc_mh_select(x.data_ptr(), Lx.data_ptr(), px.data_ptr(), u.data_ptr(),
                                          N, d, out.data_ptr(), _ffi.current_stream(x.device)))
    return out


def chain_operator(init_x, dynamics, nb_steps, aux=None, init_v=None, do_mh_step=False, *,
                   directions=None, vs=None, u=None):
    """sampler.py:57-85: `nb_steps` composed proposals with summed log-Jacobians, then one
    accept probability against (init_x, init_v).  As in the reference, the momentum threaded
    through the composition is only the returned `Lv`; every proposal draws fresh momenta.
    `directions` / `vs`: optional per-step injected draws (lists of length nb_steps)."""
    dynamics._check_aux(aux)
    init_x = as_device_f32(init_x, dynamics.device)
    if init_v is None:
        init_v = dynamics._randn_like(init_x)
    x, v = init_x, init_v
    log_jac = torch.zeros(init_x.shape[0], dtype=torch.float32, device=init_x.device)
    for k in range(int(nb_steps)):
        x, v, lj, _ = propose(x, dynamics, init_v=v, aux=aux, log_jac=True, do_mh_step=False,
                              direction=None if directions is None else directions[k],
                              v=None if vs is None else vs[k])
        log_jac = log_jac + lj
    p_accept = dynamics.p_accept(init_x, init_v, x, v, log_jac, aux=aux)
    outputs = []
    if do_mh_step:
        outputs.append(tf_accept(init_x, x, p_accept, u=u, dynamics=dynamics))
    return x, v, p_accept, outputs


def sample_chain(x, dynamics, nb_proposals, *, direction=None, v=None, u=None, record=False,
                 seed=None, proposal0=0, chain_offset=0, aux=None):
    """`nb_proposals` chained `propose(..., do_mh_step=True)` calls -- the per-MH-step
    `sess.run` loop of the notebook (SCGExperiment.ipynb raw lines 288-298) and of
    `notebook_utils.get_hmc_samples` (:25-39) -- as ONE persistent kernel launch: weights stay
    in LDS, the chain state / gradient / layer-1 partial stay in registers between proposals,
    nothing returns to the host.

    Randomness: injected draws direction (M, N) 0/1, v (M, N, d), u (M, N); or `seed=` -- then
    every draw not injected comes from the in-kernel counter-based Philox stream (proposal
    index `proposal0 + m`, global chain index `chain_offset + n`: identical numbers whatever
    the kernel geometry or the sharding of chains over GPUs); else torch's generator.
    Returns (x_final (N, d), p (M, N), x_hist (M, N, d) or None); x_hist[m] is the state AFTER
    proposal m (the notebook records the state BEFORE each step: that is [x] + x_hist[:-1]).
    `aux=`: conditioning images of the VAE posterior (split engine: the same loop, one Philox fill and one
    trajectory launch per proposal, state and draws never leave the device)."""
    dynamics._check_aux(aux)
    x = as_device_f32(x, dynamics.device)
    N, d = x.shape
    M = int(nb_proposals)
    if M < 1:
        raise ValueError("nb_proposals must be >= 1")
    gen, dev = dynamics.generator, dynamics.device
    rng = None
    if seed is not None:
        rng = {'seed': seed, 'proposal0': proposal0, 'chain_offset': chain_offset}
    else:
        if v is None:
            v = torch.randn((M, N, d), dtype=torch.float32, device=dev, generator=gen)
        if u is None:
            u = torch.rand((M, N), dtype=torch.float32, device=dev, generator=gen)
        if direction is None and not dynamics.hmc:
            direction = torch.randint(0, 2, (M, N), device=dev, dtype=torch.uint8, generator=gen)
    if dynamics.hmc:
        direction = None
    if v is not None:
        v = as_device_f32(v, dev).reshape((M, N, d) if M > 1 else (N, d))
    if u is not None:
        u = as_device_f32(u, dev).reshape((M, N) if M > 1 else (N,))
    if direction is not None:
        direction = torch.as_tensor(direction, device=dev).to(torch.uint8).reshape((M, N) if M > 1 else (N,))
    want = ('p', 'x_next') + (('x_hist',) if record else ())
    o = dynamics.run(x, v, 0, dynamics.T, direction=direction, direction_all=1, u=u, want=want,
                     n_proposals=M, rng=rng, aux=aux)
    return o['x_next'], o['p'].reshape(M, N), (o['x_hist'] if record else None)


def philox_draws(seed, n_chains, d, nb_proposals, *, proposal0=0, chain_offset=0, device=None):
    """The (v (M,N,d), direction (M,N) u8, u (M,N)) the in-kernel stream yields (`l2hmc_rng_fill`)."""
    from .layers import default_device
    dev = torch.device(device) if device is not None else default_device()
    M, N = int(nb_proposals), int(n_chains)
    v = torch.empty((M, N, d), dtype=torch.float32, device=dev)
    dr = torch.empty((M, N), dtype=torch.uint8, device=dev)
    u = torch.empty((M, N), dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib().l2hmc_rng_fill(int(seed) & 0xFFFFFFFFFFFFFFFF, int(proposal0), int(chain_offset),
                                         N, int(d), M, v.data_ptr(), dr.data_ptr(), u.data_ptr(),
                                         _ffi.current_stream(dev)))
    return v, dr, u
