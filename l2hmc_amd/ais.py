"""Annealed importance sampling -- API of the reference's utils/ais.py (Wu et al. 2016).

`ais_estimate` keeps the reference signature (ais.py:30-42).  The bridge is the reference's
geometric path  U_b = (1 - b) U_init + b U_final,  b = 1/K, 2/K, .., 1  (ais.py:43,46-47), with the
standard-normal initial energy its only caller uses (eval_vae.py:55-56).

Built-in targets: ALL K anneal steps run in ONE persistent launch of the trajectory kernel (AIS mode of
`l2hmc_trajectory`, include/l2hmc.h: bridge schedule in HBM, log-weight update, momentum refresh, HMC
transition, MH step with the momentum flip, accept statistics -- the `tf.scan` of ais.py:68 without a
single host round trip).  The decoder posterior (GEMM engine) keeps the per-step form (ais.py:48-66), all
on the device:

    l2hmc_energy            U_final(x)                 (l2hmc_vae_energy for the decoder posterior)
    l2hmc_ais_begin_step    w += db (|x|^2/2 - U_final(x));  v = fresh / partially refreshed momentum
    l2hmc_trajectory        HMC mode (`leapfrogs` steps of size `step_size`) on U_b: energy.anneal_beta = b
                            (l2hmc_trajectory_split with hmc = 1, bce_scale = b for the decoder posterior,
                             eval_vae.py:58-64: there U_b = |z|^2/2 + b BCE)
    l2hmc_ais_end_step      MH accept:  x = Lx or x;  v = Lv or -Lv (sic, ais.py:63);  alpha += p

The random draws come from the library's Philox stream (`l2hmc_rng_fill`), keyed by (seed, anneal
step, global chain index), or are injected (`draws=`, tests).  Returns what the reference returns:
(log-mean-exp of the final log-weights, summed over `num_splits` equal chain blocks; mean accept
probability).
"""
import math

import numpy as np
import torch

from . import _ffi
from .distributions import EnergyFunction, as_device_f32
from .dynamics import Dynamics


def _is_standard_normal(fn):
    if not isinstance(fn, EnergyFunction) or fn.kind not in (_ffi.ENERGY_GAUSS_DIAG, _ffi.ENERGY_GAUSS_DENSE):
        return False
    mu, prec = np.asarray(fn._host['mu']), np.asarray(fn._host['prec'])
    eye = np.ones(fn.x_dim) if fn.kind == _ffi.ENERGY_GAUSS_DIAG else np.eye(fn.x_dim)
    return not mu.any() and prec.shape == eye.shape and np.array_equal(prec, eye.astype(prec.dtype))


def ais_estimate(init_energy, final_energy, anneal_steps, initial_x, aux=None, step_size=0.5, leapfrogs=25,
                 x_dim=5, num_splits=1, refresh=False, refreshment=0.1, *, seed=0, draws=None,
                 chain_offset=0, return_state=False):
    """ais.py:30-82.  Extra keyword-only arguments: `seed` / `chain_offset` (Philox stream), `draws` =
    {'v0': (N,d), 'normals': (K,N,d), 'u': (K,N)} to inject the randomness, `return_state` to also get
    {'x', 'w', 'alpha'} (final states, final log-weights, summed accept probabilities)."""
    if not _is_standard_normal(init_energy):
        raise NotImplementedError("ais_estimate: the initial energy must be the standard normal "
                                  "`Gaussian(np.zeros(d), np.eye(d)).get_energy_function()` (eval_vae.py:55-56)")
    from .vae import ENERGY_VAE
    vae = final_energy.kind == ENERGY_VAE        # eval_vae.py:58-64: decoder posterior, aux = the images
    if vae != (aux is not None):
        raise ValueError("aux= goes with the image-conditioned VAE energy (eval_vae.py:58-64) and only with it")
    K = int(anneal_steps)
    if K < 2:
        raise ValueError("anneal_steps must be >= 2 (ais.py:44 takes beta[1] - beta[0])")
    dyn = Dynamics(x_dim, final_energy, T=leapfrogs, eps=step_size, hmc=True)
    dyn.eps_override = float(step_size)
    dev = dyn.device
    x = as_device_f32(initial_x, dev).clone()
    N, d = x.shape
    if d != x_dim:
        raise ValueError("initial_x is %d-dimensional, x_dim=%d" % (d, x_dim))
    if N % int(num_splits):
        raise ValueError("num_splits must divide the number of chains")
    L = _ffi.lib()
    s = _ffi.current_stream(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    w, alpha = torch.zeros(N, **f32), torch.zeros(N, **f32)
    z, u, U1 = torch.empty(N, d, **f32), torch.empty(N, **f32), torch.empty(N, **f32)

    def fill(step, want_u):
        # anneal step i uses Philox "proposal" index i (0 = the initial momentum)
        _ffi.check(L.l2hmc_rng_fill(int(seed), int(step), int(chain_offset), N, d, 1, z.data_ptr(), None,
                                    u.data_ptr() if want_u else None, s))

    if draws is None:
        fill(0, False)
        v = z.clone()
    else:
        v = as_device_f32(draws['v0'], dev).clone()
    # beta = linspace(0, 1, K + 1)[1:] in float32 like the graph (ais.py:43-44)
    beta = np.linspace(0.0, 1.0, K + 1, dtype=np.float32)[1:]
    dbeta = float(np.float32(beta[1] - beta[0]))
    e_final = None if vae else final_energy.c_struct(dev)
    if vae:
        aux = as_device_f32(aux, dev)
    if not vae:
        # ONE persistent launch for all K anneal steps (`tf.scan` of ais.py:68): the bridge schedule sits in HBM,
        # log-weight update, momentum refresh, HMC transition, MH step with the momentum flip and the accept
        # statistics all happen in the trajectory kernel's proposal loop (AIS mode, include/l2hmc.h)
        bt = torch.as_tensor(beta, device=dev)
        spec = {'beta': bt, 'dbeta': dbeta, 'refreshment': float(refreshment) if refresh else -1.0, 'w': w, 'alpha': alpha}
        dyn.anneal_beta = float(beta[0])
        if draws is None:
            o = dyn.run(x, None, 0, leapfrogs, direction_all=1, want=('x_next',), n_proposals=K,
                        rng={'seed': int(seed), 'proposal0': 1, 'chain_offset': int(chain_offset), 'direction': False},
                        ais=spec)
        else:
            spec['v0'] = v
            nz = as_device_f32(np.asarray(draws['normals']), dev).reshape((K, N, d) if K > 1 else (N, d))
            uu = as_device_f32(np.asarray(draws['u']), dev).reshape((K, N) if K > 1 else (N,))
            o = dyn.run(x, nz, 0, leapfrogs, direction_all=1, u=uu, want=('x_next',), n_proposals=K, ais=spec)
        x = o['x_next']
        K_loop = 0
    else:
        K_loop = K
    for i in range(K_loop):
        if draws is None:
            fill(i + 1, True)
            zi, ui = z, u
        else:
            zi, ui = as_device_f32(draws['normals'][i], dev), as_device_f32(draws['u'][i], dev)
        if vae:
            U1 = final_energy.evaluate(x, aux=aux)[0]              # l2hmc_vae_energy (split engine)
        else:
            _ffi.check(L.l2hmc_energy(e_final, x.data_ptr(), N, d, U1.data_ptr(), None, s))
        _ffi.check(L.l2hmc_ais_begin_step(x.data_ptr(), U1.data_ptr(), zi.data_ptr(),
                                          float(refreshment) if refresh else -1.0, dbeta,
                                          w.data_ptr(), v.data_ptr(), N, d, s))
        dyn.anneal_beta = float(beta[i])
        Lx, Lv, px = dyn.forward(x, init_v=v, aux=aux)
        _ffi.check(L.l2hmc_ais_end_step(Lx.data_ptr(), Lv.data_ptr(), px.data_ptr(), ui.data_ptr(),
                                        x.data_ptr(), v.data_ptr(), alpha.data_ptr(), N, d, s))

    def logmeanexp(t):
        return torch.logsumexp(t, dim=0) - math.log(t.shape[0])
    est = sum(logmeanexp(t) for t in torch.chunk(w, int(num_splits)))
    mean_alpha = alpha.sum() / (K * N)
    if return_state:
        return est, mean_alpha, {'x': x, 'w': w, 'alpha': alpha}
    return est, mean_alpha
