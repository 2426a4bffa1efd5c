"""Values of the reference's named sampler objectives (`utils/losses.py`) on device tensors.

The reference differentiates these through its TF graph.  Here the gradient of the one objective its programs use --
the `mixed` form, SCGExperiment.ipynb raw 164-169 and mnist_vae.py:207-214 -- is produced natively by
`l2hmc_amd.training`; this module only EVALUATES the objectives (monitoring, tests): plain torch arithmetic on
whatever device the arguments live on.

Everything is a reduction of one per-chain quantity, the expected squared jump (losses.py:36-37)

    jump_n = p_n * |proposal_n - state_n|^2 + 1e-4

    mixed      mean(scale / jump) - mean(jump / scale)          (:53-59)
    standard   -mean(jump)                                       (:49-51)
    inverse    -1 / mean(1 / (jump + 1e-4))                      (:44-47)
    logsumexp  log mean exp(-jump)                               (:39-42)
"""
import math

import torch


def expected_jump(state, proposal, accept_prob):
    sq = torch.sum((proposal - state) * (proposal - state), dim=1)
    return accept_prob * sq + 1e-4


_REDUCERS = {
    'mixed': lambda j, scale: torch.mean(scale / j) - torch.mean(j / scale),
    'standard': lambda j, scale: -torch.mean(j, dim=0),
    'inverse': lambda j, scale: -torch.reciprocal(torch.mean(torch.reciprocal(j + 1e-4))),
    'logsumexp': lambda j, scale: torch.logsumexp(-j, dim=0) - math.log(j.shape[0]),
}


def _named(name):
    reduce_ = _REDUCERS[name]

    def objective(x, Lx, px, scale=1.0):
        return reduce_(expected_jump(x, Lx, px), scale)
    objective.__name__ = 'loss_' + name
    return objective


# the reference's names
loss_vec = expected_jump
loss_mixed = _named('mixed')
loss_std = _named('standard')
loss_inverse = _named('inverse')
loss_logsumexp = _named('logsumexp')


def get_loss(name):
    """`utils/losses.py:26-34`: 'mixed' | 'standard' | 'inverse' | 'logsumexp' -> callable(x, Lx, px[, scale])."""
    return {'mixed': loss_mixed, 'standard': loss_std, 'inverse': loss_inverse, 'logsumexp': loss_logsumexp}[name]
