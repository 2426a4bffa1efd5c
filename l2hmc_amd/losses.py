"""The named objectives of the reference's utils/losses.py as VALUES of device tensors (`get_loss(name)(x, Lx, px)`).

The reference differentiates them through its TF graph; here the gradient of the one that is actually used --
`loss_mixed`, SCGExperiment.ipynb raw 164-169 and mnist_vae.py:207-214 -- comes from the native trainers
(`l2hmc_amd.training`), so these functions are for monitoring / evaluation: plain torch arithmetic on whatever
device the arguments live on.  Per-chain argument: v_n = |x_n - X_n|^2 p_n + 1e-4 (losses.py:36-37).
"""
import math

import torch


def loss_vec(x, X, p):
    return ((X - x) ** 2).sum(dim=1) * p + 1e-4


def loss_logsumexp(x, X, p):                      # losses.py:39-42
    v = loss_vec(x, X, p)
    return torch.logsumexp(-v, dim=0) - math.log(v.shape[0])


def loss_inverse(x, X, p):                        # losses.py:44-47
    return -1.0 / (1.0 / (loss_vec(x, X, p) + 1e-4)).mean()


def loss_std(x, X, p):                            # losses.py:49-51
    return -loss_vec(x, X, p).mean(dim=0)


def loss_mixed(x, Lx, px, scale=1.0):             # losses.py:53-59
    v1 = loss_vec(x, Lx, px) / scale
    return (1.0 / v1).mean() - v1.mean()


def get_loss(name):                               # losses.py:26-34
    return {'mixed': loss_mixed, 'standard': loss_std, 'inverse': loss_inverse, 'logsumexp': loss_logsumexp}[name]
