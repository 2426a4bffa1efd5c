"""`get_hmc_samples` of the reference's utils/notebook_utils.py:25-39 -- the HMC baseline chains of the notebook
(SCGExperiment.ipynb raw 317-319) -- on the fused kernel: the `steps` proposals run in ONE persistent launch
(`sample_chain`, in-kernel Philox draws, history kept on the device) instead of `steps` session round trips.
(The plotting helpers of that module are out of scope.)"""
import numpy as np
import torch

from .dynamics import Dynamics
from .sampler import sample_chain


def get_hmc_samples(x_dim, eps, energy_function, sess=None, T=10, steps=200, samples=None, seed=0):
    """(steps, N, x_dim) numpy array: row t = the chain states BEFORE MH step t, like the reference's loop
    (`final_samples.append(np.copy(samples))` precedes the step).  `sess` is accepted and ignored; `samples` defaults
    to 200 standard-normal starts (the reference reads a notebook global there)."""
    dyn = Dynamics(x_dim, energy_function, T=T, eps=eps, hmc=True)
    if samples is None:
        samples = np.random.randn(200, x_dim)
    x0 = torch.as_tensor(np.asarray(samples, dtype=np.float32), device=dyn.device)
    if steps <= 0:
        return np.zeros((0,) + tuple(x0.shape), dtype=np.float32)
    _, _, hist = sample_chain(x0, dyn, steps, seed=seed, record=True)
    out = torch.cat([x0[None], hist[:-1]], dim=0)
    return out.cpu().numpy()
