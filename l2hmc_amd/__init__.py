"""l2hmc_amd -- the L2HMC generalised-leapfrog hot path, MI355X-native.

Python surface = the reference's `utils/dynamics.py`, `utils/sampler.py`, `utils/layers.py`,
`utils/distributions.py` (+ the chain diagnostics of `utils/func_utils.py`, the loss values of `utils/losses.py`,
`utils/notebook_utils.get_hmc_samples`, `utils/ais.py`); compute = the
hand-written HIP kernels of `csrc/` behind the C ABI of `include/l2hmc.h`.
"""
from . import _ffi, distributions, func_utils, layers, losses  # noqa: F401
from .dynamics import Dynamics  # noqa: F401
from .sampler import chain_operator, propose, sample_chain, tf_accept  # noqa: F401

__all__ = ["Dynamics", "propose", "tf_accept", "chain_operator", "sample_chain", "layers", "distributions",
           "func_utils", "losses"]
