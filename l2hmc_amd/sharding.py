"""Chain sharding across the GPUs of a node (one process per GPU, torch.distributed).

Chains never interact on the sampling path (every tensor is (N, ...) with no
cross-chain term), so the chain batch is cut into contiguous blocks -- rank r owns rows
[r*N/W, (r+1)*N/W) -- and the leapfrog kernels run with NO data-path collective.  The only
exchanges are small statistics: the mean accept probability and the autocovariance partial
sums behind ESS (utils/func_utils.py:45-54,114-120), each ONE flat all-reduce.  Backend
"nccl" is RCCL over xGMI on the GPU box; "gloo" is used by the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_total, rank=None, world_size=None):
    """Contiguous block [lo, hi) of the n_total chains owned by `rank` (sizes differ by <= 1)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _allreduce_sum(t):
    if world()[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def mean_accept(p_local):
    """Global mean accept probability from this rank's (N_local,) probabilities."""
    p_local = torch.as_tensor(p_local)
    acc = torch.stack([p_local.double().sum(), torch.tensor(float(p_local.numel()), dtype=torch.float64,
                                                             device=p_local.device)])
    acc = _allreduce_sum(acc)
    return float(acc[0] / acc[1])


def autocov_partial_sums(X_local):
    """S(tau) = sum_t sum_{n,k} X[t,n,k] X[t+tau,n,k] for tau = 0..steps-2 on this rank's chains
    (X_local: (steps, N_local, d)); the per-lag normalisation is applied after the reduction."""
    X = torch.as_tensor(X_local, dtype=torch.float64)
    steps = X.shape[0]
    flat = X.reshape(steps, -1)
    out = torch.empty(steps - 1, dtype=torch.float64, device=X.device)
    for tau in range(steps - 1):
        out[tau] = (flat[:steps - tau] * flat[tau:]).sum()
    return out


def acl_spectrum(X_local, scale, n_total):
    """`acl_spectrum` of utils/func_utils.py:114-116 for chains sharded over ranks: one
    all-reduce of the (steps-1,) partial sums; equals the single-process value on the
    concatenated chains."""
    if isinstance(X_local, torch.Tensor) and X_local.is_cuda:      # HIP kernel, history stays in HBM
        from .func_utils import device_autocov
        steps = X_local.shape[0]
        s = _allreduce_sum(device_autocov(X_local, 1.0)[0] / (scale * scale))
    else:
        X = torch.as_tensor(X_local, dtype=torch.float64) / scale
        steps = X.shape[0]
        s = _allreduce_sum(autocov_partial_sums(X))
    lags = torch.arange(steps - 1, dtype=torch.float64, device=s.device)
    return (s / float(n_total) / (steps - lags)).cpu().numpy()


def ess(X_local, scale, n_total):
    """ESS per MH step (utils/func_utils.py:118-120) of sharded chains."""
    from .func_utils import ESS
    return float(ESS(acl_spectrum(X_local, scale, n_total)))
