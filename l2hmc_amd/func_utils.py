"""Chain diagnostics -- the sampler-related part of the reference's utils/func_utils.py
(`accept` :33-42, `autocovariance` :45-54, `acl_spectrum` :114-116, `ESS` :118-120) on recorded
chains of shape (steps, chains, dim).  numpy input -> numpy arithmetic exactly like the
reference; a ROCm tensor (e.g. the `x_hist` of `sample_chain`) -> the HIP kernel `l2hmc_autocov`,
the history never leaves the GPU.  The MNIST/VAE helpers of that file are out of scope."""
import numpy as np


def _is_device_tensor(X):
    try:
        import torch
        return isinstance(X, torch.Tensor) and X.is_cuda
    except ImportError:
        return False


def device_autocov(X, scale=1.0, n_total=None):
    """(raw sums S(tau), A(tau)) of a (steps, N, d) float32 device history via `l2hmc_autocov`;
    both are float64 device tensors of length steps-1."""
    import torch
    from . import _ffi
    X = X.detach()
    if X.dtype != torch.float32 or not X.is_contiguous():
        X = X.to(torch.float32).contiguous()
    steps, N, d = X.shape
    S = torch.empty(steps - 1, dtype=torch.float64, device=X.device)
    A = torch.empty(steps - 1, dtype=torch.float64, device=X.device)
    L = _ffi.lib()
    # per-block partial sums, added in block order: S and the thresholded ESS are bitwise reproducible
    ws = torch.empty(max(1, _ffi.check(L.l2hmc_autocov_workspace_doubles(steps, N, d))), dtype=torch.float64,
                     device=X.device)
    _ffi.check(L.l2hmc_autocov(X.data_ptr(), steps, N, d, float(scale),
                               int(N if n_total is None else n_total), S.data_ptr(),
                               A.data_ptr(), ws.data_ptr(), _ffi.current_stream(X.device)))
    return S, A


def accept(x_i, x_p, p, rng=None):
    """Numpy Metropolis select: rows with p - u >= 0 move to the proposal."""
    if x_i.shape != x_p.shape:
        raise ValueError("state and proposal shapes differ")
    u = (np.random if rng is None else rng).uniform(size=(x_i.shape[0],))
    take = (p - u >= 0)[:, None]
    return np.where(take, x_p, x_i)


def autocovariance(X, tau=0):
    """mean_t [ sum_{n,k} X[t,n,k] X[t+tau,n,k] / N ]  -- no mean subtraction, like the reference."""
    X = np.asarray(X)
    steps, chains, _ = X.shape
    a, b = X[:steps - tau], X[tau:]
    return float(np.einsum('tnk,tnk->', a, b) / chains / (steps - tau))


def acl_spectrum(X, scale):
    """A(tau) for tau = 0 .. steps-2 on X / scale."""
    if _is_device_tensor(X):
        return device_autocov(X, scale)[1].cpu().numpy()
    Xs = np.asarray(X) / scale
    return np.array([autocovariance(Xs, tau=t) for t in range(Xs.shape[0] - 1)])


def ESS(A):
    """1 / (1 + 2 sum_{tau>=1} A(tau) [A(tau) > 0.05])."""
    A = np.asarray(A)
    kept = np.where(A > 0.05, A, 0.0)
    return 1. / (1. + 2 * np.sum(kept[1:]))
