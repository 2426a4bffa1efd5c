"""Chain diagnostics -- the sampler-related part of the reference's utils/func_utils.py
(`accept` :33-42, `autocovariance` :45-54, `acl_spectrum` :114-116, `ESS` :118-120) on recorded
chains of shape (steps, chains, dim).  numpy input -> numpy arithmetic exactly like the
reference; a ROCm tensor (e.g. the `x_hist` of `sample_chain`) -> the HIP kernel `l2hmc_autocov`,
the history never leaves the GPU.  Also the small host-side helpers the VAE programs import from that file (`binarize`,
`binarize_and_shuffle`, `normal_kl`, `get_log_likelihood`, `tf_accept`); `get_data` (MNIST download) is out of scope."""
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


# ---- the small host-side helpers of utils/func_utils.py the VAE programs import (mnist_vae.py:29-30) ----------------
def binarize(x, rng=None):
    """func_utils.py:69-71: Bernoulli draw of every pixel intensity in [0, 1]."""
    x = np.asarray(x)
    if x.size and x.max() > 1.:
        raise ValueError("binarize expects intensities in [0, 1]")
    r = np.random if rng is None else rng
    draw = getattr(r, "random_sample", None) or r.random       # RandomState / np.random | a np.random.Generator
    return (draw(x.shape) < x).astype(np.float32)


def binarize_and_shuffle(x, rng=None):
    """func_utils.py:98-109: rows permuted, then binarized."""
    r = np.random if rng is None else rng
    x = np.asarray(x)
    return binarize(x[r.permutation(x.shape[0]), :], rng)


def normal_kl(q_means, q_stddevs, p_means, p_stddevs):
    """func_utils.py:77-96: KL(q || p) of diagonal normals, summed over the last dimension (torch tensors or numpy)."""
    import torch
    t = torch if any(isinstance(a, torch.Tensor) for a in (q_means, q_stddevs, p_means, p_stddevs)) else np
    as_t = (lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(a, dtype=torch.float32)) if t is torch \
        else (lambda a: np.asarray(a, dtype=np.float64))
    qm, qs, pm, ps = (as_t(a) for a in (q_means, q_stddevs, p_means, p_stddevs))
    q_entropy = 0.5 + t.log(qs)
    cross = 0.5 * (qs / ps) ** 2 + 0.5 * ((qm - pm) / ps) ** 2 + t.log(ps)
    return (cross - q_entropy).sum(-1)


def get_log_likelihood(X, gaussian):
    """func_utils.py:59-61: mean log-density of the rows of X under the target Gaussian."""
    from scipy.stats import multivariate_normal
    return multivariate_normal(mean=gaussian.mu, cov=gaussian.sigma).logpdf(np.asarray(X)).mean()


def tf_accept(x, Lx, px, u=None, dynamics=None):
    """func_utils.py:73-75 (the same function as utils/sampler.py:53-55)."""
    from .sampler import tf_accept as _tf_accept
    return _tf_accept(x, Lx, px, u=u, dynamics=dynamics)
