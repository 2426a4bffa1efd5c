"""Target distributions -- API of the reference's utils/distributions.py.

Each class keeps the reference's constructor, `get_energy_function()`, `get_samples(n)` and
`log_density(X)`.  The energy function returned is an `EnergyFunction`: calling it on an
(N, d) float32 device tensor evaluates U(x) with the HIP kernel `l2hmc_energy`, and
`Dynamics` reads its `.spec` to fuse U and grad U into the leapfrog kernel (no autograd:
the gradients are analytic, dynamics.py:217-218 used `tf.gradients`).
"""

import numpy as np
import torch

from . import _ffi
from .layers import default_device


class EnergyFunction(object):
    """fn(x, *args, **kwargs) -> (N,) energies, plus the parameters the fused kernels need."""

    def __init__(self, kind, x_dim=None, mu=None, prec=None, logc=None, n_comp=1, eta=0.0,
                 easy=False):
        self.kind, self.x_dim, self.n_comp = kind, x_dim, n_comp
        self.eta, self.easy = float(eta), bool(easy)
        # Rough Well: the reference divides by the Python-double product eps * eps, rounded to float32 once (distributions.py:93)
        self.den = float(np.float32(self.eta if self.easy else self.eta * self.eta)) if kind == _ffi.ENERGY_ROUGHWELL else 0.0
        self._host = {'mu': mu, 'prec': prec, 'logc': logc}
        self._dev = {}

    # -- device-side parameter buffers (built once per device) ---------------------------------
    def _buffers(self, device):
        key = str(device)
        if key not in self._dev:
            L = _ffi.lib()
            buf = {}
            for k in ('mu', 'logc'):
                v = self._host[k]
                buf[k] = None if v is None else torch.as_tensor(
                    np.ascontiguousarray(v, dtype=np.float32), device=device)
            prec = self._host['prec']
            if self.kind == _ffi.ENERGY_GAUSS_DIAG:
                buf['prec'] = torch.as_tensor(np.ascontiguousarray(prec, dtype=np.float32), device=device)
            elif self.kind in (_ffi.ENERGY_GAUSS_DENSE, _ffi.ENERGY_GMM):
                d = self.x_dim
                stride = L.l2hmc_packed_gaussian_floats(d)
                raw = torch.as_tensor(np.ascontiguousarray(prec, dtype=np.float32).reshape(-1, d, d),
                                      device=device)
                packed = torch.empty(raw.shape[0] * stride, dtype=torch.float32, device=device)
                s = _ffi.current_stream(device)
                for i in range(raw.shape[0]):
                    _ffi.check(L.l2hmc_pack_gaussian(raw[i].data_ptr(), d,
                                                     packed.data_ptr() + 4 * i * stride, s))
                buf['prec'], buf['_raw'] = packed, raw
            else:
                buf['prec'] = None
            self._dev[key] = buf
        return self._dev[key]

    def c_struct(self, device, temperature=1.0, anneal_beta=0.0):
        b = self._buffers(device)
        return _ffi.L2hmcEnergy(self.kind, self.n_comp, _ffi.ptr(b['mu']), _ffi.ptr(b['prec']),
                                _ffi.ptr(b['logc']), self.eta, int(self.easy), float(temperature),
                                float(anneal_beta), self.den, 0)

    def evaluate(self, x, temperature=1.0, want_U=True, want_grad=False, anneal_beta=0.0):
        x = as_device_f32(x)
        N, d = x.shape
        if self.x_dim is not None and d != self.x_dim:
            raise ValueError("energy expects x_dim=%d, got %d" % (self.x_dim, d))
        U = torch.empty(N, dtype=torch.float32, device=x.device) if want_U else None
        g = torch.empty_like(x) if want_grad else None
        e = self.c_struct(x.device, temperature, anneal_beta)
        _ffi.check(_ffi.lib().l2hmc_energy(e, x.data_ptr(), N, d, _ffi.ptr(U), _ffi.ptr(g),
                                           _ffi.current_stream(x.device)))
        return U, g

    def __call__(self, x, *args, **kwargs):
        return self.evaluate(x)[0]


ENERGY_USER = 100     # python-side tag: a caller-supplied energy (L2hmcSplitArgs.energy_cb)


class UserEnergy(EnergyFunction):
    """A caller-supplied target: ANY callable `fn(x[, aux=...]) -> (N,)` on ROCm tensors that torch can differentiate
    (the `energy_function` protocol of the reference's Dynamics, utils/dynamics.py:203-218: `self._fn(x[, aux])` and
    `tf.gradients` of it) -- e.g. the closure of mnist_vae.py:122-126, or a density that is not in
    utils/distributions.py.  `Dynamics(x_dim, fn)` wraps a plain callable in one of these by itself.

    This is the SLOW path, by construction: U and grad U are computed by the caller's torch code between kernel
    launches (`grad_fn(x[, aux=...]) -> (N, d)` if given, else autograd of `fn`), one host round trip per leapfrog step;
    the leapfrog half-updates, the S/T/Q nets, the log-determinant, the accept probability and the MH select stay on
    the library's HIP kernels (GEMM engine, `l2hmc_trajectory_split`).  Training the sampler on such a target
    (`Trainer(dynamics)`: the GEMM-engine trainer) additionally asks for Hessian-vector products of U, one per leapfrog
    step -- `hvp()` below: double backward through `fn`, or one backward through a differentiable `grad_fn`."""

    def __init__(self, fn, grad_fn=None, x_dim=None):
        EnergyFunction.__init__(self, ENERGY_USER, x_dim=x_dim)
        if not callable(fn) or (grad_fn is not None and not callable(grad_fn)):
            raise TypeError("UserEnergy needs callables")
        self.fn, self.grad_fn = fn, grad_fn

    def _buffers(self, device):
        raise NotImplementedError("a caller-supplied energy has no fused-kernel parameters")

    def c_struct(self, device, temperature=1.0, anneal_beta=0.0):
        raise NotImplementedError("a caller-supplied energy runs through L2hmcSplitArgs.energy_cb, not L2hmcEnergy")

    def _call(self, f, x, aux):
        return f(x) if aux is None else f(x, aux=aux)

    def evaluate(self, x, temperature=1.0, want_U=True, want_grad=False, aux=None, anneal_beta=0.0):
        """(U, grad U) of  [(1 - b) |x|^2 / 2 + b] U(x) / temperature  (b = anneal_beta, 0 = off: utils/ais.py:46-47;
        temperature: dynamics.py:203-212) -- U as float64 (the caller's own precision kept as far as it goes)."""
        x = as_device_f32(x)
        U = g = None
        if want_grad and self.grad_fn is None:
            with torch.enable_grad():
                xr = x.clone().requires_grad_(True)
                Ur = self._call(self.fn, xr, aux)
                if Ur.shape != (x.shape[0],):
                    raise ValueError("the energy function must return shape (N,), got %s" % (tuple(Ur.shape),))
                g, = torch.autograd.grad(Ur.sum(), xr)         # dynamics.py:217-218: tf.gradients sums over the batch
            U = Ur.detach()
        else:
            with torch.no_grad():
                if want_U:
                    U = self._call(self.fn, x, aux)
                    if U.shape != (x.shape[0],):
                        raise ValueError("the energy function must return shape (N,), got %s" % (tuple(U.shape),))
                if want_grad:
                    g = self._call(self.grad_fn, x, aux)
        b = float(anneal_beta)
        if b > 0.0 and b != 1.0:
            if U is not None:
                U = (1.0 - b) * 0.5 * x.double().square().sum(1) + b * U.double()
            if g is not None:
                g = (1.0 - b) * x + b * g
        if temperature != 1.0:
            U = None if U is None else U.double() / float(temperature)
            g = None if g is None else g / float(temperature)
        return (U.double() if (U is not None and want_U) else None), (g.to(torch.float32) if g is not None else None)

    def hvp(self, x, u, aux=None):
        """(d^2 U / dx^2)(x) u, shape (N, d): what the training loss needs where it back-propagates through grad U
        (utils/dynamics.py:218 inside the differentiated graph)."""
        x, u = as_device_f32(x), as_device_f32(u)
        with torch.enable_grad():
            xr = x.clone().requires_grad_(True)
            if self.grad_fn is None:
                Ur = self._call(self.fn, xr, aux)
                g, = torch.autograd.grad(Ur.sum(), xr, create_graph=True)
            else:
                g = self._call(self.grad_fn, xr, aux)
                if not g.requires_grad:
                    raise NotImplementedError("training needs Hessian-vector products: grad_energy must be torch code "
                                              "that autograd can differentiate (or leave it out and pass a differentiable "
                                              "energy function)")
            hv, = torch.autograd.grad((g * u).sum(), xr)
        return hv.to(torch.float32)

    def __call__(self, x, aux=None, *args, **kwargs):
        return self.evaluate(x, aux=aux)[0].to(torch.float32)


def as_device_f32(x, device=None):
    """numpy / torch input -> contiguous float32 tensor on the GPU."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32), device=device or default_device())
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.to(torch.float32).contiguous()
    if not x.is_cuda:
        raise RuntimeError("l2hmc_amd: tensors must live on a ROCm device (got %s); "
                           "there is no CPU path" % x.device)
    return x.detach()


def _rng(rng):
    """Samplers take an optional numpy RandomState; default = numpy's global stream (what the
    reference uses everywhere)."""
    return np.random if rng is None else rng


def _log_spaced_spectrum(dim, log_min, log_max, jitter, rng=None):
    """Random rotation R and eigenvalues 10**U(log_min, log_max) (+ jitter) of a tilted Gaussian."""
    from scipy.stats import ortho_group
    R = ortho_group.rvs(dim)
    eig = 10.0 ** _rng(rng).uniform(log_min, log_max, size=dim) + jitter
    return R, eig


def quadratic_gaussian(x, mu, S):
    """Row-wise 0.5 (x - mu)^T S (x - mu)  (distributions.py:31-32; the reference forms the
    whole N x N product and keeps its diagonal -- only the diagonal is computed here)."""
    S = np.asarray(S, dtype=np.float32)
    return EnergyFunction(_ffi.ENERGY_GAUSS_DENSE, S.shape[0], mu=np.asarray(mu), prec=S)(x)


class Gaussian(object):
    """N(mu, sigma)  (distributions.py:41-68).  The precision is inverted in float64 and used
    in float32, like the reference (:48,52)."""

    def __init__(self, mu, sigma):
        self.mu = np.asarray(mu)
        self.sigma = np.asarray(sigma)
        self.i_sigma = np.linalg.inv(self.sigma.copy())

    def get_energy_function(self):
        S = self.i_sigma.astype(np.float32)
        mu = self.mu.astype(np.float32)
        d = S.shape[0]
        off_diag = S[~np.eye(d, dtype=bool)]
        if d > 1 and not off_diag.any():              # exactly diagonal precision: elementwise path
            return EnergyFunction(_ffi.ENERGY_GAUSS_DIAG, d, mu=mu, prec=np.diag(S).copy())
        return EnergyFunction(_ffi.ENERGY_GAUSS_DENSE, d, mu=mu, prec=S)

    def get_samples(self, n, rng=None):
        """Exact draws mu + L z, L = chol(sigma).  (The reference omits mu; its targets are
        centred, so the two agree there.)"""
        L = np.linalg.cholesky(self.sigma)
        z = _rng(rng).randn(n, L.shape[0])
        return self.mu + z @ L.T

    def log_density(self, X):
        from scipy.stats import multivariate_normal
        return multivariate_normal(mean=self.mu, cov=self.sigma).logpdf(X)


def random_tilted_gaussian(dim, log_min=-2., log_max=2.):
    """Zero-mean Gaussian with a random rotation and log-uniform spectrum (distributions.py:34-39)."""
    R, eig = _log_spaced_spectrum(dim, log_min, log_max, 1e-6)
    return Gaussian(np.zeros(dim), (R.T * eig) @ R)


class TiltedGaussian(Gaussian):
    """distributions.py:70-82; `get_samples(n)` honours n (the reference always drew 200)."""

    def __init__(self, dim, log_min, log_max):
        self.dim = dim
        self.R, eig = _log_spaced_spectrum(dim, log_min, log_max, 1e-8)
        self.diag = np.diag(eig)
        Gaussian.__init__(self, np.zeros(dim), (self.R.T * eig) @ self.R)

    def get_samples(self, n, rng=None):
        z = _rng(rng).randn(n, self.dim)
        return (z * np.sqrt(np.diag(self.diag))) @ self.R


class RoughWell(object):
    """U(x) = |x|^2 / 2 + eps sum_k cos(x_k / eps^2)  (x_k / eps when `easy`)  -- distributions.py:84-101."""

    def __init__(self, dim, eps, easy=False):
        self.dim, self.eps, self.easy = dim, eps, easy

    def get_energy_function(self):
        return EnergyFunction(_ffi.ENERGY_ROUGHWELL, self.dim, eta=self.eps, easy=self.easy)

    def get_samples(self, n, rng=None):
        # for small eps the well is a standard normal to a good approximation (:99-101)
        return _rng(rng).randn(n, self.dim)


class GMM(object):
    """Mixture of Gaussians  (distributions.py:104-152).  Per component the reference keeps
    inv(sigma_i) and c_i = pi_i / sqrt((2 pi)^k det sigma_i) as float32 (:117-124)."""

    def __init__(self, mus, sigmas, pis):
        if len(mus) != len(sigmas) or len(mus) != len(pis):
            raise ValueError("mus, sigmas, pis must have one entry per component")
        if sum(pis) != 1.0:
            raise ValueError("mixture weights must sum to 1")
        self.mus, self.sigmas, self.pis = mus, sigmas, pis
        self.nb_mixtures = len(pis)
        self.k = int(np.asarray(mus[0]).shape[0])
        two_pi_k = (2 * np.pi) ** self.k
        self.i_sigmas = [np.linalg.inv(sg).astype(np.float32) for sg in sigmas]
        norms = [np.float32(np.sqrt(two_pi_k * np.linalg.det(sg))) for sg in sigmas]
        self.constants = [np.float32(pi / nz) for pi, nz in zip(pis, norms)]

    def get_energy_function(self):
        return EnergyFunction(_ffi.ENERGY_GMM, self.k,
                              mu=np.stack([np.asarray(m, dtype=np.float32) for m in self.mus]),
                              prec=np.stack(self.i_sigmas),
                              logc=np.log(np.asarray(self.constants, dtype=np.float32)),
                              n_comp=self.nb_mixtures)

    def get_samples(self, n, rng=None):
        r = _rng(rng)
        comp = r.choice(self.nb_mixtures, size=n, p=self.pis)
        out = np.empty((n, self.k))
        for i in range(self.nb_mixtures):
            idx = np.flatnonzero(comp == i)
            if idx.size:
                out[idx] = r.multivariate_normal(self.mus[i], self.sigmas[i], size=idx.size)
        return out

    def log_density(self, X):
        from scipy.stats import multivariate_normal
        dens = [pi * multivariate_normal(mean=m, cov=sg).pdf(X)
                for pi, m, sg in zip(self.pis, self.mus, self.sigmas)]
        return np.log(np.sum(dens, axis=0))


class GaussianFunnel(object):
    """Neal's funnel, x_0 = v ~ N(0, sigma^2), x_k ~ N(0, e^v), with the energy clipped at
    |v| > 4 sigma  (distributions.py:155-198; sigma = 2)."""

    def __init__(self, dim=2, clip=6.):
        self.dim = dim
        self.sigma = 2.0
        self.clip = 4 * self.sigma

    def get_energy_function(self):
        return EnergyFunction(_ffi.ENERGY_FUNNEL, self.dim, eta=self.sigma)

    def get_samples(self, n, rng=None):
        r = _rng(rng)
        v = self.sigma * r.randn(n)
        rest = np.exp(v / 2)[:, None] * r.randn(n, self.dim - 1)
        return np.concatenate([v[:, None], rest], axis=1)

    def log_density(self, x):
        """Same expression as the reference's `log_density` (:190-198), in numpy throughout
        (the reference mixes tf ops into it)."""
        v = x[:, 0]
        n = x.shape[1] - 1
        ss = np.sum(x[:, 1:] ** 2, axis=1)
        return 0.5 * ((v / self.sigma) ** 2 + ss * np.exp(-v) + (n / 2) * (np.log(2 * np.pi) + v))


def gen_ring(r=1.0, var=1.0, nb_mixtures=2):
    """`nb_mixtures` isotropic components equally spaced on a circle of radius r (distributions.py:201-213)."""
    ang = 2 * np.pi * np.arange(nb_mixtures) / nb_mixtures
    centres = [np.array([r * np.cos(a), r * np.sin(a)]) for a in ang]
    covs = [var * np.eye(2) for _ in range(nb_mixtures)]
    w = [1. / nb_mixtures] * nb_mixtures
    w[0] += 1 - sum(w)                       # make the weights sum to exactly 1.0
    return GMM(centres, covs, w)
