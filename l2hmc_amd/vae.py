"""The VAE latent-posterior target of the reference's mnist_vae.py (BASELINE.json config 5).

`VAEPosterior(decoder)` wraps a decoder built from the layer kit exactly like mnist_vae.py:104-111
(`Sequential([Linear, softplus, Linear, softplus, Linear])`); its energy function is
mnist_vae.py:122-126,  U(z; x) = sum_pix BCE_with_logits(x, decoder(z)) + |z|^2 / 2,  evaluated --
with its analytic gradient -- by the split engine of the HIP library (`l2hmc_vae_energy`,
`l2hmc_trajectory_split`): hand-written fp32 MFMA GEMMs with fused epilogues for the dense products, small kernels for the rest.
`sampler_net_factory` is mnist_vae.py:142-167: the S/T/Q nets whose 4th Zip branch is the shared
`encoder_sampler(aux)`.
"""
import ctypes as C

import torch

from . import _ffi, layers
from .distributions import EnergyFunction, as_device_f32

ENERGY_VAE = 6     # python-side tag; the split engine has its own argument block (L2hmcSplitArgs)


def mlp3_struct(w):
    n_in, n_h1, n_h2, n_out = w['dims']
    for k in ('W1', 'b1', 'W2', 'b2', 'W3', 'b3'):
        if not (w[k].is_cuda and w[k].is_contiguous() and w[k].dtype == torch.float32):
            raise ValueError("MLP parameter %s must be a contiguous float32 ROCm tensor" % k)
    return _ffi.L2hmcMlp3(w['W1'].data_ptr(), w['b1'].data_ptr(), w['W2'].data_ptr(), w['b2'].data_ptr(),
                          w['W3'].data_ptr(), w['b3'].data_ptr(), n_in, n_h1, n_h2, n_out)


class VAEEnergy(EnergyFunction):
    """fn(z, aux=images) -> (N,) energies; `Dynamics` routes it to the split engine."""

    def __init__(self, decoder_weights):
        EnergyFunction.__init__(self, ENERGY_VAE, x_dim=decoder_weights['dims'][0])
        self.decoder = decoder_weights
        self.n_pix = decoder_weights['dims'][3]
        self._ws = None

    def workspace(self, n_floats, device):
        if self._ws is None or self._ws.numel() < n_floats or self._ws.device != device:
            self._ws = torch.empty(int(n_floats), dtype=torch.float32, device=device)
        return self._ws

    def evaluate(self, x, temperature=1.0, want_U=True, want_grad=False, aux=None, anneal_beta=0.0):
        if aux is None:
            raise ValueError("the VAE posterior energy needs aux= (the conditioning images)")
        if temperature != 1.0:
            raise NotImplementedError("temperature is not supported by the split engine")
        x = as_device_f32(x)
        aux = as_device_f32(aux, x.device)
        N, d = x.shape
        if aux.shape != (N, self.n_pix):
            raise ValueError("aux must be (N, %d)" % self.n_pix)
        L = _ffi.lib()
        dec = mlp3_struct(self.decoder)
        need = _ffi.check(L.l2hmc_split_workspace_floats(N, d, 1, 1, None, C.byref(dec)))
        ws = self.workspace(need, x.device)
        U = torch.empty(N, dtype=torch.float32, device=x.device) if want_U else None
        g = torch.empty_like(x) if want_grad else None
        _ffi.check(L.l2hmc_vae_energy(C.byref(dec), aux.data_ptr(), x.data_ptr(), N, d, _ffi.ptr(U), _ffi.ptr(g),
                                      ws.data_ptr(), float(anneal_beta), _ffi.current_stream(x.device)))
        return U, g

    def __call__(self, x, aux=None, *args, **kwargs):
        return self.evaluate(x, aux=aux)[0]


class VAEPosterior(object):
    """Target object in the style of utils/distributions.py for the decoder posterior."""

    def __init__(self, decoder):
        w = layers.extract_mlp3(decoder)
        if w is None:
            raise NotImplementedError("decoder must be Sequential([Linear, softplus, Linear, softplus, Linear]) "
                                      "(mnist_vae.py:104-111)")
        self.decoder, self._w = decoder, w

    def get_energy_function(self):
        return VAEEnergy(self._w)


def make_decoder(latent_dim=50, hidden=1024, n_pix=784):
    """mnist_vae.py:104-111."""
    with layers.variable_scope('decoder'):
        return layers.Sequential([
            layers.Linear(latent_dim, hidden, scope='decoder_1'), layers.softplus,
            layers.Linear(hidden, hidden, scope='decoder_2'), layers.softplus,
            layers.Linear(hidden, n_pix, scope='decoder_3', factor=0.01)])


def make_encoder_sampler(n_pix=784, hidden=512, size1=200):
    """mnist_vae.py:134-140."""
    return layers.Sequential([
        layers.Linear(n_pix, hidden, scope='encoder_1'), layers.softplus,
        layers.Linear(hidden, hidden, scope='encoder_2'), layers.softplus,
        layers.Linear(hidden, size1, scope='encoder_3')])


def sampler_net_factory(latent_dim, encoder_sampler, size1=200, size2=200):
    """mnist_vae.py:142-167 (size1 == size2 == the hidden width H; one encoder_sampler object is
    shared by XNet and VNet)."""
    if size1 != size2:
        raise NotImplementedError("size1 != size2 is not supported")

    def net_factory(x_dim, scope, factor):
        with layers.variable_scope(scope):
            return layers.Sequential([
                layers.Zip([
                    layers.Linear(latent_dim, size1, scope='embed_1', factor=0.33),
                    layers.Linear(latent_dim, size1, scope='embed_2', factor=factor * 0.33),
                    layers.Linear(2, size1, scope='embed_3', factor=0.33),
                    encoder_sampler,
                ]),
                sum,
                layers.relu,
                layers.Linear(size1, size2, scope='linear_1'),
                layers.relu,
                layers.Parallel([
                    layers.Sequential([
                        layers.Linear(size2, latent_dim, scope='linear_s', factor=0.01),
                        layers.ScaleTanh(latent_dim, scope='scale_s')
                    ]),
                    layers.Linear(size2, latent_dim, scope='linear_t', factor=0.01),
                    layers.Sequential([
                        layers.Linear(size2, latent_dim, scope='linear_f', factor=0.01),
                        layers.ScaleTanh(latent_dim, scope='scale_f'),
                    ])
                ])
            ])
    return net_factory
