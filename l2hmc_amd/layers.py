"""Layer kit the S/T/Q networks are assembled from -- API of the reference's utils/layers.py.

Same constructors and call protocol as utils/layers.py:29-95 (`Linear`, `ConcatLinear`,
`Parallel`, `Sequential`, `ScaleTanh`, `Zip`), so a reference-style ``net_factory``
(SCGExperiment.ipynb raw lines 51-78) builds its net unchanged, with `variable_scope` and
`relu` below standing in for `tf.variable_scope` / `tf.nn.relu`.

These objects are PARAMETER HOLDERS: `Dynamics` recognises the S/T/Q architecture
(`extract_stq`) and hands the raw (in, out) weights to the fused HIP kernels, which
evaluate the net inside the leapfrog kernel.  Calling a layer directly evaluates it with
torch ops on the tensor's device -- that exists for API compatibility / inspection only and
is never used by `Dynamics` or `propose`.
"""
import contextlib
import math

import torch

_SCOPE = []
_DEFAULT_DEVICE = [None]


def default_device():
    if _DEFAULT_DEVICE[0] is None:
        _DEFAULT_DEVICE[0] = torch.device("cuda", torch.cuda.current_device()) \
            if torch.cuda.is_available() else torch.device("cpu")
    return _DEFAULT_DEVICE[0]


def set_default_device(device):
    _DEFAULT_DEVICE[0] = torch.device(device)


@contextlib.contextmanager
def variable_scope(name):
    """Stand-in for `tf.variable_scope`: only contributes to parameter names."""
    _SCOPE.append(name)
    try:
        yield
    finally:
        _SCOPE.pop()


def _full_name(scope, leaf):
    return "/".join(_SCOPE + [scope, leaf])


def relu(x):
    return torch.relu(x)


def softplus(x):
    return torch.nn.functional.softplus(x)


def _variance_scaling(shape, factor):
    """`tf.contrib.layers.variance_scaling_initializer(factor, 'FAN_IN', uniform=False)`
    as used at layers.py:32: truncated normal (+-2 sigma), stddev sqrt(1.3 * factor / fan_in)."""
    std = math.sqrt(1.3 * factor / shape[0])
    w = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std)
    return w


class Linear(object):
    """layers.py:29-37: y = x W + b, W (in_, out_), b (out_,)."""

    def __init__(self, in_, out_, scope='linear', factor=1.0):
        dev = default_device()
        self.in_, self.out_ = in_, out_
        self.name = _full_name(scope, '')[:-1]
        self.W = torch.nn.Parameter(_variance_scaling((in_, out_), factor * 2.0).to(dev))
        self.b = torch.nn.Parameter(torch.zeros(out_, dtype=torch.float32, device=dev))

    def parameters(self):
        return [(self.name + '/W', self.W), (self.name + '/b', self.b)]

    def __call__(self, x):
        return torch.add(torch.matmul(x, self.W), self.b)


class ConcatLinear(object):
    """layers.py:40-58."""

    def __init__(self, ins_, out_, factors=None, scope='concat_linear'):
        self.layers = []
        with variable_scope(scope):
            for i, in_ in enumerate(ins_):
                factor = 1.0 if factors is None else factors[i]
                self.layers.append(Linear(in_, out_, scope='linear_%d' % i, factor=factor))

    def parameters(self):
        return [p for l in self.layers for p in l.parameters()]

    def __call__(self, inputs):
        output = 0.
        for i, x in enumerate(inputs):
            output += self.layers[i](x)
        return output


class Parallel(object):
    """layers.py:60-66."""

    def __init__(self, layers=None):
        self.layers = [] if layers is None else layers       # (no shared mutable default)

    def add(self, layer):
        self.layers.append(layer)

    def parameters(self):
        return _collect(self.layers)

    def __call__(self, x):
        return [layer(x) for layer in self.layers]


class Sequential(object):
    """layers.py:68-79."""

    def __init__(self, layers=None):
        self.layers = [] if layers is None else layers       # (no shared mutable default)

    def add(self, layer):
        self.layers.append(layer)

    def parameters(self):
        return _collect(self.layers)

    def __call__(self, x):
        y = x
        for layer in self.layers:
            y = layer(y)
        return y


class ScaleTanh(object):
    """layers.py:81-86: exp(scale) * tanh(x), scale a (1, in_) log-scale initialised to 0."""

    def __init__(self, in_, scope='scale_tanh'):
        self.name = _full_name(scope, 'scale')
        self.log_scale = torch.nn.Parameter(
            torch.zeros((1, in_), dtype=torch.float32, device=default_device()))

    @property
    def scale(self):
        return torch.exp(self.log_scale)

    def parameters(self):
        return [(self.name, self.log_scale)]

    def __call__(self, x):
        return self.scale * torch.tanh(x)


class Zip(object):
    """layers.py:88-95."""

    def __init__(self, layers=None):
        self.layers = [] if layers is None else layers       # (no shared mutable default)

    def parameters(self):
        return _collect(self.layers)

    def __call__(self, x):
        assert len(x) == len(self.layers)
        n = len(self.layers)
        return [self.layers[i](x[i]) for i in range(n)]


def _collect(layers):
    out = []
    for l in layers:
        if hasattr(l, 'parameters'):
            out.extend(l.parameters())
    return out


def stq_network(hidden=10, embed_factor=1.0 / 3, head_factor=0.001):
    """`net_factory` for the notebook's S/T/Q architecture (SCGExperiment.ipynb `network`,
    raw lines 51-78; mnist_vae.py:142-167 uses hidden=200, head_factor=0.01 plus an aux
    branch) with hidden width `hidden`.  Returns `network(x_dim, scope, factor)`."""
    def network(x_dim, scope, factor):
        with variable_scope(scope):
            net = Sequential([
                Zip([
                    Linear(x_dim, hidden, scope='embed_1', factor=embed_factor),
                    Linear(x_dim, hidden, scope='embed_2', factor=factor * embed_factor),
                    Linear(2, hidden, scope='embed_3', factor=embed_factor),
                    lambda _: 0.,
                ]),
                sum,
                relu,
                Linear(hidden, hidden, scope='linear_1'),
                relu,
                Parallel([
                    Sequential([
                        Linear(hidden, x_dim, scope='linear_s', factor=head_factor),
                        ScaleTanh(x_dim, scope='scale_s'),
                    ]),
                    Linear(hidden, x_dim, scope='linear_t', factor=head_factor),
                    Sequential([
                        Linear(hidden, x_dim, scope='linear_f', factor=head_factor),
                        ScaleTanh(x_dim, scope='scale_f'),
                    ]),
                ]),
            ])
        return net
    return network


def extract_mlp3(seq):
    """Recognise Sequential([Linear, softplus, Linear, softplus, Linear]) -- the VAE decoder and
    the sampler's `encoder_sampler` (mnist_vae.py:104-111,134-140) -- and return its weights, or
    None."""
    try:
        ls = seq.layers
        if not (isinstance(seq, Sequential) and len(ls) == 5):
            return None
        l1, a1, l2, a2, l3 = ls
        if not (all(isinstance(l, Linear) for l in (l1, l2, l3))
                and all(callable(a) and getattr(a, '__name__', '') == 'softplus' for a in (a1, a2))):
            return None
        if not (l1.out_ == l2.in_ and l2.out_ == l3.in_):
            return None
        return {'W1': l1.W, 'b1': l1.b, 'W2': l2.W, 'b2': l2.b, 'W3': l3.W, 'b3': l3.b,
                'dims': (l1.in_, l1.out_, l2.out_, l3.out_)}
    except (AttributeError, TypeError):
        return None


_RELU_NAMES = ('relu',)


def _is_relu(f):
    return callable(f) and getattr(f, '__name__', '') in _RELU_NAMES


def extract_stq(net, x_dim):
    """Recognise the S/T/Q architecture of nb:51-78 and return its 16 parameter tensors
    keyed like include/l2hmc.h's L2hmcNet, or None if `net` has another structure."""
    try:
        seq = net.layers
        if not (isinstance(net, Sequential) and len(seq) == 6):
            return None
        z, s, r1, l1, r2, par = seq
        if not (isinstance(z, Zip) and len(z.layers) == 4 and s is sum and _is_relu(r1)
                and _is_relu(r2) and isinstance(l1, Linear) and isinstance(par, Parallel)
                and len(par.layers) == 3):
            return None
        e1, e2, e3, e4 = z.layers
        if not all(isinstance(e, Linear) for e in (e1, e2, e3)) or isinstance(e4, Linear):
            return None
        aux_encoder = extract_mlp3(e4)            # mnist_vae.py:149: image branch; else `lambda _: 0.`
        if aux_encoder is None and e4(None) != 0.:
            return None
        hs, ht, hq = par.layers
        if not (isinstance(hs, Sequential) and len(hs.layers) == 2 and isinstance(ht, Linear)
                and isinstance(hq, Sequential) and len(hq.layers) == 2):
            return None
        ls, ss = hs.layers
        lq, sq = hq.layers
        if not (isinstance(ls, Linear) and isinstance(ss, ScaleTanh) and isinstance(lq, Linear)
                and isinstance(sq, ScaleTanh)):
            return None
        H = l1.in_
        if aux_encoder is not None and aux_encoder['dims'][3] != H:
            return None
        shapes_ok = (e1.in_ == x_dim and e2.in_ == x_dim and e3.in_ == 2
                     and e1.out_ == e2.out_ == e3.out_ == l1.out_ == H
                     and ls.in_ == ht.in_ == lq.in_ == H
                     and ls.out_ == ht.out_ == lq.out_ == x_dim)
        if not shapes_ok:
            return None
        return {'W1': e1.W, 'b1': e1.b, 'W2': e2.W, 'b2': e2.b, 'W3': e3.W, 'b3': e3.b,
                'W4': l1.W, 'b4': l1.b, 'Ws': ls.W, 'bs': ls.b, 'Wt': ht.W, 'bt': ht.b,
                'Wq': lq.W, 'bq': lq.b, 'lam_s': ss.log_scale, 'lam_q': sq.log_scale, 'H': H,
                'aux_encoder': aux_encoder}
    except (AttributeError, TypeError):
        return None
