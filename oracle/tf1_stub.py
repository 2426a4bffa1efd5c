"""Test-only stand-in for the ~45 TensorFlow-1.x symbols the reference hot path touches.

TEST INFRASTRUCTURE -- container-only.  TensorFlow 1.x is not installable offline, so
`oracle/make_goldens.py` registers this module as ``sys.modules['tensorflow']`` and then
imports the reference's own ``utils/dynamics.py``, ``utils/layers.py``,
``utils/distributions.py`` (unchanged, from /root/reference at run time) and
``utils/sampler.py`` (after ``expandtabs(8)``, i.e. Python-2 tab semantics).  Every op is
executed eagerly on torch-CPU in float32, one torch op per TF op.  Nothing here is shipped
or imported by the product package, and nothing from the reference is copied.

Semantics pinned here because they are *not* in the reference tree:
  * ``tf.where`` with a rank-1 condition selects whole rows (sampler.py:55);
  * ``tf.random_uniform(maxval=2, dtype=int32)`` is in {0,1} (sampler.py:34);
  * ``tf.gradients(y, x)`` differentiates ``sum(y)`` (dynamics.py:218);
  * python floats / numpy arrays entering an op are converted to float32 first.

Every random draw is recorded in ``RANDOM_LOG`` (in call order) so goldens can carry the
exact injected randomness.
"""
import contextlib
import math
import sys
import types

import numpy as np
import torch

float32 = torch.float32
int32 = torch.int32
pi = math.pi

RANDOM_LOG = []          # list of (kind, numpy array) in call order
VARIABLES = {}           # full scoped name -> tensor
_SCOPE = []
_GEN = torch.Generator().manual_seed(0)
VARIABLE_HOOK = None     # callable(fullname, shape, default_tensor) -> tensor or None


def reset(seed=0):
    RANDOM_LOG.clear()
    VARIABLES.clear()
    del _SCOPE[:]
    _GEN.manual_seed(seed)


def _t(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x if x.dtype == dtype or not x.dtype.is_floating_point else x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


def constant(value, dtype=float32, name=None):
    return _t(value, dtype)


def placeholder(dtype, shape=None, name=None):
    # never fed in the golden generator (temperature is unused unless use_temperature)
    return torch.ones((), dtype=dtype)


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    _SCOPE.append(name)
    try:
        yield
    finally:
        _SCOPE.pop()


def constant_initializer(value, dtype=float32):
    return lambda shape: torch.full(tuple(shape), float(value), dtype=dtype)


def _variance_scaling_initializer(factor=2.0, mode='FAN_IN', uniform=False, dtype=float32):
    assert mode == 'FAN_IN' and not uniform

    def init(shape):
        fan_in = float(shape[0])
        std = math.sqrt(1.3 * factor / fan_in)
        w = torch.randn(tuple(shape), generator=_GEN, dtype=dtype)
        return torch.clamp(w, -2.0, 2.0) * std
    return init


def get_variable(name, shape=None, initializer=None, trainable=True, dtype=float32):
    full = '/'.join(_SCOPE + [name])
    if isinstance(initializer, torch.Tensor):
        val = initializer.clone()
    else:
        val = initializer(shape)
    if VARIABLE_HOOK is not None:
        new = VARIABLE_HOOK(full, tuple(val.shape), val)
        if new is not None:
            val = _t(new).reshape(val.shape).clone()
    if trainable and val.dtype.is_floating_point:
        val = val.detach().clone().requires_grad_(True)     # so tf.gradients(loss, variables) works
    VARIABLES[full] = val
    return val


# ---- elementwise / linear algebra -------------------------------------------------------
def log(x, name=None): return torch.log(_t(x))
def exp(x, name=None): return torch.exp(_t(x))
def cos(x, name=None): return torch.cos(_t(x))
def sin(x, name=None): return torch.sin(_t(x))
def square(x, name=None): return torch.square(_t(x))
def multiply(a, b, name=None): return _t(a) * _t(b)
def add(a, b, name=None): return _t(a) + _t(b)
def matmul(a, b, name=None): return torch.matmul(_t(a), _t(b))
def transpose(a, name=None): return _t(a).t()
def diag_part(a, name=None): return torch.diagonal(_t(a))
def minimum(a, b, name=None): return torch.minimum(_t(a), _t(b))
def less(a, b, name=None): return _t(a) < _t(b)
def greater(a, b, name=None): return _t(a) > _t(b)
def is_finite(x, name=None): return torch.isfinite(x)
def zeros_like(x, name=None): return torch.zeros_like(_t(x))
def stop_gradient(x, name=None): return x.detach()
def check_numerics(x, message=None): return x
def sqrt(x, name=None): return torch.sqrt(_t(x))
def linspace(start, stop, num, name=None): return torch.linspace(float(start), float(stop), int(num), dtype=float32)
def stack(values, axis=0, name=None): return torch.stack([_t(v) for v in values], dim=axis)


def split(value, num_or_size_splits, axis=0, name=None):
    return list(torch.chunk(_t(value), int(num_or_size_splits), dim=axis))


def zeros(shape, dtype=float32, name=None):
    return torch.zeros(tuple(int(s) for s in shape), dtype=dtype)


def reduce_sum(x, axis=None, name=None):
    return _t(x).sum() if axis is None else _t(x).sum(dim=axis)


def reduce_mean(x, axis=None, name=None):
    return _t(x).mean() if axis is None else _t(x).mean(dim=axis)


def reduce_logsumexp(x, axis=None, name=None):
    x = _t(x)
    return torch.logsumexp(x.reshape(-1), dim=0) if axis is None else torch.logsumexp(x, dim=axis)


def shape(x, name=None):
    return list(x.shape)


def cast(x, dtype, name=None):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.as_tensor(x).to(dtype)


def gather(params, indices, name=None):
    return params[int(indices)]


def expand_dims(x, axis, name=None): return x.unsqueeze(axis)


def squeeze(x, axis=None, name=None):
    if isinstance(x, (list, tuple)):
        x = torch.stack([_t(e) for e in x])
    return x.squeeze() if axis is None else x.squeeze(axis)


def tile(x, multiples, name=None):
    return x.repeat(*[int(m) for m in multiples])


def concat(values, axis, name=None):
    return torch.cat([_t(v) for v in values], dim=axis)


def where(cond, a, b, name=None):
    a, b = _t(a), _t(b)
    if cond.dim() == 1 and a.dim() > 1:      # TF1 rank-1 condition == row select
        cond = cond.view(-1, *([1] * (a.dim() - 1)))
    return torch.where(cond, a, b)


# ---- randomness (recorded) -----------------------------------------------------------------
def random_normal(shape, name=None, dtype=float32):
    r = torch.randn(tuple(int(s) for s in shape), generator=_GEN, dtype=dtype)
    RANDOM_LOG.append(('normal', r.numpy().copy()))
    return r


def random_uniform(shape, minval=0, maxval=None, dtype=float32, name=None):
    shape = tuple(int(s) for s in shape)
    if dtype == int32:
        r = torch.randint(int(minval), int(maxval), shape, generator=_GEN, dtype=torch.int32)
        RANDOM_LOG.append(('randint', r.numpy().copy()))
        return r
    r = torch.rand(shape, generator=_GEN, dtype=dtype)
    RANDOM_LOG.append(('uniform', r.numpy().copy()))
    return r


# ---- autodiff / control flow -----------------------------------------------------------------
def gradients(ys, xs, name=None):
    single = not isinstance(xs, (list, tuple))
    xs_ = [xs] if single else list(xs)
    # create_graph: like TF1, an outer tf.gradients differentiates THROUGH this one (the training
    # loss back-propagates through grad_energy, i.e. needs Hessian-vector products of U)
    g = torch.autograd.grad(ys.sum(), xs_, retain_graph=True, create_graph=True, allow_unused=True)
    return list(g)


def while_loop(cond, body, loop_vars, **kw):
    vs = list(loop_vars)
    while bool(cond(*vs)):
        vs = list(body(*vs))
    return vs


LAST_SCAN = []           # outputs of the most recent tf.scan (utils/ais.py:68), for the golden generator


def scan(fn, elems, initializer=None, **kw):
    """tf.scan over the leading axis of one tensor `elems` with a tuple accumulator."""
    acc, outs = tuple(initializer), []
    for e in elems:
        acc = tuple(fn(acc, e))
        outs.append(acc)
    res = tuple(torch.stack([_t(o[i]) for o in outs]) for i in range(len(acc)))
    LAST_SCAN[:] = res
    return res


def _make_module():
    m = types.ModuleType('tensorflow')
    me = sys.modules[__name__]
    for k, v in vars(me).items():
        if not k.startswith('__'):
            setattr(m, k, v)
    nn = types.ModuleType('tensorflow.nn')
    nn.relu = lambda x, name=None: torch.relu(_t(x))
    nn.tanh = lambda x, name=None: torch.tanh(_t(x))
    nn.softplus = lambda x, name=None: torch.nn.functional.softplus(_t(x))
    # TF's stable form max(l, 0) - l z + log1p(exp(-|l|))  (mnist_vae.py:124)
    nn.sigmoid_cross_entropy_with_logits = lambda labels=None, logits=None, name=None: (
        torch.clamp(_t(logits), min=0) - _t(logits) * _t(labels) + torch.log1p(torch.exp(-torch.abs(_t(logits)))))
    m.nn = nn
    contrib = types.ModuleType('tensorflow.contrib')
    layers = types.ModuleType('tensorflow.contrib.layers')
    layers.variance_scaling_initializer = _variance_scaling_initializer
    contrib.layers = layers
    m.contrib = contrib
    m._stub = me
    return m


def install():
    """Register the stub as `tensorflow` (idempotent) and return the stub module object."""
    import importlib.machinery
    m = _make_module()
    # (torch's lazy `torch._dynamo` import probes importlib.util.find_spec('tensorflow'), which raises on a module
    #  whose __spec__ is None)
    m.__spec__ = importlib.machinery.ModuleSpec('tensorflow', None)
    sys.modules['tensorflow'] = m
    sys.modules['tensorflow.nn'] = m.nn
    sys.modules['tensorflow.contrib'] = m.contrib
    sys.modules['tensorflow.contrib.layers'] = m.contrib.layers
    return m
