"""CPU: the non-obvious TensorFlow-1.x semantics `oracle/tf1_stub.py` stands in for, pinned against values
computed by hand.  Every golden fixture is (reference Python) o (this stub), so these four are the places
where a wrong stub would silently bend the parity anchor ("third-party arithmetic"):

  * `tf.where` with a rank-1 condition selects whole ROWS          (utils/sampler.py:55)
  * `tf.squeeze` of a python LIST of scalars stacks them first      (utils/dynamics.py:100-103)
  * `tf.gradients(y, x)` is d(sum y)/dx and is itself differentiable (utils/dynamics.py:218; the training loss
    back-propagates through grad_energy)
  * `tf.nn.sigmoid_cross_entropy_with_logits` is max(l,0) - l z + log1p(exp(-|l|))   (mnist_vae.py:124)
plus the integer uniform of utils/sampler.py:34 and the truncated-normal variance scaling of utils/layers.py:32.
"""
import math

import numpy as np
import torch

from oracle import tf1_stub as tf


def test_where_with_rank1_condition_selects_rows():
    cond = torch.tensor([True, False, True])
    a = torch.arange(6, dtype=torch.float32).reshape(3, 2)          # [[0,1],[2,3],[4,5]]
    b = -torch.ones(3, 2)
    out = tf.where(cond, a, b).numpy()
    assert np.array_equal(out, np.array([[0, 1], [-1, -1], [4, 5]], dtype=np.float32))
    # same-rank conditions stay elementwise
    c2 = torch.tensor([[True, False], [False, True], [True, True]])
    assert np.array_equal(tf.where(c2, a, b).numpy(), np.array([[0, -1], [-1, 3], [4, 5]], dtype=np.float32))


def test_squeeze_of_a_list_of_scalars_is_a_vector():
    t = torch.tensor(0.25)
    out = tf.squeeze([tf.cos(t), tf.sin(t)])
    assert out.shape == (2,) and out.dtype == torch.float32
    assert np.allclose(out.numpy(), [math.cos(0.25), math.sin(0.25)], atol=1e-7)
    # tile(expand_dims(.)) as dynamics.py:104-105 uses it
    tiled = tf.tile(tf.expand_dims(out, 0), (3, 1))
    assert tiled.shape == (3, 2) and np.array_equal(tiled[2].numpy(), out.numpy())


def test_gradients_differentiate_the_batch_sum_and_are_differentiable():
    x = torch.tensor([[1.0, 2.0], [3.0, -1.0]], requires_grad=True)
    y = tf.reduce_sum(tf.square(x) * x, axis=1)                     # per-row sum of x^3: shape (2,)
    (g,) = tf.gradients(y, x)
    assert np.allclose(g.detach().numpy(), 3.0 * x.detach().numpy() ** 2)          # d(sum_n y_n)/dx = 3 x^2
    # second order: d/dx sum(g * w) = 6 x w  (what the training loss needs through grad_energy)
    w = torch.tensor([[1.0, 0.5], [-2.0, 4.0]])
    (h,) = tf.gradients(tf.reduce_sum(g * w), x)
    assert np.allclose(h.detach().numpy(), 6.0 * x.detach().numpy() * w.numpy())
    # a single (non-list) `xs` still returns a list (dynamics.py:218 indexes [0])
    assert isinstance(tf.gradients(tf.reduce_sum(x), x), list)


def test_sigmoid_cross_entropy_is_tfs_stable_form():
    m = tf._make_module()
    logits = torch.tensor([-30.0, -1.0, 0.0, 2.0, 40.0, 100.0])
    labels = torch.tensor([1.0, 0.0, 1.0, 1.0, 0.0, 1.0])
    out = m.nn.sigmoid_cross_entropy_with_logits(labels=labels, logits=logits).numpy().astype(np.float64)
    l, z = logits.numpy().astype(np.float64), labels.numpy().astype(np.float64)
    want = np.maximum(l, 0) - l * z + np.log1p(np.exp(-np.abs(l)))
    assert np.all(np.isfinite(out)) and np.allclose(out, want, rtol=2e-7, atol=1e-7)
    # hand values: l=0,z=1 -> log 2; l=40,z=0 -> 40 (+4e-18); l=-30,z=1 -> 30 (+9e-14)
    assert abs(out[2] - math.log(2.0)) < 1e-7 and abs(out[4] - 40.0) < 1e-5 and abs(out[0] - 30.0) < 1e-5


def test_integer_uniform_is_a_bit_and_draws_are_logged_in_call_order():
    tf.reset(3)
    r = tf.random_uniform((4096, 1), maxval=2, dtype=tf.int32)
    assert set(np.unique(r.numpy())) == {0, 1}
    u = tf.random_uniform((5,))
    n = tf.random_normal((2, 3))
    assert [k for k, _ in tf.RANDOM_LOG] == ['randint', 'uniform', 'normal']
    assert np.array_equal(tf.RANDOM_LOG[1][1], u.numpy()) and np.array_equal(tf.RANDOM_LOG[2][1], n.numpy())
    assert u.min() >= 0.0 and u.max() < 1.0


def test_variance_scaling_is_truncated_normal_with_fan_in_std():
    tf.reset(5)
    init = tf._variance_scaling_initializer(factor=2.0 * (1.0 / 3))          # layers.py:32 with factor 1/3
    w = init((400, 50)).numpy()
    std = math.sqrt(1.3 * 2.0 / 3 / 400)
    assert np.abs(w).max() <= 2.0 * std + 1e-9                              # truncation at two sigma
    # the stub CLAMPS at two sigma (TF re-samples; std 0.95 vs 0.88 of the nominal) -- initial values never
    # enter a parity check (goldens carry explicit weights), only the scale matters
    assert 0.85 < w.std() / std < 1.0


def test_constants_and_python_scalars_enter_ops_as_float32():
    assert tf.constant(0.1).dtype == torch.float32
    assert tf.minimum(torch.tensor([1.0, -2.0]), 0.0).dtype == torch.float32
    assert tf.multiply(np.float64(0.5), torch.ones(2)).dtype == torch.float32
    assert float(tf.cast(torch.tensor(3.7), tf.int32)) == 3.0                 # mask index cast, dynamics.py:96
