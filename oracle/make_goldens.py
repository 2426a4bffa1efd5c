"""Generate golden vectors by EXECUTING THE REFERENCE'S OWN MODULES (container-only tool).

TEST INFRASTRUCTURE.  Run here (never on the GPU box -- /root/reference does not exist
there):

    python oracle/make_goldens.py            # writes ALL of tests/golden/*.npz (44 files; bit-reproducible)
    python oracle/make_goldens.py ais|wide|train_funnel|train_vae|train_vae_extra|train_wide|ess|rough_ne   # only that group
    L2HMC_GOLDEN_OUT=/tmp/gold python oracle/make_goldens.py                  # elsewhere, to diff against the committed set

How: ``oracle/tf1_stub.py`` is registered as ``tensorflow``; ``/root/reference/utils`` is
put on ``sys.path`` at run time and ``dynamics``, ``layers``, ``distributions`` are imported
unchanged; ``sampler.py`` (tab/space mixed, Python-2 only) is read, ``expandtabs(8)``-ed
(exactly Py2's rule) and exec'd in memory.  The notebook's ``network`` factory
(SCGExperiment.ipynb raw lines 51-78) is restated below in terms of the reference's own
``layers`` classes.  Nothing of the reference is copied into the repo: only inputs,
weights, recorded random draws and the outputs the reference code produced are stored.

Each .npz holds, per case: energy parameters, net weights (xnet.*, vnet.*), eps, mask, the
inputs, the recorded random draws, and the reference outputs for `_forward_step`,
`_backward_step`, `forward`, `backward`, `p_accept` and `propose`.
"""
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import tf1_stub  # noqa: E402

REF = '/root/reference/utils'
OUT = os.environ.get('L2HMC_GOLDEN_OUT') or os.path.join(os.path.dirname(HERE), 'tests', 'golden')   # (override: verification runs)

tf = tf1_stub.install()
sys.path.insert(0, REF)
import contextlib  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import dynamics as ref_dynamics            # noqa: E402  /root/reference/utils/dynamics.py
    import layers as ref_layers                # noqa: E402
    import distributions as ref_distributions  # noqa: E402

_src = open(os.path.join(REF, 'sampler.py')).read().expandtabs(8)
ref_sampler = type(sys)('ref_sampler')
exec(compile(_src, os.path.join(REF, 'sampler.py'), 'exec'), ref_sampler.__dict__)

Linear, Sequential, Zip, Parallel, ScaleTanh = (ref_layers.Linear, ref_layers.Sequential,
                                                ref_layers.Zip, ref_layers.Parallel,
                                                ref_layers.ScaleTanh)


def make_network(H):
    """The notebook's S/T/Q net (nb:51-78) with hidden width H, built from the reference's
    own layer classes."""
    def network(x_dim, scope, factor):
        with tf.variable_scope(scope):
            net = Sequential([
                Zip([
                    Linear(x_dim, H, scope='embed_1', factor=1.0 / 3),
                    Linear(x_dim, H, scope='embed_2', factor=factor * 1.0 / 3),
                    Linear(2, H, scope='embed_3', factor=1.0 / 3),
                    lambda _: 0.,
                ]),
                sum,
                tf.nn.relu,
                Linear(H, H, scope='linear_1'),
                tf.nn.relu,
                Parallel([
                    Sequential([
                        Linear(H, x_dim, scope='linear_s', factor=0.001),
                        ScaleTanh(x_dim, scope='scale_s'),
                    ]),
                    Linear(H, x_dim, scope='linear_t', factor=0.001),
                    Sequential([
                        Linear(H, x_dim, scope='linear_f', factor=0.001),
                        ScaleTanh(x_dim, scope='scale_f'),
                    ]),
                ]),
            ])
        return net
    return network


TF2KEY = {'embed_1/W': 'W1', 'embed_1/b': 'b1', 'embed_2/W': 'W2', 'embed_2/b': 'b2',
          'embed_3/W': 'W3', 'embed_3/b': 'b3', 'linear_1/W': 'W4', 'linear_1/b': 'b4',
          'linear_s/W': 'Ws', 'linear_s/b': 'bs', 'linear_t/W': 'Wt', 'linear_t/b': 'bt',
          'linear_f/W': 'Wq', 'linear_f/b': 'bq', 'scale_s/scale': 'lam_s',
          'scale_f/scale': 'lam_q'}


def variable_hook_factory(seed, head_std):
    """Replace the reference initialisation (heads ~1e-3 => S,T,Q ~ 0, i.e. plain HMC) by
    seeded values that exercise every term: head weights std `head_std`, small random biases
    and log-scales."""
    rng = np.random.RandomState(seed)

    def hook(full, shape, default):
        if full == 'alpha':
            return None
        leaf = '/'.join(full.split('/')[-2:])
        if leaf.endswith('/W'):
            if leaf.startswith('linear_s') or leaf.startswith('linear_t') or leaf.startswith('linear_f'):
                return (rng.randn(*shape) * head_std / np.sqrt(shape[0])).astype(np.float32)
            return default          # reference's own variance-scaling init for embeds / linear_1
        if leaf.endswith('/b'):
            return (0.1 * rng.randn(*shape)).astype(np.float32)
        if leaf.endswith('/scale'):
            return (0.2 * rng.randn(*shape)).astype(np.float32)
        return None
    return hook


def npy(t):
    return t.detach().numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t)


def leaf(a):
    return torch.tensor(np.asarray(a, dtype=np.float32), requires_grad=True)


def run_case(name, x_dim, H, T, eps, N, energy_fn, energy_params, seed, hmc=False,
             head_std=1.0, x_scale=1.0, steps_to_check=(0, 3, None), x0=None, temperature=None, chainop=0):
    tf1_stub.reset(seed)
    np.random.seed(seed)                      # masks come from numpy's global RNG (dynamics.py:88)
    tf1_stub.VARIABLE_HOOK = variable_hook_factory(seed + 1, head_std)
    with contextlib.redirect_stdout(io.StringIO()):
        dyn = ref_dynamics.Dynamics(x_dim, energy_fn, T=T, eps=eps, hmc=hmc,
                                    net_factory=None if hmc else make_network(H),
                                    use_temperature=temperature is not None)
    out = dict(energy_params)
    if temperature is not None:
        # dynamics.py:47,204-205: `temperature` is a placeholder the caller feeds; feeding = assigning here
        dyn.temperature = torch.tensor(float(temperature), dtype=torch.float32)
        out['temperature'] = np.float32(temperature)
    out.update(case=name, x_dim=x_dim, H=H, T=T, N=N, hmc=int(hmc),
               eps=npy(dyn.eps), mask=npy(dyn.mask))
    if not hmc:
        for full, val in tf1_stub.VARIABLES.items():
            if full == 'alpha':
                continue
            scope, rest = full.split('/', 1)
            out['%s.%s' % (scope.lower(), TF2KEY[rest])] = npy(val)

    rng = np.random.RandomState(seed + 2)
    if x0 is None:
        x0 = (x_scale * rng.randn(N, x_dim)).astype(np.float32)
    v0 = rng.randn(N, x_dim).astype(np.float32)
    out['x'], out['v'] = x0, v0

    # energy / grad / hamiltonian  (dynamics.py:203-218)
    xl = leaf(x0)
    out['energy'] = npy(dyn.energy(xl))
    out['grad_energy'] = npy(dyn.grad_energy(xl))

    # single steps (dynamics.py:115-201)
    steps = [T - 1 if s is None else s for s in steps_to_check if (s is None or s < T)]
    out['steps'] = np.array(steps, dtype=np.int32)
    for s in steps:
        st = torch.tensor(float(s))
        xo, vo, lj = dyn._forward_step(leaf(x0), leaf(v0), st)
        out['fstep%d.x' % s], out['fstep%d.v' % s], out['fstep%d.logdet' % s] = npy(xo), npy(vo), npy(lj)
        xo, vo, lj = dyn._backward_step(leaf(x0), leaf(v0), st)
        out['bstep%d.x' % s], out['bstep%d.v' % s], out['bstep%d.logdet' % s] = npy(xo), npy(vo), npy(lj)

    # full trajectories (dynamics.py:246-300)
    for nm, fn in (('fwd', dyn.forward), ('bwd', dyn.backward)):
        X, V, lj = fn(leaf(x0), init_v=leaf(v0), log_jac=True)
        out[nm + '.x'], out[nm + '.v'], out[nm + '.logjac'] = npy(X), npy(V), npy(lj)
        X, V, p = fn(leaf(x0), init_v=leaf(v0))
        out[nm + '.p'] = npy(p)
        assert np.array_equal(npy(X), out[nm + ".x"], equal_nan=True)

    # propose with do_mh_step (sampler.py:28-55); randomness recorded by the stub
    del tf1_stub.RANDOM_LOG[:]
    Lx, Lv, px, outs = ref_sampler.propose(leaf(x0), dyn, do_mh_step=True)
    log = list(tf1_stub.RANDOM_LOG)
    kinds = [k for k, _ in log]
    if hmc:
        assert kinds == ['normal', 'uniform'], kinds
        out['prop.v_fwd'], out['prop.u'] = log[0][1], log[1][1]
    else:
        assert kinds == ['randint', 'normal', 'normal', 'uniform'], kinds
        out['prop.dir'] = log[0][1][:, 0].astype(np.uint8)
        out['prop.v_fwd'], out['prop.v_bwd'], out['prop.u'] = log[1][1], log[2][1], log[3][1]
    out['prop.Lx'], out['prop.px'], out['prop.x_next'] = npy(Lx), npy(px), npy(outs[0])
    assert Lv is None or hmc
    if chainop:
        chain_operator_block(out, dyn, x0, chainop, hmc=hmc)

    for k, v in out.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float64:
            raise AssertionError('float64 leaked into golden %s/%s' % (name, k))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('%-18s N=%-4d d=%-3d T=%-3d  mean p fwd %.3f bwd %.3f  |x|max %.2f' % (
        name, N, x_dim, T, out['fwd.p'].mean(), out['bwd.p'].mean(), np.abs(out['fwd.x']).max()))


def chain_operator_block(out, dyn, x0, nb_steps, hmc=False, aux=None):
    """The reference's own `chain_operator` (sampler.py:57-85) on the case's sampler: `nb_steps` composed
    `propose(..., log_jac=True)` calls, one accept against (init_x, init_v), MH select.  `init_v=None` (the only
    form the reference's callers use, eval_sampler.py:162 / mnist_vae.py:196; `if not init_v` on a tensor would
    raise, :58): the start momentum is the first logged normal draw.  Every draw is stored in call order."""
    del tf1_stub.RANDOM_LOG[:]
    fx, fv, p, outs = ref_sampler.chain_operator(leaf(x0), dyn, nb_steps, aux=aux, do_mh_step=True)
    log = list(tf1_stub.RANDOM_LOG)
    # HMC mode (sampler.py:29-31): propose threads `init_v` into forward(), ignores `log_jac` (the summed
    # "log-Jacobian" is a sum of accept probabilities) and always draws one MH uniform, which is discarded
    per = ['uniform'] if hmc else ['randint', 'normal', 'normal']
    assert [k for k, _ in log] == ['normal'] + per * nb_steps + ['uniform'], [k for k, _ in log]
    out['chainop.K'] = np.int32(nb_steps)
    out['chainop.init_v'] = log[0][1]
    body = log[1:-1]
    if not hmc:
        out['chainop.dir'] = np.stack([body[3 * k][1][:, 0] for k in range(nb_steps)]).astype(np.uint8)
        out['chainop.v_fwd'] = np.stack([body[3 * k + 1][1] for k in range(nb_steps)])
        out['chainop.v_bwd'] = np.stack([body[3 * k + 2][1] for k in range(nb_steps)])
    out['chainop.u'] = log[-1][1]
    out['chainop.x'], out['chainop.v'], out['chainop.p'] = npy(fx), npy(fv), npy(p)
    out['chainop.x_next'] = npy(outs[0])


def train_case(name, mu, cov, H, T, eps, N, seed, head_std=0.3, dist=None, params=None, x_start=None):
    """Gradient of the notebook's training loss (SCGExperiment.ipynb raw lines 156-169) w.r.t. every
    variable, produced by the reference's own graph (propose on x with MH + propose on z,
    sampler.py:28-51) differentiated by the stub's tf.gradients (= torch autograd)."""
    tf1_stub.reset(seed)
    np.random.seed(seed)
    tf1_stub.VARIABLE_HOOK = variable_hook_factory(seed + 1, head_std)
    x_dim = len(mu)
    with contextlib.redirect_stdout(io.StringIO()):
        if dist is None:
            dist = ref_distributions.Gaussian(np.asarray(mu, dtype=np.float64), np.asarray(cov, dtype=np.float64))
            params = {'energy.kind': 'gaussian', 'energy.mu': dist.mu.astype(np.float32),
                      'energy.i_sigma': dist.i_sigma.astype(np.float32)}
        dyn = ref_dynamics.Dynamics(x_dim, dist.get_energy_function(), T=T, eps=eps, net_factory=make_network(H))
    out = dict(params)
    out.update({'case': name, 'x_dim': x_dim, 'H': H,
                'T': T, 'N': N, 'hmc': 0, 'eps': npy(dyn.eps), 'mask': npy(dyn.mask)})
    names = []
    for full, val in tf1_stub.VARIABLES.items():
        if full == 'alpha':
            continue
        scope, rest = full.split('/', 1)
        out['%s.%s' % (scope.lower(), TF2KEY[rest])] = npy(val)
        names.append((full, '%s.%s' % (scope.lower(), TF2KEY[rest])))
    rng = np.random.RandomState(seed + 2)
    if x_start is not None:
        x = np.asarray(x_start(rng), dtype=np.float32)
    else:
        C = np.linalg.cholesky(np.asarray(cov))
        x = (rng.randn(N, x_dim) @ C.T + np.asarray(mu)).astype(np.float32)
    z = rng.randn(N, x_dim).astype(np.float32)
    out['x'], out['z'] = x, z
    del tf1_stub.RANDOM_LOG[:]
    xt, zt = leaf(x), leaf(z)
    Lx, _, px, output = ref_sampler.propose(xt, dyn, do_mh_step=True)
    Lz, _, pz, _ = ref_sampler.propose(zt, dyn, do_mh_step=False)
    log = list(tf1_stub.RANDOM_LOG)
    kinds = [k for k, _ in log]
    assert kinds == ['randint', 'normal', 'normal', 'uniform', 'randint', 'normal', 'normal'], kinds
    out['x.dir'], out['x.v_fwd'], out['x.v_bwd'], out['x.u'] = log[0][1][:, 0].astype(np.uint8), log[1][1], log[2][1], log[3][1]
    out['z.dir'], out['z.v_fwd'], out['z.v_bwd'] = log[4][1][:, 0].astype(np.uint8), log[5][1], log[6][1]
    # nb raw 164-169
    v1 = (tf.reduce_sum(tf.square(xt - Lx), axis=1) * px) + 1e-4
    v2 = (tf.reduce_sum(tf.square(zt - Lz), axis=1) * pz) + 1e-4
    scale = 0.1
    loss = scale * (tf.reduce_mean(1.0 / v1) + tf.reduce_mean(1.0 / v2))
    loss = loss + (- tf.reduce_mean(v1) - tf.reduce_mean(v2)) / scale
    variables = [tf1_stub.VARIABLES[full] for full, _ in names] + [tf1_stub.VARIABLES['alpha']]
    grads = tf.gradients(loss, variables)
    out['loss'] = npy(loss)
    out['Lx'], out['px'], out['Lz'], out['pz'] = npy(Lx), npy(px), npy(Lz), npy(pz)
    for (full, key), gr in zip(names, grads[:-1]):
        out['grad.' + key] = npy(gr)
    out['grad.alpha'] = npy(grads[-1])
    assert all(np.all(np.isfinite(out[k])) for k in out if k.startswith('grad.')), 'non-finite gradient'
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    gn = np.sqrt(sum(float(np.sum(out[k].astype(np.float64) ** 2)) for k in out if k.startswith('grad.')))
    print('%-18s N=%-4d d=%-3d T=%-3d  loss %.4e  |grad| %.3e  grad.alpha %.3e  mean px %.3f pz %.3f' % (
        name, N, x_dim, T, float(out['loss']), gn, float(out['grad.alpha']), out['px'].mean(), out['pz'].mean()))


def vae_case(name, latent, H, dec_h, n_pix, enc_h, T, eps, N, seed):
    """Config-5-shaped case at fixture-friendly sizes: the VAE latent posterior of
    mnist_vae.py:104-178 -- decoder energy (:122-126), S/T/Q nets whose 4th Zip branch is the shared
    `encoder_sampler(aux)` (:134-150) -- assembled from the reference's own layer classes and run
    through the reference's Dynamics / propose with aux."""
    tf1_stub.reset(seed)
    np.random.seed(seed)
    hook = variable_hook_factory(seed + 1, 0.3)

    def vae_hook(full, shape, default):
        if full.startswith('sampler/XNet') or full.startswith('sampler/VNet'):
            return hook('/'.join(full.split('/')[1:]), shape, default)
        return None
    tf1_stub.VARIABLE_HOOK = vae_hook
    with tf.variable_scope('decoder'):
        decoder = Sequential([Linear(latent, dec_h, scope='decoder_1'), tf.nn.softplus,
                              Linear(dec_h, dec_h, scope='decoder_2'), tf.nn.softplus,
                              Linear(dec_h, n_pix, scope='decoder_3', factor=0.01)])

    def energy(z, aux=None):                                    # mnist_vae.py:122-126
        logits = decoder(z)
        log_posterior = -tf.reduce_sum(tf.nn.sigmoid_cross_entropy_with_logits(labels=aux, logits=logits), axis=1)
        log_prior = -0.5 * tf.reduce_sum(tf.square(z), axis=1)
        return (-log_posterior - log_prior)

    with tf.variable_scope('sampler'):
        encoder_sampler = Sequential([Linear(n_pix, enc_h, scope='encoder_1'), tf.nn.softplus,
                                      Linear(enc_h, enc_h, scope='encoder_2'), tf.nn.softplus,
                                      Linear(enc_h, H, scope='encoder_3')])

        def net_factory(x_dim, scope, factor):                  # mnist_vae.py:142-167
            with tf.variable_scope(scope):
                return Sequential([
                    Zip([Linear(latent, H, scope='embed_1', factor=0.33),
                         Linear(latent, H, scope='embed_2', factor=factor * 0.33),
                         Linear(2, H, scope='embed_3', factor=0.33),
                         encoder_sampler]),
                    sum, tf.nn.relu, Linear(H, H, scope='linear_1'), tf.nn.relu,
                    Parallel([Sequential([Linear(H, latent, scope='linear_s', factor=0.01),
                                          ScaleTanh(latent, scope='scale_s')]),
                              Linear(H, latent, scope='linear_t', factor=0.01),
                              Sequential([Linear(H, latent, scope='linear_f', factor=0.01),
                                          ScaleTanh(latent, scope='scale_f')])])])
        with contextlib.redirect_stdout(io.StringIO()):
            dyn = ref_dynamics.Dynamics(latent, energy, T=T, eps=eps, net_factory=net_factory)
    out = {'energy.kind': 'vae', 'case': name, 'x_dim': latent, 'H': H, 'T': T, 'N': N, 'hmc': 0,
           'eps': npy(dyn.eps), 'mask': npy(dyn.mask)}
    for full, val in tf1_stub.VARIABLES.items():
        parts = full.split('/')
        if parts[0] == 'decoder':
            out['dec.%s%s' % (parts[2], parts[1][-1])] = npy(val)            # dec.W1 ... dec.b3
        elif parts[0] == 'sampler' and parts[1].startswith('encoder_'):
            out['enc.%s%s' % (parts[2], parts[1][-1])] = npy(val)
        elif parts[0] == 'sampler' and parts[1] in ('XNet', 'VNet'):
            out['%s.%s' % (parts[1].lower(), TF2KEY['/'.join(parts[2:])])] = npy(val)
    rng = np.random.RandomState(seed + 2)
    x0 = rng.randn(N, latent).astype(np.float32)
    v0 = rng.randn(N, latent).astype(np.float32)
    aux = (rng.rand(N, n_pix) < 0.3).astype(np.float32)          # binarised "image" rows
    out['x'], out['v'], out['aux'] = x0, v0, aux
    auxt = torch.tensor(aux)
    out['aux_h'] = npy(encoder_sampler(auxt))
    xl = leaf(x0)
    out['energy'] = npy(dyn.energy(xl, aux=auxt))
    out['grad_energy'] = npy(dyn.grad_energy(xl, aux=auxt))
    st = torch.tensor(2.0)
    xo, vo, lj = dyn._forward_step(leaf(x0), leaf(v0), st, aux=auxt)
    out['fstep2.x'], out['fstep2.v'], out['fstep2.logdet'] = npy(xo), npy(vo), npy(lj)
    xo, vo, lj = dyn._backward_step(leaf(x0), leaf(v0), st, aux=auxt)
    out['bstep2.x'], out['bstep2.v'], out['bstep2.logdet'] = npy(xo), npy(vo), npy(lj)
    out['steps'] = np.array([2], dtype=np.int32)
    for nm, fn in (('fwd', dyn.forward), ('bwd', dyn.backward)):
        X, V, lj = fn(leaf(x0), init_v=leaf(v0), aux=auxt, log_jac=True)
        out[nm + '.x'], out[nm + '.v'], out[nm + '.logjac'] = npy(X), npy(V), npy(lj)
        X, V, p = fn(leaf(x0), init_v=leaf(v0), aux=auxt)
        out[nm + '.p'] = npy(p)
    del tf1_stub.RANDOM_LOG[:]
    Lx, Lv, px, outs = ref_sampler.propose(leaf(x0), dyn, aux=auxt, do_mh_step=True)
    log = list(tf1_stub.RANDOM_LOG)
    assert [k for k, _ in log] == ['randint', 'normal', 'normal', 'uniform']
    out['prop.dir'] = log[0][1][:, 0].astype(np.uint8)
    out['prop.v_fwd'], out['prop.v_bwd'], out['prop.u'] = log[1][1], log[2][1], log[3][1]
    out['prop.Lx'], out['prop.px'], out['prop.x_next'] = npy(Lx), npy(px), npy(outs[0])
    chain_operator_block(out, dyn, x0, 3, aux=auxt)          # mnist_vae.py:196's image-conditioned form
    tf1_stub.VARIABLE_HOOK = None
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('%-18s N=%-4d d=%-3d T=%-3d  mean p fwd %.3f bwd %.3f  |U| %.1f' % (
        name, N, latent, T, out['fwd.p'].mean(), out['bwd.p'].mean(), np.abs(out['energy']).mean()))


def train_vae_case(name, latent, H, dec_h, n_pix, enc_h, T, eps, N, seed, energy_scale=0.0, rlc=0):
    """Gradient of the VAE experiment's sampler loss (mnist_vae.py:185-226 with MH = 1, energy_scale = 0):
        final_x, _, px, MH = propose(init_x, dynamics, aux=inp, do_mh_step=True)
        v = sum_k (final_x - init_x)^2 / (stop_gradient(exp(2 log_sigma)) + 1e-4) * px + 1e-4
        sampler_loss = mean(1 / v) - mean(v)
    w.r.t. every variable of the `sampler` scope (XNet, VNet, the shared image branch encoder_sampler, alpha),
    from the reference's own graph (decoder energy mnist_vae.py:122-126, nets :142-167) differentiated by the stub's
    tf.gradients.  Also d loss / d init_x (what flows into the previous proposal when MH > 1 and stop_gradient is
    off, :187-190,224) and the gradients of loss + sum(final_x * R) for a fixed R (the cotangent a later proposal
    sends back into final_x).
    energy_scale != 0: + energy_scale * (mean(1 / ed) - mean(ed)), ed = (energy(final_x) - energy(init_x))^2 px + 1e-4
    (:214,218,224).  rlc > 0 (`random_lf_composition`, :193-196): the proposal is
    chain_operator(init_x, dynamics, nb_steps ~ U{1..rlc-1}, aux, do_mh_step=True) instead of propose."""
    tf1_stub.reset(seed)
    np.random.seed(seed)
    hook = variable_hook_factory(seed + 1, 0.3)

    def vae_hook(full, shape, default):
        if full.startswith('sampler/XNet') or full.startswith('sampler/VNet'):
            return hook('/'.join(full.split('/')[1:]), shape, default)
        return None
    tf1_stub.VARIABLE_HOOK = vae_hook
    with tf.variable_scope('decoder'):
        decoder = Sequential([Linear(latent, dec_h, scope='decoder_1'), tf.nn.softplus,
                              Linear(dec_h, dec_h, scope='decoder_2'), tf.nn.softplus,
                              Linear(dec_h, n_pix, scope='decoder_3', factor=0.01)])

    def energy(z, aux=None):                                    # mnist_vae.py:122-126
        logits = decoder(z)
        log_posterior = -tf.reduce_sum(tf.nn.sigmoid_cross_entropy_with_logits(labels=aux, logits=logits), axis=1)
        log_prior = -0.5 * tf.reduce_sum(tf.square(z), axis=1)
        return (-log_posterior - log_prior)

    with tf.variable_scope('sampler'):
        encoder_sampler = Sequential([Linear(n_pix, enc_h, scope='encoder_1'), tf.nn.softplus,
                                      Linear(enc_h, enc_h, scope='encoder_2'), tf.nn.softplus,
                                      Linear(enc_h, H, scope='encoder_3')])

        def net_factory(x_dim, scope, factor):                  # mnist_vae.py:142-167
            with tf.variable_scope(scope):
                return Sequential([
                    Zip([Linear(latent, H, scope='embed_1', factor=0.33),
                         Linear(latent, H, scope='embed_2', factor=factor * 0.33),
                         Linear(2, H, scope='embed_3', factor=0.33),
                         encoder_sampler]),
                    sum, tf.nn.relu, Linear(H, H, scope='linear_1'), tf.nn.relu,
                    Parallel([Sequential([Linear(H, latent, scope='linear_s', factor=0.01),
                                          ScaleTanh(latent, scope='scale_s')]),
                              Linear(H, latent, scope='linear_t', factor=0.01),
                              Sequential([Linear(H, latent, scope='linear_f', factor=0.01),
                                          ScaleTanh(latent, scope='scale_f')])])])
        with contextlib.redirect_stdout(io.StringIO()):
            dyn = ref_dynamics.Dynamics(latent, energy, T=T, eps=eps, net_factory=net_factory)
    out = {'energy.kind': 'vae', 'case': name, 'x_dim': latent, 'H': H, 'T': T, 'N': N, 'hmc': 0,
           'eps': npy(dyn.eps), 'mask': npy(dyn.mask)}
    names = []                                                   # (stub variable name, fixture key) of the trained ones
    alpha_name = None
    for full, val in tf1_stub.VARIABLES.items():
        parts = full.split('/')
        if parts[0] == 'decoder':
            out['dec.%s%s' % (parts[2], parts[1][-1])] = npy(val)            # dec.W1 ... dec.b3
        elif parts[0] == 'sampler' and parts[1].startswith('encoder_'):
            key = 'enc.%s%s' % (parts[2], parts[1][-1])
            out[key] = npy(val)
            names.append((full, key))
        elif parts[0] == 'sampler' and parts[1] in ('XNet', 'VNet'):
            key = '%s.%s' % (parts[1].lower(), TF2KEY['/'.join(parts[2:])])
            out[key] = npy(val)
            names.append((full, key))
        elif parts[-1] == 'alpha':
            alpha_name = full
    assert alpha_name is not None, list(tf1_stub.VARIABLES)
    rng = np.random.RandomState(seed + 2)
    x0 = rng.randn(N, latent).astype(np.float32)
    aux = (rng.rand(N, n_pix) < 0.3).astype(np.float32)          # binarised "image" rows
    log_sigma = (0.3 * rng.randn(N, latent) - 0.5).astype(np.float32)
    R = (0.05 * rng.randn(N, latent)).astype(np.float32)
    out['x'], out['aux'], out['log_sigma'], out['R'] = x0, aux, log_sigma, R
    auxt = torch.tensor(aux)
    del tf1_stub.RANDOM_LOG[:]
    init_x = leaf(x0)
    if rlc > 0:                                                   # mnist_vae.py:193-196
        nb_steps = tf.random_uniform((), minval=1, maxval=rlc, dtype=tf.int32)
        final_x, _, px, MH = ref_sampler.chain_operator(init_x, dyn, nb_steps, aux=auxt, do_mh_step=True)
        log = list(tf1_stub.RANDOM_LOG)
        K = int(nb_steps)
        assert [k for k, _ in log] == ['randint', 'normal'] + ['randint', 'normal', 'normal'] * K + ['uniform']
        out['chain.nb_steps'] = np.int32(K)
        out['chain.init_v'] = log[1][1]
        body = log[2:-1]
        out['chain.dir'] = np.stack([body[3 * k][1][:, 0] for k in range(K)]).astype(np.uint8)
        out['chain.v_fwd'] = np.stack([body[3 * k + 1][1] for k in range(K)])
        out['chain.v_bwd'] = np.stack([body[3 * k + 2][1] for k in range(K)])
        out['prop.u'] = log[-1][1]
    else:
        final_x, _, px, MH = ref_sampler.propose(init_x, dyn, aux=auxt, do_mh_step=True)
        log = list(tf1_stub.RANDOM_LOG)
        assert [k for k, _ in log] == ['randint', 'normal', 'normal', 'uniform']
        out['prop.dir'] = log[0][1][:, 0].astype(np.uint8)
        out['prop.v_fwd'], out['prop.v_bwd'], out['prop.u'] = log[1][1], log[2][1], log[3][1]
    # mnist_vae.py:207-224 (MH = 1)
    w = 1.0 / (tf.stop_gradient(tf.exp(2 * torch.tensor(log_sigma))) + 1e-4)
    v = tf.square(final_x - init_x) * w
    v = tf.reduce_sum(v, 1) * px + 1e-4
    loss = tf.reduce_mean(1.0 / v) - tf.reduce_mean(v)
    if energy_scale != 0.0:
        energy_diff = tf.square(energy(final_x, aux=auxt) - energy(init_x, aux=auxt)) * px + 1e-4
        loss = loss + energy_scale * (tf.reduce_mean(1.0 / energy_diff) - tf.reduce_mean(energy_diff))
        out['energy_scale'], out['ediff'] = np.float32(energy_scale), npy(energy_diff)
    variables = [tf1_stub.VARIABLES[full] for full, _ in names] + [tf1_stub.VARIABLES[alpha_name], init_x]
    grads = tf.gradients(loss, variables)
    loss2 = loss + tf.reduce_sum(final_x * torch.tensor(R))
    grads2 = tf.gradients(loss2, variables)
    out['loss'], out['v'] = npy(loss), npy(v)
    out['Lx'], out['px'], out['x_next'] = npy(final_x), npy(px), npy(MH[0])
    for pre, gs in (('grad.', grads), ('grad2.', grads2)):
        for (full, key), gr in zip(names, gs[:-2]):
            out[pre + key] = npy(gr)
        out[pre + 'alpha'] = npy(gs[-2])
        out[pre + 'x0'] = npy(gs[-1])
    assert all(np.all(np.isfinite(out[k])) for k in out if k.startswith('grad')), 'non-finite gradient'
    tf1_stub.VARIABLE_HOOK = None
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    gn = np.sqrt(sum(float(np.sum(out[k].astype(np.float64) ** 2)) for k in out if k.startswith('grad.')))
    print('%-18s N=%-4d d=%-3d T=%-3d  loss %.4e  |grad| %.3e  grad.alpha %.3e  mean px %.3f  |grad.enc.W1| %.3e' % (
        name, N, latent, T, float(out['loss']), gn, float(out['grad.alpha']), out['px'].mean(),
        float(np.abs(out['grad.enc.W1']).max())))


def train_vae_extra_cases():
    """the two optional terms of the VAE experiment's sampler objective, from the reference's own graph"""
    train_vae_case('train_vae_small_es', latent=10, H=24, dec_h=48, n_pix=40, enc_h=32, T=4, eps=0.1, N=32, seed=44,
                   energy_scale=0.5)
    train_vae_case('train_vae_small_rlc', latent=10, H=24, dec_h=48, n_pix=40, enc_h=32, T=3, eps=0.1, N=32, seed=47,
                   energy_scale=0.25, rlc=4)


def train_wide_cases(only_new=False):
    """training gradients with nets wider than the register-resident kernels take (H > 15): GEMM-engine trainer"""
    var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), 50))
    if not only_new:
        _train_wide_gaussians(var)
    _train_wide_mixture_funnel()


def _train_wide_gaussians(var):
    train_case('train_icg50_h32', np.zeros(50), np.diag(var), H=32, T=4, eps=0.05, N=16, seed=37, head_std=0.05)
    rng = np.random.RandomState(7)
    R = np.linalg.qr(rng.randn(8, 8))[0]
    cov8 = R.T.dot(np.diag(np.exp(np.log(10.) * rng.uniform(-1, 1, size=8)))).dot(R)
    train_case('train_tilted8_h24', rng.randn(8) * 0.5, cov8, H=24, T=5, eps=0.1, N=32, seed=38)
    with contextlib.redirect_stdout(io.StringIO()):
        rw_t = ref_distributions.RoughWell(6, 0.3, easy=True)
    train_case('train_rough6_h20', np.zeros(6), None, H=20, T=5, eps=0.1, N=32, seed=39, head_std=0.3, dist=rw_t,
               params={'energy.kind': 'roughwell', 'energy.eta': np.float32(0.3), 'energy.easy': np.int32(1)},
               x_start=lambda rng: rng.randn(32, 6))


def _train_wide_mixture_funnel():
    # a 3-component mixture with unequal weights and a tilted component, and the funnel, under 20-wide nets
    mus_t = [np.array([2.0, 0.0, 0.5], dtype=np.float32), np.array([-2.0, 0.0, -0.5], dtype=np.float32),
             np.array([0.0, 1.5, 0.0], dtype=np.float32)]
    A = np.array([[0.6, 0.2, 0.0], [0.2, 0.5, 0.1], [0.0, 0.1, 0.4]])
    with contextlib.redirect_stdout(io.StringIO()):
        gmm_t = ref_distributions.GMM([torch.tensor(m) for m in mus_t], [0.5 * np.eye(3), A, 0.3 * np.eye(3)], [0.5, 0.3, 0.2])
    train_case('train_mog3d_h20', np.zeros(3), None, H=20, T=5, eps=0.1, N=48, seed=40, head_std=0.3, dist=gmm_t,
               params={'energy.kind': 'gmm', 'energy.mus': np.stack(mus_t), 'energy.i_sigmas': np.stack(gmm_t.i_sigmas),
                       'energy.constants': np.array(gmm_t.constants, dtype=np.float32)},
               x_start=lambda r: np.stack(mus_t)[r.randint(0, 3, size=48)] + np.sqrt(0.4) * r.randn(48, 3))
    with contextlib.redirect_stdout(io.StringIO()):
        fun = ref_distributions.GaussianFunnel(dim=4)
    train_case('train_funnel4_h20', np.zeros(4), None, H=20, T=4, eps=0.05, N=32, seed=41, head_std=0.3, dist=fun,
               params={'energy.kind': 'funnel', 'energy.sigma': np.float32(2.0)},
               x_start=lambda rng: np.concatenate([rng.randn(32, 1) * 1.5, rng.randn(32, 3)], axis=1))


def gaussian_case(name, mu, cov, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        dist = ref_distributions.Gaussian(np.asarray(mu, dtype=np.float64), np.asarray(cov, dtype=np.float64))
    params = {'energy.kind': 'gaussian', 'energy.mu': dist.mu.astype(np.float32),
              'energy.i_sigma': dist.i_sigma.astype(np.float32)}
    run_case(name, len(mu), energy_fn=dist.get_energy_function(), energy_params=params, **kw)


def ais_case(name, d, final_fn, params, K, T, N, step_size, seed, num_splits=1, refresh=False, refreshment=0.1,
             aux=None):
    """utils/ais.py executed as is (ais_estimate, :30-82): init energy = the standard normal of
    eval_vae.py:55-56; every draw it makes is recorded in call order (v0, then normal + uniform per step)."""
    import types
    for modname in ('tensorflow.examples', 'tensorflow.examples.tutorials', 'tensorflow.examples.tutorials.mnist'):
        sys.modules.setdefault(modname, types.ModuleType(modname))
    sys.modules['tensorflow.examples.tutorials.mnist'].input_data = None
    with contextlib.redirect_stdout(io.StringIO()):
        import ais as ref_ais                      # /root/reference/utils/ais.py
        init = ref_distributions.Gaussian(np.zeros(d), np.eye(d)).get_energy_function()
    tf1_stub.reset(seed)
    np.random.seed(seed)
    tf1_stub.VARIABLE_HOOK = None
    rng = np.random.RandomState(seed + 1)
    x0 = rng.randn(N, d).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        est, mean_alpha = ref_ais.ais_estimate(init, final_fn, K, leaf(x0), step_size=step_size, leapfrogs=T,
                                               x_dim=d, num_splits=num_splits, refresh=refresh,
                                               refreshment=refreshment, aux=aux)
    log = tf1_stub.RANDOM_LOG
    kinds = [k for k, _ in log]
    assert kinds == ['normal'] + ['normal', 'uniform'] * K, kinds
    alpha, xs, ws, _ = tf1_stub.LAST_SCAN
    out = dict(params)
    out.update(case=name, x_dim=d, K=K, T=T, N=N, step_size=np.float32(step_size), num_splits=num_splits,
               refresh=int(refresh), refreshment=np.float32(refreshment), x=x0, v0=log[0][1],
               normals=np.stack([log[1 + 2 * i][1] for i in range(K)]),
               u=np.stack([log[2 + 2 * i][1] for i in range(K)]),
               estimate=npy(est), mean_alpha=npy(mean_alpha), x_final=npy(xs[-1]), w_final=npy(ws[-1]),
               alpha_sum=npy(alpha.sum(0)))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('%-18s AIS estimate %s  mean alpha %.4f' % (name, npy(est), float(mean_alpha)))


def ais_vae_case(name, latent, dec_h, n_pix, K, T, N, step_size, seed, num_splits=1):
    """eval_vae.py:43-64 at fixture-friendly sizes: decoder from the reference's layer classes, final energy
    = -log p(x|z) - log p(z) with the images as aux, through the reference's ais_estimate."""
    tf1_stub.reset(seed)
    np.random.seed(seed)
    tf1_stub.VARIABLE_HOOK = None
    with tf.variable_scope('decoder'):
        decoder = Sequential([Linear(latent, dec_h, scope='decoder_1'), tf.nn.softplus,
                              Linear(dec_h, dec_h, scope='decoder_2'), tf.nn.softplus,
                              Linear(dec_h, n_pix, scope='decoder_3', factor=0.5)])
    dec_w = {}
    for full, val in tf1_stub.VARIABLES.items():
        parts = full.split('/')
        if parts[0] == 'decoder':
            dec_w['dec.%s%s' % (parts[2], parts[1][-1])] = npy(val)
    rng = np.random.RandomState(seed + 2)
    aux = (rng.rand(N, n_pix) < 0.3).astype(np.float32)
    auxt = torch.tensor(aux)

    def final_energy(z, aux=None):                               # eval_vae.py:58-62
        logits = decoder(z)
        log_posterior = -tf.reduce_sum(tf.nn.sigmoid_cross_entropy_with_logits(labels=aux, logits=logits), axis=1)
        log_prior = -0.5 * tf.reduce_sum(tf.square(z), axis=1)
        return -log_posterior - log_prior
    params = dict(dec_w)
    params.update({'energy.kind': 'vae', 'aux': aux})
    ais_case(name, latent, final_energy, params, K=K, T=T, N=N, step_size=step_size, seed=seed + 3,
             num_splits=num_splits, aux=auxt)


def ais_cases():
    rng = np.random.RandomState(7)
    R = np.linalg.qr(rng.randn(8, 8))[0]
    cov8 = R.T.dot(np.diag(np.exp(np.log(10.) * rng.uniform(-1, 1, size=8)))).dot(R)
    mu8 = rng.randn(8) * 0.5
    with contextlib.redirect_stdout(io.StringIO()):
        g8 = ref_distributions.Gaussian(mu8, cov8)
    p8 = {'energy.kind': 'gaussian', 'energy.mu': mu8.astype(np.float32), 'energy.i_sigma': g8.i_sigma.astype(np.float32)}
    ais_case('ais_tilted8', 8, g8.get_energy_function(), p8, K=6, T=5, N=32, step_size=0.15, seed=51, num_splits=2)
    ais_case('ais_tilted8_refresh', 8, g8.get_energy_function(), p8, K=5, T=4, N=32, step_size=0.15, seed=52,
             refresh=True, refreshment=0.3)
    var = np.exp(np.linspace(np.log(1e-1), np.log(1e1), 50))
    with contextlib.redirect_stdout(io.StringIO()):
        g50 = ref_distributions.Gaussian(np.zeros(50), np.diag(var))
    p50 = {'energy.kind': 'gaussian', 'energy.mu': np.zeros(50, np.float32), 'energy.i_sigma': g50.i_sigma.astype(np.float32)}
    ais_case('ais_icg50', 50, g50.get_energy_function(), p50, K=8, T=5, N=32, step_size=0.05, seed=53, num_splits=4)
    mus = [np.array([2.0, 0.0], dtype=np.float32), np.array([-2.0, 0.0], dtype=np.float32)]
    gmm = ref_distributions.GMM([torch.tensor(m) for m in mus], [0.5 * np.eye(2), 0.5 * np.eye(2)], [0.5, 0.5])
    gfn = gmm.get_energy_function()
    pg = {'energy.kind': 'gmm', 'energy.mus': np.stack(mus), 'energy.i_sigmas': np.stack(gmm.i_sigmas),
          'energy.constants': np.array(gmm.constants, dtype=np.float32)}
    ais_case('ais_mog2d', 2, lambda z, aux=None: gfn(z), pg, K=7, T=5, N=48, step_size=0.2, seed=54)
    ais_vae_case('ais_vae', latent=10, dec_h=48, n_pix=40, K=6, T=4, N=32, step_size=0.1, seed=55, num_splits=2)


def wide_cases():
    """H > 15 (nb:51-78 with H != 10; mnist_vae.py:142-167 uses 200): the nets run on the GEMM engine."""
    var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), 50))
    rng = np.random.RandomState(5)
    x0 = (rng.randn(64, 50) * np.sqrt(var)).astype(np.float32)
    gaussian_case('icg50_h32', np.zeros(50), np.diag(var), H=32, T=10, eps=0.1, N=64, seed=61, head_std=0.05, x0=x0)
    rng = np.random.RandomState(7)
    R = np.linalg.qr(rng.randn(8, 8))[0]
    cov8 = R.T.dot(np.diag(np.exp(np.log(10.) * rng.uniform(-1, 1, size=8)))).dot(R)
    gaussian_case('tilted8_h24', rng.randn(8) * 0.5, cov8, H=24, T=7, eps=0.1, N=48, seed=62)


def train_funnel_case():
    """training gradient through the funnel's Hessian (distributions.py:155-180), start points in the free region"""
    with contextlib.redirect_stdout(io.StringIO()):
        fun = ref_distributions.GaussianFunnel(dim=3)
    train_case('train_funnel3', np.zeros(3), None, H=10, T=4, eps=0.05, N=32, seed=36, head_std=0.3, dist=fun,
               params={'energy.kind': 'funnel', 'energy.sigma': np.float32(2.0)},
               x_start=lambda rng: np.concatenate([rng.randn(32, 1) * 1.5, rng.randn(32, 2)], axis=1))


def ess_case():
    """The reference's own diagnostics (utils/func_utils.py:45-54,114-120), imported unchanged (its
    `tensorflow.examples...input_data` import is satisfied by an empty stub module), on a seeded AR(1)
    history of 8 chains x 3 dims: autocovariance at a few lags, the acl spectrum and the ESS."""
    import types
    for nm in ('tensorflow.examples', 'tensorflow.examples.tutorials', 'tensorflow.examples.tutorials.mnist',
               'tensorflow.examples.tutorials.mnist.input_data'):
        sys.modules.setdefault(nm, types.ModuleType(nm))
    sys.modules['tensorflow.examples.tutorials.mnist'].input_data = sys.modules['tensorflow.examples.tutorials.mnist.input_data']
    import func_utils as ref_func_utils         # /root/reference/utils/func_utils.py
    rng = np.random.RandomState(61)
    Tm, N, d, rho = 80, 8, 3, 0.8
    X = np.zeros((Tm, N, d))
    X[0] = rng.randn(N, d)
    for t in range(1, Tm):
        X[t] = rho * X[t - 1] + np.sqrt(1 - rho ** 2) * rng.randn(N, d)
    X = X.astype(np.float32)
    scale = np.float64(np.sqrt(d))
    out = {'X': X, 'scale': scale,
           'taus': np.array([0, 1, 5, 40], dtype=np.int32),
           'autocov': np.array([ref_func_utils.autocovariance(X, tau=t) for t in (0, 1, 5, 40)]),
           'acl': ref_func_utils.acl_spectrum(X, scale)}
    out['ess'] = np.float64(ref_func_utils.ESS(out['acl']))
    np.savez_compressed(os.path.join(OUT, 'ess_funcutils.npz'), **out)
    print('ess_funcutils      Tm=%d N=%d d=%d  ESS %.5f' % (Tm, N, d, out['ess']))


def rough_ne_cases():
    """The reference's DEFAULT Rough Well (distributions.py:84-97, `easy=False`: cos(x / eps^2)) at the scale BASELINE config 4's
    second series is benchmarked on: eta = 1e-2, arguments ~ 1e4 x.  Step sizes are the ones bench.py's pilot tunes for these
    widths (profiles/r04_bench_steps20.json: 7.8e-3 / 1.7e-3 / 3.6e-4): the curvature of this target is eta^-3 = 1e6, so a
    leapfrog step is only stable -- and two float32 evaluations of it only comparable -- for eps < 2e-3 sqrt(...) of that order.
    The first 16 chains start inside |x| < 1.29 (every argument below the 8192 pi/2 switch of the kernels' range reduction),
    the others on both sides of it.  `energy.den` = the divisor as the reference forms it: the Python-double product
    eps * eps rounded to float32 ONCE (`x / (self.eps * self.eps)`, :93); `energy.eta64` = the double the caller passed.
    rough8_eta01: eta = 0.1, where float32(0.1)^2 in float32 and float32(0.1 * 0.1) differ in the last bit."""
    for nm, d, eta, eps, N, seed in (('rough2_ne', 2, 1e-2, 3e-4, 64, 71), ('rough50_ne', 50, 1e-2, 3e-4, 64, 72),
                                     ('rough512_ne', 512, 1e-2, 3.63e-4, 32, 73), ('rough8_eta01', 8, 0.1, 0.02, 64, 74)):
        rw = ref_distributions.RoughWell(d, eta, easy=False)
        params = {'energy.kind': 'roughwell', 'energy.eta': np.float32(eta), 'energy.easy': np.int32(0),
                  'energy.den': np.float32(eta * eta), 'energy.eta64': np.float64(eta)}
        rng = np.random.RandomState(seed)
        x0 = rng.randn(N, d)
        x0[:16] = np.clip(0.3 * x0[:16], -1.25, 1.25)
        run_case(nm, d, H=10, T=10, eps=eps, N=N, energy_fn=rw.get_energy_function(), energy_params=params, seed=seed,
                 head_std=0.5, x0=x0.astype(np.float32))


def train_rough_ne_cases():
    """training gradients through the non-easy Rough Well's Hessian (1 - eta^-3 cos(x / eta^2)): eta = 0.05, arguments 400 x,
    curvature 8000 => eps = 0.01.  One case per trainer: d = 2 (one dimension per lane), d = 6 (register-resident, one wave;
    also the general kernel), d = 50 (register-resident, four waves), d = 6 with 20-wide nets (GEMM engine)."""
    for nm, d, H, N, seed in (('train_rough2_ne', 2, 10, 32, 81), ('train_rough6_ne', 6, 10, 32, 82),
                              ('train_rough50_ne', 50, 10, 16, 83), ('train_rough6_ne_h20', 6, 20, 32, 84)):
        eta = 0.05
        with contextlib.redirect_stdout(io.StringIO()):
            rw_t = ref_distributions.RoughWell(d, eta, easy=False)
        train_case(nm, np.zeros(d), None, H=H, T=5, eps=0.01, N=N, seed=seed, head_std=0.3, dist=rw_t,
                   params={'energy.kind': 'roughwell', 'energy.eta': np.float32(eta), 'energy.easy': np.int32(0),
                           'energy.den': np.float32(eta * eta), 'energy.eta64': np.float64(eta)},
                   x_start=lambda rng, N=N, d=d: rng.randn(N, d))


def main():
    if sys.argv[1:] == ['ess']:
        return ess_case()
    if sys.argv[1:] == ['rough_ne']:             # only the non-easy Rough-Well fixtures
        rough_ne_cases()
        return train_rough_ne_cases()
    if sys.argv[1:] == ['train_funnel']:         # only this fixture
        return train_funnel_case()
    if sys.argv[1:] == ['ais']:                  # only the AIS fixtures (leaves the other files untouched)
        return ais_cases()
    if sys.argv[1:] == ['wide']:                 # only the wide-net fixtures
        return wide_cases()
    if sys.argv[1:] == ['train_vae']:            # only the VAE sampler-training fixture
        return train_vae_case('train_vae_small', latent=10, H=24, dec_h=48, n_pix=40, enc_h=32, T=4, eps=0.1, N=32, seed=43)
    if sys.argv[1:] == ['train_vae_extra']:
        return train_vae_extra_cases()
    if sys.argv[1:] == ['train_wide']:
        return train_wide_cases()
    if sys.argv[1:] == ['train_wide2']:
        return train_wide_cases(only_new=True)
    # C1: Strongly-correlated Gaussian 2D, exactly the notebook's target (nb:103-108)
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    gaussian_case('scg2d', np.zeros(2), cov, H=10, T=10, eps=0.1, N=200, seed=11, x_scale=1.0, chainop=3)
    gaussian_case('scg2d_hmc', np.zeros(2), cov, H=10, T=10, eps=0.1, N=200, seed=12, hmc=True, chainop=2)

    # C2 (subset of chains): ill-conditioned Gaussian d=50, variances log-spaced 1e-2..1e2
    var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), 50))
    rng = np.random.RandomState(5)
    x0 = (rng.randn(64, 50) * np.sqrt(var)).astype(np.float32)     # start in the typical set
    gaussian_case('icg50', np.zeros(50), np.diag(var), H=10, T=10, eps=0.1, N=64, seed=13,
                  head_std=0.05, x0=x0)
    gaussian_case('icg50_hmc', np.zeros(50), np.diag(var), H=10, T=10, eps=0.05, N=64, seed=14,
                  hmc=True, x0=x0)

    # dense (rotated) Gaussian d=8 with non-zero mean: exercises the dense-precision path
    rng = np.random.RandomState(7)
    R = np.linalg.qr(rng.randn(8, 8))[0]
    cov8 = R.T.dot(np.diag(np.exp(np.log(10.) * rng.uniform(-1, 1, size=8)))).dot(R)
    rng8_mu = rng.randn(8) * 0.5
    gaussian_case('tilted8', rng8_mu, cov8, H=10, T=7, eps=0.1, N=48, seed=15, chainop=3)
    # the same target tempered: use_temperature=True with the placeholder fed 2.5 (dynamics.py:47,203-212)
    gaussian_case('tilted8_temp', rng8_mu, cov8, H=10, T=7, eps=0.1, N=48, seed=25, temperature=2.5)

    # C3 shape: 2-component MoG in 2D (paper-style: centres (+-2,0), var 0.1), T=25
    mus = [np.array([2.0, 0.0], dtype=np.float32), np.array([-2.0, 0.0], dtype=np.float32)]
    sig = [0.1 * np.eye(2), 0.1 * np.eye(2)]
    # (mus handed over as float32 torch tensors: TF would convert the numpy constants to
    #  float32 tensors at `x - mu`; torch cannot subtract a numpy array from a grad tensor)
    gmm = ref_distributions.GMM([torch.tensor(m) for m in mus], sig, [0.5, 0.5])
    params = {'energy.kind': 'gmm', 'energy.mus': np.stack(mus),
              'energy.i_sigmas': np.stack(gmm.i_sigmas),
              'energy.constants': np.array(gmm.constants, dtype=np.float32)}
    rng = np.random.RandomState(9)
    x0 = (np.stack(mus)[rng.randint(0, 2, size=96)] + np.sqrt(0.1) * rng.randn(96, 2)).astype(np.float32)
    run_case('mog2d', 2, H=10, T=25, eps=0.1, N=96, energy_fn=gmm.get_energy_function(),
             energy_params=params, seed=16, x0=x0, head_std=0.5)

    # ring of 4 (gen_ring, distributions.py:201-213), unequal-looking pis path
    ring = ref_distributions.gen_ring(r=2.0, var=0.3, nb_mixtures=4)
    ring_mus = [m.astype(np.float32) for m in ring.mus]
    ring.mus = [torch.tensor(m) for m in ring_mus]
    params = {'energy.kind': 'gmm', 'energy.mus': np.stack(ring_mus),
              'energy.i_sigmas': np.stack(ring.i_sigmas),
              'energy.constants': np.array(ring.constants, dtype=np.float32)}
    run_case('ring4', 2, H=10, T=10, eps=0.15, N=64, energy_fn=ring.get_energy_function(),
             energy_params=params, seed=17, x_scale=2.0, head_std=0.5)

    # C4 shape: Rough Well (easy: cos(x/eta)) d=8 and d=50, and the non-easy form
    for nm, d, eta, easy, N in (('rough8_easy', 8, 0.1, True, 64), ('rough50_easy', 50, 0.1, True, 32),
                                ('rough8', 8, 0.5, False, 64)):
        rw = ref_distributions.RoughWell(d, eta, easy=easy)
        params = {'energy.kind': 'roughwell', 'energy.eta': np.float32(eta),
                  'energy.easy': np.int32(easy)}
        run_case(nm, d, H=10, T=10, eps=0.1, N=N, energy_fn=rw.get_energy_function(),
                 energy_params=params, seed=18 + d, head_std=0.5)

    # Gaussian funnel d=3 incl. chains beyond both clip thresholds (|v| > 8)
    with contextlib.redirect_stdout(io.StringIO()):
        fn = ref_distributions.GaussianFunnel(dim=3).get_energy_function()
    rng = np.random.RandomState(21)
    x0 = rng.randn(64, 3).astype(np.float32)
    x0[:, 0] *= 2.0
    x0[:4, 0] = [9.0, -9.0, 8.5, -8.25]
    params = {'energy.kind': 'funnel', 'energy.sigma': np.float32(2.0)}
    run_case('funnel3', 3, H=10, T=5, eps=0.05, N=64, energy_fn=fn, energy_params=params,
             seed=22, x0=x0, head_std=0.3)

    # config-5 shape (VAE latent posterior + aux-conditioned wide nets) at fixture-friendly sizes
    vae_case('vae_small', latent=10, H=24, dec_h=48, n_pix=40, enc_h=32, T=5, eps=0.1, N=32, seed=41)

    # training-loss gradients (next-row f1): SCG (notebook), dense d=8, diagonal ICG d=50
    train_case('train_scg2d', np.zeros(2), cov, H=10, T=10, eps=0.1, N=64, seed=31)
    train_case('train_tilted8', rng8_mu, cov8, H=10, T=5, eps=0.1, N=32, seed=32)
    train_case('train_icg50', np.zeros(50), np.diag(var), H=10, T=4, eps=0.05, N=16, seed=33, head_std=0.05)

    # ... and for the non-Gaussian smooth targets (Hessian-vector products of GMM / Rough Well)
    mus_t = [np.array([2.0, 0.0], dtype=np.float32), np.array([-2.0, 0.0], dtype=np.float32)]
    gmm_t = ref_distributions.GMM([torch.tensor(m) for m in mus_t], [0.5 * np.eye(2), 0.5 * np.eye(2)], [0.5, 0.5])
    train_case('train_mog2d', np.zeros(2), None, H=10, T=6, eps=0.1, N=48, seed=34, head_std=0.3, dist=gmm_t,
               params={'energy.kind': 'gmm', 'energy.mus': np.stack(mus_t), 'energy.i_sigmas': np.stack(gmm_t.i_sigmas),
                       'energy.constants': np.array(gmm_t.constants, dtype=np.float32)},
               x_start=lambda r: np.stack(mus_t)[r.randint(0, 2, size=48)] + np.sqrt(0.5) * r.randn(48, 2))
    rw_t = ref_distributions.RoughWell(6, 0.3, easy=True)
    train_case('train_rough6', np.zeros(6), None, H=10, T=5, eps=0.1, N=32, seed=35, head_std=0.3, dist=rw_t,
               params={'energy.kind': 'roughwell', 'energy.eta': np.float32(0.3), 'energy.easy': np.int32(1)},
               x_start=lambda r: r.randn(32, 6))

    # p_accept edge cases (dynamics.py:302-309): +-inf / NaN Hamiltonian differences -> 0
    tf1_stub.reset(0)
    np.random.seed(0)
    tf1_stub.VARIABLE_HOOK = None
    with contextlib.redirect_stdout(io.StringIO()):
        dist = ref_distributions.Gaussian(np.zeros(2), np.eye(2))
        dyn = ref_dynamics.Dynamics(2, dist.get_energy_function(), T=2, eps=0.1, hmc=True)
    x0 = np.array([[0, 0], [1, 1], [0, 0], [1e20, 0], [0, 0], [0.5, -0.5]], dtype=np.float32)
    v0 = np.array([[0, 0], [1, 0], [1e20, 0], [0, 0], [0, 0], [0.1, 0.2]], dtype=np.float32)
    x1 = np.array([[1, 1], [0, 0], [1e20, 0], [1e20, 0], [np.nan, 0], [0.25, 0.1]], dtype=np.float32)
    v1 = np.array([[0, 1], [0, 0], [0, 0], [1e20, 1], [0, 0], [0.3, -0.2]], dtype=np.float32)
    lj = np.array([0.1, 0.2, 0.0, 0.0, 0.0, np.inf], dtype=np.float32)
    p = dyn.p_accept(leaf(x0), leaf(v0), leaf(x1), leaf(v1), torch.tensor(lj))
    np.savez_compressed(os.path.join(OUT, 'p_accept_edge.npz'), x0=x0, v0=v0, x1=x1, v1=v1,
                        logjac=lj, p=npy(p))
    print('p_accept_edge      p =', npy(p))
    ess_case()
    # the groups that can also be regenerated on their own (see the modes at the top of main)
    ais_cases()
    wide_cases()
    train_funnel_case()
    train_vae_case('train_vae_small', latent=10, H=24, dec_h=48, n_pix=40, enc_h=32, T=4, eps=0.1, N=32, seed=43)
    train_vae_extra_cases()
    train_wide_cases()
    rough_ne_cases()
    train_rough_ne_cases()


if __name__ == '__main__':
    main()
