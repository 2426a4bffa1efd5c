"""CPU oracle for the L2HMC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch numpy restatement of the algorithm of brain-research/l2hmc's
``utils/dynamics.py`` + ``utils/sampler.py`` (+ the S/T/Q net of SCGExperiment.ipynb and
the energies of ``utils/distributions.py``), one numpy op per TF op, in the reference's op
order, with ALL randomness injected (momenta, direction bits, accept uniforms, masks,
weights are explicit inputs).  Each function cites the reference file:line it follows
(paths relative to /root/reference).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module -- as the checker / the timed CPU baseline, never as a product path.

Pinning status: the reference has NO tests / golden vectors of its own and
TensorFlow 1.x cannot be installed here, so this oracle is pinned against the reference's
own Python modules executed under ``oracle/tf1_stub.py`` (torch-CPU fp32 stand-in for the
TF1 ops): see ``oracle/make_goldens.py`` -> ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py``.  Parity at the TF1-runtime boundary itself (Eigen kernels,
TF's Philox streams) is therefore "unpinned"; everything above that boundary is pinned.

``dtype=np.float32`` reproduces the reference arithmetic; ``dtype=np.float64`` gives the
"truth" used to show that the HIP path and the fp32 oracle bracket it.
"""
import numpy as np

# --------------------------------------------------------------------------------------------
# S/T/Q network  (SCGExperiment.ipynb `network`, raw-json lines 51-78; utils/layers.py:29-95)
# --------------------------------------------------------------------------------------------
NET_KEYS = ('W1', 'b1', 'W2', 'b2', 'W3', 'b3', 'W4', 'b4',
            'Ws', 'bs', 'Wt', 'bt', 'Wq', 'bq', 'lam_s', 'lam_q')


def net_cast(net, dtype):
    return {k: np.asarray(net[k], dtype=dtype) for k in NET_KEYS}


def net_apply(net, a, b, tau, aux_h=None):
    """[S, T, Q] = net([a, b, tau, aux])  with the notebook architecture (aux_h = None) or the
    VAE sampler's (mnist_vae.py:142-167: the 4th Zip branch `encoder_sampler(aux)` is added into
    the pre-ReLU sum; pass its output as aux_h).

    Zip of three Linear embeds + `lambda _: 0.` (nb:53-60), python `sum` (0 + e1 + e2 + e3
    + 0.), relu, Linear(H,H), relu (nb:61-64), Parallel heads (nb:65-76):
    S = exp(lam_s) * tanh(h Ws + bs)  (layers.py:81-86), T = h Wt + bt, Q likewise.
    Linear is `x @ W + b` with W of shape (in, out) (layers.py:33,37).
    """
    e1 = a @ net['W1'] + net['b1']
    e2 = b @ net['W2'] + net['b2']
    e3 = tau @ net['W3'] + net['b3']
    h = ((0 + e1) + e2) + e3 + (0.0 if aux_h is None else aux_h)
    h = np.maximum(h, 0)
    h = h @ net['W4'] + net['b4']
    h = np.maximum(h, 0)
    S = np.exp(net['lam_s']) * np.tanh(h @ net['Ws'] + net['bs'])
    T = h @ net['Wt'] + net['bt']
    Q = np.exp(net['lam_q']) * np.tanh(h @ net['Wq'] + net['bq'])
    return S, T, Q


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def mlp3(w, x):
    """Linear-softplus-Linear-softplus-Linear (mnist_vae.py decoder :104-111, encoder_sampler
    :134-140); w = dict(W1, b1, W2, b2, W3, b3), Linear = x W + b."""
    h = softplus(x @ w['W1'] + w['b1'])
    h = softplus(h @ w['W2'] + w['b2'])
    return h @ w['W3'] + w['b3']


class VAEPosterior:
    """mnist_vae.py:122-126: U(z; x) = sum_pix BCE_with_logits(x, decoder(z)) + |z|^2 / 2 with the
    decoder of :104-111.  Gradient w.r.t. z analytic (the reference uses tf.gradients)."""

    def __init__(self, dec, aux, dtype=np.float32):
        self.w = {k: np.asarray(v, np.float32).astype(dtype) for k, v in dec.items()}
        self.aux = np.asarray(aux, np.float32).astype(dtype)

    def __call__(self, z):
        w = self.w
        p1 = z @ w['W1'] + w['b1']
        a1 = softplus(p1)
        p2 = a1 @ w['W2'] + w['b2']
        a2 = softplus(p2)
        lg = a2 @ w['W3'] + w['b3']
        bce = np.maximum(lg, 0) - lg * self.aux + np.log1p(np.exp(-np.abs(lg)))
        U = np.sum(bce, axis=1) + 0.5 * np.sum(np.square(z), axis=1)
        dl = sigmoid(lg) - self.aux
        d2 = (dl @ w['W3'].T) * sigmoid(p2)
        d1 = (d2 @ w['W2'].T) * sigmoid(p1)
        return U, d1 @ w['W1'].T + z


def zero_net(a, b, tau):
    """HMC mode: nets return three zero tensors (dynamics.py:73-76)."""
    z = np.zeros_like(a)
    return z, z, z


# --------------------------------------------------------------------------------------------
# Energies: each returns (U(x) of shape (N,), grad U(x) of shape (N,d))
# --------------------------------------------------------------------------------------------
class Gaussian:
    """distributions.py:41-57 + quadratic_gaussian :31-32.

    U = diag(0.5 (x-mu) S (x-mu)^T); S = float32(inv(sigma)) (:48,52).
    `faithful_nxn=True` literally forms the N x N product like the reference (only for
    small N); otherwise the row-wise equivalent.  grad = what tf.gradients gives:
    0.5 ((x-mu) S^T + (x-mu) S).
    """

    def __init__(self, mu, i_sigma, dtype=np.float32, faithful_nxn=False):
        self.mu = np.asarray(mu, dtype=np.float32).astype(dtype)
        self.S = np.asarray(i_sigma, dtype=np.float32).astype(dtype)
        self.faithful_nxn = faithful_nxn

    def __call__(self, x):
        dx = x - self.mu
        y = dx @ self.S
        if self.faithful_nxn:
            U = np.diagonal(0.5 * (y @ dx.T)).copy()
        else:
            U = 0.5 * np.sum(y * dx, axis=1)
        g = 0.5 * (y + dx @ self.S.T)
        return U, g


class GMM:
    """distributions.py:104-134.  U = -logsumexp_i(-q_i + log c_i),
    q_i = quadratic_gaussian(x, mu_i, S_i), c_i = pi_i / sqrt((2 pi)^k det sigma_i) in f32."""

    def __init__(self, mus, i_sigmas, constants, dtype=np.float32):
        self.mus = [np.asarray(m, dtype=np.float32).astype(dtype) for m in mus]
        self.S = [np.asarray(s, dtype=np.float32).astype(dtype) for s in i_sigmas]
        self.logc = [np.log(np.asarray(c, dtype=np.float32)).astype(dtype) for c in constants]

    def __call__(self, x):
        V, ys = [], []
        for mu, S, lc in zip(self.mus, self.S, self.logc):
            dx = x - mu
            y = dx @ S
            V.append(-(0.5 * np.sum(y * dx, axis=1)) + lc)
            ys.append(0.5 * (y + dx @ S.T))
        V = np.stack(V, axis=1)
        m = V.max(axis=1, keepdims=True)
        e = np.exp(V - m)
        lse = np.log(e.sum(axis=1)) + m[:, 0]
        w = e / e.sum(axis=1, keepdims=True)
        g = sum(w[:, i:i + 1] * ys[i] for i in range(len(ys)))
        return -lse, g


class RoughWell:
    """distributions.py:84-97.  U = 0.5|x|^2 + eta sum cos(x/eta^2)   (or x/eta if easy).
    `eta` is the Python double the caller hands the reference; the divisor is formed like there: the DOUBLE product
    eta * eta rounded to the working precision once (`x / (self.eps * self.eps)`, :93 -- TF converts the Python scalar to
    a float32 constant), not float32(eta) squared in float32 (the two differ in the last bit for e.g. eta = 0.1)."""

    def __init__(self, eta, easy=False, dtype=np.float32):
        self.eta = dtype(eta)
        self.den = dtype(float(eta)) if easy else dtype(float(eta) * float(eta))
        self.easy = easy
        self.dtype = dtype

    def __call__(self, x):
        eta, den = self.eta, self.den
        n = np.sum(np.square(x), axis=1)
        arg = x / den
        U = 0.5 * n + eta * np.sum(np.cos(arg), axis=1)
        g = x - (eta * np.sin(arg)) / den          # the graph's gradient: (-sin(arg) * eta) / den, TF's RealDiv gradient
        return U.astype(self.dtype), g.astype(self.dtype)


class GaussianFunnel:
    """distributions.py:155-180 (sigma=2, clip = 4 sigma; `where`-clipped)."""

    def __init__(self, sigma=2.0, dtype=np.float32):
        self.sigma = dtype(sigma)
        self.clip = dtype(4 * sigma)
        self.dtype = dtype

    def __call__(self, x):
        dt = self.dtype
        v = x[:, 0]
        log_p_v = np.square(v / self.sigma)
        s = np.exp(v)
        sum_sq = np.sum(np.square(x[:, 1:]), axis=1)
        n = dt(x.shape[1] - 1)
        two_pi = dt(2.0 * np.pi)
        E = 0.5 * (log_p_v + sum_sq / s + n * np.log(two_pi * s))
        s_min = np.exp(-self.clip)
        s_max = np.exp(self.clip)
        E1 = 0.5 * (log_p_v + sum_sq / s_max + n * np.log(two_pi * s_max))
        E2 = 0.5 * (log_p_v + sum_sq / s_min + n * np.log(two_pi * s_min))
        hi = v > self.clip
        lo = -self.clip > v
        U = np.where(hi, E1, E)
        U = np.where(lo, E2, U)
        # gradient of the selected branch (tf.where routes the gradient)
        s_eff = np.where(hi, s_max, np.where(lo, s_min, s))
        g = np.empty_like(x)
        g[:, 1:] = x[:, 1:] / s_eff[:, None]
        gv_free = v / (self.sigma * self.sigma) + 0.5 * (-sum_sq / s + n)
        gv_clip = v / (self.sigma * self.sigma)
        g[:, 0] = np.where(hi | lo, gv_clip, gv_free)
        return U.astype(dt), g.astype(dt)


# --------------------------------------------------------------------------------------------
# Dynamics  (utils/dynamics.py)
# --------------------------------------------------------------------------------------------
def init_mask(T, x_dim, rng):
    """dynamics.py:84-93: T independent masks with floor(d/2) ones at permutation(d)[:d//2].
    `rng` is a numpy RandomState (the reference uses the global numpy RNG)."""
    rows = []
    for _ in range(T):
        ind = rng.permutation(np.arange(x_dim))[:int(x_dim / 2)]
        m = np.zeros((x_dim,))
        m[ind] = 1
        rows.append(m)
    return np.stack(rows).astype(np.float32)


def format_time(step, T, dtype=np.float32):
    """dynamics.py:99-105: [cos(2 pi t / T), sin(2 pi t / T)], t a float32 scalar."""
    t = dtype(step)
    ang = dtype(2 * np.pi) * t / dtype(T)
    return np.array([np.cos(ang), np.sin(ang)], dtype=dtype)


class Dynamics:
    """Functional restatement of utils/dynamics.py:34-309.

    energy(x) -> (U, gradU); xnet / vnet: dicts of NET_KEYS arrays, None for HMC mode, or -- the reference's own generality,
    dynamics.py:69-79 -- ANY callables net(a, b, tau) -> (S, T, Q) on (N, d) / (N, d) / (N, 2) arrays.
    """

    def __init__(self, x_dim, energy, T, eps, mask, xnet=None, vnet=None,
                 temperature=1.0, dtype=np.float32, aux_h=None):
        self.x_dim, self.T, self.dtype = x_dim, int(T), dtype
        self.eps = dtype(eps)
        self.mask = np.asarray(mask, dtype=dtype)
        self._energy = energy
        self.temperature = dtype(temperature)
        self.hmc = xnet is None
        if self.hmc:
            self.XNet = self.VNet = zero_net
        elif callable(xnet) and callable(vnet):
            self.XNet, self.VNet = xnet, vnet
        else:
            xn, vn = net_cast(xnet, dtype), net_cast(vnet, dtype)
            ah = None if aux_h is None else np.asarray(aux_h, dtype=dtype)
            self.XNet = lambda a, b, t: net_apply(xn, a, b, t, ah)
            self.VNet = lambda a, b, t: net_apply(vn, a, b, t, ah)

    # dynamics.py:203-218
    def energy(self, x):
        return self._energy(x)[0] / self.temperature

    def grad_energy(self, x):
        return self._energy(x)[1] / self.temperature

    def kinetic(self, v):                       # dynamics.py:107-108
        return 0.5 * np.sum(np.square(v), axis=1)

    def hamiltonian(self, x, v):                # dynamics.py:214-215
        return self.energy(x) + self.kinetic(v)

    def _get_mask(self, step):                  # dynamics.py:95-97
        m = self.mask[int(step)]
        return m, 1. - m

    def _tau(self, step, n):                    # dynamics.py:99-105 tiled to (N,2)
        return np.tile(format_time(step, self.T, self.dtype)[None, :], (n, 1))

    def forward_step(self, x, v, step):
        """dynamics.py:115-157."""
        eps = self.eps
        t = self._tau(step, x.shape[0])
        grad1 = self.grad_energy(x)
        S1 = self.VNet(x, grad1, t)
        sv1 = 0.5 * eps * S1[0]
        tv1 = S1[1]
        fv1 = eps * S1[2]
        v_h = v * np.exp(sv1) + 0.5 * eps * (-(np.exp(fv1) * grad1) + tv1)

        m, mb = self._get_mask(step)
        X1 = self.XNet(v_h, m * x, t)
        sx1 = eps * X1[0]
        tx1 = X1[1]
        fx1 = eps * X1[2]
        y = m * x + mb * (x * np.exp(sx1) + eps * (np.exp(fx1) * v_h + tx1))

        X2 = self.XNet(v_h, mb * y, t)
        sx2 = eps * X2[0]
        tx2 = X2[1]
        fx2 = eps * X2[2]
        x_o = mb * y + m * (y * np.exp(sx2) + eps * (np.exp(fx2) * v_h + tx2))

        grad2 = self.grad_energy(x_o)
        S2 = self.VNet(x_o, grad2, t)
        sv2 = 0.5 * eps * S2[0]
        tv2 = S2[1]
        fv2 = eps * S2[2]
        v_o = v_h * np.exp(sv2) + 0.5 * eps * (-(np.exp(fv2) * grad2) + tv2)
        log_jac = np.sum(sv1 + sv2 + mb * sx1 + m * sx2, axis=1)
        return x_o, v_o, log_jac

    def backward_step(self, x_o, v_o, step):
        """dynamics.py:159-201."""
        eps = self.eps
        t = self._tau(step, x_o.shape[0])
        grad1 = self.grad_energy(x_o)
        S1 = self.VNet(x_o, grad1, t)
        sv2 = -0.5 * eps * S1[0]
        tv2 = S1[1]
        fv2 = eps * S1[2]
        v_h = (v_o - 0.5 * eps * (-(np.exp(fv2) * grad1) + tv2)) * np.exp(sv2)

        m, mb = self._get_mask(step)
        X1 = self.XNet(v_h, mb * x_o, t)
        sx2 = -eps * X1[0]
        tx2 = X1[1]
        fx2 = eps * X1[2]
        y = mb * x_o + m * (np.exp(sx2) * (x_o - eps * (np.exp(fx2) * v_h + tx2)))

        X2 = self.XNet(v_h, m * y, t)
        sx1 = -eps * X2[0]
        tx1 = X2[1]
        fx1 = eps * X2[2]
        x = m * y + mb * (np.exp(sx1) * (y - eps * (np.exp(fx1) * v_h + tx1)))

        grad2 = self.grad_energy(x)
        S2 = self.VNet(x, grad2, t)
        sv1 = -0.5 * eps * S2[0]
        tv1 = S2[1]
        fv1 = eps * S2[2]
        v = np.exp(sv1) * (v_h - 0.5 * eps * (-(np.exp(fv1) * grad2) + tv1))
        return x, v, np.sum(sv1 + sv2 + mb * sx1 + m * sx2, axis=1)

    def p_accept(self, x0, v0, x1, v1, log_jac):
        """dynamics.py:302-309: exp(min(H0 - H1 + logjac, 0)); non-finite -> 0."""
        e_new = self.hamiltonian(x1, v1)
        e_old = self.hamiltonian(x0, v0)
        with np.errstate(all='ignore'):
            val = e_old - e_new + log_jac
            p = np.exp(np.minimum(val, 0.0))
        return np.where(np.isfinite(p), p, np.zeros_like(p)).astype(self.dtype)

    def forward(self, x, init_v, log_jac=False):
        """dynamics.py:246-272 (init_v is mandatory here: randomness is injected)."""
        X, V = x, init_v
        j = np.zeros((x.shape[0],), dtype=self.dtype)
        t = self.dtype(0.)
        while t < self.T:
            X, V, lj = self.forward_step(X, V, t)
            t, j = t + 1, j + lj
        if log_jac:
            return X, V, j
        return X, V, self.p_accept(x, init_v, X, V, j)

    def backward(self, x, init_v, log_jac=False):
        """dynamics.py:274-300: steps T-1 ... 0."""
        X, V = x, init_v
        j = np.zeros((x.shape[0],), dtype=self.dtype)
        t = self.dtype(0.)
        while t < self.T:
            X, V, lj = self.backward_step(X, V, self.T - t - 1)
            t, j = t + 1, j + lj
        if log_jac:
            return X, V, j
        return X, V, self.p_accept(x, init_v, X, V, j)


# --------------------------------------------------------------------------------------------
# AIS  (utils/ais.py:30-82, Wu et al. 2016)
# --------------------------------------------------------------------------------------------
def ais_estimate(init_energy, final_energy, anneal_steps, initial_x, v0, normals, u, step_size=0.5,
                 leapfrogs=25, num_splits=1, refresh=False, refreshment=0.1, dtype=np.float32):
    """ais.py:30-82 with the randomness injected: v0 (N,d) is the scan's initial momentum (:72),
    normals (K,N,d) the per-step momentum draws (:55,57), u (K,N) the accept uniforms (:62).
    init_energy / final_energy: callables x -> (U, grad U).  Returns (estimate, mean alpha, state)."""
    K = int(anneal_steps)
    x = np.asarray(initial_x, dtype)
    N, d = x.shape
    beta = np.linspace(0.0, 1.0, K + 1, dtype=np.float32)[1:].astype(dtype)      # :43
    dbeta = beta[1] - beta[0]                                                    # :44
    w = np.zeros(N, dtype)
    v = np.asarray(v0, dtype)
    alphas = []
    mask = np.zeros((leapfrogs, d), dtype)        # HMC mode: the nets are zero, the mask drops out
    for i in range(K):
        b = beta[i]

        def curr(z, b=b):                                                        # :46-47
            U0, g0 = init_energy(z)
            U1, g1 = final_energy(z)
            return (1 - b) * U0 + b * U1, (1 - b) * g0 + b * g1
        z = np.asarray(normals[i], dtype)
        rv = v * np.sqrt(dtype(1 - refreshment)) + z * np.sqrt(dtype(refreshment)) if refresh else z   # :54-57
        w = w + dbeta * (-final_energy(x)[0] + init_energy(x)[0])                # :58-59
        dyn = Dynamics(d, curr, leapfrogs, step_size, mask, dtype=dtype)         # :60 (hmc=True)
        Lx, Lv, px = dyn.forward(x, rv)                                          # :61
        acc = (px - np.asarray(u[i], dtype)) >= 0                                # :63
        x = np.where(acc[:, None], Lx, x)
        v = np.where(acc[:, None], Lv, -Lv)                                      # :65 (sic)
        alphas.append(px)

    def logmeanexp(t):
        m = t.max()
        return m + np.log(np.sum(np.exp(t - m))) - np.log(dtype(t.shape[0]))
    est = sum(logmeanexp(t) for t in np.split(w, int(num_splits)))
    return est, float(np.mean(alphas)), {'x': x, 'w': w, 'alpha': np.sum(alphas, axis=0)}


# --------------------------------------------------------------------------------------------
# Sampler  (utils/sampler.py)
# --------------------------------------------------------------------------------------------
def tf_accept(x, Lx, px, u):
    """sampler.py:53-55: rows with px - u >= 0 take the proposal."""
    mask = (px - u >= 0.)
    return np.where(mask[:, None], Lx, x)


def propose(x, dyn, v_fwd, v_bwd=None, direction=None, u=None, log_jac=False,
            both_directions=True):
    """sampler.py:28-51 with injected randomness.

    HMC mode (:29-31): forward only with `v_fwd`.  L2HMC mode (:33-51): `direction` is the
    (N,) 0/1 draw of :34 (1 = forward); forward uses v_fwd, backward uses v_bwd (each
    direction draws its own momenta, dynamics.py:247-250,275-278).  `both_directions=True`
    evaluates both trajectories for ALL chains and mixes them like the reference (:35-44);
    False evaluates each chain only in its drawn direction (what the HIP path does) --
    identical results wherever the discarded direction is finite.
    Returns Lx, Lv, px, x_next (x_next None if u is None).
    """
    if dyn.hmc:                              # :29-31 -- log_jac is not forwarded there
        Lx, Lv, px = dyn.forward(x, v_fwd)
        return Lx, Lv, px, (tf_accept(x, Lx, px, u) if u is not None else None)
    mask = np.asarray(direction).astype(dyn.dtype)[:, None]
    if both_directions:
        Lx1, Lv1, px1 = dyn.forward(x, v_fwd, log_jac=log_jac)
        Lx2, Lv2, px2 = dyn.backward(x, v_bwd, log_jac=log_jac)
        Lx = mask * Lx1 + (1 - mask) * Lx2
        Lv = mask * Lv1 + (1 - mask) * Lv2
        px = mask[:, 0] * px1 + (1 - mask[:, 0]) * px2
    else:
        # each chain keeps only its drawn direction: a SELECT instead of the reference's 0/1
        # weighted sum, so a non-finite discarded trajectory cannot leak (0 * inf = NaN, :38)
        f = mask > 0
        with np.errstate(all='ignore'):
            Lx1, Lv1, px1 = dyn.forward(x, v_fwd, log_jac=log_jac)
            Lx2, Lv2, px2 = dyn.backward(x, v_bwd, log_jac=log_jac)
        Lx, Lv = np.where(f, Lx1, Lx2), np.where(f, Lv1, Lv2)
        px = np.where(f[:, 0], px1, px2)
    return Lx, Lv, px, (tf_accept(x, Lx, px, u) if u is not None else None)


def chain_operator(init_x, dyn, nb_steps, init_v, v_fwd_list, v_bwd_list, directions, u=None):
    """sampler.py:57-85.  Quirk kept: in L2HMC mode propose() ignores the
    momentum handed to it -- every composed proposal draws fresh momenta (v_*_list[k]);
    only the returned Lv is threaded on, and p_accept uses the caller's init_v.  In HMC mode
    (:29-31) propose DOES start from the threaded momentum and ignores `log_jac`: the summed
    "log-Jacobian" is then a sum of accept probabilities (the lists may be None there).
    Pinned by the `chainop.*` keys of the goldens (the reference's own chain_operator)."""
    x, v = init_x, init_v
    lj = np.zeros((init_x.shape[0],), dtype=dyn.dtype)
    for k in range(int(nb_steps)):
        if dyn.hmc:
            x, v, px, _ = propose(x, dyn, v, log_jac=True)
        else:
            x, v, px, _ = propose(x, dyn, v_fwd_list[k], v_bwd_list[k], directions[k], log_jac=True)
        lj = lj + px
    p = dyn.p_accept(init_x, init_v, x, v, lj)
    return x, v, p, (tf_accept(init_x, x, p, u) if u is not None else None)


# --------------------------------------------------------------------------------------------
# Diagnostics  (utils/func_utils.py:45-54,114-120)
# --------------------------------------------------------------------------------------------
def autocovariance(X, tau=0):
    dT, dN, dX = np.shape(X)
    s = 0.
    for t in range(dT - tau):
        s += np.sum(X[t] * X[t + tau]) / dN
    return s / (dT - tau)


def acl_spectrum(X, scale):
    n = X.shape[0]
    return np.array([autocovariance(X / scale, tau=t) for t in range(n - 1)])


def ESS(A):
    A = A * (A > 0.05)
    return 1. / (1. + 2 * np.sum(A[1:]))


# --------------------------------------------------------------------------------------------
# Counter-based randomness of the sampler loop (not in the reference, which uses TF's streams:
# dynamics.py:247-250, sampler.py:34,54).  Philox4x32-10 as published (Salmon, Moraes, Dror,
# Shaw, SC'11; constants as in Random123).  Known-answer vectors: tests/test_oracle_golden.py.
# --------------------------------------------------------------------------------------------
def philox4x32_10(c, k0, k1):
    """c: (..., 4) uint32 counters; k0, k1: uint32 key words -> (..., 4) uint32."""
    c = np.asarray(c, dtype=np.uint64).copy()
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M0, M1, m32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[..., 0], M1 * c[..., 2]
        n = np.stack([(p1 >> np.uint64(32)) ^ c[..., 1] ^ k0, p1 & m32,
                      (p0 >> np.uint64(32)) ^ c[..., 3] ^ k1, p0 & m32], axis=-1)
        c = n & m32
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c.astype(np.uint32)


def philox_draws(seed, n_chains, d, nb_proposals, proposal0=0, chain_offset=0):
    """(v (M,N,d) f32, direction (M,N) u8, u (M,N) f32) of the in-kernel stream: counter =
    (global chain, dim // 4, proposal lo, (proposal hi << 1) | stream); 24-bit uniforms,
    Box-Muller pairs (float32 arithmetic)."""
    M, N = int(nb_proposals), int(n_chains)
    nblk = (d + 3) // 4
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    prop = proposal0 + np.arange(M, dtype=np.uint64)
    gch = (chain_offset + np.arange(N, dtype=np.int64)).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    plo, phi = prop & np.uint64(0xFFFFFFFF), (prop >> np.uint64(32)) << np.uint64(1)
    c = np.zeros((M, N, nblk, 4), dtype=np.uint64)
    c[..., 0] = gch[None, :, None]
    c[..., 1] = np.arange(nblk, dtype=np.uint64)[None, None, :]
    c[..., 2] = plo[:, None, None]
    c[..., 3] = phi[:, None, None]
    r = philox4x32_10(c, k0, k1)
    k = np.float32(2.0 ** -24)
    u1 = ((r[..., 0] >> 8) + 1).astype(np.float32) * k
    u2 = (r[..., 1] >> 8).astype(np.float32) * k
    u3 = ((r[..., 2] >> 8) + 1).astype(np.float32) * k
    u4 = (r[..., 3] >> 8).astype(np.float32) * k
    two_pi = np.float32(6.283185307179586)
    ra, rb = np.sqrt(np.float32(-2.0) * np.log(u1)), np.sqrt(np.float32(-2.0) * np.log(u3))
    z = np.stack([ra * np.cos(two_pi * u2), ra * np.sin(two_pi * u2),
                  rb * np.cos(two_pi * u4), rb * np.sin(two_pi * u4)], axis=-1).astype(np.float32)
    v = z.reshape(M, N, nblk * 4)[:, :, :d].copy()
    c1 = np.zeros((M, N, 4), dtype=np.uint64)
    c1[..., 0] = gch[None, :]
    c1[..., 2] = plo[:, None]
    c1[..., 3] = phi[:, None] | np.uint64(1)
    r1 = philox4x32_10(c1, k0, k1)
    direction = (r1[..., 0] & 1).astype(np.uint8)
    u = (r1[..., 1] >> 8).astype(np.float32) * k
    return v, direction, u
