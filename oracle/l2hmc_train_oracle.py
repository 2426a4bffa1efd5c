"""CPU oracle for the L2HMC training gradient -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Hand-derived reverse mode of the notebook's training loss (SCGExperiment.ipynb raw lines
156-169: two direction-mixed proposals, sampler.py:28-51, ESJD-style loss) through the
generalised leapfrog trajectory (dynamics.py:115-201, 246-309), including the
Hessian-vector path through `grad_energy` (TF1 differentiates through `tf.gradients`).
Targets with analytic Hessian-vector products: Gaussian, GMM, Rough Well.  The HIP training kernel follows exactly this
derivation; this file is pinned against gradients produced by the reference's own graph under
oracle/tf1_stub.py (tests/golden/train_*.npz, tests/test_oracle_golden.py).

Per chain the trajectory runs only in its drawn direction (the reference's other direction
carries a zero mixing weight, hence a zero gradient).
"""
import numpy as np

from oracle.l2hmc_oracle import NET_KEYS, format_time


def _net_fwd(net, a, b, tau):
    h1p = a @ net['W1'] + net['b1'] + b @ net['W2'] + net['b2'] + tau @ net['W3'] + net['b3']
    h1 = np.maximum(h1p, 0)
    h2p = h1 @ net['W4'] + net['b4']
    h2 = np.maximum(h2p, 0)
    ts, tq = np.tanh(h2 @ net['Ws'] + net['bs']), np.tanh(h2 @ net['Wq'] + net['bq'])
    es, eq = np.exp(net['lam_s']), np.exp(net['lam_q'])
    S, T, Q = es * ts, h2 @ net['Wt'] + net['bt'], eq * tq
    return (S, T, Q), dict(a=a, b=b, tau=tau, h1=h1, h2=h2, ts=ts, tq=tq, es=es, eq=eq, S=S, Q=Q)


def _net_bwd(net, c, dS, dT, dQ, grad):
    """Accumulates parameter gradients into `grad` (dict like NET_KEYS); returns (d a, d b)."""
    dzs = dS * c['es'] * (1 - c['ts'] ** 2)
    dzq = dQ * c['eq'] * (1 - c['tq'] ** 2)
    dzt = dT
    grad['lam_s'] += np.sum(dS * c['S'], axis=0, keepdims=True)
    grad['lam_q'] += np.sum(dQ * c['Q'], axis=0, keepdims=True)
    for W, b, dz in (('Ws', 'bs', dzs), ('Wt', 'bt', dzt), ('Wq', 'bq', dzq)):
        grad[W] += c['h2'].T @ dz
        grad[b] += dz.sum(0)
    dh2 = dzs @ net['Ws'].T + dzt @ net['Wt'].T + dzq @ net['Wq'].T
    da2 = dh2 * (c['h2'] > 0)
    grad['W4'] += c['h1'].T @ da2
    grad['b4'] += da2.sum(0)
    da1 = (da2 @ net['W4'].T) * (c['h1'] > 0)
    grad['W1'] += c['a'].T @ da1
    grad['W2'] += c['b'].T @ da1
    grad['W3'] += c['tau'].T @ da1
    for b in ('b1', 'b2', 'b3'):
        grad[b] += da1.sum(0)
    return da1 @ net['W1'].T, da1 @ net['W2'].T


# A net OUTSIDE the notebook's architecture (the reference's `net_factory` may return any callable, dynamics.py:69-79): an
# object with  fwd(a, b, tau) -> ((S, T, Q), cache)  and  bwd(cache, dS, dT, dQ) -> (da, db)  that accumulates its own
# parameter gradients in `.grads` (a dict).  `TanhSigmoidNet` below is the one the tests use; dict nets are the notebook's.
def _fwd(net, a, b, tau):
    return net.fwd(a, b, tau) if hasattr(net, 'fwd') else _net_fwd(net, a, b, tau)


def _bwd(net, c, dS, dT, dQ, grad):
    return net.bwd(c, dS, dT, dQ) if hasattr(net, 'bwd') else _net_bwd(net, c, dS, dT, dQ, grad)


class TanhSigmoidNet:
    """h = tanh(a A + b B + c) * (1 + tau C);  S = s (sigmoid(h Ws) - 1/2);  T = h Wt + bt;  Q = q tanh(h Wq)  -- one hidden
    layer, the time input entering multiplicatively, S bounded by a sigmoid: nothing the fused kernels have.  Hand-derived
    reverse mode, pinned by finite differences (tests/test_oracle_golden.py)."""
    KEYS = ('A', 'B', 'c', 'C', 'Ws', 'Wt', 'bt', 'Wq')

    def __init__(self, w, s, q, dtype=np.float64):
        self.w = {k: np.asarray(w[k], dtype) for k in self.KEYS}
        self.s, self.q = dtype(s), dtype(q)
        self.grads = {k: np.zeros_like(self.w[k]) for k in self.KEYS}

    def fwd(self, a, b, tau):
        w = self.w
        t = np.tanh(a @ w['A'] + b @ w['B'] + w['c'])
        m = 1.0 + tau @ w['C']
        h = t * m
        sg = 1.0 / (1.0 + np.exp(-(h @ w['Ws'])))
        tq = np.tanh(h @ w['Wq'])
        return (self.s * (sg - 0.5), h @ w['Wt'] + w['bt'], self.q * tq), dict(a=a, b=b, tau=tau, t=t, m=m, h=h, sg=sg, tq=tq)

    def bwd(self, c, dS, dT, dQ):
        w, g = self.w, self.grads
        dzs = dS * self.s * c['sg'] * (1.0 - c['sg'])
        dzq = dQ * self.q * (1.0 - c['tq'] ** 2)
        g['Ws'] += c['h'].T @ dzs
        g['Wt'] += c['h'].T @ dT
        g['bt'] += dT.sum(0)
        g['Wq'] += c['h'].T @ dzq
        dh = dzs @ w['Ws'].T + dT @ w['Wt'].T + dzq @ w['Wq'].T
        g['C'] += c['tau'].T @ (dh * c['t'])
        dpre = dh * c['m'] * (1.0 - c['t'] ** 2)
        g['A'] += c['a'].T @ dpre
        g['B'] += c['b'].T @ dpre
        g['c'] += dpre.sum(0)
        return dpre @ w['A'].T, dpre @ w['B'].T


def _v_half(vin, g, S, T, Q, eps, sgn, fwd):
    ES, EQ = np.exp(sgn * 0.5 * eps * S), np.exp(eps * Q)
    cc = 0.5 * eps * (T - EQ * g)
    return np.where(fwd, vin * ES + cc, (vin - cc) * ES), (ES, EQ, cc)


def _v_half_bwd(dout, lam_ld, vin, g, S, T, Q, eps, sgn, fwd, aux):
    ES, EQ, cc = aux
    dvin = dout * ES
    dES = np.where(fwd, dout * vin, dout * (vin - cc))
    dcc = np.where(fwd, dout, -dout * ES)
    ds = dES * ES + lam_ld
    dS = ds * sgn * 0.5 * eps
    deps = np.sum(ds * sgn * 0.5 * S, axis=1)
    dT = dcc * 0.5 * eps
    dEQ = -dcc * 0.5 * eps * g
    dg = -dcc * 0.5 * eps * EQ
    deps += np.sum(dcc * 0.5 * (T - EQ * g), axis=1)
    dq = dEQ * EQ
    deps += np.sum(dq * Q, axis=1)
    return dvin, dg, dS, dT, dq * eps, deps


def _x_half(zin, kp, vh, S, T, Q, eps, sgn, fwd):
    ES, EQ = np.exp(sgn * eps * S), np.exp(eps * Q)
    tr = eps * (EQ * vh + T)
    nw = np.where(fwd, zin * ES + tr, ES * (zin - tr))
    return kp * zin + (1 - kp) * nw, (ES, EQ, tr)


def _x_half_bwd(dout, lam_ld, zin, kp, vh, S, T, Q, eps, sgn, fwd, aux):
    ES, EQ, tr = aux
    up = 1 - kp
    dnw = up * dout
    dzin = kp * dout + dnw * ES
    dES = np.where(fwd, dnw * zin, dnw * (zin - tr))
    dtr = np.where(fwd, dnw, -dnw * ES)
    dsx = dES * ES + up * lam_ld
    dS = dsx * sgn * eps
    deps = np.sum(dsx * sgn * S, axis=1)
    dEQ = dtr * eps * vh
    dvh = dtr * eps * EQ
    dT = dtr * eps
    deps += np.sum(dtr * (EQ * vh + T), axis=1)
    dq = dEQ * EQ
    deps += np.sum(dq * Q, axis=1)
    return dzin, dvh, dS, dT, dq * eps, deps


class GaussianTarget:
    """U, grad U and Hessian-vector products of the reference's Gaussian (distributions.py:41-57)."""

    def __init__(self, mu, i_sigma, dtype):
        self.mu = np.asarray(mu, np.float32).astype(dtype)
        self.S = np.asarray(i_sigma, np.float32).astype(dtype)
        self.G = 0.5 * (self.S + self.S.T)

    def energy(self, x):
        dx = x - self.mu
        return 0.5 * np.sum((dx @ self.S) * dx, axis=1)

    def grad(self, x):
        return (x - self.mu) @ self.G

    def hessvec(self, x, v):
        return v @ self.G


class GMMTarget:
    """distributions.py:104-134.  grad = sum_i w_i y_i, y_i = G_i (x - mu_i), w = softmax(V);
    H v = sum_i w_i G_i v - sum_i w_i y_i (y_i . v) + g (g . v)."""

    def __init__(self, mus, i_sigmas, constants, dtype):
        self.mus = [np.asarray(m, np.float32).astype(dtype) for m in mus]
        self.S = [np.asarray(s_, np.float32).astype(dtype) for s_ in i_sigmas]
        self.G = [0.5 * (s_ + s_.T) for s_ in self.S]
        self.logc = [np.log(np.asarray(c, np.float32)).astype(dtype) for c in constants]

    def _parts(self, x):
        ys = [(x - m) @ G for m, G in zip(self.mus, self.G)]
        V = np.stack([-0.5 * np.sum(((x - m) @ S) * (x - m), axis=1) + lc
                      for m, S, lc in zip(self.mus, self.S, self.logc)], axis=1)
        mx = V.max(axis=1, keepdims=True)
        e = np.exp(V - mx)
        return ys, e / e.sum(axis=1, keepdims=True), np.log(e.sum(axis=1)) + mx[:, 0]

    def energy(self, x):
        return -self._parts(x)[2]

    def grad(self, x):
        ys, w, _ = self._parts(x)
        return sum(w[:, i:i + 1] * ys[i] for i in range(len(ys)))

    def hessvec(self, x, v):
        ys, w, _ = self._parts(x)
        g = sum(w[:, i:i + 1] * ys[i] for i in range(len(ys)))
        out = g * np.sum(g * v, axis=1, keepdims=True)
        for i, y in enumerate(ys):
            out = out + w[:, i:i + 1] * (v @ self.G[i]) - w[:, i:i + 1] * y * np.sum(y * v, axis=1, keepdims=True)
        return out


class RoughWellTarget:
    """distributions.py:84-97: grad = x - (eta/den) sin(x/den), H = diag(1 - (eta/den^2) cos(x/den))."""

    def __init__(self, eta, easy, dtype):
        self.eta = dtype(eta)
        self.den = dtype(float(eta)) if easy else dtype(float(eta) * float(eta))     # the double product rounded once (distributions.py:93)

    def energy(self, x):
        return 0.5 * np.sum(x * x, axis=1) + self.eta * np.sum(np.cos(x / self.den), axis=1)

    def grad(self, x):
        return x - (self.eta / self.den) * np.sin(x / self.den)

    def hessvec(self, x, v):
        return (1.0 - (self.eta / (self.den * self.den)) * np.cos(x / self.den)) * v


class FunnelTarget:
    """distributions.py:155-180 (sigma = 2, clip = 4 sigma; `tf.where` selects the branch and routes its gradient).
    Free branch, s = e^v, q = sum_{k>=1} x_k^2, n = d - 1:  U = (v^2 / sigma^2 + q / s + n log(2 pi s)) / 2;
    grad_k = x_k / s,  grad_0 = v / sigma^2 + (n - q / s) / 2;
    H u:  (H u)_k = u_k / s - x_k u_0 / s,   (H u)_0 = u_0 (1 / sigma^2 + q / (2 s)) - sum_k x_k u_k / s.
    Clipped branches: s is a constant, so the cross terms and the q-term of (H u)_0 vanish."""

    def __init__(self, sigma, dtype):
        self.sigma = dtype(sigma)
        self.clip = dtype(4 * sigma)
        self.dtype = dtype

    def _parts(self, x):
        v = x[:, 0]
        hi, lo = v > self.clip, -self.clip > v
        s = np.exp(v)
        s_eff = np.where(hi, np.exp(self.clip), np.where(lo, np.exp(-self.clip), s))
        return v, hi | lo, s_eff, np.sum(np.square(x[:, 1:]), axis=1)

    def energy(self, x):
        v, _, s_eff, q = self._parts(x)
        n = self.dtype(x.shape[1] - 1)
        return 0.5 * (np.square(v / self.sigma) + q / s_eff + n * np.log(self.dtype(2.0 * np.pi) * s_eff))

    def grad(self, x):
        v, clipped, s_eff, q = self._parts(x)
        n = self.dtype(x.shape[1] - 1)
        g = x / s_eff[:, None]
        g[:, 0] = v / (self.sigma * self.sigma) + np.where(clipped, 0.0, 0.5 * (n - q / s_eff))
        return g

    def hessvec(self, x, u):
        v, clipped, s_eff, q = self._parts(x)
        free = (~clipped).astype(x.dtype)
        out = u / s_eff[:, None] - free[:, None] * x * (u[:, 0] / s_eff)[:, None]
        dot = np.sum(x[:, 1:] * u[:, 1:], axis=1)
        out[:, 0] = u[:, 0] * (1.0 / (self.sigma * self.sigma) + free * 0.5 * q / s_eff) - free * dot / s_eff
        return out


def target_of(g, dtype=np.float64):
    kind = str(g['energy.kind'])
    if kind == 'gaussian':
        return GaussianTarget(g['energy.mu'], g['energy.i_sigma'], dtype)
    if kind == 'gmm':
        return GMMTarget(g['energy.mus'], g['energy.i_sigmas'], g['energy.constants'], dtype)
    if kind == 'roughwell':
        return RoughWellTarget(float(g['energy.eta64']) if 'energy.eta64' in g else float(g['energy.eta']), bool(g['energy.easy']), dtype)
    if kind == 'funnel':
        return FunnelTarget(float(g['energy.sigma']), dtype)
    raise ValueError(kind)


def propose_loss_and_grad(x0, v0, direction, target, xnet, vnet, eps, mask, T, scale=0.1,
                          dtype=np.float64, float32_weights=True):
    """One direction-mixed proposal from x0 with momenta v0 (each chain in its own direction) and
    its loss term  scale * mean(1/v1) - mean(v1)/scale,  v1 = |x0 - Lx|^2 p + 1e-4  (nb 164-169).
    Returns (loss, Lx, p, grads) with grads = {'xnet': {...}, 'vnet': {...}, 'eps': d loss/d eps}."""
    x0, v0 = np.asarray(x0, dtype), np.asarray(v0, dtype)
    N, d = x0.shape
    # (the fixtures' weights ARE float32; a training replay in `dtype` -- tools/ess_seed_study.py -- keeps its own precision)
    wt = np.float32 if float32_weights else dtype
    xn = xnet if hasattr(xnet, 'fwd') else {k: np.asarray(xnet[k], wt).astype(dtype) for k in NET_KEYS}
    vn = vnet if hasattr(vnet, 'fwd') else {k: np.asarray(vnet[k], wt).astype(dtype) for k in NET_KEYS}
    eps = dtype(eps)
    mask = np.asarray(mask, dtype)
    fwd = (np.asarray(direction) != 0)[:, None]
    sgn = np.where(fwd, 1.0, -1.0).astype(dtype)
    taus = np.stack([format_time(t, T, np.float32) for t in range(T)]).astype(dtype)

    gradU, energy = target.grad, target.energy

    # ---- forward, keeping what the reverse sweep needs ------------------------------------------
    x, v = x0, v0
    ld = np.zeros(N, dtype)
    tape = []
    for it in range(T):
        srow = np.where(fwd[:, 0], it, T - 1 - it)
        tau, m = taus[srow], mask[srow]
        k1 = np.where(fwd, m, 1 - m)
        k2 = 1 - k1
        g1 = gradU(x)
        (S1, T1, Q1), c1 = _fwd(vn, x, g1, tau)
        vh, a1 = _v_half(v, g1, S1, T1, Q1, eps, sgn, fwd)
        (Sa, Ta, Qa), ca = _fwd(xn, vh, k1 * x, tau)
        y, aa = _x_half(x, k1, vh, Sa, Ta, Qa, eps, sgn, fwd)
        (Sb, Tb, Qb), cb = _fwd(xn, vh, k2 * y, tau)
        xo, ab = _x_half(y, k2, vh, Sb, Tb, Qb, eps, sgn, fwd)
        g2 = gradU(xo)
        (S2, T2, Q2), c2 = _fwd(vn, xo, g2, tau)
        vo, a2 = _v_half(vh, g2, S2, T2, Q2, eps, sgn, fwd)
        ld = ld + np.sum(sgn * 0.5 * eps * (S1 + S2) + k2 * sgn * eps * Sa + k1 * sgn * eps * Sb, axis=1)
        tape.append(dict(x=x, v=v, k1=k1, k2=k2, g1=g1, c1=c1, a1=a1, vh=vh, ca=ca, aa=aa, y=y,
                         cb=cb, ab=ab, xo=xo, g2=g2, c2=c2, a2=a2,
                         out=((S1, T1, Q1), (Sa, Ta, Qa), (Sb, Tb, Qb), (S2, T2, Q2))))
        x, v = xo, vo
    val = (energy(x0) + 0.5 * np.sum(v0 * v0, 1)) - (energy(x) + 0.5 * np.sum(v * v, 1)) + ld
    with np.errstate(all='ignore'):
        p = np.exp(np.minimum(val, 0.0))
    finite = np.isfinite(p)
    p = np.where(finite, p, 0.0)
    sq = np.sum((x0 - x) ** 2, axis=1)
    v1 = sq * p + 1e-4
    loss = scale * np.mean(1.0 / v1) - np.mean(v1) / scale

    # ---- reverse sweep ---------------------------------------------------------------------------
    dv1 = (scale * (-1.0 / v1 ** 2) - 1.0 / scale) / N
    dp = dv1 * sq
    dval = np.where(finite & (val < 0), dp * p, 0.0)
    lx = (dv1 * p)[:, None] * 2 * (x - x0) - dval[:, None] * gradU(x)
    lv = -dval[:, None] * v
    lam_ld = dval[:, None]
    gx = xn.grads if hasattr(xn, 'fwd') else {k: np.zeros_like(xn[k]) for k in NET_KEYS}
    gv = vn.grads if hasattr(vn, 'fwd') else {k: np.zeros_like(vn[k]) for k in NET_KEYS}
    deps = np.zeros(N, dtype)
    for it in reversed(range(T)):
        t = tape[it]
        (S1, T1, Q1), (Sa, Ta, Qa), (Sb, Tb, Qb), (S2, T2, Q2) = t['out']
        # v' = v_half(vh, g2, V(x', g2))
        dvh, dg2, dS, dT, dQ, de = _v_half_bwd(lv, lam_ld, t['vh'], t['g2'], S2, T2, Q2, eps, sgn, fwd, t['a2'])
        deps += de
        da, db = _bwd(vn, t['c2'], dS, dT, dQ, gv)
        dxo = lx + da + target.hessvec(t['xo'], dg2 + db)
        # x' = x_half(y, k2, vh, X(vh, k2 y))
        dy, dvh2, dS, dT, dQ, de = _x_half_bwd(dxo, lam_ld, t['y'], t['k2'], t['vh'], Sb, Tb, Qb, eps, sgn, fwd, t['ab'])
        deps += de
        da, db = _bwd(xn, t['cb'], dS, dT, dQ, gx)
        dvh = dvh + dvh2 + da
        dy = dy + t['k2'] * db
        # y = x_half(x, k1, vh, X(vh, k1 x))
        dx, dvh2, dS, dT, dQ, de = _x_half_bwd(dy, lam_ld, t['x'], t['k1'], t['vh'], Sa, Ta, Qa, eps, sgn, fwd, t['aa'])
        deps += de
        da, db = _bwd(xn, t['ca'], dS, dT, dQ, gx)
        dvh = dvh + dvh2 + da
        dx = dx + t['k1'] * db
        # vh = v_half(v, g1, V(x, g1))
        dv, dg1, dS, dT, dQ, de = _v_half_bwd(dvh, lam_ld, t['v'], t['g1'], S1, T1, Q1, eps, sgn, fwd, t['a1'])
        deps += de
        da, db = _bwd(vn, t['c1'], dS, dT, dQ, gv)
        lx = dx + da + target.hessvec(t['x'], dg1 + db)
        lv = dv
    return loss, x, p, {'xnet': gx, 'vnet': gv, 'eps': float(np.sum(deps))}


def training_loss_and_grad(g, dtype=np.float64, target=None):
    """Full notebook loss on a golden-shaped dict `g` (x with its draws + z with its draws):
    returns (loss, grads keyed like the golden: 'xnet.W1', ..., 'alpha').  `target`: an object with energy / grad /
    hessvec (like the classes above) instead of the one `g` names -- tests of caller-supplied energies."""
    xn = {k: g['xnet.' + k] for k in NET_KEYS}
    vn = {k: g['vnet.' + k] for k in NET_KEYS}
    total, out = 0.0, {}
    for tag, start in (('x', g['x']), ('z', g['z'])):
        dr = g[tag + '.dir']
        v0 = np.where(dr[:, None] != 0, g[tag + '.v_fwd'], g[tag + '.v_bwd'])
        loss, Lx, p, gr = propose_loss_and_grad(start, v0, dr, target if target is not None else target_of(g, dtype), xn, vn,
                                                g['eps'], g['mask'], int(g['T']), dtype=dtype)
        total += loss
        for net in ('xnet', 'vnet'):
            for k in NET_KEYS:
                out[net + '.' + k] = out.get(net + '.' + k, 0) + gr[net][k]
        out['alpha'] = out.get('alpha', 0.0) + gr['eps'] * float(g['eps'])     # eps = exp(alpha)
        out['L' + tag], out['p' + tag] = Lx, p
    return total, out
