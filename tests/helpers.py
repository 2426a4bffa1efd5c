"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np

from oracle import l2hmc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))
               if not f.endswith("p_accept_edge.npz") and not os.path.basename(f).startswith(("train_", "ais_", "ess_")))
CHAINOP_CASES = ["scg2d", "scg2d_hmc", "tilted8", "vae_small"]     # goldens that carry `chainop.*` keys
TRAIN_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "train_*.npz")))
AIS_CASES = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "ais_*.npz")))


def load(case):
    return np.load(os.path.join(GOLDEN, case + ".npz"))


def oracle_energy(g, dtype=np.float32):
    kind = str(g["energy.kind"])
    if kind == "gaussian":
        return O.Gaussian(g["energy.mu"], g["energy.i_sigma"], dtype)
    if kind == "gmm":
        return O.GMM(g["energy.mus"], g["energy.i_sigmas"], g["energy.constants"], dtype)
    if kind == "roughwell":
        return O.RoughWell(rough_eta(g), bool(g["energy.easy"]), dtype)
    if kind == "funnel":
        return O.GaussianFunnel(float(g["energy.sigma"]), dtype)
    if kind == "vae":
        return O.VAEPosterior(mlp_weights(g, "dec."), g["aux"], dtype)
    raise ValueError(kind)


def rough_eta(g):
    """the Python double the reference's caller passed (newer fixtures keep it; the older ones' etas are float32-exact in effect)"""
    return float(g["energy.eta64"]) if "energy.eta64" in g else float(g["energy.eta"])


def stiffness(g):
    """Curvature amplitude of the Rough Well's cosine term, eta / den^2 (= eta^-3 for the reference's default form): the
    factor by which an error in x comes back through grad U.  >= 1e3 marks the STIFF fixtures (rough*_ne, rough8_eta01,
    train_rough*_ne): two float32 evaluations of the same T-step map then differ by (ulps of x) * curvature * eps per step,
    amplified by ~exp(eps sqrt(curvature)) per step -- far above the 3e-5 / 1e-4 gates of the well-conditioned cases, for
    the reference's own arithmetic as much as for any restatement of it."""
    if str(g["energy.kind"]) != "roughwell":
        return 0.0
    eta = rough_eta(g)
    den = eta if bool(g["energy.easy"]) else eta * eta
    return eta / (den * den)


def is_stiff(g):
    return stiffness(g) >= 500.0     # (eta = 0.1: 1000 up to rounding)


_ORACLE_DISTANCE = {}


def oracle_distance(case):
    """How far the float32 numpy oracle lands from a STIFF fixture (the reference's own float32 run), per stored output:
    the yardstick of the GPU gates for these cases -- `stiff_tol`.  Both are float32 evaluations of the same op sequence,
    so this distance IS the conditioning of the map at float32, measured rather than bounded."""
    if case not in _ORACLE_DISTANCE:
        g = load(case)
        d = oracle_dynamics(g)
        x, v = g["x"], g["v"]
        out = {}
        with np.errstate(all="ignore"):
            for s in g["steps"]:
                f = d.forward_step(x, v, np.float32(s))
                b = d.backward_step(x, v, np.float32(s))
                for pre, r in (("fstep%d" % s, f), ("bstep%d" % s, b)):
                    for key, val in zip((".x", ".v", ".logdet"), r):
                        out[pre + key] = rel_err(val, g[pre + key])
            for nm, fn in (("fwd", d.forward), ("bwd", d.backward)):
                X, V, lj = fn(x, v, log_jac=True)
                p = fn(x, v)[2]
                out[nm + ".x"], out[nm + ".v"] = rel_err(X, g[nm + ".x"]), rel_err(V, g[nm + ".v"])
                out[nm + ".logjac"], out[nm + ".p"] = rel_err(lj, g[nm + ".logjac"]), abs_err(p, g[nm + ".p"])
            Lx, _, px, _ = O.propose(x, d, g["prop.v_fwd"], g["prop.v_bwd"], g["prop.dir"], g["prop.u"])
            out["prop.Lx"], out["prop.px"] = rel_err(Lx, g["prop.Lx"]), abs_err(px, g["prop.px"])
        _ORACLE_DISTANCE[case] = out
    return _ORACLE_DISTANCE[case]


def stiff_tol(case, g, key, base):
    """Gate of a GPU-vs-fixture comparison: `base` (the suite's 3e-5 / 1e-4 / 1e-4) for the well-conditioned fixtures; for a
    STIFF one, 4x the float32 oracle's own distance to the same fixture value, never below `base`."""
    if not is_stiff(g):
        return base
    return max(base, 4.0 * oracle_distance(case)[key])


def truth_dynamics(g):
    """The same T-step map evaluated in float64 WITH THE REFERENCE'S float32 CONSTANTS: for the Rough Well the divisor is
    the float32 constant the reference graph holds (`x / (self.eps * self.eps)`, distributions.py:93 -- a Python double
    converted to a float32 tensor) and eta the float32 scalar, both widened exactly; everything else (quotient, cosines,
    nets, sums) is float64.  This is the "truth" a float32 run of the reference's op sequence approximates: a pure-float64
    Rough Well (divisor 1e-4 instead of float32(1e-4), relative difference 2.5e-8) is a DIFFERENT map -- arguments 1e4 x move
    by 2.5e-4 rad, the gradient by 2.5e-2."""
    d = oracle_dynamics(g, np.float64)
    if str(g["energy.kind"]) == "roughwell":
        e32 = O.RoughWell(rough_eta(g), bool(g["energy.easy"]), np.float32)
        d._energy.eta, d._energy.den = np.float64(e32.eta), np.float64(e32.den)
    return d


_STIFF_BRACKET = {}
BRACKET_KEYS = ("x", "v", "logdet")


def stiff_bracket(case):
    """For a STIFF fixture: per stored output, (truth, e_o32) -- the float64 value (`truth_dynamics`) and the float32 numpy
    oracle's distance from it (rel_err for positions / momenta / log-dets, abs_err for accept probabilities).  The GPU gate
    `assert_bracket` holds the HIP path to 3x that distance (+ the suite's base tolerance): it says on which side of the
    truth the kernel sits, which `stiff_tol` (4x the oracle's distance to the reference's own float32 run) cannot."""
    if case not in _STIFF_BRACKET:
        g = load(case)
        d64, d32 = truth_dynamics(g), oracle_dynamics(g)
        x, v = g["x"], g["v"]
        x64, v64 = x.astype(np.float64), v.astype(np.float64)
        truth, e32 = {}, {}

        def put(key, t, o, absolute=False):
            truth[key] = np.asarray(t, np.float64)
            e32[key] = (abs_err if absolute else rel_err)(o, t)
        with np.errstate(all="ignore"):
            for s in g["steps"]:
                for pre, f in (("fstep%d" % s, "forward_step"), ("bstep%d" % s, "backward_step")):
                    t, o = getattr(d64, f)(x64, v64, np.float64(s)), getattr(d32, f)(x, v, np.float32(s))
                    for key, tv, ov in zip((".x", ".v", ".logdet"), t, o):
                        put(pre + key, tv, ov)
            for nm, fn in (("fwd", "forward"), ("bwd", "backward")):
                t, o = getattr(d64, fn)(x64, v64, log_jac=True), getattr(d32, fn)(x, v, log_jac=True)
                for key, tv, ov in zip((".x", ".v", ".logjac"), t, o):
                    put(nm + key, tv, ov)
                put(nm + ".p", getattr(d64, fn)(x64, v64)[2], getattr(d32, fn)(x, v)[2], True)
            a = (g["prop.v_fwd"], g["prop.v_bwd"], g["prop.dir"], g["prop.u"])
            tL, _, tp, _ = O.propose(x64, d64, a[0].astype(np.float64), a[1].astype(np.float64), a[2], a[3].astype(np.float64))
            oL, _, op, _ = O.propose(x, d32, *a)
            put("prop.Lx", tL, oL)
            put("prop.px", tp, op, True)
        _STIFF_BRACKET[case] = (truth, e32)
    return _STIFF_BRACKET[case]


def assert_bracket(case, key, got, base, factor=3.0, what=""):
    """|got - truth| <= factor * |oracle32 - truth| + base for output `key` of STIFF fixture `case` (see stiff_bracket)."""
    truth, e32 = stiff_bracket(case)
    e = (abs_err if key.endswith((".p", ".px")) else rel_err)(got, truth[key])
    assert e <= factor * e32[key] + base, (case, key, what, "hip %.2e vs oracle32 %.2e from the float64 map" % (e, e32[key]))
    return e, e32[key]


# ---- training gradients, per tensor --------------------------------------------------------------------------------------
# Cases whose reference-graph gradient is itself ill-conditioned in float32: the default Rough Well at eta = 0.05 (`_ne`:
# an ulp of the float32 quotient x / eta^2 moves the cosines by 2e-5) and the d = 50 ill-conditioned Gaussian under 32-wide nets
# (the reference graph's own rounding is ~1e-4 of a tensor there: the float64 restatement sits 1e-4, the float32 one 3e-4 from
# the fixture, tests/test_oracle_golden.py).  Their per-tensor gate is widened to 4x the float32 oracle's own per-tensor
# distance from the fixture, measured (`train_yardstick`), never below the plain gate.
CONDITIONED_TRAIN_CASES = ("train_rough2_ne", "train_rough6_ne", "train_rough50_ne", "train_rough6_ne_h20", "train_icg50_h32")
_TRAIN_YARD = {}


def _train_truth_target(g):
    """float64 target with the reference's float32 constants (the Rough Well's divisor and eta: see `truth_dynamics`)"""
    from oracle import l2hmc_train_oracle as TO
    if str(g["energy.kind"]) != "roughwell":
        return None
    t32, t64 = TO.target_of(g, np.float32), TO.target_of(g, np.float64)
    t64.eta, t64.den = np.float64(t32.eta), np.float64(t32.den)
    return t64


def _train_distances(case):
    """(truth, d_o32_fix, d_fix_truth, d_o32_truth): the float64 evaluation of the notebook loss's gradient (reference's
    float32 constants) per tensor, and per tensor the max-norm distances between it, the float32 numpy restatement and the
    fixture (the reference graph's own float32 run)."""
    if case not in _TRAIN_YARD:
        from oracle import l2hmc_train_oracle as TO
        g = load(case)
        with np.errstate(all="ignore"):
            _, o32 = TO.training_loss_and_grad(g, np.float32)
            _, t64 = TO.training_loss_and_grad(g, np.float64, target=_train_truth_target(g))
        names = [n + "." + k for n in ("xnet", "vnet") for k in O.NET_KEYS] + ["alpha"]
        truth, a, b, c = {}, {}, {}, {}
        for k in names:
            ref = np.asarray(g["grad." + k], np.float64)
            o, t = np.asarray(o32[k], np.float64).reshape(ref.shape), np.asarray(t64[k], np.float64).reshape(ref.shape)
            truth[k] = t
            a[k], b[k], c[k] = float(np.abs(o - ref).max()), float(np.abs(ref - t).max()), float(np.abs(o - t).max())
        _TRAIN_YARD[case] = (truth, a, b, c)
    return _TRAIN_YARD[case]


def train_yardstick(case):
    """Per tensor (and 'alpha'): how far a float32 evaluation of the same reverse-mode op sequence may be expected from the
    reference graph's own float32 gradient -- the larger of (float32 numpy restatement - fixture) and (fixture - float64
    truth).  The second term matters: numpy and the reference's torch-CPU stub round the SAME op order, so their errors are
    correlated (train_rough50_ne vnet.Ws: 1.1e-4 apart, both 5.7e-4 from the truth); an implementation with another
    summation order -- every GPU kernel -- lands an independent 5e-4 from the truth and so up to 1e-3 from the fixture."""
    _, a, b, _ = _train_distances(case)
    return {k: max(a[k], b[k]) for k in a}


def train_bracket(case):
    """(truth, e): the float64 gradient and, per tensor, the larger of the two float32 CPU runs' distances from it -- the
    bracket a GPU gradient is held to (3x, tests: `check_grads_per_tensor(..., truth, yard=e, yard_factor=3)`)."""
    truth, _, b, c = _train_distances(case)
    return truth, {k: max(b[k], c[k]) for k in b}


def check_grads_per_tensor(label, got, ref, rel=2e-4, floor=1e-6, yard=None, yard_factor=4.0):
    """Every tensor against ITS OWN size:  max |got_t - ref_t| < rel * max |ref_t| + floor * scale   (scale = the largest
    entry of any tensor: the absolute floor below which float32 sums of O(scale) terms carry no information), widened for
    the ill-conditioned fixtures to yard_factor * yard[t] (`train_yardstick`).  `got` / `ref`: dicts name -> array (or
    scalar, e.g. 'alpha').  All failing tensors are reported at once.  Returns (worst ratio err / gate, its tensor)."""
    names = [k for k in ref if k in got]
    assert names, "no tensors to compare"
    scale = max(float(np.abs(np.asarray(ref[k], np.float64)).max()) for k in names if k != "alpha")
    bad, worst = [], (0.0, None)
    for k in names:
        r = np.asarray(ref[k], np.float64)
        gt = np.asarray(got[k], np.float64).reshape(r.shape)
        m = float(np.abs(r).max())
        gate = rel * m + floor * scale
        if yard is not None:
            gate = max(gate, yard_factor * yard[k])
        err = float(np.abs(gt - r).max())
        if err / gate > worst[0]:
            worst = (err / gate, k)
        if not err < gate:
            bad.append("%s: |d| %.2e (%.1e of its max %.2e) > gate %.2e" % (k, err, err / max(m, 1e-300), m, gate))
    assert not bad, "%s: %d tensor(s) outside their per-tensor gate (scale %.2e):\n  %s" % (label, len(bad), scale, "\n  ".join(bad))
    return worst


def net_grads(dyn, extra=None):
    """name -> gradient array of a HIP Dynamics' nets (+ 'alpha'), named like the fixtures' `grad.*` keys."""
    out = {}
    for n, w in (("xnet", dyn._xw), ("vnet", dyn._vw)):
        for k in O.NET_KEYS:
            out[n + "." + k] = to_np(w[k].grad)
    out["alpha"] = float(dyn.alpha.grad)
    if extra is not None:
        for k, t in extra.items():
            out[k] = to_np(t.grad)
    return out


def fixture_grads(g, pre="grad."):
    """the `grad.*` entries of a training fixture (or of an oracle's output dict with the same names) as name -> array"""
    return {k[len(pre):]: g[k] for k in (g.keys() if hasattr(g, "keys") else g) if k.startswith(pre) and k != pre + "x0"}


def mlp_weights(g, prefix):
    return {k: g[prefix + k] for k in ("W1", "b1", "W2", "b2", "W3", "b3")}


def golden_nets(g):
    if int(g["hmc"]):
        return None, None
    return ({k: g["xnet." + k] for k in O.NET_KEYS}, {k: g["vnet." + k] for k in O.NET_KEYS})


def oracle_dynamics(g, dtype=np.float32):
    xn, vn = golden_nets(g)
    aux_h = None
    if str(g["energy.kind"]) == "vae":        # shared encoder_sampler(aux) branch, mnist_vae.py:134-150
        w = {k: v.astype(dtype) for k, v in mlp_weights(g, "enc.").items()}
        aux_h = O.mlp3(w, g["aux"].astype(dtype))
    temp = float(g["temperature"]) if "temperature" in g else 1.0     # fed placeholder, dynamics.py:47,204-205
    return O.Dynamics(int(g["x_dim"]), oracle_energy(g, dtype), int(g["T"]), g["eps"], g["mask"],
                      xn, vn, temperature=temp, dtype=dtype, aux_h=aux_h)


def rel_err(a, b):
    """max |a-b| / max(1,|b|) over the entries where the reference value b is finite."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(1.0, np.abs(b[m]))))


def abs_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    return float(np.max(np.abs(a[m] - b[m]))) if m.any() else 0.0


def check_x_next(x_next, x, Lx, px, u, tol):
    """x_next rows must equal the proposal where px-u >= 0 (sampler.py:53-55) and the old
    state elsewhere; rows whose |px-u| < tol are ties that fp32 noise may flip."""
    x_next, x, Lx = np.asarray(x_next), np.asarray(x), np.asarray(Lx)
    margin = np.asarray(px, dtype=np.float64) - np.asarray(u, dtype=np.float64)
    acc, rej = margin >= tol, margin < -tol
    fin = np.all(np.isfinite(Lx), axis=1)
    assert rel_err(x_next[acc & fin], Lx[acc & fin]) <= tol
    assert np.array_equal(x_next[rej], x[rej])


# ---- HIP side ------------------------------------------------------------------------------
def hip_energy(g):
    """l2hmc_amd energy function carrying exactly the golden's fp32 parameters."""
    from l2hmc_amd import distributions as D
    kind = str(g["energy.kind"])
    if kind == "gaussian":
        obj = D.Gaussian.__new__(D.Gaussian)
        obj.mu, obj.sigma, obj.i_sigma = g["energy.mu"], None, g["energy.i_sigma"]
        return obj.get_energy_function()
    if kind == "gmm":
        obj = D.GMM.__new__(D.GMM)
        obj.mus = list(g["energy.mus"])
        obj.i_sigmas = list(g["energy.i_sigmas"])
        obj.constants = list(g["energy.constants"])
        obj.nb_mixtures, obj.k = len(obj.mus), obj.mus[0].shape[0]
        return obj.get_energy_function()
    if kind == "roughwell":
        return D.RoughWell(int(g["x_dim"]), rough_eta(g), bool(g["energy.easy"])).get_energy_function()
    if kind == "funnel":
        return D.GaussianFunnel(int(g["x_dim"])).get_energy_function()
    if kind == "vae":
        from l2hmc_amd import vae
        dec = vae.make_decoder(int(g["x_dim"]), g["dec.W1"].shape[1], g["dec.W3"].shape[1])
        _load_mlp(dec, g, "dec.")
        return vae.VAEPosterior(dec).get_energy_function()
    raise ValueError(kind)


def _load_mlp(seq, g, prefix):
    import torch
    from l2hmc_amd import layers
    w = layers.extract_mlp3(seq)
    with torch.no_grad():
        for k in ("W1", "b1", "W2", "b2", "W3", "b3"):
            w[k].copy_(torch.as_tensor(g[prefix + k]).reshape(w[k].shape))


def aux_of(g):
    """Device tensor of the conditioning images for the VAE-shaped cases, else None."""
    return to_dev(g["aux"]) if str(g["energy.kind"]) == "vae" else None


def hip_dynamics(g, variant=0):
    """l2hmc_amd.Dynamics loaded with the golden's weights, mask and step size."""
    import torch
    from l2hmc_amd import Dynamics, layers
    hmc = bool(int(g["hmc"]))
    if str(g["energy.kind"]) == "vae":
        from l2hmc_amd import vae
        d, H = int(g["x_dim"]), int(g["H"])
        dec = vae.make_decoder(d, g["dec.W1"].shape[1], g["dec.W3"].shape[1])
        enc = vae.make_encoder_sampler(g["enc.W1"].shape[0], g["enc.W1"].shape[1], H)
        _load_mlp(dec, g, "dec.")
        _load_mlp(enc, g, "enc.")
        energy, factory = vae.VAEPosterior(dec).get_energy_function(), vae.sampler_net_factory(d, enc, H, H)
    else:
        energy, factory = hip_energy(g), (None if hmc else layers.stq_network(int(g["H"])))
    dyn = Dynamics(int(g["x_dim"]), energy, T=int(g["T"]), eps=float(g["eps"]), hmc=hmc, net_factory=factory,
                   use_temperature="temperature" in g)
    if "temperature" in g:
        dyn.temperature = float(g["temperature"])
    dyn.mask = g["mask"]
    dyn.eps_override = float(g["eps"])
    dyn.variant = variant
    if not hmc:
        with torch.no_grad():
            for w, pre in ((dyn._xw, "xnet."), (dyn._vw, "vnet.")):
                for k in O.NET_KEYS:
                    w[k].copy_(torch.as_tensor(g[pre + k]).reshape(w[k].shape))
    return dyn


def to_dev(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def to_np(t):
    return t.detach().cpu().numpy()


def synthetic_case(kind, d, H=10, T=10, N=64, seed=0, eps=0.1, head_std=0.3):
    """A golden-shaped dict (weights, mask, energy parameters, inputs) for sizes the committed
    fixtures do not cover; consumed by oracle_dynamics() and hip_dynamics() alike."""
    rng = np.random.RandomState(seed)
    g = {"x_dim": d, "H": H, "T": T, "N": N, "hmc": 0, "eps": np.float32(eps),
         "mask": O.init_mask(T, d, rng)}
    for net, fac in (("xnet", 2.0), ("vnet", 1.0)):
        def vs(shape, f):
            return (np.clip(rng.randn(*shape), -2, 2) * np.sqrt(1.3 * 2 * f / shape[0])).astype(np.float32)
        w = {"W1": vs((d, H), 1 / 3.), "W2": vs((d, H), fac / 3.), "W3": vs((2, H), 1 / 3.), "W4": vs((H, H), 1.)}
        for k in ("Ws", "Wt", "Wq"):
            w[k] = (head_std * rng.randn(H, d) / np.sqrt(H)).astype(np.float32)
        for k, n in (("b1", H), ("b2", H), ("b3", H), ("b4", H), ("bs", d), ("bt", d), ("bq", d)):
            w[k] = (0.1 * rng.randn(n)).astype(np.float32)
        w["lam_s"] = (0.2 * rng.randn(1, d)).astype(np.float32)
        w["lam_q"] = (0.2 * rng.randn(1, d)).astype(np.float32)
        for k, v in w.items():
            g[net + "." + k] = v
    scale = np.ones(d, dtype=np.float32)
    if kind == "gauss_diag":
        var = np.exp(np.linspace(np.log(1e-1), np.log(1e1), d))
        g.update({"energy.kind": "gaussian", "energy.mu": (0.3 * rng.randn(d)).astype(np.float32),
                  "energy.i_sigma": np.diag(1.0 / var).astype(np.float32)})
        scale = np.sqrt(var).astype(np.float32)
    elif kind == "gauss_dense":
        R = np.linalg.qr(rng.randn(d, d))[0]
        prec = (R.T * np.exp(rng.uniform(-1, 1, size=d))) @ R
        g.update({"energy.kind": "gaussian", "energy.mu": (0.3 * rng.randn(d)).astype(np.float32),
                  "energy.i_sigma": prec.astype(np.float32)})
    elif kind == "roughwell_easy":
        g.update({"energy.kind": "roughwell", "energy.eta": np.float32(0.1), "energy.easy": np.int32(1)})
    elif kind == "roughwell_ne":                     # the reference's default form at BASELINE config 4's second series
        g.update({"energy.kind": "roughwell", "energy.eta": np.float32(1e-2), "energy.easy": np.int32(0),
                  "energy.eta64": np.float64(1e-2)})
    elif kind.startswith("gmm"):                     # "gmm<K>": K components, slightly non-symmetric raw precisions
        K = int(kind[3:] or 2)
        mus, i_sigmas, consts = [], [], []
        for k in range(K):
            R = np.linalg.qr(rng.randn(d, d))[0]
            if d <= 8:
                ev = np.exp(rng.uniform(-0.7, 0.7, size=d))
            else:      # the reference keeps c_i = pi_i / sqrt((2 pi)^d det Sigma_i) in float32 (distributions.py:117-124): it
                ev = np.exp(rng.uniform(np.log(4.0), np.log(10.0), size=d))   # only exists for precisions around 2 pi at this d
            prec = (R.T * ev) @ R
            consts.append((1.0 + 0.3 * k) / K * np.exp(0.5 * (np.sum(np.log(ev)) - d * np.log(2 * np.pi))))
            i_sigmas.append(prec + 0.05 * np.triu(rng.randn(d, d), 1))
            mus.append(1.5 * rng.randn(d) / max(1.0, np.sqrt(d / 4.0)))     # components overlap at any d
        g.update({"energy.kind": "gmm", "energy.mus": np.asarray(mus, np.float32),
                  "energy.i_sigmas": np.asarray(i_sigmas, np.float32), "energy.constants": np.asarray(consts, np.float32)})
        scale = 1.5 * scale
    else:
        raise ValueError(kind)
    g["x"] = (rng.randn(N, d) * scale).astype(np.float32)
    if kind.startswith("gmm") and d > 8:             # start in the typical set of a component
        pick = rng.randint(0, K, size=N)
        g["x"] = (g["energy.mus"][pick] + 0.4 * rng.randn(N, d)).astype(np.float32)
    g["v"] = rng.randn(N, d).astype(np.float32)
    return g


def synthetic_vae_case(latent=50, H=200, dec_h=1024, n_pix=784, enc_h=512, T=5, N=256, seed=0, eps=0.1):
    """Config-5 sized golden-shaped dict (random weights at init-like scales; no checkpoint or
    MNIST exists offline): consumed by oracle_dynamics() and hip_dynamics()."""
    rng = np.random.RandomState(seed)
    g = synthetic_case("roughwell_easy", latent, H=H, T=T, N=N, seed=seed, eps=eps, head_std=0.05)
    g["energy.kind"] = "vae"

    def lin(i, o, f=1.0):
        return (np.clip(rng.randn(i, o), -2, 2) * np.sqrt(1.3 * 2 * f / i)).astype(np.float32)
    for pre, dims, f3 in (("dec.", (latent, dec_h, dec_h, n_pix), 0.01), ("enc.", (n_pix, enc_h, enc_h, H), 1.0)):
        g[pre + "W1"], g[pre + "W2"], g[pre + "W3"] = lin(dims[0], dims[1]), lin(dims[1], dims[2]), lin(dims[2], dims[3], f3)
        for i, n in enumerate(dims[1:], 1):
            g[pre + "b%d" % i] = (0.05 * rng.randn(n)).astype(np.float32)
    g["aux"] = (rng.rand(N, n_pix) < 0.13).astype(np.float32)
    return g
