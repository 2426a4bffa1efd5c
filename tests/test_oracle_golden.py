"""CPU: pin the numpy oracle against the golden vectors produced by the reference's own
modules (executed under oracle/tf1_stub.py by oracle/make_goldens.py).

Tolerances: both sides are fp32 implementations of the same op sequence (numpy vs torch-CPU
kernels), so single steps agree to ~1e-6 and T-step trajectories to ~2e-5 relative
(measured max 2.0e-5 on ring4); gates are 5e-6-ish x margin: 2e-5 / 1e-4.
"""
import numpy as np
import pytest

from oracle import l2hmc_oracle as O
from tests.helpers import CASES, CHAINOP_CASES, abs_err, check_x_next, is_stiff, load, oracle_dynamics, rel_err, stiff_bracket

STEP_TOL = 3e-5     # one generalised leapfrog step (funnel's |grad| ~ 5e3 amplifies: 1.3e-5 seen)
TRAJ_TOL = 1e-4     # T steps
P_TOL = 5e-5        # accept probability (north_star asks 1e-4)

# The STIFF fixtures: the reference's default Rough Well cos(x / eta^2) (distributions.py:84-97) at eta = 1e-2 (curvature
# eta^-3 = 1e6) and eta = 0.1 (1e3).  Energy, gradient and the POSITION after one step still agree to float32 rounding (the
# same float32 quotient x / den goes into cos / sin on both sides), but the step's last momentum half-update reads grad U at
# the new position: an ulp of x' (1e-7) comes back as curvature * eps / 2 * 1e-7 in v', and every further step multiplies it
# by ~exp(eps sqrt(curvature)) (eps = 3e-4: 1.35).  numpy and torch-CPU float32 differ by exactly such ulps (libm vs
# SLEEF cos / sin, fma contraction).  Gates = 4x the distance measured between the two (columns: one step x, v, logdet;
# T steps x, v, logjac, p).  Measured (max over the stored steps / both directions): rough2_ne 2.1e-7 6.5e-5 4.0e-5 | 8.8e-7
# 1.0e-3 3.2e-5 4.6e-4; rough50_ne 2.3e-7 7.9e-5 1.8e-5 | 1.8e-6 1.7e-3 2.0e-4 1.8e-3; rough512_ne 2.4e-7 1.2e-4 2.8e-5 |
# 2.4e-6 2.5e-3 4.0e-4 2.2e-3; rough8_eta01 2.3e-7 4.6e-6 2.6e-6 | 6.1e-6 3.8e-4 4.1e-5 6.7e-5.
STIFF_TOL = {"rough2_ne": (3e-5, 3e-4, 1.6e-4, 1e-4, 4e-3, 2e-4, 2e-3),
             "rough50_ne": (3e-5, 3.2e-4, 8e-5, 1e-4, 7e-3, 8e-4, 8e-3),
             "rough512_ne": (3e-5, 5e-4, 1.2e-4, 1e-4, 1e-2, 1.6e-3, 9e-3),
             "rough8_eta01": (3e-5, 3e-5, 3e-5, 1e-4, 1.6e-3, 2e-4, 3e-4)}


def tols(case):
    """(step x, step v, step logdet, traj x, traj v, traj logjac, p)"""
    return STIFF_TOL.get(case, (STEP_TOL, STEP_TOL, STEP_TOL, TRAJ_TOL, TRAJ_TOL, TRAJ_TOL, P_TOL))


@pytest.mark.parametrize("case", CASES)
def test_energy_and_grad(case):
    g = load(case)
    d = oracle_dynamics(g)
    assert rel_err(d.energy(g["x"]), g["energy"]) < 2e-6
    assert rel_err(d.grad_energy(g["x"]), g["grad_energy"]) < 2e-6


@pytest.mark.parametrize("case", CASES)
def test_single_steps(case):
    g = load(case)
    d = oracle_dynamics(g)
    x, v = g["x"], g["v"]
    for s in g["steps"]:
        with np.errstate(all="ignore"):
            xo, vo, lj = d.forward_step(x, v, np.float32(s))
            xb, vb, ljb = d.backward_step(x, v, np.float32(s))
        sx, sv, sl = tols(case)[:3]
        for got, key, tol in ((xo, "fstep%d.x", sx), (vo, "fstep%d.v", sv), (lj, "fstep%d.logdet", sl),
                              (xb, "bstep%d.x", sx), (vb, "bstep%d.v", sv), (ljb, "bstep%d.logdet", sl)):
            assert rel_err(got, g[key % s]) < tol, (case, key % s)


@pytest.mark.parametrize("case", CASES)
def test_trajectories_and_accept_prob(case):
    g = load(case)
    d = oracle_dynamics(g)
    x, v = g["x"], g["v"]
    with np.errstate(all="ignore"):
        for nm, fn in (("fwd", d.forward), ("bwd", d.backward)):
            X, V, lj = fn(x, v, log_jac=True)
            _, _, p = fn(x, v)
            tx, tv, tl, tp = tols(case)[3:]
            assert rel_err(X, g[nm + ".x"]) < tx
            assert rel_err(V, g[nm + ".v"]) < tv
            assert rel_err(lj, g[nm + ".logjac"]) < tl
            assert abs_err(p, g[nm + ".p"]) < tp
            # NaN trajectories of the reference must be rejected (dynamics.py:309)
            bad = ~np.all(np.isfinite(g[nm + ".x"]), axis=1)
            assert np.all(p[bad] == 0)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("both", [True, False])
def test_propose(case, both):
    g = load(case)
    d = oracle_dynamics(g)
    x = g["x"]
    with np.errstate(all="ignore"):
        if int(g["hmc"]):
            Lx, Lv, px, xn = O.propose(x, d, g["prop.v_fwd"], u=g["prop.u"])
        else:
            Lx, Lv, px, xn = O.propose(x, d, g["prop.v_fwd"], g["prop.v_bwd"], g["prop.dir"],
                                       g["prop.u"], both_directions=both)
    tx, _, _, tp = tols(case)[3:]
    assert rel_err(Lx, g["prop.Lx"]) < tx
    assert abs_err(px, g["prop.px"]) < tp
    check_x_next(xn, x, g["prop.Lx"], g["prop.px"], g["prop.u"], max(1e-4, tp))
    check_x_next(g["prop.x_next"], x, g["prop.Lx"], g["prop.px"], g["prop.u"], 1e-4)


def chainop_inputs(g):
    """(init_v, v_fwd list, v_bwd list, directions list, u) of the recorded chain_operator run."""
    K = int(g["chainop.K"])
    if int(g["hmc"]):
        return K, g["chainop.init_v"], None, None, None, g["chainop.u"]
    return (K, g["chainop.init_v"], list(g["chainop.v_fwd"]), list(g["chainop.v_bwd"]), list(g["chainop.dir"]),
            g["chainop.u"])


@pytest.mark.parametrize("case", CHAINOP_CASES)
def test_chain_operator(case):
    """sampler.py:57-85 -- the reference's own `chain_operator` run on its recorded draws (init_v = its first
    normal draw): composed proposals with summed log-Jacobians, one accept against the start point, MH select."""
    g = load(case)
    d = oracle_dynamics(g)
    K, init_v, vf, vb, dirs, u = chainop_inputs(g)
    with np.errstate(all="ignore"):
        fx, fv, p, xn = O.chain_operator(g["x"], d, K, init_v, vf, vb, dirs, u)
    fin = np.all(np.isfinite(g["chainop.x"]), axis=1)
    assert fin.mean() > 0.9
    assert rel_err(fx[fin], g["chainop.x"][fin]) < 3 * TRAJ_TOL
    assert rel_err(fv[fin], g["chainop.v"][fin]) < 3 * TRAJ_TOL
    assert abs_err(p[fin], g["chainop.p"][fin]) < 3 * P_TOL
    assert np.all(p[~fin] == 0) and np.all(g["chainop.p"][~fin] == 0)
    check_x_next(xn[fin], g["x"][fin], g["chainop.x"][fin], g["chainop.p"][fin], u[fin], 3e-4)
    check_x_next(g["chainop.x_next"][fin], g["x"][fin], g["chainop.x"][fin], g["chainop.p"][fin], u[fin], 3e-4)


def test_tempered_case_is_really_tempered():
    """`tilted8_temp` was produced with use_temperature=True and the placeholder fed 2.5 (dynamics.py:203-212):
    its energy is the plain Gaussian energy / 2.5."""
    g = load("tilted8_temp")
    assert float(g["temperature"]) == 2.5
    U, _ = O.Gaussian(g["energy.mu"], g["energy.i_sigma"])(g["x"])
    assert rel_err(U / np.float32(2.5), g["energy"]) < 1e-6


def test_p_accept_edge_cases():
    """dynamics.py:302-309: non-finite accept probabilities become 0."""
    g = load("p_accept_edge")
    d = O.Dynamics(2, O.Gaussian(np.zeros(2), np.eye(2)), 2, 0.1, np.zeros((2, 2)))
    p = d.p_accept(g["x0"], g["v0"], g["x1"], g["v1"], g["logjac"])
    assert np.all(np.isfinite(p))
    assert abs_err(p, g["p"]) < 1e-6


def test_reversibility_fp64():
    """backward(forward(x,v)) == (x,v) and the log-Jacobians cancel."""
    g = load("scg2d")
    d = oracle_dynamics(g, np.float64)
    x, v = g["x"].astype(np.float64), g["v"].astype(np.float64)
    X, V, lj = d.forward(x, v, log_jac=True)
    x2, v2, lj2 = d.backward(X, V, log_jac=True)
    assert np.max(np.abs(x2 - x)) < 1e-9 and np.max(np.abs(v2 - v)) < 1e-9
    assert np.max(np.abs(lj + lj2)) < 1e-10


def test_logdet_is_log_abs_det_jacobian_fp64():
    """logdet of one forward step == log|det d(x',v')/d(x,v)| by central differences."""
    g = load("tilted8")
    d = oracle_dynamics(g, np.float64)
    x, v = g["x"][:1].astype(np.float64), g["v"][:1].astype(np.float64)
    n = x.shape[1]
    z0 = np.concatenate([x, v], axis=1)

    def f(z):
        xo, vo, lj = d.forward_step(z[:, :n], z[:, n:], np.float64(2))
        return np.concatenate([xo, vo], axis=1)[0], lj[0]

    h = 1e-6
    J = np.zeros((2 * n, 2 * n))
    for i in range(2 * n):
        dz = np.zeros_like(z0)
        dz[0, i] = h
        J[:, i] = (f(z0 + dz)[0] - f(z0 - dz)[0]) / (2 * h)
    _, logabsdet = np.linalg.slogdet(J)
    assert abs(logabsdet - f(z0)[1]) < 1e-6


def test_hmc_limit_logjac_zero():
    g = load("scg2d_hmc")
    d = oracle_dynamics(g)
    X, V, lj = d.forward(g["x"], g["v"], log_jac=True)
    assert np.all(lj == 0)
    assert d.p_accept(g["x"], g["v"], X, V, lj).mean() > 0.99


def test_masks_have_floor_half_ones():
    m = O.init_mask(7, 9, np.random.RandomState(0))
    assert m.shape == (7, 9) and np.all(m.sum(axis=1) == 4)


def test_ess_helpers():
    rng = np.random.RandomState(0)
    X = rng.randn(50, 8, 2)
    A = O.acl_spectrum(X, 1.0)
    assert A.shape == (49,)
    assert abs(A[0] - np.mean(np.sum(X * X, axis=(1, 2)) / 8)) < 1e-12
    assert 0 < O.ESS(A / A[0]) <= 1.0 + 1e-9


def test_ess_helpers_match_the_reference_func_utils():
    """utils/func_utils.py:45-54,114-120 executed unchanged (oracle/make_goldens.py `ess_case`) on a seeded
    AR(1) history: autocovariance at four lags, the whole acl spectrum and the ESS."""
    g = load("ess_funcutils")
    # scale as the generator passed it: an np.float64 scalar (under numpy 2 `X / scale` is then float64; the
    # numpy-1.x value-based casting of the reference's day kept float32 -- covered by the 1e-6 check below)
    X, scale = g["X"], np.float64(g["scale"])
    for tau, ref in zip(g["taus"], g["autocov"]):
        assert abs(O.autocovariance(X, int(tau)) - ref) <= 1e-12 * max(1.0, abs(ref))
    A = O.acl_spectrum(X, scale)
    assert A.shape == g["acl"].shape and np.max(np.abs(A - g["acl"])) < 1e-12
    assert abs(O.ESS(A) - float(g["ess"])) < 1e-12
    assert np.max(np.abs(O.acl_spectrum(X, float(scale)) - g["acl"])) < 1e-6          # float32 arithmetic
    # the product's host-side mirror (numpy path) is the same arithmetic
    from l2hmc_amd import func_utils as F
    assert np.max(np.abs(np.asarray(F.acl_spectrum(X, scale)) - g["acl"])) < 1e-6
    assert abs(float(F.ESS(np.asarray(g["acl"]))) - float(g["ess"])) < 1e-9


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors published with Random123 (kat_vectors: counters/keys
    all-zero, all-ones, and the pi digits)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = O.philox4x32_10(np.array(c, dtype=np.uint32), k[0], k[1])
        assert tuple(int(g) for g in got) == want


def test_philox_draws_are_sharding_invariant_and_standard():
    v, dr, u = O.philox_draws(1234, 64, 7, 5)
    v2, dr2, u2 = O.philox_draws(1234, 32, 7, 3, proposal0=2, chain_offset=32)
    assert np.array_equal(v[2:, 32:], v2) and np.array_equal(dr[2:, 32:], dr2) and np.array_equal(u[2:, 32:], u2)
    vb, db, ub = O.philox_draws(7, 4096, 8, 8)
    assert abs(vb.mean()) < 0.01 and abs(vb.std() - 1) < 0.01 and abs(db.mean() - 0.5) < 0.02
    assert 0 <= ub.min() and ub.max() < 1 and abs(ub.mean() - 0.5) < 0.01


@pytest.mark.parametrize("case", ["train_scg2d", "train_tilted8", "train_icg50", "train_mog2d", "train_rough6", "train_funnel3",
                                  "train_icg50_h32", "train_tilted8_h24", "train_rough6_h20", "train_mog3d_h20",
                                  "train_funnel4_h20", "train_rough2_ne", "train_rough6_ne", "train_rough50_ne",
                                  "train_rough6_ne_h20"])
def test_training_gradient_oracle_matches_reference_graph(case):
    """oracle/l2hmc_train_oracle.py (hand-derived reverse mode incl. the Hessian-vector path)
    vs tf.gradients of the notebook loss evaluated by the reference's own graph (stub)."""
    from oracle import l2hmc_train_oracle as TO
    g = load(case)
    loss, out = TO.training_loss_and_grad(g, np.float64)
    # the non-easy Rough Well at eta = 0.05 (arguments 400 x, curvature 8000): the reference graph divides in float32, this
    # restatement in float64, so the cosine arguments differ by an ulp of 400 x = 2e-5 -- measured against the four fixtures:
    # loss 6e-5, px 1.5e-4, gradients 4e-4 of their scale (float32 restatement: 6e-4).  Gates: 2e-4 / 1e-3 / 2e-3.
    stiff = case.startswith("train_rough") and "_ne" in case
    assert abs(loss - float(g["loss"])) < (2e-4 if stiff else 2e-5) * max(1.0, abs(float(g["loss"])))
    assert rel_err(out["Lx"], g["Lx"]) < TRAJ_TOL and abs_err(out["px"], g["px"]) < (1e-3 if stiff else P_TOL)
    # per tensor (round 6: before, every tensor was gated against the largest entry of ANY tensor): 2e-4 of the tensor's own max
    # + 1e-6 of the scale -- measured <= 3.3e-5 (float64) / 5.7e-5 (float32 restatement) on the well-conditioned fixtures; the
    # ill-conditioned ones (the default Rough Well at eta = 0.05; the d = 50 Gaussian under 32-wide nets, where the reference
    # graph's own float32 rounding is 1e-4 of a tensor) at 4x `train_yardstick`
    from tests.helpers import CONDITIONED_TRAIN_CASES, check_grads_per_tensor, fixture_grads, train_yardstick
    yard = train_yardstick(case) if case in CONDITIONED_TRAIN_CASES else None
    got = {k: out[k] for k in out if k.startswith(("xnet.", "vnet.")) or k == "alpha"}
    check_grads_per_tensor(case + " float64 oracle", got, fixture_grads(g), yard=yard)
    with np.errstate(all="ignore"):
        _, out32 = TO.training_loss_and_grad(g, np.float32)
    got = {k: out32[k] for k in out32 if k.startswith(("xnet.", "vnet.")) or k == "alpha"}
    check_grads_per_tensor(case + " float32 oracle", got, fixture_grads(g), yard=yard)


def test_vae_sampler_objective_oracle_matches_reference_graph():
    """oracle/vae_train_oracle.py (torch-CPU restatement of mnist_vae.py:185-226's sampler loss, float64, autograd)
    vs tf.gradients of the reference's own graph: every sampler variable (XNet, VNet, the image branch, alpha), the
    start point, and the variant with a cotangent R on the proposal."""
    from oracle import vae_train_oracle as V
    g = load("train_vae_small")
    dr = [{"v_fwd": g["prop.v_fwd"], "v_bwd": g["prop.v_bwd"], "dir": g["prop.dir"], "u": g["prop.u"]}]
    for pre, R in (("grad.", None), ("grad2.", g["R"])):
        o = V.sampler_loss_and_grad(g, dr, MH=1, R=R)
        if R is None:
            assert abs(o["loss"] - float(g["loss"])) < 2e-5 * max(1.0, abs(float(g["loss"])))
            assert rel_err(o["Lx"], g["Lx"]) < TRAJ_TOL and abs_err(o["px"], g["px"]) < P_TOL
            assert rel_err(o["x_next"], g["x_next"]) < TRAJ_TOL
        keys = [k[len(pre):] for k in g if k.startswith(pre)]
        assert len(keys) == 2 * 16 + 6 + 2
        for k in keys:
            ref = g[pre + k]
            got = np.asarray(o["grad." + k]).reshape(ref.shape)
            assert np.abs(got - ref).max() < 2e-5 * max(float(np.abs(ref).max()), 1e-3), (pre, k)


def vae_train_draws(g):
    """the recorded randomness of a train_vae_* fixture in the oracle's format (one MH iteration)"""
    if "chain.nb_steps" in g:
        return [{"nb_steps": int(g["chain.nb_steps"]), "init_v": g["chain.init_v"], "v_fwd": g["chain.v_fwd"],
                 "v_bwd": g["chain.v_bwd"], "dir": g["chain.dir"], "u": g["prop.u"]}]
    return [{"v_fwd": g["prop.v_fwd"], "v_bwd": g["prop.v_bwd"], "dir": g["prop.dir"], "u": g["prop.u"]}]


@pytest.mark.parametrize("case", ["train_vae_small_es", "train_vae_small_rlc"])
def test_vae_sampler_objective_optional_terms_match_reference_graph(case):
    """mnist_vae.py's `energy_scale` term (:214,218,224) and its `random_lf_composition` branch (:193-196: the proposal is
    the reference's chain_operator with a drawn number of composed links) -- oracle vs tf.gradients of the reference's
    own graph for every sampler variable and the start point."""
    from oracle import vae_train_oracle as V
    g = load(case)
    o = V.sampler_loss_and_grad(g, vae_train_draws(g), MH=1, energy_scale=float(g["energy_scale"]))
    assert abs(o["loss"] - float(g["loss"])) < 5e-5 * max(1.0, abs(float(g["loss"])))
    assert rel_err(o["Lx"], g["Lx"]) < 3 * TRAJ_TOL and abs_err(o["px"], g["px"]) < P_TOL
    assert rel_err(o["ediff"], g["ediff"]) < 1e-3
    keys = [k[len("grad."):] for k in g if k.startswith("grad.")]
    assert len(keys) == 2 * 16 + 6 + 2
    # the reference side is float32: a few chains with v ~ 1e-4 carry 1 / v^2 ~ 1e8 weights, so compare on the scale of
    # the whole gradient rather than per array
    scale = max(float(np.abs(g["grad." + k]).max()) for k in keys)
    for k in keys:
        ref = g["grad." + k]
        got = np.asarray(o["grad." + k]).reshape(ref.shape)
        assert np.abs(got - ref).max() < 2e-4 * scale, (k, np.abs(got - ref).max(), scale)


def test_vae_aux_branch_matches_reference_layers():
    g = load("vae_small")
    from tests.helpers import mlp_weights
    assert rel_err(O.mlp3(mlp_weights(g, "enc."), g["aux"]), g["aux_h"]) < 2e-6


@pytest.mark.parametrize("case", __import__("tests.helpers", fromlist=["AIS_CASES"]).AIS_CASES)
def test_ais_oracle_matches_reference(case):
    """oracle.ais_estimate vs the reference's own utils/ais.py (run under the TF1 stub by
    oracle/make_goldens.py) on the recorded draws: final states, log-weights, estimate, mean accept."""
    from tests.helpers import oracle_energy
    g = load(case)
    d = int(g["x_dim"])
    init = O.Gaussian(np.zeros(d), np.eye(d))
    est, mean_alpha, st = O.ais_estimate(init, oracle_energy(g), int(g["K"]), g["x"], g["v0"], g["normals"], g["u"],
                                         step_size=float(g["step_size"]), leapfrogs=int(g["T"]),
                                         num_splits=int(g["num_splits"]), refresh=bool(int(g["refresh"])),
                                         refreshment=float(g["refreshment"]))
    assert rel_err(st["x"], g["x_final"]) < 2e-5
    assert abs_err(st["w"], g["w_final"]) < 2e-4 * max(1.0, float(np.abs(g["w_final"]).max()))
    assert abs(float(est) - float(g["estimate"])) < 2e-4 * max(1.0, abs(float(g["estimate"])))
    assert abs(mean_alpha - float(g["mean_alpha"])) < 1e-5


@pytest.mark.parametrize("case", ["scg2d", "icg50", "tilted8"])
@pytest.mark.parametrize("nxn", [False, True])
def test_torch_cpu_baseline_matches_goldens(case, nxn):
    """The multi-threaded torch-CPU restatement timed by bench.py's cpu_baseline (both forms of the
    Gaussian energy: the reference's literal N x N product and the row-wise one) reproduces the
    reference's own `propose` outputs."""
    import torch
    from oracle import ref_cpu_torch as R
    g = load(case)
    nets = tuple({k: g[p + k] for k in R.NET_KEYS} for p in ("xnet.", "vnet."))
    en = R.GaussianRef(g["energy.mu"], g["energy.i_sigma"], nxn)
    dyn = R.DynamicsRef(int(g["x_dim"]), en, int(g["T"]), float(g["eps"]), g["mask"], *nets)
    t = lambda k: torch.as_tensor(g[k])
    Lx, px, x_next = R.propose(t("x"), dyn, t("prop.v_fwd"), t("prop.v_bwd"), t("prop.dir").reshape(-1), t("prop.u"))
    assert rel_err(Lx.numpy(), g["prop.Lx"]) < TRAJ_TOL
    assert abs_err(px.numpy(), g["prop.px"]) < P_TOL
    check_x_next(x_next.numpy(), g["x"], g["prop.Lx"], g["prop.px"], g["prop.u"], P_TOL)


# ---- the bf16x3 arithmetic of the GEMM engine (oracle/bf16x3_oracle.py): claims checked without a GPU ------------------------
def test_bf16x3_split_is_exact():
    """x = h + m + l exactly, every term a bf16 number, for fp32 values of every magnitude the decoder sees and well beyond
    (the split needs 16 more exponent steps below x: true for |x| >= 2^-110)."""
    from oracle import bf16x3_oracle as B
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(20000), rng.randn(2000) * 1e-20, rng.randn(2000) * 1e20, rng.rand(2000) * 2.0 ** -100,
                        [0.0, -0.0, 1.0, -1.0, 3.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0 + 255.0 / 256, 2.0 ** 100 * (2 - 2.0 ** -23)]]).astype(np.float32)
    h, m, l = B.split3(x)
    for t in (h, m, l):
        assert np.all(t.view(np.uint32) & 0xFFFF == 0)
    s = (h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64))
    assert np.array_equal(s, x.astype(np.float64))
    assert np.array_equal(((h + m).astype(np.float32) + l).astype(np.float32), x)       # ... and in fp32, in the kernels' order
    # each level is a round-to-nearest: |x - h| <= ulp_bf16(x) / 2 = 2^-8 |x| (2^-9 relative to the next power of two above)
    nz = x != 0
    assert np.all(np.abs(x[nz] - h[nz]) <= 2.0 ** -8 * np.abs(x[nz]))
    assert np.all(np.abs((x - h - m)[nz]) <= 2.0 ** -16 * np.abs(x[nz]))


def test_bf16x3_planes_layout_round_trips():
    """The plane layout the kernels agree on: (3, rows_pad, ld) bf16 bit patterns h | m | l, zero beyond the matrix; h + m + l
    gives the matrix back bit for bit (784 columns: ld = 800; 784 weight rows: 896)."""
    from oracle import bf16x3_oracle as B
    rng = np.random.RandomState(1)
    W = (rng.randn(784, 1000) * 0.05).astype(np.float32)
    P = B.to_planes(W, rows_pad=896)
    assert P.shape == (3, 896, 1024) and P.dtype == np.uint16
    assert not P[:, 784:, :].any() and not P[:, :, 1000:].any()
    assert np.array_equal(B.from_planes(P, 784, 1000), W)


def test_bf16x3_six_products_are_fp32_accurate_three_are_not():
    """A decoder-sized contraction (K = 1024, softplus-like activations, 0.05-scale weights) against float64: the six products
    of weight >= 2^-16 land where an fp32 GEMM lands (a few 1e-6 at |C| ~ 3), the three largest alone are an order of
    magnitude worse -- why the kernels issue six MFMAs per block and not three."""
    from oracle import bf16x3_oracle as B
    rng = np.random.RandomState(2)
    A = (rng.rand(96, 1024) * 2.0 - 0.5).astype(np.float32)
    W = ((rng.rand(80, 1024) - 0.5) * 0.1).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    e6 = np.abs(B.gemm_nt(A, W, B.SIX) - ref).max()
    e3 = np.abs(B.gemm_nt(A, W, B.THREE) - ref).max()
    e32 = np.abs((A @ W.T).astype(np.float64) - ref).max()
    print("K = 1024: max |err| vs float64: six products %.2e, three %.2e, fp32 GEMM %.2e (|C| <= %.2f)" % (e6, e3, e32, np.abs(ref).max()))
    assert e6 < 4e-6 and e6 < 3 * e32 + 1e-6 and e3 > 5 * e6


@pytest.mark.parametrize("case", [c for c in CASES if is_stiff(load(c))])
def test_stiff_fixtures_sit_in_the_float64_bracket(case):
    """The yardstick of tests/test_gpu_round6.py's bracket, checked on the reference's own float32 run: every stored output of a
    STIFF fixture lies within 3x the float32 numpy oracle's distance from the same map evaluated in float64 with the
    reference's float32 constants (`truth_dynamics`), + the suite's base tolerance.  Measured ratios 0.2 ... 2.2: the float64
    evaluation is the truth both float32 runs scatter around, and a pure-float64 Rough Well (divisor 1e-4 instead of
    float32(1e-4)) is not -- with that divisor the reference's own run would sit 10 ... 100x outside."""
    g = load(case)
    truth, e32 = stiff_bracket(case)
    for key in truth:
        prob = key.endswith((".p", ".px"))
        base = P_TOL if prob else (STEP_TOL if "step" in key else TRAJ_TOL)
        e = (abs_err if prob else rel_err)(g[key], truth[key])
        assert e <= 3.0 * e32[key] + base, (case, key, e, e32[key])


def test_generic_net_protocol_of_the_training_oracle_by_finite_differences():
    """oracle/l2hmc_train_oracle.py takes nets outside the notebook's architecture as objects with fwd / bwd (`TanhSigmoidNet`: the
    yardstick of tests/test_gpu_round6.py's arbitrary-net training test).  Its hand-derived reverse mode -- and the generic path
    through `propose_loss_and_grad` -- against central differences of the float64 loss: every parameter tensor's largest entry and
    three random ones, eps too."""
    from oracle import l2hmc_train_oracle as TO
    d, N, T, eps, Hh = 4, 24, 3, 0.15, 7
    rng = np.random.RandomState(3)
    shapes = (("A", (d, Hh)), ("B", (d, Hh)), ("c", (Hh,)), ("C", (2, Hh)), ("Ws", (Hh, d)), ("Wt", (Hh, d)), ("bt", (d,)), ("Wq", (Hh, d)))
    W = {net: {k: 0.4 * rng.randn(*shp) for k, shp in shapes} for net in ("X", "V")}
    prec = np.diag(np.exp(np.linspace(-0.5, 0.5, d)))
    target = TO.GaussianTarget(0.1 * rng.randn(d), prec, np.float64)
    mask = O.init_mask(T, d, np.random.RandomState(1))
    x0, v0 = rng.randn(N, d), rng.randn(N, d)
    dr = rng.randint(0, 2, N)

    def loss_of(Wd, e):
        xn, vn = TO.TanhSigmoidNet(Wd["X"], 0.7, 0.4), TO.TanhSigmoidNet(Wd["V"], 0.5, 0.3)
        ls, _, p, gr = TO.propose_loss_and_grad(x0, v0, dr, target, xn, vn, e, mask, T)
        return ls, xn.grads, vn.grads, gr["eps"], p
    ls, gx, gv, ge, p = loss_of(W, eps)
    assert 0.05 < p.mean() < 0.999
    h = 1e-6
    for net, gr in (("X", gx), ("V", gv)):
        for k, shp in shapes:
            idx = [np.unravel_index(np.abs(gr[k]).argmax(), gr[k].shape)] + [tuple(rng.randint(0, s) for s in shp) for _ in range(3)]
            for i in idx:
                Wp = {n: {kk: v.copy() for kk, v in W[n].items()} for n in W}
                Wm = {n: {kk: v.copy() for kk, v in W[n].items()} for n in W}
                Wp[net][k][i] += h
                Wm[net][k][i] -= h
                fd = (loss_of(Wp, eps)[0] - loss_of(Wm, eps)[0]) / (2 * h)
                assert abs(fd - gr[k][i]) < 1e-5 * max(1.0, np.abs(gr[k]).max()), (net, k, i, fd, gr[k][i])
    fd = (loss_of(W, eps + h)[0] - loss_of(W, eps - h)[0]) / (2 * h)
    assert abs(fd - ge) < 1e-5 * max(1.0, abs(ge)), (fd, ge)
