"""GPU tests of the GEMM-engine trainer (csrc/train_split.hpp, `l2hmc_train_split_grad`): wide nets on the built-in
targets, and the image-conditioned VAE sampler of mnist_vae.py (BASELINE.json config 5's "trained sampler")."""
import numpy as np
import pytest

from oracle import l2hmc_oracle as O
from tests.helpers import (CONDITIONED_TRAIN_CASES, abs_err, check_grads_per_tensor, fixture_grads, hip_dynamics, load, net_grads,
                           rel_err, to_dev, to_np, train_bracket, train_yardstick)

pytestmark = pytest.mark.gpu
TRAJ_TOL = 2e-4     # T steps, fp32 on the GPU vs the float32 reference graph
P_TOL = 1e-4        # accept probability (north_star)


def _trainer(g, force_split=False):
    import torch
    from l2hmc_amd.training import SplitTrainer, Trainer
    dyn = hip_dynamics(g)
    dyn.eps_override = None
    with torch.no_grad():
        dyn.alpha.fill_(float(np.log(g["eps"])))
    tr = SplitTrainer(dyn) if force_split else Trainer(dyn)
    assert isinstance(tr, SplitTrainer)
    return dyn, tr


def _check_net_grads(g, dyn, pre="grad.", tol=2e-4, yard=None, label=""):
    """every sampler variable -- XNet, VNet, the shared image branch where the fixture has it, alpha -- against ITS OWN size
    (tests/helpers.py `check_grads_per_tensor`: tol of the tensor's max + 1e-6 of the scale)"""
    ref = fixture_grads(g, pre)
    extra = None
    if any(k.startswith("enc.") for k in ref):
        enc = dyn._xw["aux_encoder"]
        extra = {"enc." + k: enc[k] for k in ("W1", "b1", "W2", "b2", "W3", "b3")}
    return check_grads_per_tensor(label, net_grads(dyn, extra), ref, rel=tol, yard=yard)


@pytest.mark.parametrize("case,force", [("train_icg50_h32", False), ("train_tilted8_h24", False), ("train_rough6_h20", False),
                                        ("train_mog3d_h20", False), ("train_funnel4_h20", False), ("train_mog2d", True),
                                        ("train_funnel3", True),
                                        ("train_scg2d", True), ("train_tilted8", True), ("train_icg50", True),
                                        ("train_rough6", True), ("train_rough6_ne_h20", False), ("train_rough6_ne", True),
                                        ("train_rough50_ne", True), ("train_rough2_ne", True)])
def test_gemm_engine_training_gradient_matches_reference_graph(case, force):
    """tf.gradients of the notebook loss from the reference's own graph (SCGExperiment.ipynb raw 156-169) vs the
    GEMM-engine trainer: the wide-net fixtures (H > 15: `Trainer` picks the engine itself) and, forced onto the engine,
    the H = 10 fixtures the register-resident kernels are pinned by."""
    g = load(case)
    dyn, tr = _trainer(g, force)
    draws = {"z": g["z"], "x_dir": g["x.dir"], "z_dir": g["z.dir"],
             "x_v": np.where(g["x.dir"][:, None] != 0, g["x.v_fwd"], g["x.v_bwd"]),
             "z_v": np.where(g["z.dir"][:, None] != 0, g["z.v_fwd"], g["z.v_bwd"])}
    loss, Lx, px = tr.loss_and_grad(to_dev(g["x"]), draws=draws)
    stiff = "_ne" in case      # the default Rough Well at eta = 0.05: gates as in tests/test_gpu_parity.py's training test
    assert abs(float(loss) - float(g["loss"])) < (2e-4 if stiff else 1e-4) * max(1.0, abs(float(g["loss"])))
    assert rel_err(to_np(Lx), g["Lx"]) < TRAJ_TOL and abs_err(to_np(px), g["px"]) < (1e-3 if stiff else P_TOL)
    worst = _check_net_grads(g, dyn, yard=train_yardstick(case) if case in CONDITIONED_TRAIN_CASES else None, label=case)
    if case in CONDITIONED_TRAIN_CASES:        # bracketed by the float64 evaluation (tests/helpers.py `train_bracket`)
        truth, e = train_bracket(case)
        check_grads_per_tensor(case + " vs float64", net_grads(dyn), truth, yard=e, yard_factor=3.0)
    print("%s: loss %.6e  worst tensor %s at %.2f of its gate" % (case, float(loss), worst[1], worst[0]))


def _vae_draws(g):
    return {"v": np.where(g["prop.dir"][:, None] != 0, g["prop.v_fwd"], g["prop.v_bwd"]), "dir": g["prop.dir"], "u": g["prop.u"]}


def test_vae_sampler_gradient_matches_reference_graph():
    """mnist_vae.py:185-226's sampler loss (MH = 1) differentiated by the reference's own graph vs the HIP trainer:
    loss, proposal, accept probability, the gradient of every sampler variable incl. the shared image branch
    (encoder_sampler) and alpha."""
    g = load("train_vae_small")
    dyn, tr = _trainer(g)
    loss, x_T, px = tr.sampler_loss_and_grad(to_dev(g["x"]), to_dev(g["aux"]), to_dev(g["log_sigma"]), MH=1, draws=[_vae_draws(g)])
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    assert abs_err(to_np(px), g["px"]) < P_TOL and rel_err(to_np(x_T), g["x_next"]) < TRAJ_TOL
    worst = _check_net_grads(g, dyn, label="train_vae_small")
    print("train_vae_small: loss %.6e  worst tensor %s at %.2f of its gate" % (float(loss), worst[1], worst[0]))


def test_vae_sampler_cotangent_in_and_start_point_gradient_out():
    """dLx_in / dx0_out of l2hmc_train_split_grad: the reference graph's gradients of loss + sum(final_x * R) w.r.t. the
    sampler variables and w.r.t. init_x (what chained proposals exchange, mnist_vae.py:185-224)."""
    import torch
    g = load("train_vae_small")
    dyn, tr = _trainer(g)
    dr = _vae_draws(g)
    N, d = g["x"].shape
    wgt = (1.0 / (torch.exp(2.0 * to_dev(g["log_sigma"])) + 1e-4)).contiguous()
    for pre, R in (("grad.", None), ("grad2.", to_dev(g["R"]))):
        tr.flat.zero_()
        dx0 = torch.empty((N, d), dtype=torch.float32, device="cuda")
        tr._propose_grad(to_dev(g["x"]), to_dev(dr["v"]), to_dev(dr["dir"]), N, aux=to_dev(g["aux"]), dist_weight=wgt,
                         dLx_in=R, dx0_out=dx0)
        tr._publish_grads()
        _check_net_grads(g, dyn, pre)
        ref = g[pre + "x0"]
        assert np.abs(to_np(dx0) - ref).max() < 2e-4 * float(np.abs(ref).max()), pre


@pytest.mark.parametrize("stop", [False, True])
def test_vae_sampler_chained_proposals_match_the_autograd_oracle(stop):
    """MH = 3 proposals chained through MH selects (mnist_vae.py:185-224) with and without stop_gradient: the trainer's
    reverse sweep over the chain vs torch autograd of the float64 CPU restatement (oracle/vae_train_oracle.py, itself
    pinned by the reference-graph fixture)."""
    from oracle import vae_train_oracle as V
    g = load("train_vae_small")
    dyn, tr = _trainer(g)
    N, d = g["x"].shape
    rng = np.random.RandomState(3)
    od, hd = [], []
    for t in range(3):
        dr = {"v_fwd": rng.randn(N, d).astype(np.float32), "v_bwd": rng.randn(N, d).astype(np.float32),
              "dir": rng.randint(0, 2, size=N).astype(np.uint8), "u": rng.rand(N).astype(np.float32)}
        od.append(dr)
        hd.append({"v": np.where(dr["dir"][:, None] != 0, dr["v_fwd"], dr["v_bwd"]), "dir": dr["dir"], "u": dr["u"]})
    o = V.sampler_loss_and_grad(g, od, MH=3, stop_gradient=stop)
    loss, x_T, px = tr.sampler_loss_and_grad(to_dev(g["x"]), to_dev(g["aux"]), to_dev(g["log_sigma"]), MH=3,
                                             stop_gradient=stop, draws=hd)
    assert abs(float(loss) - o["loss"]) < 2e-4 * max(1.0, abs(o["loss"]))
    assert rel_err(to_np(x_T), o["x_next"]) < 5e-4
    gg = {"grad." + k[5:]: v for k, v in o.items() if k.startswith("grad.")}
    _check_net_grads(gg, dyn, tol=5e-4)


def _hip_vae_draws(od):
    """oracle-format draws (both momenta per direction) -> the HIP trainer's (each chain's own direction only)"""
    if "nb_steps" in od:
        K = int(od["nb_steps"])
        return {"nb_steps": K, "init_v": od["init_v"], "u": od["u"], "dir": [od["dir"][k] for k in range(K)],
                "v": [np.where(od["dir"][k][:, None] != 0, od["v_fwd"][k], od["v_bwd"][k]) for k in range(K)]}
    return {"v": np.where(od["dir"][:, None] != 0, od["v_fwd"], od["v_bwd"]), "dir": od["dir"], "u": od["u"]}


@pytest.mark.parametrize("case", ["train_vae_small_es", "train_vae_small_rlc"])
def test_vae_sampler_optional_terms_match_reference_graph(case):
    """mnist_vae.py's `energy_scale` term (:214,218,224) and its `random_lf_composition` branch (:193-196, the proposal is
    `chain_operator` with a drawn number of composed links, sampler.py:57-85): tf.gradients of the reference's own graph
    vs the HIP trainer -- loss, accept probability, next state, every sampler variable's gradient."""
    from tests.test_oracle_golden import vae_train_draws
    g = load(case)
    dyn, tr = _trainer(g)
    R = 4 if "chain.nb_steps" in g else 0
    loss, x_T, px = tr.sampler_loss_and_grad(to_dev(g["x"]), to_dev(g["aux"]), to_dev(g["log_sigma"]), MH=1,
                                             draws=[_hip_vae_draws(vae_train_draws(g)[0])],
                                             energy_scale=float(g["energy_scale"]), random_lf_composition=R)
    assert abs(float(loss) - float(g["loss"])) < 5e-4 * max(1.0, abs(float(g["loss"]))), (float(loss), float(g["loss"]))
    assert abs_err(to_np(px), g["px"]) < P_TOL and rel_err(to_np(x_T), g["x_next"]) < 3 * TRAJ_TOL
    worst = _check_net_grads(g, dyn, tol=5e-4, label=case)
    print("%s: loss %.6e (ref %.6e)  worst tensor %s at %.2f of its gate" % (case, float(loss), float(g["loss"]), worst[1], worst[0]))


@pytest.mark.parametrize("stop", [False, True])
def test_vae_sampler_compositions_chained_over_mh_iterations_match_the_autograd_oracle(stop):
    """MH = 2 iterations, each a composition of a different number of links (2, then 3), energy_scale = 0.3, with and without
    stop_gradient: the reverse pass walks the links of both compositions and the MH select between them; float64 autograd
    of the CPU restatement is the checker (pinned by the two reference-graph fixtures above)."""
    from oracle import vae_train_oracle as V
    g = load("train_vae_small_rlc")
    dyn, tr = _trainer(g)
    N, d = g["x"].shape
    rng = np.random.RandomState(8)
    od = []
    for K in (2, 3):
        od.append({"nb_steps": K, "init_v": rng.randn(N, d).astype(np.float32),
                   "v_fwd": rng.randn(K, N, d).astype(np.float32), "v_bwd": rng.randn(K, N, d).astype(np.float32),
                   "dir": rng.randint(0, 2, size=(K, N)).astype(np.uint8), "u": rng.rand(N).astype(np.float32)})
    o = V.sampler_loss_and_grad(g, od, MH=2, stop_gradient=stop, energy_scale=0.3)
    loss, x_T, px = tr.sampler_loss_and_grad(to_dev(g["x"]), to_dev(g["aux"]), to_dev(g["log_sigma"]), MH=2,
                                             stop_gradient=stop, draws=[_hip_vae_draws(x) for x in od], energy_scale=0.3,
                                             random_lf_composition=4)
    assert abs(float(loss) - o["loss"]) < 5e-4 * max(1.0, abs(o["loss"])), (float(loss), o["loss"])
    assert rel_err(to_np(x_T), o["x_next"]) < 1e-3 and abs_err(to_np(px), o["px"]) < 2 * P_TOL
    gg = {"grad." + k[5:]: v for k, v in o.items() if k.startswith("grad.")}
    _check_net_grads(gg, dyn, tol=1e-3)


def test_gemm_engine_gradient_is_bitwise_reproducible_and_shards_add_up():
    """no atomics: two runs give identical bits; per-shard gradients with inv_n = 1 / (global count) sum to the full one"""
    import torch
    g = load("train_vae_small")
    dyn, tr = _trainer(g)
    dr = _vae_draws(g)
    N = g["x"].shape[0]
    x, v, db, aux = to_dev(g["x"]), to_dev(dr["v"]), to_dev(dr["dir"]), to_dev(g["aux"])
    wgt = (1.0 / (torch.exp(2.0 * to_dev(g["log_sigma"])) + 1e-4)).contiguous()
    runs = []
    for _ in range(2):
        tr.flat.zero_()
        tr._propose_grad(x, v, db, N, aux=aux, dist_weight=wgt)
        runs.append(tr.flat.clone())
    assert torch.equal(runs[0], runs[1])
    tr.flat.zero_()
    h = N // 2 + 3
    tr._propose_grad(x[:h].contiguous(), v[:h].contiguous(), db[:h].contiguous(), N, aux=aux[:h].contiguous(), dist_weight=wgt[:h].contiguous())
    tr._propose_grad(x[h:].contiguous(), v[h:].contiguous(), db[h:].contiguous(), N, aux=aux[h:].contiguous(), dist_weight=wgt[h:].contiguous())
    assert float((tr.flat - runs[0]).abs().max()) < 2e-5 * float(runs[0].abs().max())


def test_vae_sampler_gradient_at_config5_widths_matches_the_autograd_oracle():
    """the layer widths of BASELINE.json config 5 (latent 50, H = 200, decoder 1024, 784 pixels, image branch 512, Lf = 5)
    on a few chains: the HIP trainer vs the float64 autograd restatement"""
    from oracle import vae_train_oracle as V
    from tests.helpers import synthetic_vae_case
    N = 24
    g = synthetic_vae_case(N=N, seed=4)
    rng = np.random.RandomState(5)
    g["log_sigma"] = (0.3 * rng.randn(N, 50) - 0.5).astype(np.float32)
    dyn, tr = _trainer(g)
    dr = {"v_fwd": rng.randn(N, 50).astype(np.float32), "v_bwd": rng.randn(N, 50).astype(np.float32),
          "dir": rng.randint(0, 2, size=N).astype(np.uint8), "u": rng.rand(N).astype(np.float32)}
    o = V.sampler_loss_and_grad(g, [dr], MH=1)
    hd = {"v": np.where(dr["dir"][:, None] != 0, dr["v_fwd"], dr["v_bwd"]), "dir": dr["dir"], "u": dr["u"]}
    loss, x_T, px = tr.sampler_loss_and_grad(to_dev(g["x"]), to_dev(g["aux"]), to_dev(g["log_sigma"]), MH=1, draws=[hd])
    assert abs(float(loss) - o["loss"]) < 2e-4 * max(1.0, abs(o["loss"]))
    assert abs_err(to_np(px), o["px"]) < P_TOL
    gg = {k: v for k, v in o.items() if k.startswith("grad.")}
    worst = _check_net_grads(gg, dyn, tol=5e-4, label="config-5 widths")
    print("config-5 widths: loss %.6e  worst tensor %s at %.2f of its gate, mean p %.3f" % (float(loss), worst[1], worst[0], float(px.mean())))


def test_short_vae_sampler_training_run_improves_the_objective():
    """mnist_vae.py:185-262's sampler update (MH = 2 chained proposals, global-norm clipping, Adam) for 150 steps on a
    small synthetic decoder posterior: the objective falls, everything stays finite, and the trained sampler moves
    further per accepted proposal than the untrained one at the same step size."""
    import torch
    from l2hmc_amd.training import Trainer
    from tests.helpers import synthetic_vae_case
    N = 256
    g = synthetic_vae_case(latent=10, H=24, dec_h=48, n_pix=40, enc_h=32, T=4, N=N, seed=2)
    rng = np.random.RandomState(0)
    g["dec.W3"] = (g["dec.W3"] * 30.0).astype(np.float32)        # a decoder that actually constrains the latent
    dyn = hip_dynamics(g)
    dyn.eps_override = None
    dyn.generator = torch.Generator(device="cuda").manual_seed(0)
    tr = Trainer(dyn, decay_steps=0)
    aux = to_dev(g["aux"])
    log_sigma = to_dev(np.full((N, 10), -0.3, dtype=np.float32))

    def objective(k=8):
        tot, jump = 0.0, 0.0
        for i in range(k):
            x = torch.randn((N, 10), device="cuda", generator=dyn.generator) * 0.7
            loss, xT, px = tr.sampler_loss_and_grad(x, aux, log_sigma, MH=1)
            tot += float(loss)
            jump += float(((xT - x) ** 2).sum(1).mean())
        return tot / k, jump / k
    l0, j0 = objective()
    losses = []
    for it in range(150):
        x = torch.randn((N, 10), device="cuda", generator=dyn.generator) * 0.7
        loss, xT, px, lr = tr.sampler_step(x, aux, log_sigma, MH=2)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and bool(torch.isfinite(tr.theta).all())
    l1, j1 = objective()
    print("VAE sampler training: objective %.3f -> %.3f, mean squared jump %.3f -> %.3f, eps %.3f -> %.3f"
          % (l0, l1, j0, j1, float(g["eps"]), float(torch.exp(dyn.alpha.detach()))))
    assert l1 < l0 - 0.05 * abs(l0) and j1 > j0


@pytest.mark.parametrize("kind,d,H,T,N", [("gauss_diag", 5, 7, 2, 21), ("gauss_dense", 3, 15, 3, 18), ("roughwell_easy", 17, 10, 2, 19),
                                          ("gauss_diag", 33, 18, 1, 37), ("gauss_dense", 7, 33, 2, 5), ("roughwell_easy", 50, 64, 3, 130)])
def test_gemm_engine_trainer_on_odd_shapes_matches_the_numpy_reverse_mode_oracle(kind, d, H, T, N):
    """ragged chain counts, d and H that are no multiples of 4 / 16 (unaligned rows in every GEMM form), T = 1, nets both
    below and above the register-resident kernels' H = 15: loss, proposals and the whole flat gradient against
    oracle/l2hmc_train_oracle.py (float64; pinned by the reference-graph fixtures)"""
    from oracle import l2hmc_train_oracle as TO
    from tests.helpers import synthetic_case
    g = synthetic_case(kind, d, H=H, T=T, N=N, seed=3 * d + H, head_std=0.2)
    rng = np.random.RandomState(2)
    g["z"] = rng.randn(N, d).astype(np.float32)
    for pre in ("x.", "z."):
        g[pre + "dir"] = rng.randint(0, 2, N).astype(np.uint8)
        g[pre + "v_fwd"] = rng.randn(N, d).astype(np.float32)
        g[pre + "v_bwd"] = rng.randn(N, d).astype(np.float32)
    ref_loss, ref = TO.training_loss_and_grad(g, np.float64)
    dyn, tr = _trainer(g, force_split=True)
    draws = {"z": g["z"], "x_dir": g["x.dir"], "z_dir": g["z.dir"],
             "x_v": np.where(g["x.dir"][:, None] != 0, g["x.v_fwd"], g["x.v_bwd"]),
             "z_v": np.where(g["z.dir"][:, None] != 0, g["z.v_fwd"], g["z.v_bwd"])}
    loss, Lx, px = tr.loss_and_grad(to_dev(g["x"]), draws=draws)
    assert abs(float(loss) - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    assert rel_err(to_np(Lx), ref["Lx"]) < TRAJ_TOL and abs_err(to_np(px), ref["px"]) < P_TOL
    gg = {"grad.%s.%s" % (n, k): np.asarray(ref["%s.%s" % (n, k)]) for n in ("xnet", "vnet") for k in O.NET_KEYS}
    gg["grad.alpha"] = np.float64(ref["alpha"])
    _check_net_grads(gg, dyn, tol=3e-4)
