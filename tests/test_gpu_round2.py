"""GPU tests added in round 2: the per-step C entry point, checkpoint round trip, order-independent
autocovariance, the GMM zero-weight guard, the shared-aux-branch check, and the RCCL leg of bench.py."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import abs_err, hip_dynamics, load, rel_err, to_dev, to_np

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", ["icg50", "scg2d", "mog2d", "scg2d_hmc"])
def test_l2hmc_step_entry_point_matches_reference_steps(case):
    """l2hmc_step (raw reference-layout weights, one schedule row passed by value) against the
    reference's recorded `_forward_step` / `_backward_step` outputs (dynamics.py:115-201)."""
    import torch
    from l2hmc_amd import _ffi
    g = load(case)
    dyn = hip_dynamics(g)
    L = _ffi.lib()
    N, d = g["x"].shape
    H = max(dyn.H, 1)
    x, v = to_dev(g["x"]), to_dev(g["v"])
    ws = torch.empty(_ffi.check(L.l2hmc_workspace_bytes(N, d, H)) // 4 + 4, dtype=torch.float32, device=x.device)
    en = dyn._fn.c_struct(x.device, 1.0)
    nets = (None, None)
    if not dyn.hmc:
        nets = tuple(C.pointer(_ffi.L2hmcNet(*[w[k].data_ptr() for k in _ffi.NET_FIELDS])) for w in (dyn._xw, dyn._vw))
    T = int(g["T"])
    for s in g["steps"]:
        s = int(s)
        ang = np.float32(2 * np.pi) * np.float32(s) / np.float32(T)
        mrow = dyn._mask[s].contiguous()
        for fwd, tag in ((1, "fstep%d"), (0, "bstep%d")):
            xo, vo = torch.empty_like(x), torch.empty_like(v)
            lj = torch.full((N,), 0.5, dtype=torch.float32, device=x.device)          # accumulated into
            _ffi.check(L.l2hmc_step(nets[0], nets[1], C.byref(en), x.data_ptr(), v.data_ptr(), xo.data_ptr(),
                                    vo.data_ptr(), lj.data_ptr(), mrow.data_ptr(), float(np.cos(ang)),
                                    float(np.sin(ang)), float(g["eps"]), None, fwd, N, d, H, ws.data_ptr(),
                                    _ffi.current_stream(x.device)))
            assert rel_err(to_np(xo), g[(tag % s) + ".x"]) < 3e-5, (case, s, fwd)
            assert rel_err(to_np(vo), g[(tag % s) + ".v"]) < 3e-5, (case, s, fwd)
            assert rel_err(to_np(lj) - 0.5, g[(tag % s) + ".logdet"]) < 3e-5, (case, s, fwd)


def test_checkpoint_round_trip_continues_training_bit_for_bit():
    """train 25 steps, save (parameters, Adam moments, step counter, seed, masks), reload into a FRESH sampler,
    train 25 more == 50 straight.  (mnist_vae.py:290,334 Saver; eval_sampler.py:52-59,156 mask hack.)"""
    import io
    import torch
    from l2hmc_amd import Dynamics, distributions, layers
    from l2hmc_amd.training import Trainer

    def fresh(seed_np):
        np.random.seed(seed_np)          # masks come from numpy's global RNG like the reference's
        torch.manual_seed(seed_np)
        dist = distributions.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
        dyn = Dynamics(2, dist.get_energy_function(), T=10, eps=0.1, net_factory=layers.stq_network(10))
        return dyn, Trainer(dyn, seed=5)

    x0 = torch.randn(200, 2, generator=torch.Generator().manual_seed(1)).cuda()
    dyn_a, tr_a = fresh(0)
    xs = x0.clone()
    for _ in range(50):
        _, _, xs, _ = tr_a.step(xs)
    dyn_b, tr_b = fresh(0)
    xb = x0.clone()
    for _ in range(25):
        _, _, xb, _ = tr_b.step(xb)
    buf = io.BytesIO()
    torch.save({"trainer": tr_b.state_dict(), "x": xb.cpu()}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    dyn_c, tr_c = fresh(123)             # different initial weights AND masks: everything must come from the file
    tr_c.load_state_dict(ck["trainer"])
    xc = ck["x"].cuda()
    for _ in range(25):
        _, _, xc, _ = tr_c.step(xc)
    assert torch.equal(tr_c.theta, tr_a.theta) and torch.equal(tr_c.m, tr_a.m) and torch.equal(tr_c.v, tr_a.v)
    assert torch.equal(xc, xs) and tr_c.global_step == 50
    assert torch.equal(dyn_c.mask, dyn_a.mask)
    # Dynamics.state_dict alone restores a sampler (weights, alpha, masks)
    dyn_d, _ = fresh(77)
    dyn_d.load_state_dict(dyn_a.state_dict())
    from l2hmc_amd import propose
    v = torch.randn(200, 2, generator=torch.Generator().manual_seed(2)).cuda()
    dr = torch.randint(0, 2, (200,), generator=torch.Generator().manual_seed(3)).cuda()
    u = torch.rand(200, generator=torch.Generator().manual_seed(4)).cuda()
    a = propose(x0, dyn_a, do_mh_step=True, direction=dr, v=v, u=u)
    b = propose(x0, dyn_d, do_mh_step=True, direction=dr, v=v, u=u)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


def test_device_autocov_is_order_independent():
    """per-block partial sums added in block order: bitwise identical across runs."""
    import torch
    from l2hmc_amd import func_utils
    X = torch.randn(300, 2000, 3, generator=torch.Generator().manual_seed(0)).cuda()
    S0, A0 = func_utils.device_autocov(X, 2.0)
    for _ in range(3):
        S1, A1 = func_utils.device_autocov(X, 2.0)
        assert torch.equal(S0, S1) and torch.equal(A0, A1)


def test_gmm_zero_weight_component_matches_logsumexp():
    """a first component with pi = 0 (log c = -inf) must not poison the online softmax."""
    import torch
    from l2hmc_amd import Dynamics, distributions
    mus = [np.array([-2.0, 0.0]), np.array([2.0, 0.0]), np.array([0.0, 3.0])]
    covs = [0.1 * np.eye(2), 0.1 * np.eye(2), 0.3 * np.eye(2)]
    with np.errstate(divide="ignore"):
        full = distributions.GMM(mus, covs, [0.0, 0.6, 0.4])
    live = distributions.GMM(mus[1:], covs[1:], [0.6, 0.4])
    x = torch.randn(256, 2, generator=torch.Generator().manual_seed(0)).cuda() * 2
    da = Dynamics(2, full.get_energy_function(), T=5, eps=0.1, hmc=True)
    db = Dynamics(2, live.get_energy_function(), T=5, eps=0.1, hmc=True)
    Ua, Ub = to_np(da.energy(x)), to_np(db.energy(x))
    ga, gb = to_np(da.grad_energy(x)), to_np(db.grad_energy(x))
    assert np.all(np.isfinite(Ua)) and np.all(np.isfinite(ga))
    assert rel_err(Ua, Ub) < 1e-6 and rel_err(ga, gb) < 1e-6


def test_separate_aux_branches_take_the_general_path():
    """mnist_vae.py:134-150 shares ONE encoder_sampler between XNet and VNet, and that is what the fused form evaluates (once per
    trajectory).  A factory that builds one image branch per net must not silently run VNet on XNet's encoder:
    rounds 1-5 refused it, round 6 runs such nets as what they are -- callables, each evaluating its own branch -- on the
    general path (tests/test_gpu_round6.py holds that path against the oracle)."""
    from l2hmc_amd import Dynamics, vae
    d, H = 6, 16
    dec = vae.make_decoder(d, 32, 20)
    energy = vae.VAEPosterior(dec).get_energy_function()

    def per_net_factory(x_dim, scope, factor):
        return vae.sampler_net_factory(d, vae.make_encoder_sampler(20, 24, H), H, H)(x_dim, scope=scope, factor=factor)
    sep = Dynamics(d, energy, T=3, eps=0.1, net_factory=per_net_factory)
    assert sep._user_nets and sep._split and sep._vae
    shared = vae.sampler_net_factory(d, vae.make_encoder_sampler(20, 24, H), H, H)
    assert not Dynamics(d, energy, T=3, eps=0.1, net_factory=shared)._user_nets


def test_bench_dist_leg_over_rccl_world_size_1():
    """bench.py --force-dist: process group on backend nccl (= RCCL), the sharded-ESS all-reduces and the
    flat-gradient all-reduce of a training step run on the GPU box's one GPU; --steps 20 exercises the
    short-run repeat logic."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-sweep"], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-800:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["dist"]["ranks"] == 1 and out["dist"]["backend"] == "nccl" and out["dist"]["sharded_training"]["parameters_identical_across_ranks"]
    assert 4e-3 < out["dist"]["sharded_ess"]["ess_per_mh_step"] < 8e-3         # notebook: 5.63e-3
    assert out["config"]["repeats"] > 1 and out["config"]["timed_steps"] == 20 * out["config"]["repeats"]
    assert out["value"] > 1e8 and out["config"]["state_finite"]


def test_vae_sampler_loop_with_library_philox_matches_injected_draws_and_oracle():
    """sample_chain on the split engine (config 5's sampler loop, mnist_vae.py:185-224): `seed=` draws every
    proposal's momenta / direction / uniforms from the library's Philox stream -- bit-identical to injecting the
    stream's own draws, sharding-invariant, and equal to the oracle's loop on those draws."""
    import torch
    from oracle import l2hmc_oracle as O
    from l2hmc_amd import sample_chain
    from l2hmc_amd.sampler import philox_draws
    from tests.helpers import aux_of, oracle_dynamics
    g = load("vae_small")
    dyn = hip_dynamics(g)
    x, aux = to_dev(g["x"]), aux_of(g)
    N, d, M = g["x"].shape[0], int(g["x_dim"]), 4
    v, dr, u = philox_draws(31, N, d, M)
    xf, p, xh = sample_chain(x, dyn, M, seed=31, record=True, aux=aux)
    xi, pi, xhi = sample_chain(x, dyn, M, v=v, u=u, direction=dr, record=True, aux=aux)
    assert torch.equal(xf, xi) and torch.equal(p, pi) and torch.equal(xh, xhi)
    lo = N // 2
    xs, ps, _ = sample_chain(x[lo:], dyn, M, seed=31, chain_offset=lo, aux=aux[lo:])
    assert torch.equal(xs, xf[lo:]) and torch.equal(ps, p[:, lo:])
    od = oracle_dynamics(g)
    xo = g["x"]
    with np.errstate(all="ignore"):
        for m in range(M):
            _, _, rpx, xo_next = O.propose(xo, od, to_np(v[m]), to_np(v[m]), to_np(dr[m]), to_np(u[m]), both_directions=False)
            assert abs_err(to_np(p[m]), rpx) < 2e-4, m
            flip = np.abs(rpx - to_np(u[m])) < 2e-4          # ties fp32 noise may flip
            assert rel_err(to_np(xh[m])[~flip], xo_next[~flip]) < 2e-4, m
            xo = to_np(xh[m])                                # continue from the HIP state (no tie divergence)


@pytest.mark.parametrize("kind,d,H,T,N", [
    ("gauss_diag", 1, 10, 3, 37), ("gauss_dense", 3, 15, 2, 21), ("roughwell_easy", 4, 3, 1, 16),
    ("gauss_dense", 4, 10, 5, 50), ("gauss_diag", 5, 15, 3, 33), ("roughwell_easy", 16, 7, 2, 17),
    ("gauss_diag", 17, 15, 3, 40), ("gauss_dense", 20, 10, 2, 19), ("roughwell_easy", 33, 4, 1, 23),
    ("gauss_diag", 64, 12, 2, 35), ("gauss_diag", 100, 15, 2, 18)])
def test_new_kernels_agree_with_the_general_kernel_and_the_oracle_on_odd_shapes(kind, d, H, T, N):
    """Ragged chain counts, d not a multiple of 4 / 16, H = 15 (4 hidden k-steps), T = 1: the small (d <= 4) and the
    instruction-lean kernels (variant 0) against the general kernel (variant 100) and the numpy oracle, direction-mixed
    propose with MH."""
    import torch
    from oracle import l2hmc_oracle as O
    from l2hmc_amd import propose
    from tests.helpers import oracle_dynamics, synthetic_case
    g = synthetic_case(kind, d, H=H, T=T, N=N, seed=d + H)
    rng = np.random.RandomState(1)
    dr, u = rng.randint(0, 2, N).astype(np.uint8), rng.rand(N).astype(np.float32)
    outs = {}
    for var in (0, 100):
        dyn = hip_dynamics(g, var)
        Lx, _, px, o = propose(to_dev(g["x"]), dyn, do_mh_step=True, direction=to_dev(dr), v=to_dev(g["v"]), u=to_dev(u))
        outs[var] = (to_np(Lx), to_np(px), to_np(o[0]))
    with np.errstate(all="ignore"):
        rLx, _, rpx, _ = O.propose(g["x"], oracle_dynamics(g), g["v"], g["v"], dr, u, both_directions=False)
    for var in (0, 100):
        assert rel_err(outs[var][0], rLx) < 1e-4 and abs_err(outs[var][1], rpx) < 1e-4, (var, kind, d, H, T)
    assert rel_err(outs[0][0], outs[100][0]) < 5e-5 and abs_err(outs[0][1], outs[100][1]) < 5e-5


@pytest.mark.parametrize("kind,d,H,T,N", [("gauss_diag", 5, 7, 2, 21), ("gauss_dense", 3, 15, 3, 18),
                                          ("roughwell_easy", 17, 10, 2, 19), ("gauss_diag", 33, 15, 2, 40),
                                          ("gauss_diag", 48, 12, 3, 16), ("gmm2", 2, 10, 3, 37), ("gmm5", 4, 15, 2, 21),
                                          ("gmm8", 3, 12, 2, 16), ("gmm3", 1, 9, 2, 20)])
def test_training_kernels_agree_on_odd_shapes(kind, d, H, T, N):
    """register-resident training kernel (variant 0) vs the general tile kernel (variant 100): loss, proposals and
    every parameter gradient on shapes the goldens do not cover (ragged N, d % 4 != 0, H = 15)."""
    import torch
    from l2hmc_amd.training import Trainer
    from tests.helpers import synthetic_case
    g = synthetic_case(kind, d, H=H, T=T, N=N, seed=3 * d + H, head_std=0.2)
    rng = np.random.RandomState(2)
    draws = {"z": rng.randn(N, d).astype(np.float32), "x_dir": rng.randint(0, 2, N), "z_dir": rng.randint(0, 2, N),
             "x_v": rng.randn(N, d).astype(np.float32), "z_v": rng.randn(N, d).astype(np.float32)}
    res = {}
    for var in (0, 100):
        dyn = hip_dynamics(g)
        dyn.eps_override = None
        with torch.no_grad():
            dyn.alpha.fill_(float(np.log(g["eps"])))
        tr = Trainer(dyn)
        tr.variant = var
        loss, Lx, px = tr.loss_and_grad(to_dev(g["x"]), draws=draws)
        res[var] = (float(loss), to_np(Lx), to_np(px), to_np(tr.flat).copy())
    a, b = res[0], res[100]
    scale = max(1.0, float(np.abs(b[3]).max()))
    assert abs(a[0] - b[0]) < 1e-4 * max(1.0, abs(b[0]))
    assert rel_err(a[1], b[1]) < 5e-5 and abs_err(a[2], b[2]) < 5e-5
    assert np.abs(a[3] - b[3]).max() < 3e-4 * scale, (np.abs(a[3] - b[3]).max(), scale)


def test_split_engine_workspace_reuse_is_invisible():
    """L2hmcSplitArgs.reuse: repeated launches skip the weight preparation and the image branch while nothing changed --
    and redo them as soon as a parameter or the images change (in place, or a new tensor)"""
    import torch
    from l2hmc_amd import propose
    g = load("vae_small")
    N = g["x"].shape[0]
    x, v, aux = to_dev(g["x"]), to_dev(g["v"]), to_dev(g["aux"])
    dr, u = to_dev(g["prop.dir"]), to_dev(g["prop.u"])

    def run(dyn, aux_t):
        Lx, _, px, o = propose(x, dyn, do_mh_step=True, aux=aux_t, direction=dr, v=v, u=u)
        return Lx.clone(), px.clone(), o[0].clone()

    def same(a, b):
        return all(torch.equal(p, q) for p, q in zip(a, b))
    dyn = hip_dynamics(g)
    first = run(dyn, aux)
    assert dyn._split_key is not None
    assert same(run(dyn, aux), first) and same(run(dyn, aux), first)          # reuse = 3 on these
    assert dyn._last_reuse == 3      # (round 5: it was 1 -- every launch gets a fresh detach() view of aux, identity never matched)
    with torch.no_grad():
        dyn._xw["W1"].mul_(1.25)                                               # a parameter changes (version counter)
        dyn._xw["aux_encoder"]["W3"].mul_(0.5)
    fresh = hip_dynamics(g)
    with torch.no_grad():
        fresh._xw["W1"].mul_(1.25)
        fresh._xw["aux_encoder"]["W3"].mul_(0.5)
    want = run(fresh, aux)
    assert not same(want, first)
    assert same(run(dyn, aux), want) and dyn._last_reuse == 0
    aux2 = aux.clone()
    aux2[:, ::3] = 1.0 - aux2[:, ::3]                                          # other images, another tensor
    want2 = run(fresh, aux2)
    assert not same(want2, want) and same(run(dyn, aux2), want2) and dyn._last_reuse == 1
    aux2[:, 1::3] = 1.0 - aux2[:, 1::3]                                        # ... and the same tensor modified in place
    assert same(run(dyn, aux2), run(hip_dynamics_like(fresh, g), aux2))


def hip_dynamics_like(src, g):
    """a fresh Dynamics with the weights `src` holds now"""
    import torch
    dyn = hip_dynamics(g)
    with torch.no_grad():
        for a, b in ((dyn._xw, src._xw), (dyn._vw, src._vw)):
            for k in ("W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4", "Ws", "bs", "Wt", "bt", "Wq", "bq", "lam_s", "lam_q"):
                a[k].copy_(b[k])
        for k in ("W1", "b1", "W2", "b2", "W3", "b3"):
            dyn._xw["aux_encoder"][k].copy_(src._xw["aux_encoder"][k])
    return dyn


def test_get_hmc_samples_is_the_notebook_baseline_loop():
    """utils/notebook_utils.py:25-39 on one persistent launch: shape, the states BEFORE each step, and the ESS of
    HMC(eps = 0.15) on the notebook's target (raw 388: 5.63e-3 per MH step)"""
    from l2hmc_amd import distributions as D, func_utils
    from l2hmc_amd.notebook_utils import get_hmc_samples
    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    dist = D.Gaussian(np.zeros(2), cov)
    np.random.seed(0)
    start = dist.get_samples(n=200)
    S = get_hmc_samples(2, 0.15, dist.get_energy_function(), None, T=10, steps=2000, samples=start)
    assert S.shape == (2000, 200, 2) and np.array_equal(S[0], start.astype(np.float32))
    ess = func_utils.ESS(func_utils.acl_spectrum(S, float(np.sqrt(np.trace(cov)))))
    assert 4.5e-3 < ess < 7.0e-3, ess
    assert get_hmc_samples(2, 0.15, dist.get_energy_function(), None, steps=0, samples=start).shape == (0, 200, 2)
