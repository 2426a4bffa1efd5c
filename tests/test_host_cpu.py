"""CPU: host-side logic that needs no GPU -- the layer kit / net recognition, distributions'
host parameters, diagnostics vs the oracle's restatement of utils/func_utils.py."""
import os
import numpy as np
import pytest
import torch

from l2hmc_amd import _ffi, distributions as D, func_utils, layers
from oracle import l2hmc_oracle as O


@pytest.fixture(autouse=True)
def cpu_params():
    layers.set_default_device("cpu")      # parameter holders only; nothing is computed here
    yield


def test_stq_network_is_recognised_with_reference_names():
    net = layers.stq_network(10)(50, scope="XNet", factor=2.0)
    w = layers.extract_stq(net, 50)
    assert w is not None and w["H"] == 10
    assert w["W1"].shape == (50, 10) and w["W3"].shape == (2, 10) and w["Ws"].shape == (10, 50)
    assert w["lam_s"].shape == (1, 50)
    names = [n for n, _ in net.parameters()]
    assert "XNet/embed_1/W" in names and "XNet/linear_f/b" in names and "XNet/scale_s/scale" in names
    assert len(names) == 16
    # parameter count of SURVEY.md 2.2: 5dH + H^2 + 6H + 5d
    assert sum(p.numel() for _, p in net.parameters()) == 5 * 50 * 10 + 100 + 60 + 250


def test_init_follows_variance_scaling_and_heads_start_near_zero():
    torch.manual_seed(0)
    net = layers.stq_network(10)(50, scope="VNet", factor=1.0)
    w = layers.extract_stq(net, 50)
    std = float(w["W1"].std())
    assert abs(std - np.sqrt(1.3 * (2.0 / 3) / 50) * 0.88) < 0.03      # truncated normal ~0.88 sigma
    assert float(w["Ws"].abs().max()) < 0.1 and float(w["b1"].abs().max()) == 0.0


def test_foreign_structures_are_rejected():
    bad = layers.Sequential([layers.Linear(4, 4, scope="a"), layers.relu])
    assert layers.extract_stq(bad, 4) is None
    net = layers.stq_network(10)(4, scope="X", factor=1.0)
    assert layers.extract_stq(net, 5) is None                          # wrong x_dim


def test_layer_call_protocol_matches_oracle_net():
    """Zip/Sequential/Parallel call protocol (layers.py:60-95): evaluating the holder objects
    with torch reproduces the oracle's net_apply on the same weights."""
    torch.manual_seed(1)
    net = layers.stq_network(10)(6, scope="XNet", factor=2.0)
    w = layers.extract_stq(net, 6)
    with torch.no_grad():
        for k in ("Ws", "Wt", "Wq"):
            w[k].normal_(0, 0.3)
        w["lam_s"].normal_(0, 0.2)
    rng = np.random.RandomState(0)
    a, b, tau = rng.randn(5, 6).astype(np.float32), rng.randn(5, 6).astype(np.float32), rng.randn(5, 2).astype(np.float32)
    S, T, Q = net([torch.as_tensor(a), torch.as_tensor(b), torch.as_tensor(tau), None])
    onet = {k: w[k].detach().numpy() for k in O.NET_KEYS}
    rS, rT, rQ = O.net_apply(onet, a, b, tau)
    assert np.allclose(S.detach().numpy(), rS, atol=1e-6) and np.allclose(T.detach().numpy(), rT, atol=1e-6)
    assert np.allclose(Q.detach().numpy(), rQ, atol=1e-6)


def test_gaussian_kind_selection_and_parameters():
    var = np.exp(np.linspace(np.log(1e-2), np.log(1e2), 50))
    e = D.Gaussian(np.zeros(50), np.diag(var)).get_energy_function()
    assert e.kind == _ffi.ENERGY_GAUSS_DIAG and np.allclose(e._host["prec"], 1.0 / var, rtol=1e-6)
    scg = D.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]])).get_energy_function()
    assert scg.kind == _ffi.ENERGY_GAUSS_DENSE
    assert np.allclose(scg._host["prec"], [[5.005, 4.995], [4.995, 5.005]], atol=1e-4)   # nb:105 / SURVEY E1


def test_gmm_constants_match_reference_formula():
    g = D.gen_ring(r=2.0, var=0.3, nb_mixtures=4)
    assert g.nb_mixtures == 4 and sum(g.pis) == 1.0
    c = 0.25 / np.sqrt((2 * np.pi) ** 2 * 0.3 ** 2)
    assert np.allclose(g.constants, c, rtol=1e-6)
    e = g.get_energy_function()
    assert e.kind == _ffi.ENERGY_GMM and e.n_comp == 4 and e._host["mu"].shape == (4, 2)
    s = g.get_samples(500, rng=np.random.RandomState(0))
    assert s.shape == (500, 2) and abs(np.linalg.norm(s, axis=1).mean() - 2.0) < 0.3


def test_samplers_have_the_right_moments():
    rng = np.random.RandomState(0)
    gs = D.Gaussian(np.array([1.0, -1.0]), np.array([[2.0, 0.5], [0.5, 1.0]]))
    s = gs.get_samples(20000, rng=rng)
    assert np.allclose(s.mean(0), [1, -1], atol=0.05) and np.allclose(np.cov(s.T), gs.sigma, atol=0.08)
    f = D.GaussianFunnel(dim=3).get_samples(20000, rng=rng)
    assert abs(f[:, 0].std() - 2.0) < 0.06
    assert D.RoughWell(4, 0.1).get_samples(7, rng=rng).shape == (7, 4)


def test_diagnostics_match_oracle_restatement():
    X = np.random.RandomState(3).randn(30, 6, 2)
    for tau in (0, 1, 7):
        assert abs(func_utils.autocovariance(X, tau) - O.autocovariance(X, tau)) < 1e-12
    A = func_utils.acl_spectrum(X, 1.3)
    assert np.allclose(A, O.acl_spectrum(X, 1.3))
    assert abs(func_utils.ESS(A) - O.ESS(A)) < 1e-12
    x, xp = np.zeros((4, 2)), np.ones((4, 2))
    out = func_utils.accept(x, xp, np.array([1.0, 0.0, 1.0, 0.0]), rng=np.random.RandomState(0))
    assert np.array_equal(out[:, 0], [1, 0, 1, 0])


def test_product_refuses_cpu_tensors_and_non_callable_energies():
    with pytest.raises(RuntimeError, match="no CPU path"):
        D.as_device_f32(torch.zeros(2, 2))
    from l2hmc_amd import Dynamics
    with pytest.raises(TypeError, match="energy function"):
        Dynamics(2, "not an energy", T=3, eps=0.1, hmc=True, device="cpu")
    with pytest.raises(TypeError, match="grad_energy"):
        Dynamics(2, D.Gaussian(np.zeros(2), np.eye(2)).get_energy_function(), T=3, eps=0.1, hmc=True, device="cpu",
                 grad_energy=lambda x: x)
    # a plain callable is wrapped as the caller-supplied (slow-path) energy -- and still has no CPU path
    dyn = Dynamics(2, lambda x: (x * x).sum(1), T=3, eps=0.1, hmc=True, device="cpu")
    assert isinstance(dyn._fn, D.UserEnergy) and dyn._split and dyn._user
    with pytest.raises(RuntimeError, match="no CPU path"):
        dyn.energy(torch.zeros(2, 2))
    with pytest.raises(RuntimeError, match="no CPU path"):          # ... nor do its Hessian-vector products (training)
        dyn._fn.hvp(torch.zeros(2, 2), torch.ones(2, 2))


def test_bench_host_helpers():
    """bench.py pieces that need no GPU: the usable-core count honours affinity and cgroup quota, the algorithmic
    work per chain-step is SURVEY 8(d)'s formula, the synthetic problem is seeded."""
    import bench
    assert 1 <= bench.usable_cores() <= (os.cpu_count() or 1)
    assert bench.algorithmic_flops_per_chain_step(50, 10, 10, 150) == 4 * 2 * 10 * (5 * 50 + 10 + 2) + 1.1 * 150 + 30 * 50
    a, b = bench.make_problem(0, 8, None), bench.make_problem(0, 8, None)
    assert np.array_equal(a["mask"], b["mask"]) and np.array_equal(a["nets"]["xnet"]["W1"], b["nets"]["xnet"]["W1"])
    assert a["mask"].shape == (bench.T, bench.D) and np.all(a["mask"].sum(axis=1) == bench.D // 2)


def test_small_func_utils_and_losses_helpers():
    """utils/func_utils.py:59-109 and utils/losses.py:26-59 as values: closed forms on small inputs"""
    import torch
    from l2hmc_amd import func_utils as F, losses
    rng = np.random.RandomState(0)
    qm, qs, pm, ps = rng.randn(5, 3), np.exp(0.3 * rng.randn(5, 3)), rng.randn(5, 3), np.exp(0.2 * rng.randn(5, 3))
    kl = F.normal_kl(qm, qs, pm, ps)
    ref = (np.log(ps / qs) + (qs ** 2 + (qm - pm) ** 2) / (2 * ps ** 2) - 0.5).sum(-1)
    assert np.allclose(kl, ref, rtol=1e-12)
    klt = F.normal_kl(torch.tensor(qm, dtype=torch.float32), torch.tensor(qs, dtype=torch.float32), 0., 1.)
    assert np.allclose(klt.numpy(), (-np.log(qs) + (qs ** 2 + qm ** 2) / 2 - 0.5).sum(-1), rtol=1e-5)
    x = rng.rand(200, 50)
    b = F.binarize(x, np.random.RandomState(1))
    assert b.dtype == np.float32 and set(np.unique(b)) <= {0.0, 1.0} and abs(b.mean() - x.mean()) < 0.02
    with pytest.raises(ValueError):
        F.binarize(2.0 * x)
    sh = F.binarize_and_shuffle(np.eye(6), np.random.RandomState(2))
    assert sh.shape == (6, 6) and np.all(sh.sum(0) == 1) and np.all(sh.sum(1) == 1)
    xs, Xs, p = torch.tensor(rng.randn(7, 2)), torch.tensor(rng.randn(7, 2)), torch.tensor(rng.rand(7))
    v = ((Xs - xs) ** 2).sum(1) * p + 1e-4
    assert torch.allclose(losses.loss_vec(xs, Xs, p), v)
    assert torch.allclose(losses.get_loss('mixed')(xs, Xs, p, scale=0.1), (0.1 / v).mean() - (v / 0.1).mean())
    assert torch.allclose(losses.get_loss('standard')(xs, Xs, p), -v.mean())
    assert torch.allclose(losses.get_loss('inverse')(xs, Xs, p), -1.0 / (1.0 / (v + 1e-4)).mean())
    assert torch.allclose(losses.get_loss('logsumexp')(xs, Xs, p), torch.logsumexp(-v, 0) - np.log(7.0))

    class G(object):
        mu, sigma = np.zeros(2), np.eye(2)
    assert abs(F.get_log_likelihood(np.zeros((3, 2)), G) + np.log(2 * np.pi)) < 1e-12


def test_trainer_picks_the_engine_from_the_shape_without_a_gpu():
    """`Trainer(dynamics)` asks the library which shapes a fused training kernel holds (`l2hmc_train_fused_lds_bytes`,
    host logic) and hands the others -- d > 64 at H = 10, wide nets, caller-supplied energies -- to the GEMM-engine
    trainer; construction needs no device work."""
    from l2hmc_amd import Dynamics, layers
    from l2hmc_amd.training import SplitTrainer, Trainer

    def make(d, H=10, energy=None):
        e = energy if energy is not None else D.RoughWell(d, 0.1, easy=True).get_energy_function()
        return Dynamics(d, e, T=5, eps=0.1, net_factory=layers.stq_network(H), device="cpu")
    assert type(Trainer(make(50))) is Trainer
    assert type(Trainer(make(2))) is Trainer
    assert isinstance(Trainer(make(128)), SplitTrainer)
    assert isinstance(Trainer(make(50, H=32)), SplitTrainer)
    tr = Trainer(make(3, energy=lambda x: (x * x).sum(1)))
    assert isinstance(tr, SplitTrainer) and tr.user and not tr.image_sampler



def test_exec_prologue_check_finds_the_defect_it_was_written_for():
    """tools/check_exec_prologue.py on two 40-line excerpts of this tree's own compiler output (tests/golden/asm/): the
    train_fast_kernel<2,1,3> build in which the register allocator's copy of the hidden-unit index ran before the exec restore of
    its reconvergence block (16 lanes kept garbage: gradient rows stored to wrong addresses), and the hand-corrected listing
    that was verified bit-equal on the GPU.  `make -C l2hmc_amd/csrc lint` (run by __graft_entry__.build()) applies the same
    check to every translation unit of the shipped library."""
    import importlib.util
    from tests.helpers import ROOT
    spec = importlib.util.spec_from_file_location("check_exec_prologue", os.path.join(ROOT, "tools", "check_exec_prologue.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.check(os.path.join(ROOT, "tests", "golden", "asm", "exec_prologue_bad.s"))
    assert len(bad) == 1 and "v_accvgpr_write_b32 a76, v51" in bad[0][3] and "s[16:17]" in bad[0][5]
    assert mod.check(os.path.join(ROOT, "tests", "golden", "asm", "exec_prologue_good.s")) == []
    # an `if` body with its exec restore at its own end, and SGPR-spill lane traffic ahead of a restore, are not findings
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("_Zk:\n\ts_and_saveexec_b64 s[0:1], vcc\n; %bb.1:\n\tv_add_f32_e32 v0, v1, v2\n\ts_or_b64 exec, exec, s[0:1]\n"
                "\ts_and_saveexec_b64 s[0:1], vcc\n\ts_cbranch_execz .LBB0_3\n; %bb.2:\n\tv_mov_b32_e32 v0, 0\n.LBB0_3:\n"
                "\tv_readlane_b32 s4, v255, 0\n\ts_or_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v1, v0\n\ts_endpgm\n")
    try:
        assert mod.check(f.name) == []
    finally:
        os.unlink(f.name)


def test_aux_key_follows_the_storage_not_the_tensor_object():
    """`Dynamics._aux_key` (what `L2hmcSplitArgs.reuse` bit 1 is decided by): every launch sees a fresh `detach()` view of the
    caller's images -- object identity never matches (the round-2 key; the image branch was recomputed on every launch until
    round 5) -- while (storage address, version counter, shape) identifies unchanged content and notices in-place writes,
    through any view, and other storages."""
    from l2hmc_amd.dynamics import Dynamics
    a = torch.rand(8, 12)
    v1, v2 = a.detach(), a.detach()
    assert v1 is not v2 and Dynamics._aux_key(v1) == Dynamics._aux_key(v2)
    k0 = Dynamics._aux_key(v1)
    a[:, ::3] = 0.5                                   # in place, through the base tensor
    assert Dynamics._aux_key(a.detach()) != k0
    k1 = Dynamics._aux_key(a.detach())
    v1.mul_(2.0)                                      # in place, through an old view
    assert Dynamics._aux_key(a.detach()) != k1
    assert Dynamics._aux_key(a.clone()) != Dynamics._aux_key(a.detach())          # same content, another storage
    assert Dynamics._aux_key(a[:4].detach()) != Dynamics._aux_key(a.detach())     # same address, another shape
