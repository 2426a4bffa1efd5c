"""GPU, round 4: the refusals the round-3 review listed on the contract surface are gone.

* a tempered target (`use_temperature`, dynamics.py:203-212) under nets too wide for the fused kernels (H > 15: the GEMM
  engine) -- propose + MH and the energy entry points against the oracle;
* a caller-supplied energy under the AIS bridge (`anneal_beta`, utils/ais.py:46-47): the callback anneals, the library
  is told nothing (it used to refuse the call);
* HMC-mode `_backward_step` / `backward` on the GEMM engine (the inverse leapfrog, dynamics.py:159-201 with S = T = Q = 0).
"""
import numpy as np
import pytest

from oracle import l2hmc_oracle as O
from tests.helpers import (abs_err, check_x_next, hip_dynamics, load, oracle_dynamics, rel_err, synthetic_case, to_dev, to_np)
from tests.test_gpu_round3 import _banana_np, _banana_torch

pytestmark = pytest.mark.gpu

TRAJ_TOL, P_TOL = 1e-4, 1e-4


@pytest.mark.parametrize("kind,d,H", [("gauss_diag", 50, 24), ("roughwell_easy", 12, 20), ("gauss_dense", 10, 32)])
def test_tempered_energy_under_wide_nets(kind, d, H):
    from l2hmc_amd import propose
    N = 96
    g = synthetic_case(kind, d, H=H, T=4, N=N, seed=d + H)
    g["temperature"] = np.float32(2.5)
    rng = np.random.RandomState(4)
    dr, u = rng.randint(0, 2, N).astype(np.uint8), rng.rand(N).astype(np.float32)
    dyn, od = hip_dynamics(g, 0), oracle_dynamics(g)
    assert dyn.use_temperature and dyn.temperature == 2.5 and dyn.H == H
    Lx, _, px, o = propose(to_dev(g["x"]), dyn, do_mh_step=True, direction=to_dev(dr), v=to_dev(g["v"]), u=to_dev(u))
    with np.errstate(all="ignore"):
        rLx, _, rpx, _ = O.propose(g["x"], od, g["v"], g["v"], dr, u, both_directions=False)
        cold = O.propose(g["x"], oracle_dynamics({k: v for k, v in g.items() if k != "temperature"}), g["v"], g["v"],
                         dr, u, both_directions=False)
    assert rel_err(cold[0], rLx) > 1e-2          # the temperature really changes the trajectory
    assert rel_err(to_np(Lx), rLx) < TRAJ_TOL and abs_err(to_np(px), rpx) < P_TOL, (kind, d, H)
    check_x_next(to_np(o[0]), g["x"], rLx, rpx, u, P_TOL)
    assert rel_err(to_np(dyn.energy(to_dev(g["x"]))), od.energy(g["x"])) < 1e-5
    assert rel_err(to_np(dyn.grad_energy(to_dev(g["x"]))), od.grad_energy(g["x"])) < 1e-5


@pytest.mark.parametrize("beta", [0.35, 1.0])
def test_user_energy_under_the_ais_bridge(beta):
    """`dyn.anneal_beta = b` on a callable-energy HMC Dynamics: trajectories, accept probabilities, energy and gradient
    are those of (1 - b) |x|^2 / 2 + b U(x) (utils/ais.py:46-47)."""
    import torch
    from l2hmc_amd import Dynamics, propose
    d, T, N = 6, 5, 80
    rng = np.random.RandomState(11)
    x0 = (rng.randn(N, d) * np.array([2.0] + [1.0] * (d - 1))).astype(np.float32)
    v0, u = rng.randn(N, d).astype(np.float32), rng.rand(N).astype(np.float32)
    dyn = Dynamics(d, _banana_torch, T=T, eps=0.07, hmc=True)
    dyn.eps_override = 0.07
    dyn.anneal_beta = beta

    def bridged(x):
        U, g = _banana_np(x)
        one = x.dtype.type(1.0)
        return ((one - x.dtype.type(beta)) * x.dtype.type(0.5) * np.sum(x * x, axis=1) + x.dtype.type(beta) * U,
                (one - x.dtype.type(beta)) * x + x.dtype.type(beta) * g)
    od = O.Dynamics(d, bridged, T, np.float32(0.07), np.zeros((T, d), np.float32), None, None)
    assert rel_err(to_np(dyn.energy(to_dev(x0))), od.energy(x0)) < 1e-5
    assert rel_err(to_np(dyn.grad_energy(to_dev(x0))), od.grad_energy(x0)) < 1e-5
    Lx, Lv, px, outs = propose(to_dev(x0), dyn, init_v=to_dev(v0), do_mh_step=True, u=to_dev(u))
    rLx, rLv, rpx, _ = O.propose(x0, od, v0, u=u)
    assert rel_err(to_np(Lx), rLx) < TRAJ_TOL and rel_err(to_np(Lv), rLv) < TRAJ_TOL and abs_err(to_np(px), rpx) < P_TOL
    check_x_next(to_np(outs[0]), x0, rLx, rpx, u, P_TOL)
    if beta < 1.0:                               # the bridge really changes the dynamics
        plain = O.Dynamics(d, _banana_np, T, np.float32(0.07), np.zeros((T, d), np.float32), None, None)
        assert rel_err(O.propose(x0, plain, v0, u=u)[0], rLx) > 1e-3


def test_hmc_mode_goes_backwards_on_the_gemm_engine():
    """HMC mode with a caller-supplied energy (the GEMM engine): `_backward_step` inverts `_forward_step`, `backward` inverts
    `forward`, both match the oracle's inverse leapfrog, log-det 0."""
    import torch
    from l2hmc_amd import Dynamics
    d, T, N = 6, 5, 64
    rng = np.random.RandomState(3)
    x0 = (rng.randn(N, d) * np.array([2.0] + [1.0] * (d - 1))).astype(np.float32)
    v0 = rng.randn(N, d).astype(np.float32)
    dyn = Dynamics(d, _banana_torch, T=T, eps=0.05, hmc=True)
    dyn.eps_override = 0.05
    od = O.Dynamics(d, _banana_np, T, np.float32(0.05), np.zeros((T, d), np.float32), None, None)
    hx, hv, hl = dyn._backward_step(to_dev(x0), to_dev(v0), 2)
    rx, rv, rl = od.backward_step(x0, v0, 2)
    assert rel_err(to_np(hx), rx) < 3e-5 and rel_err(to_np(hv), rv) < 3e-5 and float(hl.abs().max()) == 0.0
    fx, fv, _ = dyn._forward_step(hx, hv, 2)
    assert rel_err(to_np(fx), x0) < 3e-5 and rel_err(to_np(fv), v0) < 3e-5
    X, V, p = dyn.backward(to_dev(x0), init_v=to_dev(v0))
    rX, rV, rp = od.backward(x0, v0)
    assert rel_err(to_np(X), rX) < TRAJ_TOL and rel_err(to_np(V), rV) < TRAJ_TOL and abs_err(to_np(p), rp) < P_TOL
    Xf, Vf, _ = dyn.forward(X, init_v=V)
    assert rel_err(to_np(Xf), x0) < TRAJ_TOL and rel_err(to_np(Vf), v0) < TRAJ_TOL


def test_tile_kernel_with_eight_tiles_per_workgroup_and_the_kernel_name_query():
    """A long schedule (T = 25) makes the one-wave-per-tile kernel's staged tables -- the split bf16 head fragments are 51 KB of
    them -- too large for two workgroups per CU; from 128 chains per CU on the dispatcher then puts EIGHT tiles into one
    workgroup.  Same numbers as the oracle on the chains we can afford to check, bit-identical to the four-tile form on the
    rest (a forced variant 4 -> the four-wave kernel is a different summation order, so the comparison there is the tolerance),
    and `l2hmc_last_kernel` names what ran."""
    import torch
    from l2hmc_amd import _ffi, propose
    N, T = 32768, 25
    g = synthetic_case("gauss_diag", 50, H=10, T=T, N=256, seed=77)
    rng = np.random.RandomState(8)
    reps = N // 256
    x = np.tile(g["x"], (reps, 1)) + 0.01 * rng.randn(N, 50).astype(np.float32)
    v = rng.randn(N, 50).astype(np.float32)
    dr, u = rng.randint(0, 2, N).astype(np.uint8), rng.rand(N).astype(np.float32)
    dyn = hip_dynamics(g, 0)
    Lx, _, px, outs = propose(to_dev(x), dyn, do_mh_step=True, direction=to_dev(dr), v=to_dev(v), u=to_dev(u))
    name = _ffi.last_kernel()
    assert name == "traj_tile_kernel<1, 4, 3, 8, true>", name      # (d = 50: the last slice holds 2 dimensions -> the HALF form)
    od = oracle_dynamics(g)
    k = 192                                                   # oracle on the first chains (they span two workgroups)
    rLx, _, rpx, _ = O.propose(x[:k], od, v[:k], v[:k], dr[:k], u[:k], both_directions=False)
    assert rel_err(to_np(Lx)[:k], rLx) < TRAJ_TOL and abs_err(to_np(px)[:k], rpx) < P_TOL
    check_x_next(to_np(outs[0])[:k], x[:k], rLx, rpx, u[:k], P_TOL)
    # the same chains through the four-tile workgroups (16 384 chains at a time: below the 8-tile threshold)
    h = N // 2
    for lo in (0, h):
        Lh, _, ph, _ = propose(to_dev(x[lo:lo + h]), dyn, do_mh_step=True, direction=to_dev(dr[lo:lo + h]), v=to_dev(v[lo:lo + h]),
                               u=to_dev(u[lo:lo + h]))
        assert _ffi.last_kernel() == "traj_tile_kernel<1, 4, 3, 4, true>"
        assert torch.equal(Lh, Lx[lo:lo + h]) and torch.equal(ph, px[lo:lo + h])
    # and the names of the other families
    small = synthetic_case("gauss_diag", 50, H=10, T=5, N=64, seed=3)
    propose(to_dev(small["x"]), hip_dynamics(small, 0), direction=to_dev(dr[:64]), v=to_dev(small["v"]))
    assert _ffi.last_kernel() == "traj_fast_kernel<1, 1, 4, 3, 1>"          # (f16x2 contractions: traj_fast.hpp)
    two = synthetic_case("gauss_dense", 2, H=10, T=5, N=64, seed=3)
    propose(to_dev(two["x"]), hip_dynamics(two, 0), direction=to_dev(dr[:64]), v=to_dev(two["v"]))
    assert _ffi.last_kernel().startswith("traj_small_kernel<2")


@pytest.mark.parametrize("case", ["train_scg2d", "train_icg50", "train_mog2d"])
def test_fused_optimiser_step_equals_the_sequence_of_library_calls(case):
    """`Trainer.step` = `l2hmc_train_step` (three launches: the slot reduction overwrites the gradient and carries the loss
    terms, the Metropolis select and Adam) against the same step spelt as round 3 did -- Philox fill, [x; z] staged side
    by side, `l2hmc_train_propose_grad` into a zeroed buffer, `l2hmc_loss_terms`, `l2hmc_adam_step`, `l2hmc_mh_select`:
    parameters, Adam moments, selected states, accept probabilities and the loss are bit-identical, over three steps, on the
    d <= 4 kernel, the register-resident one and the mixture target."""
    import torch
    from l2hmc_amd import _ffi
    from l2hmc_amd.training import Trainer
    g = load(case)

    def fresh():
        dyn = hip_dynamics(g)
        dyn.eps_override = None
        with torch.no_grad():
            dyn.alpha.fill_(float(np.log(g["eps"])))
        return Trainer(dyn, seed=3)
    ta, tb = fresh(), fresh()
    L, dev = _ffi.lib(), ta.dyn.device
    s = _ffi.current_stream(dev)
    N, d = g["x"].shape
    xa = xb = to_dev(g["x"])
    for it in range(3):
        loss_a, p_a, xa, lr = ta.step(xa)
        # ---- the unfused sequence on the second trainer --------------------------------------------------------------
        W = torch.empty((4, N, d), dtype=torch.float32, device=dev)
        dirs = torch.empty((3, N), dtype=torch.uint8, device=dev)
        us = torch.empty((3, N), dtype=torch.float32, device=dev)
        _ffi.check(L.l2hmc_rng_fill(tb.seed, 3 * tb.global_step, 0, N, d, 3, W[1].data_ptr(), dirs.data_ptr(), us.data_ptr(), s))
        W[0].copy_(xb)
        tb.flat.zero_()
        Lx, p12, v1 = tb._propose_grad(W[0:2].view(2 * N, d), W[2:4].view(2 * N, d), dirs[1:3].view(2 * N), N)
        lt = torch.empty(3, dtype=torch.float64, device=dev)
        _ffi.check(L.l2hmc_loss_terms(v1.data_ptr(), v1.numel(), tb.scale, 1.0 / N, lt.data_ptr(), s))
        lr_b = tb.lr_at(tb.global_step)
        tb.global_step += 1
        n_par = tb.n_grad if tb.train_alpha else tb.n_grad - 1
        _ffi.check(L.l2hmc_adam_step(tb.theta.data_ptr(), tb.flat.data_ptr(), tb.m.data_ptr(), tb.v.data_ptr(), n_par, lr_b,
                                     tb.beta1, tb.beta2, tb.epsilon, tb.global_step, int(tb.train_alpha), s))
        tb.dyn._packed_key = None
        x_next = torch.empty_like(xb)
        _ffi.check(L.l2hmc_mh_select(xb.data_ptr(), Lx.data_ptr(), p12.data_ptr(), us[0].data_ptr(), N, d, x_next.data_ptr(), s))
        xb = x_next
        assert lr == lr_b
        assert torch.equal(ta.flat, tb.flat), (case, it)
        assert torch.equal(ta.theta, tb.theta) and torch.equal(ta.m, tb.m) and torch.equal(ta.v, tb.v), (case, it)
        assert torch.equal(xa, xb) and torch.equal(p_a, p12[:N]) and float(loss_a) == float(lt[2]), (case, it)


def test_adam_after_the_all_reduce_forms_the_global_loss():
    """`l2hmc_adam_step_terms`: the same update as `l2hmc_adam_step`, and the loss of the global batch from the (hi, lo) float
    pairs a sharded step all-reduces behind its gradient."""
    import torch
    from l2hmc_amd import _ffi
    L = _ffi.lib()
    dev = torch.device("cuda", 0)
    s = _ffi.current_stream(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    n = 541
    th = torch.randn(n, device=dev, generator=gen)
    gr = torch.randn(n, device=dev, generator=gen)
    m = 0.1 * torch.randn(n, device=dev, generator=gen)
    v = torch.rand(n, device=dev, generator=gen)
    th2, m2, v2 = th.clone(), m.clone(), v.clone()
    A, B, cnt, scale = 123456.789012345, 98.7654321, 400.0, float(np.float32(0.1))     # the entry point takes a float scale
    t6 = torch.tensor([np.float32(A), A - float(np.float32(A)), np.float32(B), B - float(np.float32(B)), cnt, 0.0],
                      dtype=torch.float32, device=dev)
    out = torch.zeros(3, dtype=torch.float64, device=dev)
    _ffi.check(L.l2hmc_adam_step(th.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 7, 1, s))
    _ffi.check(L.l2hmc_adam_step_terms(th2.data_ptr(), gr.data_ptr(), m2.data_ptr(), v2.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, 7, 1,
                                       t6.data_ptr(), scale, out.data_ptr(), s))
    assert torch.equal(th, th2) and torch.equal(m, m2) and torch.equal(v, v2)
    o = out.cpu().numpy()
    assert abs(o[0] - A) < 1e-9 * A and abs(o[1] - B) < 1e-9 * B
    assert abs(o[2] - (scale * A - B / scale) / cnt) < 1e-9 * abs(o[2])


@pytest.mark.parametrize("N", [6144, 6200, 8192, 8205])
def test_config5_on_presplit_planes(N):
    """Config 5's widths with the decoder-sized products on pre-split bf16 planes (csrc/gemm_xl.hpp: 256 x 128 tiles, weights
    split once per call, activations by the epilogue that produces them -- same six bf16 products per block as the in-loop
    split, so the same fp32-level accuracy) at MORE than one tile per XCD slot: 6144 chains = 24 row tiles (the XCD-aware tile
    order engages from 16), 6200: the last row tile is partial (56 rows), its loads clamp to the last chain, and 8192: BASELINE
    config 5's own chain count (32 row tiles, the shape bench.py times), 8205 (round 5): the 32-chain forms of the net kernels
    and of the split-K product with a last workgroup of 13 chains.  Positions
    2e-4 relative, accept probability 1e-4 absolute against the float64 evaluation of the same map (the tolerances of the
    3072-chain test, tests/test_gpu_round3.py)."""
    from l2hmc_amd import _ffi, propose
    from tests.helpers import aux_of, synthetic_vae_case
    g = synthetic_vae_case(N=N, seed=6)
    dyn = hip_dynamics(g)
    assert dyn.gemm_mode == 3                                 # (f16x2 planes; the bf16x3 planes of round 4 are re-run below)
    rng = np.random.RandomState(3)
    direction = rng.randint(0, 2, size=N).astype(np.uint8)
    u = rng.rand(N).astype(np.float32)
    aux = aux_of(g)
    Lx, _, px, outs = propose(to_dev(g["x"]), dyn, do_mh_step=True, direction=to_dev(direction), v=to_dev(g["v"]),
                              u=to_dev(u), aux=aux)
    assert _ffi.last_kernel() == "gemm_xlp_kernel"
    od64 = oracle_dynamics(g, np.float64)
    with np.errstate(all="ignore"):
        tLx, _, tpx, _ = O.propose(g["x"].astype(np.float64), od64, g["v"].astype(np.float64), g["v"].astype(np.float64),
                                   direction, u.astype(np.float64), both_directions=False)
    ex, ep = rel_err(to_np(Lx), tLx), abs_err(to_np(px), tpx)
    print("config 5 @ %d chains on planes: mean p %.3f  max rel err x %.2e  |p - p64| max %.2e" % (N, float(tpx.mean()), ex, ep))
    assert ex < 2e-4 and ep < 1e-4, (ex, ep)
    check_x_next(to_np(outs[0]), g["x"], tLx, tpx, u, 5e-4)
    # the in-loop form of the same products (gemm_mode 0 = f32-input MFMA) agrees to the same tolerances
    dyn.gemm_mode = 0
    Lx0, _, px0, _ = propose(to_dev(g["x"]), dyn, do_mh_step=True, direction=to_dev(direction), v=to_dev(g["v"]), u=to_dev(u), aux=aux)
    assert _ffi.last_kernel() == "gemm_nt_kernel"
    assert rel_err(to_np(Lx), to_np(Lx0)) < 2e-4 and abs_err(to_np(px), to_np(px0)) < 1e-4
    # ... and so do the bf16x3 planes (gemm_mode 1), against the float64 map
    dyn.gemm_mode = 1
    Lx1, _, px1, _ = propose(to_dev(g["x"]), dyn, do_mh_step=True, direction=to_dev(direction), v=to_dev(g["v"]), u=to_dev(u), aux=aux)
    assert _ffi.last_kernel() == "gemm_xlp_kernel"
    e1x, e1p = rel_err(to_np(Lx1), tLx), abs_err(to_np(px1), tpx)
    print("          bf16x3 planes: max rel err x %.2e  |p - p64| max %.2e   (f16x2 planes above: %.2e / %.2e)" % (e1x, e1p, ex, ep))
    assert e1x < 2e-4 and e1p < 1e-4


def test_device_bf16_planes_equal_the_numpy_restatement():
    """`l2hmc_bf16_planes` (= `to_planes` of csrc/gemm_xl.hpp, what the GEMM engine turns the decoder weights into) against
    oracle/bf16x3_oracle.py bit for bit: the round-to-nearest-even three-way split, the plane order h | m | l, the zero padding
    of rows and columns -- incl. magnitudes from 1e-30 to 1e30, zeros and a ragged K (1000 -> row stride 1024)."""
    import torch
    from l2hmc_amd import _ffi
    from oracle import bf16x3_oracle as B
    rng = np.random.RandomState(7)
    rows, K, rows_pad, ldp = 300, 1000, 384, 1024
    W = (rng.randn(rows, K) * np.exp(rng.uniform(-69, 69, size=(rows, 1)))).astype(np.float32)
    W[5, :7] = 0.0
    W[6, 3] = -0.0
    dW = to_dev(W)
    P = torch.full((3, rows_pad, ldp), 0x7FFF, dtype=torch.int16, device=dW.device)          # poison: the padding must be WRITTEN
    _ffi.check(_ffi.lib().l2hmc_bf16_planes(dW.data_ptr(), K, rows, K, P.data_ptr(), rows_pad, ldp, _ffi.current_stream(dW.device)))
    got = P.cpu().numpy().view(np.uint16)
    ref = B.to_planes(W, rows_pad=rows_pad, ld=ldp)
    assert np.array_equal(got, ref), int((got != ref).sum())
    assert np.array_equal(B.from_planes(got, rows, K), W)
