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
from tests.helpers import (abs_err, check_x_next, hip_dynamics, oracle_dynamics, rel_err, synthetic_case, to_dev, to_np)
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
    assert _ffi.last_kernel() == "traj_fast_kernel<1, 1, 4, 3>"
    two = synthetic_case("gauss_dense", 2, H=10, T=5, N=64, seed=3)
    propose(to_dev(two["x"]), hip_dynamics(two, 0), direction=to_dev(dr[:64]), v=to_dev(two["v"]))
    assert _ffi.last_kernel().startswith("traj_small_kernel<2")
