"""CPU, world_size 2 over gloo: the N>1 path -- contiguous chain shards, no data-path
collective, one all-reduce for accept-rate / autocovariance statistics -- reproduces the
single-process numbers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from l2hmc_amd import func_utils, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, X, p, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = X.shape[1]
        lo, hi = sharding.shard_range(n)
        acl = sharding.acl_spectrum(X[:, lo:hi], 1.7, n)
        ma = sharding.mean_accept(torch.as_tensor(p[lo:hi]))
        es = sharding.ess(X[:, lo:hi], 1.7, n)
        if rank == 0:
            out.put((acl, ma, es, (lo, hi)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    for n, w in ((4096, 8), (65536, 8), (10, 3), (7, 8), (0, 2)):
        spans = [sharding.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_two_rank_statistics_match_single_process():
    rng = np.random.RandomState(0)
    steps, n, d = 40, 37, 3                     # odd chain count: ragged shards
    X = np.cumsum(rng.randn(steps, n, d) * 0.3, axis=0) * 0.2 + rng.randn(1, n, d)
    p = rng.rand(n).astype(np.float32)
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, X, p, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(100)
        assert pr.exitcode == 0
    acl, ma, es, span = out.get()
    ref = func_utils.acl_spectrum(X, 1.7)
    assert span == (0, 19)
    assert np.allclose(acl, ref, rtol=1e-12, atol=1e-12)
    assert abs(ma - float(p.astype(np.float64).mean())) < 1e-12
    assert abs(es - func_utils.ESS(ref)) < 1e-12


def test_single_process_fallthrough():
    X = np.random.RandomState(1).randn(10, 4, 2)
    assert np.allclose(sharding.acl_spectrum(X, 1.0, 4), func_utils.acl_spectrum(X, 1.0))


def _layout_worker(rank, world, port, out):
    """The host logic of `Trainer._shard` / `_allreduce_flat` on CPU tensors.  The layout (global chain count, this
    rank's offset) is exchanged ONCE -- by every rank, on its first step -- and a step then issues exactly ONE collective:
    the all-reduce of [gradient | loss sums | count].  A rank whose local count changes afterwards raises instead of
    re-entering an exchange no other rank takes part in; `set_sharding(None, None)` on EVERY rank re-opens the exchange."""
    import types
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(int(t.numel())), real(t, *a, **k))[1]
    try:
        from l2hmc_amd.training import Trainer
        n_grad = 7
        tr = types.SimpleNamespace(dyn=types.SimpleNamespace(device=torch.device("cpu")), _layout=None, _auto_layout=None,
                                   _world=lambda: world, n_grad=n_grad, N_TAIL=Trainer.N_TAIL, _stale=None, _reduced=None,
                                   _STALE_MSG=Trainer._STALE_MSG)
        tr._check_reduced_count = lambda: Trainer._check_reduced_count(tr)
        tr._flat_ext = torch.zeros(n_grad + Trainer.N_TAIL)
        tr.flat = tr._flat_ext[:n_grad]
        seen = []

        def step(n_local):
            n_total, off = Trainer._shard(tr, n_local)
            tr.flat.fill_(float(rank + 1))                          # "the rank's gradient"
            # (hi parts 1e9 + 64 and 1e9 + 128: their float32 sum is NOT exact -- the reduced sums are float32-accurate)
            sums = torch.tensor([1e9 + 64.0 * (rank + 1) + 0.125 * (rank + 1), 3.0 * (rank + 1)], dtype=torch.float64)
            red, cnt = Trainer._allreduce_flat(tr, sums, n_local)
            Trainer._note_reduced_count(tr, tr._flat_ext[n_grad + 4:n_grad + 6], n_total, hi_scale=4096.0)
            return n_total, off, float(tr.flat[0]), float(red[0]), float(red[1]), float(cnt)

        n0 = 501 if rank == 0 else 500                              # ragged shards
        seen.append(step(n0))                                        # layout exchange + the step's collective
        seen.append(step(n0))                                        # the step's collective ONLY
        seen.append(tuple(calls))
        # only rank 0's count changes: it must NOT re-exchange on its own and must NOT raise before the step's collective (the
        # other rank would wait in it); it raises once the collective is behind it, rank 1 at its next `_shard` from the count
        raised = []
        step(500 if rank == 0 else n0)
        try:
            Trainer._raise_if_stale(tr)
            raised.append(False)
        except RuntimeError:
            raised.append(True)
        try:
            Trainer._shard(tr, 500 if rank == 0 else n0)
            raised.append(False)
        except RuntimeError:
            raised.append(True)
        seen.append(tuple(raised))
        Trainer.set_sharding(tr, None, None)                        # every rank re-opens the exchange ...
        seen.append(step(500))                                      # ... and all see the new layout
        del calls[:]
        Trainer.set_sharding(tr, 1234, 617 * rank)                  # declared layout: no layout collective at all
        seen.append(Trainer._shard(tr, 617))
        seen.append(tuple(calls))
        out.put((rank, seen))
        dist.barrier()
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_shard_layout_is_exchanged_once_and_a_step_is_one_collective():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_layout_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(100)
        assert pr.exitcode == 0
    res = dict(out.get() for _ in range(2))
    big = 2e9 + 192.375          # the double sums travel as float (hi, lo) pairs and are ADDED in float32: 2^-23 of the sum
    for rank, off in ((0, 0), (1, 501)):
        first, second, calls, raised, third, declared, calls2 = res[rank]
        for got in (first, second):
            assert got[:3] == (1001, off, 3.0) and got[4:] == (9.0, 1001.0)
            assert abs(got[3] - big) <= 2.0 ** -23 * big and got[3] != big
        assert calls == (2, 7 + 6, 7 + 6)                           # layout exchange once, then ONE all-reduce per step
        # rank 0 (whose count changed) raises after the collective; BOTH ranks at their next step, from the reduced count
        # (1000 != the cached 1001) -- nobody is left waiting in a collective
        assert raised == ((True, True) if rank == 0 else (False, True))
        assert third[:3] == (1000, 500 * rank, 3.0) and third[4:] == (9.0, 1000.0)
        assert declared == (1234, 617 * rank) and calls2 == ()


def _train_worker(rank, world, port, out):
    """One rank of a 2-process training step on the SAME GPU (gloo all-reduce of the flat gradient):
    exercises `Trainer`'s world_size > 1 branch end to end against the reference's full-batch gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import l2hmc_oracle as O
        from l2hmc_amd.training import Trainer
        from tests.helpers import hip_dynamics, load, to_dev, to_np
        g = load("train_tilted8")
        dyn = hip_dynamics(g)
        dyn.eps_override = None
        with torch.no_grad():
            dyn.alpha.fill_(float(np.log(g["eps"])))
        tr = Trainer(dyn)
        N = g["x"].shape[0]
        lo, hi = sharding.shard_range(N)
        pick = lambda d_, f, b: np.where(d_[:, None] != 0, f, b)[lo:hi]
        draws = {"z": g["z"][lo:hi], "x_dir": g["x.dir"][lo:hi], "z_dir": g["z.dir"][lo:hi],
                 "x_v": pick(g["x.dir"], g["x.v_fwd"], g["x.v_bwd"]), "z_v": pick(g["z.dir"], g["z.v_fwd"], g["z.v_bwd"])}
        loss, _, _ = tr.loss_and_grad(to_dev(g["x"][lo:hi]), draws=draws)
        from tests.helpers import check_grads_per_tensor, fixture_grads, net_grads
        worst = check_grads_per_tensor("rank %d" % rank, net_grads(dyn), fixture_grads(g))[0] * 2e-4      # per tensor (round 6)
        scale = 1.0
        out.put((rank, float(loss), float(g["loss"]), worst / scale, float(dyn.alpha.grad), float(g["grad.alpha"])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_rank_training_step_matches_the_full_batch_gradient():
    """Trainer with chains sharded over 2 processes (both on cuda:0, gloo): every rank ends up with the
    reference's full-batch loss and gradient after the ONE flat all-reduce."""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(280)
        assert pr.exitcode == 0
    res = sorted(out.get() for _ in range(2))
    for rank, loss, ref_loss, rel, ga, ref_ga in res:
        assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss)), (rank, loss, ref_loss)
        assert rel < 2e-4, (rank, rel)
        assert abs(ga - ref_ga) < 2e-4 * max(1.0, abs(ref_ga)), (rank, ga, ref_ga)


def _step_worker(rank, world, port, out):
    """Three whole optimiser steps (`Trainer.step`) with ragged shards on the SAME GPU: counts the collectives and
    compares the sharded run with the same steps on the full batch in one process."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(int(t.numel())), real(t, *a, **k))[1]
    try:
        from l2hmc_amd.training import Trainer
        from tests.helpers import hip_dynamics, load, to_dev
        g = load("train_tilted8")
        N = 37                                                       # 19 + 18 chains
        x_all = np.tile(g["x"], (8, 1))[:N].copy()

        def fresh():
            dyn = hip_dynamics(g)
            dyn.eps_override = None
            with torch.no_grad():
                dyn.alpha.fill_(float(np.log(g["eps"])))
            return Trainer(dyn, seed=5)
        tr = fresh()
        lo, hi = sharding.shard_range(N)
        x = to_dev(x_all[lo:hi])
        losses = []
        for _ in range(3):
            loss, px, x, lr = tr.step(x)
            losses.append(float(loss))
        n_calls = tuple(calls)
        ref = fresh()
        ref._world = lambda: 1                                       # the same three steps on the full batch, one process
        xr = to_dev(x_all)
        ref_losses = []
        for _ in range(3):
            loss, px, xr, lr = ref.step(xr)
            ref_losses.append(float(loss))
        out.put((rank, n_calls, tr.n_grad, losses, ref_losses, float((tr.theta - ref.theta).abs().max()),
                 float(tr.theta.double().sum()), float((x - xr[lo:hi]).abs().max())))
        dist.barrier()
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_rank_optimiser_steps_issue_one_collective_each():
    """north_star: RCCL 'only to all-reduce the training-loss gradient': after the one-time layout exchange a sharded
    `Trainer.step` is exactly ONE all-reduce (gradient, loss sums and chain count in one buffer), and three such steps
    on ragged shards reproduce the single-process steps on the full batch (same Philox draws per global chain)."""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(280)
        assert pr.exitcode == 0
    res = sorted(out.get() for _ in range(2))
    for rank, n_calls, n_grad, losses, ref_losses, dtheta, tsum, dx in res:
        assert n_calls == (2, n_grad + 6, n_grad + 6, n_grad + 6), (rank, n_calls)
        assert np.allclose(losses, ref_losses, rtol=1e-5, atol=1e-6), (losses, ref_losses)
        assert dtheta < 2e-5 and dx < 1e-4, (rank, dtheta, dx)
    assert res[0][6] == res[1][6]                                    # identical parameters on both ranks


def _vae_train_worker(rank, world, port, out):
    """One rank of a 2-process VAE sampler update on the SAME GPU: chains (and their images) sharded, the flat gradient
    [XNet | VNet | d/d eps | encoder_sampler] of the GEMM-engine trainer all-reduced once (gloo), then clipped Adam."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import l2hmc_oracle as O
        from l2hmc_amd.training import SplitTrainer, Trainer
        from tests.helpers import hip_dynamics, load, to_dev, to_np
        g = load("train_vae_small")
        dyn = hip_dynamics(g)
        dyn.eps_override = None
        with torch.no_grad():
            dyn.alpha.fill_(float(np.log(g["eps"])))
        tr = Trainer(dyn, decay_steps=0)
        assert isinstance(tr, SplitTrainer)
        N = g["x"].shape[0]
        lo, hi = sharding.shard_range(N)
        dr = {"v": np.where(g["prop.dir"][:, None] != 0, g["prop.v_fwd"], g["prop.v_bwd"])[lo:hi],
              "dir": g["prop.dir"][lo:hi], "u": g["prop.u"][lo:hi]}
        loss, xT, px = tr.sampler_loss_and_grad(to_dev(g["x"][lo:hi]), to_dev(g["aux"][lo:hi]), to_dev(g["log_sigma"][lo:hi]),
                                                MH=1, draws=[dr])
        from tests.helpers import check_grads_per_tensor, fixture_grads, net_grads
        enc = dyn._xw["aux_encoder"]
        got = net_grads(dyn, extra={"enc." + k: enc[k] for k in ("W1", "b1", "W2", "b2", "W3", "b3")})
        worst = check_grads_per_tensor("rank %d" % rank, got, fixture_grads(g))[0] * 2e-4                 # per tensor (round 6)
        scale = 1.0
        tr._adam(tr.lr_at(0))                       # every rank applies the same update to its replica
        out.put((rank, float(loss), float(g["loss"]), worst / scale, float(dyn.alpha.grad), float(g["grad.alpha"]),
                 float(tr.theta.double().sum()), float(tr.theta.double().abs().sum())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_rank_vae_sampler_update_matches_the_full_batch_gradient():
    """the GEMM-engine trainer with chains sharded 16 + 16 over 2 processes: every rank ends up with the reference graph's
    full-batch sampler loss and gradient (incl. the image branch), and with identical parameters after the update"""
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_vae_train_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(280)
        assert pr.exitcode == 0
    res = sorted(out.get() for _ in range(2))
    for rank, loss, ref_loss, rel, ga, ref_ga, s1, s2 in res:
        assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss)), (rank, loss, ref_loss)
        assert rel < 2e-4, (rank, rel)
        assert abs(ga - ref_ga) < 2e-4 * max(1.0, abs(ref_ga)), (rank, ga, ref_ga)
    assert res[0][6] == res[1][6] and res[0][7] == res[1][7]


def _ragged_split_worker(rank, world, port, out):
    """SplitTrainer's two sharded entry points (`sampler_step`, `step`) when ONE rank's chain count changes under a
    discovered layout (a ragged last mini-batch): that rank must pass the step's collective and raise at the END of its
    call; the other rank learns of it from the reduced chain count and raises at the START of its next call, before any
    collective -- nobody waits in an all-reduce the other never enters (training.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from l2hmc_amd.training import SplitTrainer, Trainer
        from tests.helpers import hip_dynamics, load, to_dev
        seen = {}
        for name, case in (("sampler", "train_vae_small"), ("step", "train_tilted8_h24")):
            g = load(case)
            dyn = hip_dynamics(g)
            dyn.eps_override = None
            with torch.no_grad():
                dyn.alpha.fill_(float(np.log(g["eps"])))
            tr = Trainer(dyn, decay_steps=0) if name == "sampler" else Trainer(dyn)
            assert isinstance(tr, SplitTrainer)
            N = g["x"].shape[0]
            lo, hi = sharding.shard_range(N)

            def call(hi_):
                if name == "sampler":
                    return tr.sampler_step(to_dev(g["x"][lo:hi_]), to_dev(g["aux"][lo:hi_]), to_dev(g["log_sigma"][lo:hi_]), MH=1)
                return tr.step(to_dev(g["x"][lo:hi_]))
            call(hi)                                   # layout exchange + a healthy step
            call(hi)
            events = []
            try:                                       # rank 0 arrives one chain short; rank 1 as before
                call(hi - 1 if rank == 0 else hi)
                events.append("passed")
            except RuntimeError as e:
                events.append("stale" if "chain count changed" in str(e) else "other: %s" % e)
            try:                                       # the next call: both ranks raise before any collective (the reduced count)
                call(hi - 1 if rank == 0 else hi)
                events.append("passed")
            except RuntimeError as e:
                events.append("count" if "counted" in str(e) else ("stale" if "chain count changed" in str(e) else "other: %s" % e))
            tr.set_sharding(None, None)                # every rank re-opens the exchange: healthy again on the new layout
            call(hi - 1 if rank == 0 else hi)
            events.append("recovered")
            seen[name] = tuple(events)
        out.put((rank, seen))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_rank_gemm_engine_trainer_survives_a_ragged_shard():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_split_worker, args=(r, 2, port, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(280)
        assert pr.exitcode == 0
    res = dict(out.get() for _ in range(2))
    for name in ("sampler", "step"):
        assert res[0][name] == ("stale", "count", "recovered"), (name, res[0][name])
        assert res[1][name] == ("passed", "count", "recovered"), (name, res[1][name])
