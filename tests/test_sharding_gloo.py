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
