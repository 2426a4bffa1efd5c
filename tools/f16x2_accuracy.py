#!/usr/bin/env python
"""f16x2 against the f32-input MFMA, both against the float64 oracle (GPU box): direction-mixed propose on seeded draws at the
bench's sizes -- per chain max |Lx - Lx64| / max(1, |Lx64|) and |p - p64| -- for the kernels that carry f16x2 contractions
(traj_fast_kernel<., ., ., ., 1>, traj_tile_kernel) and the same geometry with the f32-input MFMA forced (variant 200 + v).
    python tools/f16x2_accuracy.py            -> the table of profiles/r06_f16x2.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from l2hmc_amd import _ffi, propose
from oracle import l2hmc_oracle as O
from tests.helpers import load, hip_dynamics, oracle_dynamics, to_dev, to_np


def big_case(case, N, seed):
    g = dict(load(case))
    rng = np.random.RandomState(seed)
    d = int(g["x_dim"])
    g["x"] = (rng.randn(N, d) * g["x"].std(axis=0, keepdims=True)).astype(np.float32)
    g["v"] = rng.randn(N, d).astype(np.float32)
    return g


def main():
    rows = [("icg50", 4096, 4, 204), ("icg50", 8192, 4, 204), ("icg50", 16384, 16, 204), ("rough50_easy", 4096, 4, 204),
            ("rough50_ne", 4096, 4, 204), ("rough8_eta01", 4096, 0, 200)]
    print("%-14s %6s | %-34s %9s %9s %9s | %9s %9s" % ("case", "chains", "kernel", "x median", "x 99.9%", "x max", "p 99.9%", "p max"))
    for case, N, var, var32 in rows:
        g = big_case(case, N, 123)
        rng = np.random.RandomState(7)
        direction = rng.randint(0, 2, size=N).astype(np.uint8)
        u = rng.rand(N).astype(np.float32)
        od64 = oracle_dynamics(g, np.float64)
        od32 = oracle_dynamics(g)
        with np.errstate(all="ignore"):
            tLx, _, tpx, _ = O.propose(g["x"].astype(np.float64), od64, g["v"].astype(np.float64), g["v"].astype(np.float64), direction,
                                       u.astype(np.float64), both_directions=False)
            rLx, _, rpx, _ = O.propose(g["x"], od32, g["v"], g["v"], direction, u, both_directions=False)
        fin = np.all(np.isfinite(tLx), axis=1) & (np.abs(tLx).max(axis=1) < 1e4)
        scale = np.maximum(1.0, np.abs(tLx).max(axis=1))

        def line(name, Lx, px):
            e = (np.abs(Lx - tLx).max(axis=1) / scale)[fin]
            ep = np.abs(px - tpx)[fin]
            print("%-14s %6d | %-34s %9.2e %9.2e %9.2e | %9.2e %9.2e" % (case, N, name, np.median(e), np.quantile(e, 0.999), e.max(),
                                                                        np.quantile(ep, 0.999), ep.max()))
        for v in (var, var32):
            dyn = hip_dynamics(g, variant=v)
            Lx, _, px, _ = propose(to_dev(g["x"]), dyn, do_mh_step=True, direction=to_dev(direction), v=to_dev(g["v"]), u=to_dev(u))
            line(_ffi.last_kernel(), to_np(Lx), to_np(px))
        line("numpy float32 oracle", rLx, rpx)


if __name__ == "__main__":
    main()
