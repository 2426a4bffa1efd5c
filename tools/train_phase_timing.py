#!/usr/bin/env python
"""Per-phase cycle breakdown of the training kernel (block 0).  `build` (in the container) makes a
profiling copy of the library with -DL2HMC_TRAIN_TIMING under csrc/variants/; on the GPU box
`python tools/train_phase_timing.py scg2d|icg50 [chains]` runs one x- plus one z-proposal."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "l2hmc_amd", "csrc")
OUT = os.path.join(CSRC, "variants", "libl2hmc_hip_tt.so")
KINDS = ["stage", "fwd L1 partials", "fwd bias+relu", "fwd L2", "fwd heads", "bwd heads/dW/part", "bwd relu'",
         "bwd W4 / d1", "bwd W1,W2 / da,db", "elementwise passes", "other (energy, seeds)"]


def main():
    if sys.argv[1] == "build":
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        import glob
        srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                        "-DL2HMC_TRAIN_TIMING", "-Wno-return-type", "-Wno-pass-failed", "-shared", "-o", OUT] + srcs
                       + [], check=True)
        return
    from l2hmc_amd import _ffi
    _ffi.LIB_PATH = OUT
    import torch
    from tools.bench_train import make
    from l2hmc_amd.training import Trainer
    case = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    dyn, x, _ = make(case, n, dev)
    tr = Trainer(dyn)
    v = torch.randn_like(x)
    dr = torch.randint(0, 2, (n,), device=dev, dtype=torch.uint8)
    L = _ffi.lib()
    buf = (ctypes.c_ulonglong * 16)()
    for _ in range(2):
        tr._propose_grad(x, v, dr, n)
    L.l2hmc_train_read_timers(buf)
    reps = 5
    for _ in range(reps):
        tr._propose_grad(x, v, dr, n)
    L.l2hmc_train_read_timers(buf)
    tot = sum(buf[i] for i in range(11))
    print("%s, %d chains: s_memtime ticks per launch (one proposal of 16 chains), wave 0 of block 0" % (case, n))
    kinds = KINDS
    if case in ("scg2d", "mog2d"):       # train_small_kernel (d <= 4): its own phases
        kinds = ["stage", "forward trajectory", "reductions / seeds", "rev: net re-evaluation (+ grad U)", "rev: half-update adjoint",
                 "rev: net back-propagation (critical path)", "rev: operand transposes (LDS)", "rev: weight-gradient MFMAs",
                 "rev: step head / hessvec / hand-over", "-", "-"]
    if case == "icg50":                  # train_fast_kernel (register-resident, 4 waves per tile): coarse phases
        kinds = ["stage", "forward trajectory", "reductions / seeds", "rev: checkpoints, grad U, hessvec, hand-over", "flush",
                 "rev: net re-evaluation", "rev: net back-propagation (adjoints, cross-wave sum)", "rev: weight gradients (transposes + MFMAs)",
                 "rev: half-update adjoint", "-", "-"]
    for i, k in enumerate(kinds):
        print("  %-24s %10.0f  %5.1f%%" % (k, buf[i] / reps, 100.0 * buf[i] / tot))
    print("  %-24s %10.0f" % ("total", tot / reps))


if __name__ == "__main__":
    main()
