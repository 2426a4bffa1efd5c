#!/usr/bin/env python
"""Per-phase cycle breakdown of one traj_kernel workgroup (GPU box).  Builds a PROFILING copy of
the library with -DL2HMC_PHASE_TIMING (`phase_timing.py build`, in the container; the product .so is
untouched) under csrc/variants/, runs the bench
problem once and prints s_memtime cycles per phase per leapfrog step for each wave of block 0."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["prologue", "step head", "VNet tail #1", "XNet L1 (a,b)", "xchg", "XNet tail #1",
          "XNet L1 (b)", "xchg", "XNet tail #2", "grad+VNet L1", "xchg", "VNet tail #2"]


def main():
    csrc = os.path.join(ROOT, "l2hmc_amd", "csrc")
    out = os.environ.get("L2HMC_PT_LIB", os.path.join(csrc, "variants", "libl2hmc_hip_pt.so"))
    if len(sys.argv) > 1 and sys.argv[1] == "build":      # run this in the container, NOT on the GPU box
        os.makedirs(os.path.dirname(out), exist_ok=True)
        import glob
        srcs = sorted(glob.glob(os.path.join(csrc, "*.hip")))
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                        "-DL2HMC_PHASE_TIMING", "-Wno-return-type", "-shared", "-o", out] + srcs
                       + [], check=True)
        return
    if not os.path.exists(out):
        raise SystemExit("build the profiling library first: python tools/phase_timing.py build")
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from l2hmc_amd import _ffi
    _ffi.LIB_PATH = out
    import numpy as np
    import torch
    import bench
    from l2hmc_amd import Dynamics, distributions, layers
    from oracle import l2hmc_oracle as O
    dev = torch.device("cuda", 0)
    prob = bench.make_problem(0, chains, dev)
    dyn = Dynamics(bench.D, distributions.Gaussian(np.zeros(bench.D), np.diag(prob["var"])).get_energy_function(),
                   T=bench.T, eps=0.1, net_factory=layers.stq_network(bench.H), device=dev)
    dyn.mask = prob["mask"]
    dyn.variant = variant
    with torch.no_grad():
        for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
            for k in O.NET_KEYS:
                w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
    x = torch.as_tensor(prob["x0"], device=dev)
    v = torch.randn_like(x)
    dbg = torch.zeros(64, dtype=torch.int64, device=dev)
    L = _ffi.lib()
    L.l2hmc_set_debug_buffer.argtypes = [ctypes.c_void_p]
    L.l2hmc_set_debug_buffer(dbg.data_ptr())
    for _ in range(3):
        dyn.run(x, v, 0, bench.T, direction_all=1, want=("x", "v", "p"))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(4, 16)[:, :12]
    print("cycles (s_memtime ticks) per phase, chains=%d variant=%d; per leapfrog step except prologue" % (chains, variant))
    for i, name in enumerate(PHASES):
        div = 1 if i == 0 else bench.T
        print("  %-16s " % name + "  ".join("w%d %7.0f" % (w, d[w, i] / div) for w in range({4: 4, 2: 2}.get(variant, 1))))
    print("  %-16s " % "sum/step" + "  ".join("w%d %7.0f" % (w, d[w, 1:].sum() / bench.T) for w in range({4: 4, 2: 2}.get(variant, 1))))


if __name__ == "__main__":
    main()
