#!/usr/bin/env python
"""Timing-only ablations of traj_kernel (GPU box): builds variants of the library with pieces of
the step removed (results are WRONG by construction -- this only attributes time) and prints the
time per 4096-chain trajectory for each."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [("baseline", []), ("no LDS exchange", ["-DL2HMC_ABL_NOXCHG"]),
            ("no transcendentals", ["-DL2HMC_ABL_NOTRANS"]), ("no head MFMAs", ["-DL2HMC_ABL_NOHEADS"]),
            ("no xchg + no trans + no heads", ["-DL2HMC_ABL_NOXCHG", "-DL2HMC_ABL_NOTRANS", "-DL2HMC_ABL_NOHEADS"])]


def build(tag, flags):
    out = "/tmp/libl2hmc_abl_%s.so" % tag
    csrc = os.path.join(ROOT, "l2hmc_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("l2hmc_abi.hip", "traj_ek1.hip")]
    stub = "/tmp/abl_stub.hip"
    open(stub, "w").write('#include "%s/l2hmc_kernels.hpp"\nnamespace l2hmc {\n' % csrc + "".join(
        "template <> int launch_ek<%d>(int, const KArgs&, int, int, int, long long, hipStream_t) { return -2; }\n" % k
        for k in (2, 3, 4, 5)) + "}\n")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", "-Wno-return-type", "-shared", "-o", out]
                   + flags + srcs + [stub], check=True)
    return out


def main():
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    code = r'''
import sys, time; sys.path.insert(0, %r)
from l2hmc_amd import _ffi
_ffi.LIB_PATH = sys.argv[1]
import numpy as np, torch, bench
from l2hmc_amd import Dynamics, distributions, layers
from oracle import l2hmc_oracle as O
dev = torch.device("cuda", 0); n = int(sys.argv[2])
prob = bench.make_problem(0, n, dev)
dyn = Dynamics(bench.D, distributions.Gaussian(np.zeros(bench.D), np.diag(prob["var"])).get_energy_function(), T=bench.T, eps=0.1, net_factory=layers.stq_network(bench.H), device=dev)
dyn.mask = prob["mask"]; dyn.variant = 4
with torch.no_grad():
    for w, key in ((dyn._xw, "xnet"), (dyn._vw, "vnet")):
        for k in O.NET_KEYS: w[k].copy_(torch.as_tensor(prob["nets"][key][k]).reshape(w[k].shape))
x = torch.as_tensor(prob["x0"], device=dev); v = torch.randn_like(x)
for _ in range(5): dyn.run(x, v, 0, bench.T, direction_all=1, want=("x", "v", "p"))
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): dyn.run(x, v, 0, bench.T, direction_all=1, want=("x", "v", "p"))
e1.record(); torch.cuda.synchronize(); print("%%.1f" %% (e0.elapsed_time(e1) * 1e3 / 50))
''' % ROOT
    for i, (name, flags) in enumerate(VARIANTS):
        lib = build(str(i), flags)
        r = subprocess.run([sys.executable, "-c", code, lib, str(chains)], capture_output=True, text=True)
        print("%-32s %s us / trajectory" % (name, r.stdout.strip() or ("FAILED: " + r.stderr[-300:])))


if __name__ == "__main__":
    main()
