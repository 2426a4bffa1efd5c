#!/usr/bin/env python
"""Timing-only ablations / compiler-flag experiments of traj_kernel.  Variants are BUILT in the
build container (`python tools/ablate.py 4096 [flags] build` -> l2hmc_amd/csrc/variants/*.so) and
only TIMED on the GPU box (`python tools/ablate.py 4096 [flags]`).  Ablation variants remove pieces
of the step (results WRONG by construction -- they only attribute time)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ABLATIONS = [("baseline", []), ("no LDS exchange", ["-DL2HMC_ABL_NOXCHG"]),
             ("no transcendentals", ["-DL2HMC_ABL_NOTRANS"]), ("no head MFMAs", ["-DL2HMC_ABL_NOHEADS"]),
             ("no xchg + no trans + no heads", ["-DL2HMC_ABL_NOXCHG", "-DL2HMC_ABL_NOTRANS", "-DL2HMC_ABL_NOHEADS"])]
# compiler-level experiments (results stay correct): `python tools/ablate.py 4096 flags`
FLAG_EXPERIMENTS = [("baseline", []),
                    ("sched max-ilp", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
                    ("sched max-memory-clause", ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]),
                    ("fast-math", ["-ffast-math"]),
                    ("iglp_opt(0) in tails", ["-DL2HMC_IGLP=0"]),
                    # ("iglp_opt(1) in tails", ["-DL2HMC_IGLP=1"]),   # clang 22 (ROCm 7.2) explodes on this one
                    ("no sched groups", ["-DL2HMC_NO_SCHED_GROUPS"])]
VARIANTS = FLAG_EXPERIMENTS if (len(sys.argv) > 2 and sys.argv[2] == "flags") else ABLATIONS


VARDIR = os.path.join(ROOT, "l2hmc_amd", "csrc", "variants")     # *.so are git-ignored but travel with gpurun


def build(tag, flags):
    os.makedirs(VARDIR, exist_ok=True)
    out = os.path.join(VARDIR, "libl2hmc_abl_%s.so" % tag)
    csrc = os.path.join(ROOT, "l2hmc_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("l2hmc_abi.hip", "traj_ek1.hip", "train.hip", "split.hip")]
    stub = "/tmp/abl_stub.hip"
    open(stub, "w").write('#include "%s/l2hmc_kernels.hpp"\nnamespace l2hmc {\n' % csrc + "".join(
        "template <> int launch_ek<%d>(int, const KArgs&, int, int, int, long long, hipStream_t) { return -2; }\n" % k
        for k in (2, 3, 4, 5)) +
        "long long plan_lds_wide(KArgs&) { return 1LL << 40; }\n"
        "int launch_wide(const KArgs&, int, long long, hipStream_t) { return -2; }\n}\n")
    r = subprocess.run(["timeout", "600", "/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                        "-Wno-return-type", "-shared", "-o", out,
                        "-Wl,-rpath,/opt/rocm/lib"] + flags + srcs + [stub])
    return out if r.returncode == 0 else None


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
x = torch.as_tensor(prob["x0"], device=dev)
go = lambda: dyn.run(x, None, 0, bench.T, want=("p", "x_next"), n_proposals=25, rng={"seed": 1})
for _ in range(3): go()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8): go()
e1.record(); torch.cuda.synchronize(); print("%%.2f" %% (e0.elapsed_time(e1) * 1e3 / 8 / 25))
''' % ROOT
    mode = sys.argv[3] if len(sys.argv) > 3 else "run"
    for i, (name, flags) in enumerate(VARIANTS):
        tag = ("f%d" if VARIANTS is FLAG_EXPERIMENTS else "a%d") % i
        if mode == "build":          # in the build container (no GPU): compile only
            print("%-32s %s" % (name, "built" if build(tag, flags) else "BUILD FAILED"))
            continue
        lib = os.path.join(VARDIR, "libl2hmc_abl_%s.so" % tag)
        if not os.path.exists(lib):
            print("%-32s (not built)" % name)
            continue
        r = subprocess.run(["timeout", "120", sys.executable, "-c", code, lib, str(chains)], capture_output=True, text=True)
        print("%-32s %s us / proposal (25 per launch)" % (name, r.stdout.strip() or ("FAILED: " + r.stderr[-300:])))


if __name__ == "__main__":
    main()
