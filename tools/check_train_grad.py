#!/usr/bin/env python
"""Training-gradient check against the committed goldens (GPU box), optionally with an alternative
build of the library (L2HMC_DBG_LIB=path/to/lib.so):

    python tools/check_train_grad.py train_scg2d:full:1 train_icg50:chunk:2 ...

case:mode:dup -- mode `full` = one launch over all chains, `chunk` = one launch per 16-chain tile
(the flat gradient is accumulated across launches); dup = replicate the chains (the gradient of the
mean loss must not change).  Prints the worst relative error over all parameter tensors."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from l2hmc_amd import _ffi
if os.environ.get("L2HMC_DBG_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["L2HMC_DBG_LIB"])
from tests.helpers import load, hip_dynamics, to_dev, to_np
from oracle import l2hmc_oracle as O
from l2hmc_amd.training import Trainer
print("imported", flush=True)

def run(case, mode, dup=1):
    g = load(case)
    dyn = hip_dynamics(g); dyn.eps_override = None
    with torch.no_grad(): dyn.alpha.fill_(float(np.log(g["eps"])))
    tr = Trainer(dyn)
    rep = lambda a: np.concatenate([a] * dup, 0)
    N = g["x"].shape[0] * dup
    xs = {"x": to_dev(rep(g["x"])), "z": to_dev(rep(g["z"]))}
    vs = {"x": to_dev(rep(np.where(g["x.dir"][:, None] != 0, g["x.v_fwd"], g["x.v_bwd"]))),
          "z": to_dev(rep(np.where(g["z.dir"][:, None] != 0, g["z.v_fwd"], g["z.v_bwd"])))}
    ds = {"x": torch.as_tensor(rep(g["x.dir"]), device="cuda").to(torch.uint8), "z": torch.as_tensor(rep(g["z.dir"]), device="cuda").to(torch.uint8)}
    tr.flat.zero_()
    step = 16 if mode == "chunk" else N
    for key in ("x", "z"):
        for lo in range(0, N, step):
            tr._propose_grad(xs[key][lo:lo+step].contiguous(), vs[key][lo:lo+step].contiguous(), ds[key][lo:lo+step].contiguous(), N)
            torch.cuda.synchronize()
    flat = tr.flat.cpu().numpy()
    off = 0
    worst = 0.0
    for n in ("xnet", "vnet"):
        for k in O.NET_KEYS:
            ref = g["grad.%s.%s" % (n, k)]
            got = flat[off:off + ref.size].reshape(ref.shape); off += ref.size
            worst = max(worst, np.abs(got - ref).max() / np.abs(ref).max())
    print(case, mode, "dup", dup, "worst rel err %.3e" % worst, " eps-grad", flat[-1] * float(g["eps"]), float(g["grad.alpha"]), flush=True)

for spec in sys.argv[1:]:
    c, m, dup = spec.split(":")
    run(c, m, int(dup))
