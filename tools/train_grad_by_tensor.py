#!/usr/bin/env python
"""Per-tensor distance of the training gradient from a reference-graph fixture, for an alternative build of the library
(L2HMC_DBG_LIB=path/to/lib.so) -- the diagnosis tool of DESIGN section 1 row f1 (round 5):

    python tools/train_grad_by_tensor.py train_tilted8 [train_rough6 ...]

Prints, per parameter tensor, max |got - ref| / max |ref| of that tensor and the position of the worst entry."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from l2hmc_amd import _ffi
if os.environ.get("L2HMC_DBG_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["L2HMC_DBG_LIB"])
from tests.helpers import load, hip_dynamics, to_dev
from oracle import l2hmc_oracle as O
from l2hmc_amd.training import Trainer

def run(case, variant=0):
    g = load(case)
    dyn = hip_dynamics(g); dyn.eps_override = None
    with torch.no_grad(): dyn.alpha.fill_(float(np.log(g["eps"])))
    tr = Trainer(dyn); tr.variant = variant
    draws = {"z": g["z"], "x_dir": g["x.dir"], "z_dir": g["z.dir"],
             "x_v": np.where(g["x.dir"][:, None] != 0, g["x.v_fwd"], g["x.v_bwd"]),
             "z_v": np.where(g["z.dir"][:, None] != 0, g["z.v_fwd"], g["z.v_bwd"])}
    loss, Lx, px = tr.loss_and_grad(to_dev(g["x"]), draws=draws)
    torch.cuda.synchronize()
    out = []
    for n, w in (("xnet", dyn._xw), ("vnet", dyn._vw)):
        for k in O.NET_KEYS:
            ref = g["grad.%s.%s" % (n, k)]
            got = w[k].grad.detach().cpu().numpy().reshape(ref.shape)
            e = np.abs(got - ref)
            out.append("%s.%s %.1e@%s" % (n[0], k, e.max() / max(np.abs(ref).max(), 1e-30), np.unravel_index(e.argmax(), e.shape)))
    print(case, "v%d" % variant, _ffi.last_kernel(), "loss %.6e (ref %.6e) |dLx| %.1e" % (float(loss), float(g["loss"]), np.abs(Lx.cpu().numpy() - g["Lx"]).max()))
    print("   " + "  ".join(out), flush=True)

for c in sys.argv[1:]:
    run(c)
