#!/usr/bin/env python
"""Dump the per-workgroup gradient slots of one l2hmc_train_propose_grad launch (the x half of a training fixture) for an
alternative build (L2HMC_DBG_LIB), or compare two dumps:
    python tools/train_slots_dump.py dump train_tilted8 out.npz
    python tools/train_slots_dump.py cmp a.npz b.npz"""
import os, sys, numpy as np
sys.path.insert(0, '.')
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    names = list(a["names"]); offs = list(a["offs"]) + [int(a["P"])]
    for blk in range(a["slots"].shape[0]):
        sa, sb = a["slots"][blk], b["slots"][blk]
        bad = np.nonzero(sa != sb)[0]
        print("block", blk, "differing entries:", len(bad))
        for i in bad[:40]:
            net, j = ("x", i) if i < int(a["P"]) else ("v", i - int(a["P"]))
            if i == 2 * int(a["P"]): print("   eps", sa[i], sb[i]); continue
            k = max(t for t in range(len(names)) if offs[t] <= j)
            print("   %s.%s[%d]  %.8e  vs  %.8e" % (net, names[k], j - offs[k], sa[i], sb[i]))
    sys.exit(0)
import torch
from l2hmc_amd import _ffi
if os.environ.get("L2HMC_DBG_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["L2HMC_DBG_LIB"])
from tests.helpers import load, hip_dynamics, to_dev
from l2hmc_amd.training import Trainer
case, out = sys.argv[2], sys.argv[3]
g = load(case)
dyn = hip_dynamics(g); dyn.eps_override = None
with torch.no_grad(): dyn.alpha.fill_(float(np.log(g["eps"])))
tr = Trainer(dyn)
x = to_dev(g["x"]); v = to_dev(np.where(g["x.dir"][:, None] != 0, g["x.v_fwd"], g["x.v_bwd"]))
dr = torch.as_tensor(g["x.dir"], device="cuda").to(torch.uint8)
N, d = x.shape; T, H = int(g["T"]), int(g["H"])
need = _ffi.check(_ffi.lib().l2hmc_train_workspace_floats(N, d, H, T))
tr._ws = torch.full((int(need),), float("nan"), dtype=torch.float32, device="cuda")      # a slot entry nobody writes stays NaN
tr._propose_grad(x, v, dr, N)
torch.cuda.synchronize()
blocks = (N + 15) // 16
P = (tr.flat.numel() - 1) // 2
NW = 1 if d <= 16 else 4
base = blocks * T * 13 * NW * 256
ws = tr._ws.cpu().numpy()
slots = np.stack([ws[base + b * (2 * P + 1): base + (b + 1) * (2 * P + 1)] for b in range(blocks)])
# flat layout of one net (train.hip net_off): W1 b1 W2 b2 W3 b3 W4 b4 Ws bs Wt bt Wq bq lam_s lam_q
names = ["W1", "b1", "W2", "b2", "W3", "b3", "W4", "b4", "Ws", "bs", "Wt", "bt", "Wq", "bq", "lam_s", "lam_q"]
sizes = [d * H, H, d * H, H, 2 * H, H, H * H, H, H * d, d, H * d, d, H * d, d, d, d]
offs = np.cumsum([0] + sizes[:-1])
assert sum(sizes) == P, (sum(sizes), P)
np.savez(out, slots=slots, flat=tr.flat.cpu().numpy(), names=np.array(names), offs=offs, P=P, kernel=_ffi.last_kernel())
print(case, _ffi.last_kernel(), "blocks", blocks, "P", P)
