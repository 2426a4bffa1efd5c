#!/usr/bin/env python
"""debug: one forward step of the icg50 golden with the library given by L2HMC_LIB; saves outputs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests.helpers import load, hip_dynamics, to_dev, to_np, aux_of
g = load("icg50")
dyn = hip_dynamics(g, 4)
x, v = to_dev(g["x"]), to_dev(g["v"])
s = int(g["steps"][0])
xo, vo, lj = dyn._forward_step(x, v, s, aux=aux_of(g))
xo, vo, lj = to_np(xo), to_np(vo), to_np(lj)
ex, ev = np.abs(xo - g["fstep%d.x" % s]), np.abs(vo - g["fstep%d.v" % s])
np.set_printoptions(linewidth=250, precision=2)
print("x err by dim (max over chains):\n", ex.max(0))
print("v err by dim:\n", ev.max(0))
print("x err by chain:\n", ex.max(1))
print("logdet err", np.abs(lj - g["fstep%d.logdet" % s]).max())
