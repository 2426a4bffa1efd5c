"""GPU tests added in round 6.

1. The float64 BRACKET of the stiff Rough-Well fixtures (rough{2,50,512}_ne, rough8_eta01).  The generic tests of
   tests/test_gpu_parity.py hold these cases to 4x the float32 numpy oracle's distance from the reference's own float32
   run (`stiff_tol`) -- up to 1e-2 in the momenta at d = 512 -- which does not say on which side of the truth the kernel
   sits.  Here every output of every kernel family is held against the SAME MAP EVALUATED IN float64 with the reference's
   float32 constants (tests/helpers.py `truth_dynamics`):  |HIP - truth| <= 3 |oracle32 - truth| + the suite's base tolerance.
   (The reference's own float32 run sits at 0.2 ... 2.2x the oracle's distance from that truth: tests/test_oracle_golden.py
   `test_stiff_fixtures_sit_in_the_float64_bracket`.)
2. Training an arbitrary `net_factory` (ABI 6, `net_vjp_cb`): further down."""
import numpy as np
import pytest

from oracle import l2hmc_oracle as O
from tests.helpers import (CASES, abs_err, assert_bracket, hip_dynamics, is_stiff, load, rel_err, stiff_bracket, to_dev, to_np)
from tests.test_gpu_parity import variants

pytestmark = pytest.mark.gpu

STEP_TOL, TRAJ_TOL, P_TOL = 3e-5, 1e-4, 1e-4
STIFF_CASES = [c for c in CASES if is_stiff(load(c))]


@pytest.mark.parametrize("case", STIFF_CASES)
def test_stiff_fixtures_sit_in_the_float64_bracket_on_every_kernel_family(case):
    from l2hmc_amd import propose
    g = load(case)
    report = {}
    for var in variants(g):
        dyn = hip_dynamics(g, var)
        x, v = to_dev(g["x"]), to_dev(g["v"])
        worst = 0.0

        def chk(key, got, base):
            nonlocal worst
            e, e32 = assert_bracket(case, key, to_np(got), base, what="variant %d" % var)
            worst = max(worst, e / (3.0 * e32 + base))
        for s in g["steps"]:
            xo, vo, lj = dyn._forward_step(x, v, int(s))
            xb, vb, ljb = dyn._backward_step(x, v, int(s))
            for got, key in ((xo, "fstep%d.x"), (vo, "fstep%d.v"), (lj, "fstep%d.logdet"),
                             (xb, "bstep%d.x"), (vb, "bstep%d.v"), (ljb, "bstep%d.logdet")):
                chk(key % s, got, STEP_TOL)
        for nm, fn in (("fwd", dyn.forward), ("bwd", dyn.backward)):
            X, V, lj = fn(x, init_v=v, log_jac=True)
            p = fn(x, init_v=v)[2]
            chk(nm + ".x", X, TRAJ_TOL)
            chk(nm + ".v", V, TRAJ_TOL)
            chk(nm + ".logjac", lj, TRAJ_TOL)
            chk(nm + ".p", p, P_TOL)
        Lx, _, px, _ = propose(x, dyn, do_mh_step=True, direction=to_dev(g["prop.dir"]),
                               v=(to_dev(g["prop.v_fwd"]), to_dev(g["prop.v_bwd"])), u=to_dev(g["prop.u"]))
        chk("prop.Lx", Lx, TRAJ_TOL)
        chk("prop.px", px, P_TOL)
        report[var] = worst
    print("%s: worst (|hip - fp64| / gate) per variant: %s" % (case, {k: round(v, 2) for k, v in report.items()}))
