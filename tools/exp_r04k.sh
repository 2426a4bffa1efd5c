#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04k
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2 3; do
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_libm.so $n 25 2>/dev/null | sed "s/^/libm normals     /"; done
  for n in 16384 65536; do L2HMC_VARIANT=0 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_libm.so $n 25 2>/dev/null | sed "s/^/libm normals     /"; done
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/hardware normals /"; done
  for n in 16384 65536; do L2HMC_VARIANT=0 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/hardware normals /"; done
done
} | tee $OUT/timing.txt
python - <<'PY' | tee $OUT/normals.txt
import numpy as np, sys
sys.path.insert(0, ".")
from l2hmc_amd.sampler import philox_draws
from oracle import l2hmc_oracle as O
v, dr, u = philox_draws(99, 4096, 50, 8)
rv, rd, ru = O.philox_draws(99, 4096, 50, 8)
e = np.abs(v.cpu().numpy() - rv)
print("hardware Box-Muller vs numpy over %d normals: max |diff| %.3e, 99.99%% %.3e, mean %.3e; mean %.5f var %.5f" % (e.size, e.max(), np.quantile(e, 0.9999), e.mean(), float(v.mean()), float(v.var())))
PY
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.txt
