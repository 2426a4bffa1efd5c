#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04l
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2 3; do
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_halftr.so $n 25 2>/dev/null | sed "s/^/half transcendentals only    /"; done
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/half transcendentals + packed /"; done
done
} | tee $OUT/timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x -m gpu -p no:cacheprovider \
   -k "full_size or config4 or tempered or tile_kernel or sample_chain or reversibility or odd_shapes" 2>&1 | tail -4 | tee $OUT/tests.txt
