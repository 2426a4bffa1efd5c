#!/bin/bash
# round 4, experiment s: from which chain count do the planes GEMMs pay?  (variants: noplanes = in-loop split everywhere; tiles84 = planes from 84 tiles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04s
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
for n in 3072 4096 5120; do
  echo "$n chains, in-loop:"; L2HMC_LIB=$V/libl2hmc_hip_noplanes.so timeout 200 python tools/bench_vae.py $n 1 2>&1 | grep -v amdgpu
  echo "$n chains, planes:"; L2HMC_LIB=$V/libl2hmc_hip_tiles84.so timeout 200 python tools/bench_vae.py $n 1 2>&1 | grep -v amdgpu
done | tee $OUT/threshold.txt
