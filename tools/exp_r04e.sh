#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04e
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
  echo "pk subs (round 3):"; L2HMC_LIB=$V/libl2hmc_hip_pksub.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "no packed f32 in the whole TU:"; L2HMC_LIB=$V/libl2hmc_hip_nopk.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "plain v_sub_f32 in the split (in-tree):"; timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
done
} | tee $OUT/vae_split_sub.txt
# the LDS-poison pass: every dynamic-LDS kernel fills its LDS with NaN patterns first; the whole GPU suite must still pass
echo "== LDS-poison build, whole GPU suite" | tee $OUT/lds_poison.txt
timeout 1500 python tools/pytest_with_lib.py $V/libl2hmc_hip_poison.so tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee -a $OUT/lds_poison.txt
