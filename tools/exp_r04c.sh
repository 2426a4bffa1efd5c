#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
T="tests/test_gpu_parity.py -q -x -m gpu -k training_gradient_matches -p no:cacheprovider"
{
echo "== in-tree"; timeout 300 python -m pytest $T 2>&1 | tail -4
echo "== in-tree again"; timeout 300 python -m pytest $T 2>&1 | tail -4
echo "== tilted8 alone"; timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "training_gradient_matches and tilted8" -p no:cacheprovider 2>&1 | tail -4
echo "== no MH block"; timeout 300 python tools/pytest_with_lib.py $V/libl2hmc_hip_tr_nomh.so $T 2>&1 | tail -4
echo "== no MH block, no x_head"; timeout 300 python tools/pytest_with_lib.py $V/libl2hmc_hip_tr_noxh.so $T 2>&1 | tail -4
} 2>&1 | tee $OUT/train_debug.txt
{
echo "== phase timing, f32 heads"; L2HMC_PT_LIB=$R/$V/libl2hmc_hip_pt.so timeout 200 python tools/phase_timing.py 4096 4 2>&1 | grep -v amdgpu
echo "== phase timing, bf16x3 pipelined heads"; L2HMC_PT_LIB=$R/$V/libl2hmc_hip_ptbfp.so timeout 200 python tools/phase_timing.py 4096 4 2>&1 | grep -v amdgpu
} 2>&1 | tee $OUT/phase.txt
ASAN_TIMEOUT=600 bash tools/collect_r04_asan.sh
