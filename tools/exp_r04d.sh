#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04d
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
  for lib in dp2m0 dp4m0 dp2m1 dp4m1; do
    for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v4 /"; done
  done
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/v4 in-tree /"; done
  for n in 16384 32768 65536; do L2HMC_VARIANT=0 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/auto in-tree /"; done
done
} | tee $OUT/timing.txt
echo "== training / sharding tests (in-tree library)" | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_sharding_gloo.py tests/test_gpu_round4.py -q -x -m gpu \
   -k "training or checkpoint or adam or two_rank or round4 or bench_dist or user_energy or full_size or config4 or tempered" -p no:cacheprovider 2>&1 | tail -8 | tee -a $OUT/tests.txt
timeout 300 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu | tee $OUT/train_timing.txt
ASAN_TIMEOUT=700 bash tools/collect_r04_asan.sh
