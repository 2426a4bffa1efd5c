#!/bin/bash
# Round-4 experiment B (GPU box): software-pipelined bf16x3 heads (bfp / bfpt) vs base and the unpipelined form (bfh);
# the fused training step (l2hmc_train_step) -- tests and step times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
for lib in ${LIBS:-base bfh bfp bfpt}; do
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v4 /"; done
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v16 /"; done
done
done
} | tee $OUT/timing.txt
echo "== parity with bfp" | tee $OUT/parity.txt
timeout 900 python tools/pytest_with_lib.py $V/libl2hmc_hip_bfp.so tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x -m gpu \
   -k "single_steps or trajectories or propose_matches or full_size or reversibility or sample_chain or config4 or tempered or odd_shapes" 2>&1 | tail -5 | tee -a $OUT/parity.txt
echo "== training / sharding tests (in-tree library, l2hmc_train_step)" | tee -a $OUT/parity.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_sharding_gloo.py tests/test_gpu_round4.py -q -x -m gpu \
   -k "training or checkpoint or adam or two_rank or round4 or bench_dist or user_energy" 2>&1 | tail -8 | tee -a $OUT/parity.txt
timeout 300 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu | tee $OUT/train_timing.txt
