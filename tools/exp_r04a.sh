#!/bin/bash
# Round-4 kernel experiment A (GPU box): K-packed bf16x3 heads / layer 1 against the f32-MFMA product kernels.
# Variant libraries are built in the container by tools/build_variant_full.sh (csrc/variants/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04a
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
for lib in ${LIBS:-base bfh bfh8 bfl}; do
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v4 /"; done
  L2HMC_VARIANT=2 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so 8192 25 2>/dev/null | sed "s/^/v2 /"
  for n in 16384 32768 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v16 /"; done
done
done
} | tee $OUT/timing.txt
for lib in ${PLIBS:-bfh bfl}; do
  echo "== parity with $lib" | tee -a $OUT/parity.txt
  timeout 900 python tools/pytest_with_lib.py $V/libl2hmc_hip_$lib.so tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x -m gpu \
     -k "single_steps or trajectories or propose_matches or full_size or reversibility or sample_chain or config4 or tempered or odd_shapes" 2>&1 | tail -5 | tee -a $OUT/parity.txt
done
echo "== round-4 contract tests (in-tree library)" | tee -a $OUT/parity.txt
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -q -x -m gpu -k "round4 or banana or tempered" 2>&1 | tail -8 | tee -a $OUT/parity.txt
