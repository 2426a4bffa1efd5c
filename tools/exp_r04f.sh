#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04f
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
echo "== config-5 tests with the pre-split (planes) GEMMs (in-tree)" | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_train_split.py -q -x -m gpu \
   -k "config5 or vae or bf16x3 or split_engine" -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/tests.txt
{
for rep in 1 2; do
  echo "in-loop split (round 3 form):"; L2HMC_LIB=$V/libl2hmc_hip_pksub.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "pre-split planes (in-tree):"; timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "f32 MFMA (gemm_mode 0):"; timeout 200 python tools/bench_vae.py 8192 0 2>&1 | grep -v amdgpu
done
echo "3072 chains:"; timeout 200 python tools/bench_vae.py 3072 1 2>&1 | grep -v amdgpu
} | tee $OUT/vae_planes.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1; cp $OUT/vtrace/v_kernel_stats.csv $OUT/vae_kernel_stats.csv; rm -rf $OUT/vtrace)
head -12 $OUT/vae_kernel_stats.csv | cut -c1-200
echo "== LDS-poison build: the three two-process tests again (workers now load the same build)" | tee $OUT/lds_poison_rest.txt
timeout 900 python tools/pytest_with_lib.py $V/libl2hmc_hip_poison.so tests/test_sharding_gloo.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee -a $OUT/lds_poison_rest.txt
