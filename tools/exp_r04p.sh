#!/bin/bash
# round 4, experiment p: config 5 with the decoder products on pre-split planes (gemm_xlp_kernel, in-tree) vs the in-loop split
# (variants/libl2hmc_hip_noplanes.so = the same tree with -DL2HMC_NO_PLANES)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04p
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
echo "== standalone (tools/ubench_gemm_bf3.hip)" | tee $OUT/ubench.txt
timeout 120 tools/bin/ubench_gemm_bf3 2>&1 | cut -c1-330 | tee -a $OUT/ubench.txt
echo "== tests" | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -m gpu \
   -k "config5 or vae or bf16x3 or split_engine or planes" -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/tests.txt
{
for rep in 1 2; do
  echo "in-loop split:"; L2HMC_LIB=$V/libl2hmc_hip_noplanes.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "pre-split planes (in-tree):"; timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
done
echo "6144 chains, in-loop:"; L2HMC_LIB=$V/libl2hmc_hip_noplanes.so timeout 200 python tools/bench_vae.py 6144 1 2>&1 | grep -v amdgpu
echo "6144 chains, planes:"; timeout 200 python tools/bench_vae.py 6144 1 2>&1 | grep -v amdgpu
} | tee $OUT/vae_planes.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1; cp $OUT/vtrace/v_kernel_stats.csv $OUT/vae_kernel_stats.csv; rm -rf $OUT/vtrace)
head -12 $OUT/vae_kernel_stats.csv | cut -c1-200
