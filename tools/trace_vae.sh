#!/bin/bash
# Kernel traces of one config-5 proposal loop and of the sampler-training step (GPU box): rocprofv3 --kernel-trace --stats of
# tools/bench_vae.py and tools/bench_vae_train.py, the two stats tables copied to gpurun_out/$1/.
#   gpurun -- 'bash tools/trace_vae.sh r05n'
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-trace_vae}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tv_p /tmp/tv_t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tv_p -o p -- python $R/tools/bench_vae.py > $OUT/prop.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tv_t -o t -- python $R/tools/bench_vae_train.py 8192 1 ${2:-0} > $OUT/train.log 2>&1
cp /tmp/tv_p/p_kernel_stats.csv $OUT/vae_kernel_stats.csv
cp /tmp/tv_t/t_kernel_stats.csv $OUT/vae_train_kernel_stats.csv
grep "config 5" $OUT/prop.log $OUT/train.log
