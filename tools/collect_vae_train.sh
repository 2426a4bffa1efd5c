#!/bin/bash
# Kernel trace of the config-5 sampler-training step (one gpurun call; everything under its own timeout):
#   gpurun --timeout 600 -- 'bash tools/collect_vae_train.sh'
# -> gpurun_out/vtr/{train8192.txt,train512.txt,kernel_stats.csv}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/vtr
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 python $R/tools/bench_vae_train.py 8192 1 2>/dev/null | tail -1 > $OUT/train8192.txt
timeout 150 python $R/tools/bench_vae_train.py 8192 5 2>/dev/null | tail -1 >> $OUT/train8192.txt
timeout 150 python $R/tools/bench_vae_train.py 512 5 2>/dev/null | tail -1 > $OUT/train512.txt
timeout 150 python $R/tools/bench_vae_train.py 512 1 2>/dev/null | tail -1 >> $OUT/train512.txt
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/tools/bench_vae_train.py 8192 1 > /dev/null 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats.csv; fi
rm -rf $OUT/trace
cat $OUT/train8192.txt $OUT/train512.txt
if [ -f $OUT/kernel_stats.csv ]; then head -25 $OUT/kernel_stats.csv | cut -c1-200; fi
