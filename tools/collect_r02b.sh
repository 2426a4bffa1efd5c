#!/bin/bash
# Second collection of round 2 (GEMM-engine changes only; the fused-kernel profiles of tools/collect_r02.sh stay valid):
#   gpurun --timeout 900 -- 'timeout 850 bash tools/collect_r02b.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r02b
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 100 python tools/bench_vae.py 8192 2>/dev/null | tail -1 > $OUT/vae.txt
timeout 100 python tools/bench_vae.py 512 2>/dev/null | tail -1 >> $OUT/vae.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 > /dev/null 2>&1
f=$(find $OUT/vtrace -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/vae_kernel_stats.csv; fi
rm -rf $OUT/vtrace
cd $R
timeout 400 bash tools/collect_vae_train.sh > /dev/null 2>&1
cp $R/gpurun_out/vtr/*.txt $R/gpurun_out/vtr/kernel_stats.csv $OUT/ 2>/dev/null
timeout 200 python examples/vae_sampler_training.py --steps 200 2>&1 | grep -v amdgpu.ids > $OUT/vae_train_example.txt
ls -la $OUT; cat $OUT/vae.txt $OUT/train8192.txt $OUT/train512.txt | cut -c1-160; tail -3 $OUT/vae_train_example.txt
