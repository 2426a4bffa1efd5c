#!/bin/bash
# round 4, final pass on the tree with the planes GEMMs: the whole GPU suite, smoke, the default bench line, config-5 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04r
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/gpu_tests.txt
cat $OUT/gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
for rep in 1 2; do timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu; done | tee $OUT/vae.txt
timeout 200 python tools/bench_vae.py 6144 1 2>&1 | grep -v amdgpu | tee -a $OUT/vae.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1; cp $OUT/vtrace/v_kernel_stats.csv $OUT/vae_kernel_stats.csv; rm -rf $OUT/vtrace)
head -8 $OUT/vae_kernel_stats.csv | cut -c1-160
