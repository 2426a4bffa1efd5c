#!/bin/bash
# round 4, last pass: config-5 / VAE tests on the final tree, then the bench evidence set (tools/collect_r04.sh PART=bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04t
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_train_split.py -q -m gpu \
   -k "config5 or vae or bf16x3 or split_engine or planes" -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/tests.txt
PART="bench" bash tools/collect_r04.sh
tail -c 300 gpurun_out/r04/bench.json
