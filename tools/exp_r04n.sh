#!/bin/bash
# round 4, experiment n: the 256 x 128 GEMM with its memory instructions spread over the stages of the k-tile
mkdir -p gpurun_out/r04m
for b in spread spread_noload spread_nosgb; do
  echo "== $b" >> gpurun_out/r04m/gemm_xl_spread.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | cut -c1-330 >> gpurun_out/r04m/gemm_xl_spread.txt
done
cat gpurun_out/r04m/gemm_xl_spread.txt
