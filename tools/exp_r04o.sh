#!/bin/bash
# round 4, experiment o: the planes GEMM on 256 x 128 tiles -- 4 waves (one per SIMD, 128 x 64 blocks) vs 8 waves (two per SIMD, 64 x 64)
mkdir -p gpurun_out/r04m
for b in w4 w8; do
  echo "== $b" >> gpurun_out/r04m/gemm_xlp_waves.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | grep "^M=\|planes" | cut -c1-260 >> gpurun_out/r04m/gemm_xlp_waves.txt
done
cat gpurun_out/r04m/gemm_xlp_waves.txt
