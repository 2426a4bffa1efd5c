#!/bin/bash
# round 4, experiment m: the 256 x 128 one-wave-per-SIMD bf16x3 GEMM (csrc/gemm_xl.hpp) against the 128 x 128 form, standalone
# (tools/ubench_gemm_bf3.hip, -DL2HMC_XL_TIMING: shader cycles of the k loop), with the timing ablations of gemm_f32.hpp
# (no split VALU / no MFMA / neither = staging only; wrong numbers by construction) and without packed-f32 VALU
mkdir -p gpurun_out/r04m
for b in base nosplit nomfma neither nopk; do
  echo "== $b" >> gpurun_out/r04m/gemm_xl_ablate.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | grep -v "^M=8192 N=784\|^M=8192 N=1024 K=784" | cut -c1-330 >> gpurun_out/r04m/gemm_xl_ablate.txt
done
cat gpurun_out/r04m/gemm_xl_ablate.txt
