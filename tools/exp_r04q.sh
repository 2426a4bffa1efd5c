#!/bin/bash
# round 4, experiment q: gemm_xlp_kernel -- the second wave of every SIMD stages in the last stages of the k-tile (XLP_DEPHASE)
mkdir -p gpurun_out/r04q
for b in s3 s3_dephase s4_dephase s2_dephase; do
  echo "== $b" >> gpurun_out/r04q/xlp_dephase.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | grep "planes" | cut -c1-200 >> gpurun_out/r04q/xlp_dephase.txt
done
cat gpurun_out/r04q/xlp_dephase.txt
