#!/bin/bash
# round 4, experiment u: gemm_xlp_kernel -- MFMAs one stage after their fragments (o1) or in the same stage (o0), LDS-only barrier
# (rawbar), ONE barrier in the middle of the k-tile with the next tile's first fragments requested behind it (midbar)
mkdir -p gpurun_out/r04u
for b in o1a4 o0a4 o0a3 o0a2 rawbar midbar; do
  echo "== $b" >> gpurun_out/r04u/xlp_boundary.txt
  timeout 60 tools/bin/ubx_$b 2>&1 | grep "planes" | head -2 | cut -c1-220 >> gpurun_out/r04u/xlp_boundary.txt
done
cat gpurun_out/r04u/xlp_boundary.txt
