#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04g
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
  for lib in v2f32 v2bf; do
    for var in 2 4; do
      for n in 4096 8192 12288 16384; do L2HMC_VARIANT=$var timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v$var /"; done
    done
  done
done
} | tee $OUT/timing.txt
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-config5 --no-config4 2>/dev/null | grep "^{" > $OUT/bench_force_dist.json
python - <<PY
import json
o=json.load(open("$OUT/bench_force_dist.json"))
print(json.dumps(o["dist"]["sharded_training"]))
PY
