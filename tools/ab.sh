# A/B timing of library variants on the bench workload (GPU box): tools/ab.sh "<variants>" "<chain counts>"
for rep in 1 2; do
for lib in ${1:-base}; do
  for n in ${2:-4096}; do
    python tools/time_lib.py l2hmc_amd/csrc/variants/libl2hmc_hip_$lib.so $n 25 2>/dev/null
  done
done
done
