#!/bin/bash
# Collect the evidence kept under profiles/ (run on the GPU box via gpurun):
#   tools/collect_profiles.sh r01      -> gpurun_out/r01/*
# Counter passes are separate rocprofv3 runs (--pmc never combined with trace domains).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"

timeout 400 $BENCH > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- $BENCH --no-cpu-baseline --no-ess --no-sweep --no-config5 > /dev/null 2>&1
cp $OUT/trace/b_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
head -3 $OUT/kernel_stats.csv

timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_sq -o c -- $BENCH --steps 25 --warmup 25 --no-cpu-baseline --no-ess --no-sweep --no-config5 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/pmc_sq2 -o c -- $BENCH --steps 25 --warmup 25 --no-cpu-baseline --no-ess --no-sweep --no-config5 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- $BENCH --steps 25 --warmup 25 --no-cpu-baseline --no-ess --no-sweep --no-config5 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- $BENCH --steps 25 --warmup 25 --no-cpu-baseline --no-ess --no-sweep --no-config5 > /dev/null 2>&1
# the many-chains regime (262144 chains, 10 proposals per launch): what saturates there
BIG="$BENCH --chains 262144 --proposals-per-launch 10 --steps 10 --warmup 10 --no-cpu-baseline --no-ess --no-sweep --no-config5"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_big -o c -- $BIG > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/pmc_big2 -o c -- $BIG > /dev/null 2>&1
python - <<PY > $OUT/pmc_summary.txt
import csv, glob, collections
print("4096-chain passes: ten dispatches of 25 chained proposals each (preheat 1000, --steps 25 --warmup 25); pmc_big*: 262144 chains, two dispatches of 10 proposals")
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_big", "pmc_big2"):
    fs = glob.glob("$OUT/%s/*counter_collection.csv" % d)
    if not fs:
        print(d, "no output"); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "traj_" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(d, "traj_*_kernel per-dispatch means:", {c: round(v / n[c], 1) for c, v in sorted(acc.items())})
PY
cat $OUT/pmc_summary.txt
rm -rf $OUT/trace/*.db $OUT/pmc_*/*.db

cd $R
timeout 300 python tools/phase_timing.py 4096 4 > $OUT/phase_timing.txt 2>&1; tail -15 $OUT/phase_timing.txt
timeout 300 python tools/bench_train.py > $OUT/train_timing.txt 2>&1; cat $OUT/train_timing.txt
( timeout 100 python tools/train_phase_timing.py scg2d; timeout 100 python tools/train_phase_timing.py icg50 ) > $OUT/train_phase_timing.txt 2>&1
tools/sweep.sh "4096 4 25" "4096 4 1" "4096 1 25" "8192 4 25" "16384 4 25" "65536 4 25" "65536 1 25" "262144 4 10" "262144 1 10" "1048576 4 5" > $OUT/sweep.txt 2>&1; cat $OUT/sweep.txt
