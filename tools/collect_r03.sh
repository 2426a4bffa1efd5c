#!/bin/bash
# Round-3 evidence kept under profiles/ (run on the GPU box via gpurun): gpurun_out/r03/*
# Counter passes are separate rocprofv3 runs (--pmc never combined with trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
LEAN="--no-cpu-baseline --no-ess --no-sweep --no-config5"

timeout 500 $BENCH > $OUT/bench.json 2> $OUT/bench.err
timeout 300 $BENCH --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/bench.err
timeout 300 $BENCH --gpus 2 --one-device --backend gloo --steps 20 --warmup 5 > $OUT/bench_2rank_rehearsal.json 2>> $OUT/bench.err

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- $BENCH $LEAN > /dev/null 2>&1
cp $OUT/trace/b_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace20 -o b -- $BENCH --steps 20 --warmup 5 $LEAN > /dev/null 2>&1
cp $OUT/trace20/b_kernel_stats.csv $OUT/kernel_stats_steps20.csv 2>/dev/null

P25="$BENCH --steps 25 --warmup 25 $LEAN"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_sq -o c -- $P25 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/pmc_sq2 -o c -- $P25 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- $P25 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- $P25 > /dev/null 2>&1
python - <<PY > $OUT/pmc_summary.txt
import csv, glob, collections
print("4096-chain passes: dispatches of 25 chained proposals each (preheat 1000, --steps 25 --warmup 25)")
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
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

# config 5 (GEMM engine, bf16x3 decoder products): kernel trace + matrix-pipe counters
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vae -o v -- python $R/tools/bench_vae.py 8192 1 > $OUT/vae_run.txt 2>&1
cp $OUT/vae/v_kernel_stats.csv $OUT/vae_kernel_stats.csv 2>/dev/null
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU \
    --output-format csv -d $OUT/vae_pmc -o c -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1
python - <<PY > $OUT/vae_pmc_summary.txt
import csv, glob, collections
fs = glob.glob("$OUT/vae_pmc/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])) if fs else []:
    k = r["Kernel_Name"].split("(")[0][:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("config 5, 8192 chains, gemm_mode 1 (bf16x3): per-dispatch counter means by kernel; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES per SE...) -- see profiles/README.md")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    print(k, {c: round(v / n[k][c], 1) for c, v in sorted(acc[k].items())}, "dispatches", max(n[k].values()))
PY
rm -rf $OUT/trace $OUT/trace20 $OUT/vae $OUT/pmc_*/*.db $OUT/vae_pmc/*.db

cd $R
timeout 200 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu > $OUT/train_timing.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trk -o t -- python $R/tools/bench_train.py --no-cpu > /dev/null 2>&1; cp $OUT/trk/t_kernel_stats.csv $OUT/train_kernel_stats.csv 2>/dev/null; rm -rf $OUT/trk)
timeout 200 python tools/bench_vae_train.py 2>&1 | grep -v amdgpu > $OUT/vae_train_timing.txt
timeout 100 python tools/bench_vae.py 8192 0 2>&1 | grep -v amdgpu > $OUT/vae_modes.txt
timeout 100 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu >> $OUT/vae_modes.txt
timeout 100 python tools/bench_vae.py 512 1 2>&1 | grep -v amdgpu >> $OUT/vae_modes.txt
timeout 400 python tools/bench_configs.py 2>&1 | grep -v amdgpu > $OUT/configs.txt
timeout 300 python tools/probe_dense_wide.py 2>&1 | grep -v amdgpu > $OUT/dense_wide.txt
tail -c 1500 $OUT/bench.json; cat $OUT/pmc_summary.txt $OUT/train_timing.txt $OUT/vae_train_timing.txt $OUT/vae_modes.txt; head -5 $OUT/kernel_stats.csv
