#!/bin/bash
# counters of the one-wave-per-tile kernel at 65 536 chains (where is the time of a tile-step?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --chains 65536 --steps 25 --warmup 25 --preheat 100 --no-cpu-baseline --no-ess --no-sweep --no-config5 --no-config4"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o c -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o c -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p3 -o c -- $B > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1","p2","p3"):
    fs = glob.glob("$OUT/%s/*counter_collection.csv" % d)
    if not fs: print(d, "no output"); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "traj_tile" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(d, {c: round(v / n[c], 1) for c, v in sorted(acc.items())}, "dispatches", max(n.values()) if n else 0)
PY
rm -rf $OUT/p*/*.db
