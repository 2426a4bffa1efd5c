#!/bin/bash
# MFMA-pipe counters of the config-5 sampler (GEMM engine):  gpurun --timeout 600 -- 'timeout 500 bash tools/collect_vae_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/vae_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o c -- python $R/tools/bench_vae.py 8192 > /dev/null 2>&1
python - <<PY > $OUT/summary.txt
import csv, glob, collections
fs = glob.glob("$OUT/p1/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("config 5 sampler (tools/bench_vae.py 8192), per-dispatch means.  GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (gui / 8 / 2.4 GHz = the kernel's duration),")
print("so  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs);  executed MFMA flops = insts x 2048")
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
    c = {m: acc[k][m] / n[k][m] for m in acc[k]}
    if "GRBM_GUI_ACTIVE" not in c or c.get("SQ_INSTS_MFMA", 0) == 0: continue
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    print("%-62s calls %4d  MFMA insts %.3e (%.2f GFLOP)  gui cycles / 8 %.3e (%.0f us)  mfma-pipe busy %.2f" % (k, n[k]["GRBM_GUI_ACTIVE"], c["SQ_INSTS_MFMA"], c["SQ_INSTS_MFMA"] * 2048e-9, c["GRBM_GUI_ACTIVE"] / 8, c["GRBM_GUI_ACTIVE"] / 8 / 2400.0, busy))
PY
cat $OUT/summary.txt
rm -rf $OUT/p1/*.db
