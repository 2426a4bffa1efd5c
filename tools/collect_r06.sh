#!/bin/bash
# Round-6 evidence kept under profiles/ (run on the GPU box via gpurun): gpurun_out/r06/*  (collect_r05.sh with the round's names;
# new: traffic.json also carries the matrix pipe's busy share of the bench workload -- bench.py's roofline.mfma_pipe_busy -- and the
# config-5 summary prices a bf16 16x16x32 MFMA at 16 384 flop;
# the 20-step kernel trace of round 4 is gone: it averaged calibration launches with the timed ones -- VERDICT r04 -- the roofline
# reading uses the 25-proposal trace; PART=vae: the matrix-pipe counters of config 5's kernels incl. net_eval_kernel)
# Counter passes are separate rocprofv3 runs (--pmc never combined with trace domains).
#   PART=bench    bench lines + kernel trace + PMC of the bench workload (-> traffic.json from the SAME pass as the summary)
#   PART=config4  row J: PMC + kernel trace of the config-4 kernels (Rough Well, 16 384 chains: d = 2 / 50 / 512), easy and non-easy
#   PART=rest     training / VAE / config tables
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"
LEAN="--no-cpu-baseline --no-ess --no-sweep --no-config5 --no-config4"
PART=${PART:-bench config4 rest vae}

summarise() {   # summarise <outfile> <title> <dirs...>: per-dispatch counter means of the traj_* kernels
python - "$@" <<'PY'
import csv, glob, collections, sys
out, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
with open(out, "a") as f:
    f.write(title + "\n")
    for d in dirs:
        fs = glob.glob(d + "/*counter_collection.csv")
        if not fs:
            f.write("%s: no output\n" % d); continue
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"].split("(")[0]
            if "traj_" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        for k in acc:
            f.write("%s %s per-dispatch means over %d dispatches: %s\n" % (d.split("/")[-1], k.replace("void l2hmc::", ""), max(n[k].values()),
                    {c: round(v / n[k][c], 1) for c, v in sorted(acc[k].items())}))
PY
}

if [[ " $PART " == *" bench "* ]]; then
timeout 600 $BENCH > $OUT/bench.json 2> $OUT/bench.err
timeout 300 $BENCH --steps 20 --warmup 5 > $OUT/bench_steps20.json 2>> $OUT/bench.err
timeout 300 $BENCH --gpus 2 --one-device --backend gloo --steps 20 --warmup 5 > $OUT/bench_2rank_rehearsal.json 2>> $OUT/bench.err
timeout 300 $BENCH --force-dist --steps 20 --warmup 5 $LEAN --no-ess > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- $BENCH $LEAN > /dev/null 2>&1
cp $OUT/trace/b_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sweep -o b -- $BENCH --no-cpu-baseline --no-ess --no-config5 --no-config4 > /dev/null 2>&1
cp $OUT/trace_sweep/b_kernel_stats.csv $OUT/kernel_stats_with_sweep.csv 2>/dev/null
P25="$BENCH --steps 25 --warmup 25 $LEAN"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_sq -o c -- $P25 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT/pmc_sq2 -o c -- $P25 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c -- $P25 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c -- $P25 > /dev/null 2>&1
rm -f $OUT/pmc_summary.txt
summarise $OUT/pmc_summary.txt "4096-chain passes: dispatches of 25 chained proposals each (preheat 1000, --steps 25 --warmup 25)" $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_fetch $OUT/pmc_write
# traffic.json from THIS pass (bench.py reads it for roofline.traffic / hbm_frac)
python - <<PY
import csv, glob, json
def mean(d, name):
    fs = glob.glob("$OUT/%s/*counter_collection.csv" % d)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "traj_" in r["Kernel_Name"] and r["Counter_Name"] == name] if fs else []
    return sum(v) / len(v) if v else None
f, w = mean("pmc_fetch", "FETCH_SIZE"), mean("pmc_write", "WRITE_SIZE")
mb, wc = mean("pmc_sq2", "SQ_VALU_MFMA_BUSY_CYCLES"), mean("pmc_sq", "SQ_WAVE_CYCLES")
if f is not None and w is not None:
    json.dump({"mfma_pipe_busy": (mb / (4.0 * wc)) if (mb and wc) else None,
               "mfma_pipe_busy_source": "profiles/r06_pmc_summary.txt: SQ_VALU_MFMA_BUSY_CYCLES (cycles) / (4 x SQ_WAVE_CYCLES (quad-cycles)) of "
                                        "traj_fast_kernel<1,1,4,3,1> (f16x2 contractions), one wave per SIMD, per-dispatch means of two separate --pmc passes",
               "source": "profiles/r06_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of tools/collect_r06.sh, "
                         "per-dispatch means of the trajectory kernel; written by the same script run as that summary)",
               "workload_chains": 4096, "proposals_per_launch": 25, "fetch_kb": round(f, 1), "write_kb": round(w, 1),
               "note": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch; all dispatches of the PMC run are 25-proposal launches (preheat 1000, "
                       "--steps 25 --warmup 25). 8-byte-per-lane state loads (d = 50 rows are 8-byte aligned), so the gfx950 x2 FETCH "
                       "correction for 16-byte streaming reads is not applied (uncalibrated for this width); algorithmic bytes per launch: "
                       "x in 0.78 MiB + x_next 0.78 MiB + 25 x p 0.39 MiB"}, open("$OUT/traffic.json", "w"), indent=1)
PY
rm -rf $OUT/trace $OUT/trace20 $OUT/trace_sweep $OUT/pmc_*/*.db
tail -c 1200 $OUT/bench.json; cat $OUT/pmc_summary.txt; head -4 $OUT/kernel_stats.csv
fi

if [[ " $PART " == *" config4 "* ]]; then
# ---- row J: config 4 (Rough Well, 16 384 chains), one kernel family per d: traj_small (d = 2), traj_tile (d = 50), traj_wide (d = 512)
rm -f $OUT/config4_pmc.txt
for spec in "2 easy" "50 easy" "512 easy" "2 noneasy" "50 noneasy" "512 noneasy"; do
  set -- $spec; d=$1; kind=$2; tag=${d}_${kind}
  C4="python $R/tools/bench_configs.py one $d $kind"
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4t_$tag -o t -- $C4 > $OUT/c4_run_$tag.txt 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4f_$tag -o c -- $C4 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c4w_$tag -o c -- $C4 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c4s_$tag -o c -- $C4 > /dev/null 2>&1
  grep "C4 RoughWell" $OUT/c4_run_$tag.txt >> $OUT/config4_pmc.txt
  grep "traj_" $OUT/c4t_$tag/t_kernel_stats.csv | head -2 >> $OUT/config4_pmc.txt
  summarise $OUT/config4_pmc.txt "counters, d = $d ($kind), 16 384 chains, 10 proposals per launch (pilot launches of the step-size search included in the means)" $OUT/c4f_$tag $OUT/c4w_$tag $OUT/c4s_$tag
  rm -rf $OUT/c4t_$tag $OUT/c4f_$tag/*.db $OUT/c4w_$tag/*.db $OUT/c4s_$tag/*.db
done
cat $OUT/config4_pmc.txt
fi

if [[ " $PART " == *" rest "* ]]; then
cd $R
timeout 200 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu > $OUT/train_timing.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trk -o t -- python $R/tools/bench_train.py --no-cpu > /dev/null 2>&1; cp $OUT/trk/t_kernel_stats.csv $OUT/train_kernel_stats.csv 2>/dev/null; rm -rf $OUT/trk)
timeout 200 python tools/bench_vae_train.py 2>&1 | grep -v amdgpu > $OUT/vae_train_timing.txt
(for m in 1 3; do timeout 100 python tools/bench_vae.py 8192 $m 2>&1; done) | grep -v amdgpu > $OUT/vae_modes.txt
timeout 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu > $OUT/configs.txt
cat $OUT/train_timing.txt $OUT/vae_train_timing.txt $OUT/vae_modes.txt $OUT/configs.txt
fi

if [[ " $PART " == *" vae "* ]]; then
# ---- config 5: matrix-pipe counters per kernel (net_eval_kernel was not re-measured in round 4)
cd /tmp
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/vp1 -o c -- python $R/tools/bench_vae.py 8192 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtr -o t -- python $R/tools/bench_vae.py 8192 > /dev/null 2>&1
cp $OUT/vtr/t_kernel_stats.csv $OUT/vae_kernel_stats.csv 2>/dev/null
python - <<PY > $OUT/vae_pmc_summary.txt
import csv, glob, collections
fs = glob.glob("$OUT/vp1/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("config 5 sampler (tools/bench_vae.py 8192), per-dispatch means.  GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (gui / 8 = the kernel's cycles),")
print("so  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs);  f32 MFMA = 2048 flop, bf16 / f16 16x16x32 = 16384 flop per instruction (gemm_xlp_kernel<..., 1>: f16x2 planes, three products)")
for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0)):
    c = {m: acc[k][m] / n[k][m] for m in acc[k]}
    if "GRBM_GUI_ACTIVE" not in c or c.get("SQ_INSTS_MFMA", 0) == 0: continue
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    bf16 = ("gemm_xlp" in k) or ("gemm_nt_kernel<" in k and k.rstrip().endswith(", 1>"))   # bf16 16x16x32 MFMAs (16 384 flop); the rest f32 16x16x4 (2048)
    print("%-62s calls %4d  MFMA insts %.3e (%s: %.2f GFLOP executed)  gui cycles / 8 %.3e  mfma-pipe busy %.2f" % (
        k, n[k]["GRBM_GUI_ACTIVE"], c["SQ_INSTS_MFMA"], "bf16/f16 16x16x32" if bf16 else "f32 16x16x4", c["SQ_INSTS_MFMA"] * (16384 if bf16 else 2048) / 1e9,
        c["GRBM_GUI_ACTIVE"] / 8, busy))
PY
cat $OUT/vae_pmc_summary.txt
rm -rf $OUT/vp1 $OUT/vtr
fi
