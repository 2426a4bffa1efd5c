#!/bin/bash
# experiments_r04.sh <letter>: the round-4 kernel experiments (GPU box), one function per experiment -- exactly what was run;
# results: profiles/r04_* (index: tools/README.md "Round-4 experiment records").  The variant libraries / ubench binaries they name
# were builds of the tree at the time with the -D switches stated in each header (tools/build_variant*.sh); switches of refused
# experiments are gone from the sources, the commits are named in DESIGN.md.  (Round 4 kept these as 21 files exp_r04[a-u].sh;
# bodies are unindented so that their here-documents stay intact.)

exp_a() {
# Round-4 kernel experiment A (GPU box): K-packed bf16x3 heads / layer 1 against the f32-MFMA product kernels.
# Variant libraries are built in the container by tools/build_variant_full.sh (csrc/variants/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04a
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
for lib in ${LIBS:-base bfh bfh8 bfl}; do
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v4 /"; done
  L2HMC_VARIANT=2 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so 8192 25 2>/dev/null | sed "s/^/v2 /"
  for n in 16384 32768 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v16 /"; done
done
done
} | tee $OUT/timing.txt
for lib in ${PLIBS:-bfh bfl}; do
  echo "== parity with $lib" | tee -a $OUT/parity.txt
  timeout 900 python tools/pytest_with_lib.py $V/libl2hmc_hip_$lib.so tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x -m gpu \
     -k "single_steps or trajectories or propose_matches or full_size or reversibility or sample_chain or config4 or tempered or odd_shapes" 2>&1 | tail -5 | tee -a $OUT/parity.txt
done
echo "== round-4 contract tests (in-tree library)" | tee -a $OUT/parity.txt
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -q -x -m gpu -k "round4 or banana or tempered" 2>&1 | tail -8 | tee -a $OUT/parity.txt
}

exp_b() {
# Round-4 experiment B (GPU box): software-pipelined bf16x3 heads (bfp / bfpt) vs base and the unpipelined form (bfh);
# the fused training step (l2hmc_train_step) -- tests and step times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
for lib in ${LIBS:-base bfh bfp bfpt}; do
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v4 /"; done
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v16 /"; done
done
done
} | tee $OUT/timing.txt
echo "== parity with bfp" | tee $OUT/parity.txt
timeout 900 python tools/pytest_with_lib.py $V/libl2hmc_hip_bfp.so tests/test_gpu_parity.py tests/test_gpu_round3.py -q -x -m gpu \
   -k "single_steps or trajectories or propose_matches or full_size or reversibility or sample_chain or config4 or tempered or odd_shapes" 2>&1 | tail -5 | tee -a $OUT/parity.txt
echo "== training / sharding tests (in-tree library, l2hmc_train_step)" | tee -a $OUT/parity.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_sharding_gloo.py tests/test_gpu_round4.py -q -x -m gpu \
   -k "training or checkpoint or adam or two_rank or round4 or bench_dist or user_energy" 2>&1 | tail -8 | tee -a $OUT/parity.txt
timeout 300 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu | tee $OUT/train_timing.txt
}

exp_c() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
T="tests/test_gpu_parity.py -q -x -m gpu -k training_gradient_matches -p no:cacheprovider"
{
echo "== in-tree"; timeout 300 python -m pytest $T 2>&1 | tail -4
echo "== in-tree again"; timeout 300 python -m pytest $T 2>&1 | tail -4
echo "== tilted8 alone"; timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "training_gradient_matches and tilted8" -p no:cacheprovider 2>&1 | tail -4
echo "== no MH block"; timeout 300 python tools/pytest_with_lib.py $V/libl2hmc_hip_tr_nomh.so $T 2>&1 | tail -4
echo "== no MH block, no x_head"; timeout 300 python tools/pytest_with_lib.py $V/libl2hmc_hip_tr_noxh.so $T 2>&1 | tail -4
} 2>&1 | tee $OUT/train_debug.txt
{
echo "== phase timing, f32 heads"; L2HMC_PT_LIB=$R/$V/libl2hmc_hip_pt.so timeout 200 python tools/phase_timing.py 4096 4 2>&1 | grep -v amdgpu
echo "== phase timing, bf16x3 pipelined heads"; L2HMC_PT_LIB=$R/$V/libl2hmc_hip_ptbfp.so timeout 200 python tools/phase_timing.py 4096 4 2>&1 | grep -v amdgpu
} 2>&1 | tee $OUT/phase.txt
ASAN_TIMEOUT=600 bash tools/collect_r04_asan.sh
}

exp_d() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04d
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
  for lib in dp2m0 dp4m0 dp2m1 dp4m1; do
    for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_$lib.so $n 25 2>/dev/null | sed "s/^/v4 /"; done
  done
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/v4 in-tree /"; done
  for n in 16384 32768 65536; do L2HMC_VARIANT=0 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/auto in-tree /"; done
done
} | tee $OUT/timing.txt
echo "== training / sharding tests (in-tree library)" | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_sharding_gloo.py tests/test_gpu_round4.py -q -x -m gpu \
   -k "training or checkpoint or adam or two_rank or round4 or bench_dist or user_energy or full_size or config4 or tempered" -p no:cacheprovider 2>&1 | tail -8 | tee -a $OUT/tests.txt
timeout 300 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu | tee $OUT/train_timing.txt
ASAN_TIMEOUT=700 bash tools/collect_r04_asan.sh
}

exp_e() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04e
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2; do
  echo "pk subs (round 3):"; L2HMC_LIB=$V/libl2hmc_hip_pksub.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "no packed f32 in the whole TU:"; L2HMC_LIB=$V/libl2hmc_hip_nopk.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "plain v_sub_f32 in the split (in-tree):"; timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
done
} | tee $OUT/vae_split_sub.txt
# the LDS-poison pass: every dynamic-LDS kernel fills its LDS with NaN patterns first; the whole GPU suite must still pass
echo "== LDS-poison build, whole GPU suite" | tee $OUT/lds_poison.txt
timeout 1500 python tools/pytest_with_lib.py $V/libl2hmc_hip_poison.so tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee -a $OUT/lds_poison.txt
}

exp_f() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04f
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
echo "== config-5 tests with the pre-split (planes) GEMMs (in-tree)" | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_train_split.py -q -x -m gpu \
   -k "config5 or vae or bf16x3 or split_engine" -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/tests.txt
{
for rep in 1 2; do
  echo "in-loop split (round 3 form):"; L2HMC_LIB=$V/libl2hmc_hip_pksub.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "pre-split planes (in-tree):"; timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "f32 MFMA (gemm_mode 0):"; timeout 200 python tools/bench_vae.py 8192 0 2>&1 | grep -v amdgpu
done
echo "3072 chains:"; timeout 200 python tools/bench_vae.py 3072 1 2>&1 | grep -v amdgpu
} | tee $OUT/vae_planes.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1; cp $OUT/vtrace/v_kernel_stats.csv $OUT/vae_kernel_stats.csv; rm -rf $OUT/vtrace)
head -12 $OUT/vae_kernel_stats.csv | cut -c1-200
echo "== LDS-poison build: the three two-process tests again (workers now load the same build)" | tee $OUT/lds_poison_rest.txt
timeout 900 python tools/pytest_with_lib.py $V/libl2hmc_hip_poison.so tests/test_sharding_gloo.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | tee -a $OUT/lds_poison_rest.txt
}

exp_g() {
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
}

exp_h() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04h
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2 3; do
  for n in 16384 32768 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_tnp.so $n 25 2>/dev/null | sed "s/^/no-prefetch /"; done
  for n in 16384 32768 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/prefetch    /"; done
done
} | tee $OUT/timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x -m gpu -p no:cacheprovider \
   -k "full_size or config4 or tempered or tile_kernel or sample_chain or reversibility" 2>&1 | tail -6 | tee $OUT/tests.txt
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-config5 --no-config4 2>/dev/null | grep "^{" > $OUT/bench_force_dist.json
python - <<PY
import json
o=json.load(open("$OUT/bench_force_dist.json"))
print(json.dumps(o["dist"]["sharded_training"]), o["dist"]["backend"])
PY
}

exp_i() {
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
}

exp_j() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04j
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2 3; do
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_nohalf.so $n 25 2>/dev/null | sed "s/^/all four components /"; done
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/two live components  /"; done
done
} | tee $OUT/timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x -m gpu -p no:cacheprovider \
   -k "full_size or config4 or tempered or tile_kernel or sample_chain or reversibility or odd_shapes" 2>&1 | tail -6 | tee $OUT/tests.txt
}

exp_k() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04k
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2 3; do
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_libm.so $n 25 2>/dev/null | sed "s/^/libm normals     /"; done
  for n in 16384 65536; do L2HMC_VARIANT=0 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_libm.so $n 25 2>/dev/null | sed "s/^/libm normals     /"; done
  for n in 4096 8192; do L2HMC_VARIANT=4 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/hardware normals /"; done
  for n in 16384 65536; do L2HMC_VARIANT=0 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/hardware normals /"; done
done
} | tee $OUT/timing.txt
python - <<'PY' | tee $OUT/normals.txt
import numpy as np, sys
sys.path.insert(0, ".")
from l2hmc_amd.sampler import philox_draws
from oracle import l2hmc_oracle as O
v, dr, u = philox_draws(99, 4096, 50, 8)
rv, rd, ru = O.philox_draws(99, 4096, 50, 8)
e = np.abs(v.cpu().numpy() - rv)
print("hardware Box-Muller vs numpy over %d normals: max |diff| %.3e, 99.99%% %.3e, mean %.3e; mean %.5f var %.5f" % (e.size, e.max(), np.quantile(e, 0.9999), e.mean(), float(v.mean()), float(v.var())))
PY
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.txt
}

exp_l() {
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04l
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
{
for rep in 1 2 3; do
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py $V/libl2hmc_hip_halftr.so $n 25 2>/dev/null | sed "s/^/half transcendentals only    /"; done
  for n in 16384 65536; do L2HMC_VARIANT=16 timeout 120 python tools/time_lib.py l2hmc_amd/csrc/libl2hmc_hip.so $n 25 2>/dev/null | sed "s/^/half transcendentals + packed /"; done
done
} | tee $OUT/timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -x -m gpu -p no:cacheprovider \
   -k "full_size or config4 or tempered or tile_kernel or sample_chain or reversibility or odd_shapes" 2>&1 | tail -4 | tee $OUT/tests.txt
}

exp_m() {
# round 4, experiment m: the 256 x 128 one-wave-per-SIMD bf16x3 GEMM (csrc/gemm_xl.hpp) against the 128 x 128 form, standalone
# (tools/ubench_gemm_bf3.hip, -DL2HMC_XL_TIMING: shader cycles of the k loop), with the timing ablations of gemm_f32.hpp
# (no split VALU / no MFMA / neither = staging only; wrong numbers by construction) and without packed-f32 VALU
mkdir -p gpurun_out/r04m
for b in base nosplit nomfma neither nopk; do
  echo "== $b" >> gpurun_out/r04m/gemm_xl_ablate.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | grep -v "^M=8192 N=784\|^M=8192 N=1024 K=784" | cut -c1-330 >> gpurun_out/r04m/gemm_xl_ablate.txt
done
cat gpurun_out/r04m/gemm_xl_ablate.txt
}

exp_n() {
# round 4, experiment n: the 256 x 128 GEMM with its memory instructions spread over the stages of the k-tile
mkdir -p gpurun_out/r04m
for b in spread spread_noload spread_nosgb; do
  echo "== $b" >> gpurun_out/r04m/gemm_xl_spread.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | cut -c1-330 >> gpurun_out/r04m/gemm_xl_spread.txt
done
cat gpurun_out/r04m/gemm_xl_spread.txt
}

exp_o() {
# round 4, experiment o: the planes GEMM on 256 x 128 tiles -- 4 waves (one per SIMD, 128 x 64 blocks) vs 8 waves (two per SIMD, 64 x 64)
mkdir -p gpurun_out/r04m
for b in w4 w8; do
  echo "== $b" >> gpurun_out/r04m/gemm_xlp_waves.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | grep "^M=\|planes" | cut -c1-260 >> gpurun_out/r04m/gemm_xlp_waves.txt
done
cat gpurun_out/r04m/gemm_xlp_waves.txt
}

exp_p() {
# round 4, experiment p: config 5 with the decoder products on pre-split planes (gemm_xlp_kernel, in-tree) vs the in-loop split
# (variants/libl2hmc_hip_noplanes.so = the same tree with -DL2HMC_NO_PLANES)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04p
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
echo "== standalone (tools/ubench_gemm_bf3.hip)" | tee $OUT/ubench.txt
timeout 120 tools/bin/ubench_gemm_bf3 2>&1 | cut -c1-330 | tee -a $OUT/ubench.txt
echo "== tests" | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py -q -x -m gpu \
   -k "config5 or vae or bf16x3 or split_engine or planes" -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/tests.txt
{
for rep in 1 2; do
  echo "in-loop split:"; L2HMC_LIB=$V/libl2hmc_hip_noplanes.so timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
  echo "pre-split planes (in-tree):"; timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu
done
echo "6144 chains, in-loop:"; L2HMC_LIB=$V/libl2hmc_hip_noplanes.so timeout 200 python tools/bench_vae.py 6144 1 2>&1 | grep -v amdgpu
echo "6144 chains, planes:"; timeout 200 python tools/bench_vae.py 6144 1 2>&1 | grep -v amdgpu
} | tee $OUT/vae_planes.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1; cp $OUT/vtrace/v_kernel_stats.csv $OUT/vae_kernel_stats.csv; rm -rf $OUT/vtrace)
head -12 $OUT/vae_kernel_stats.csv | cut -c1-200
}

exp_q() {
# round 4, experiment q: gemm_xlp_kernel -- the second wave of every SIMD stages in the last stages of the k-tile (XLP_DEPHASE)
mkdir -p gpurun_out/r04q
for b in s3 s3_dephase s4_dephase s2_dephase; do
  echo "== $b" >> gpurun_out/r04q/xlp_dephase.txt
  timeout 120 tools/bin/ubx_$b 2>&1 | grep "planes" | cut -c1-200 >> gpurun_out/r04q/xlp_dephase.txt
done
cat gpurun_out/r04q/xlp_dephase.txt
}

exp_r() {
# round 4, final pass on the tree with the planes GEMMs: the whole GPU suite, smoke, the default bench line, config-5 kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04r
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/gpu_tests.txt
cat $OUT/gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
for rep in 1 2; do timeout 200 python tools/bench_vae.py 8192 1 2>&1 | grep -v amdgpu; done | tee $OUT/vae.txt
timeout 200 python tools/bench_vae.py 6144 1 2>&1 | grep -v amdgpu | tee -a $OUT/vae.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vtrace -o v -- python $R/tools/bench_vae.py 8192 1 > /dev/null 2>&1; cp $OUT/vtrace/v_kernel_stats.csv $OUT/vae_kernel_stats.csv; rm -rf $OUT/vtrace)
head -8 $OUT/vae_kernel_stats.csv | cut -c1-160
}

exp_s() {
# round 4, experiment s: from which chain count do the planes GEMMs pay?  (variants: noplanes = in-loop split everywhere; tiles84 = planes from 84 tiles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04s
mkdir -p $OUT
cd $R
V=l2hmc_amd/csrc/variants
for n in 3072 4096 5120; do
  echo "$n chains, in-loop:"; L2HMC_LIB=$V/libl2hmc_hip_noplanes.so timeout 200 python tools/bench_vae.py $n 1 2>&1 | grep -v amdgpu
  echo "$n chains, planes:"; L2HMC_LIB=$V/libl2hmc_hip_tiles84.so timeout 200 python tools/bench_vae.py $n 1 2>&1 | grep -v amdgpu
done | tee $OUT/threshold.txt
}

exp_t() {
# round 4, last pass: config-5 / VAE tests on the final tree, then the bench evidence set (tools/collect_r04.sh PART=bench)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04t
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_train_split.py -q -m gpu \
   -k "config5 or vae or bf16x3 or split_engine or planes" -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/tests.txt
PART="bench" bash tools/collect_r04.sh
tail -c 300 gpurun_out/r04/bench.json
}

exp_u() {
# round 4, experiment u: gemm_xlp_kernel -- MFMAs one stage after their fragments (o1) or in the same stage (o0), LDS-only barrier
# (rawbar), ONE barrier in the middle of the k-tile with the next tile's first fragments requested behind it (midbar)
mkdir -p gpurun_out/r04u
for b in o1a4 o0a4 o0a3 o0a2 rawbar midbar; do
  echo "== $b" >> gpurun_out/r04u/xlp_boundary.txt
  timeout 60 tools/bin/ubx_$b 2>&1 | grep "planes" | head -2 | cut -c1-220 >> gpurun_out/r04u/xlp_boundary.txt
done
cat gpurun_out/r04u/xlp_boundary.txt
}

case "$1" in
  a|b|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s|t|u) exp_$1 ;;
  *) echo "usage: $0 {a,b,c,d,e,f,g,h,i,j,k,l,m,n,o,p,q,r,s,t,u}" >&2; exit 2 ;;
esac
