cd $GRAFT_REPO_ROOT
timeout 600 bash tools/collect_profiles.sh r02 > gpurun_out/r02_collect.log 2>&1
timeout 200 python tools/bench_configs.py > gpurun_out/r02/configs.txt 2>&1
timeout 100 python tools/bench_vae.py 8192 >> gpurun_out/r02/configs.txt 2>&1
timeout 200 python tools/bf16_state_study.py > gpurun_out/r02/bf16_state_study.txt 2>&1
cd /tmp && export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/vae -o v -- python $GRAFT_REPO_ROOT/tools/bench_vae.py 8192 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; cp gpurun_out/r02/vae/*kernel_stats.csv gpurun_out/r02/vae_kernel_stats.csv; rm -rf gpurun_out/r02/vae gpurun_out/r02/trace gpurun_out/r02/pmc_*/*.db
tail -30 gpurun_out/r02_collect.log; cat gpurun_out/r02/configs.txt gpurun_out/r02/bf16_state_study.txt
timeout 100 python tools/bench_ais.py > gpurun_out/r02/ais_timing.txt 2>&1; cat gpurun_out/r02/ais_timing.txt
# GEMM-engine timings / traces (config 5 sampling and sampler training), the lane-kernel table, training timings
timeout 600 bash tools/collect_r02b.sh > gpurun_out/collect_r02b.log 2>&1
mkdir -p gpurun_out/r02b
timeout 300 python tools/bench_lane.py 2>&1 | grep -v amdgpu > gpurun_out/r02b/lane.txt
timeout 120 python tools/bench_train.py --no-cpu 2>&1 | grep -v amdgpu > gpurun_out/r02b/train_timing.txt
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r02b/bench_s20.json 2>/dev/null
