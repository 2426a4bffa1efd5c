#!/bin/bash
# AddressSanitizer pass over the LDS-heavy kernels (SURVEY section 5's sanitizer row; GPU box).  The library is built by
# tools/build_asan.sh in the container (host + device instrumentation, gfx950:xnack+).  Log -> gpurun_out/r04/asan.txt.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
cd $R
export HSA_XNACK=1
ASAN_LIB=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_LIBRARY_PATH=$(dirname $ASAN_LIB):${LD_LIBRARY_PATH:-}
{
echo "# rocminfo xnack:"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -i -m3 "xnack\|gfx950"
echo "# 1. probe (tools/bin/asan_probe: a kernel writes one float past a 64-float hipMalloc): ASan must report it"
ASAN_OPTIONS=detect_leaks=0 timeout 120 tools/bin/asan_probe 2>&1 | head -30
echo "probe exit code: ${PIPESTATUS[0]}"
echo "# 2. parity tests on the instrumented library (LD_PRELOAD=$ASAN_LIB)"
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 timeout ${ASAN_TIMEOUT:-900} \
  python tools/pytest_with_lib.py l2hmc_amd/csrc/variants/libl2hmc_hip_asan.so tests/test_gpu_parity.py -q -x -m gpu \
  -k "${ASAN_K:-single_steps or trajectories or training_gradient}" -p no:cacheprovider > $OUT/asan_pytest.log 2>&1
echo "pytest exit code: $?"
grep -n "ERROR: AddressSanitizer\|SUMMARY: AddressSanitizer\|passed\|failed" $OUT/asan_pytest.log | head -20
head -c 3000 $OUT/asan_pytest.log; echo; echo ...; tail -c 3000 $OUT/asan_pytest.log
} > $OUT/asan.txt 2>&1
tail -50 $OUT/asan.txt
