#!/bin/bash
# build_asan.sh: csrc/variants/libl2hmc_hip_asan.so = the whole library with AddressSanitizer on host AND device code
# (gfx950:xnack+; run with HSA_XNACK=1 and LD_PRELOAD of clang's libclang_rt.asan-x86_64.so: tools/collect_r04_asan.sh).
set -e
src="$(cd "$(dirname "$0")/../l2hmc_amd/csrc" && pwd)"
root=/tmp/l2hmc_variants/asan
work=$root/l2hmc_amd/csrc
mkdir -p "$work" "$root/include" "$src/variants"
cp "$src"/../../include/*.h "$root/include/"
cp "$src"/*.hip "$src"/*.hpp "$src"/Makefile "$work"/
make -C "$work" -j8 HIPCC=/opt/rocm/bin/hipcc ARCH=gfx950:xnack+ \
  CXXFLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -Wno-unused-function -Wno-return-type" > $root/build.log 2>&1 || { tail -30 $root/build.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -shared -fPIC -o "$src/variants/libl2hmc_hip_asan.so" "$work"/*.o
echo built
