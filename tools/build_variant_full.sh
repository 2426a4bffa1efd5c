#!/bin/bash
# build_variant_full.sh <name> <extra hipcc flags...>: csrc/variants/libl2hmc_hip_<name>.so = the WHOLE library rebuilt with the
# extra flags in a scratch copy of csrc (for switches that change a layout shared by several translation units, e.g.
# -DL2HMC_BFH=1: the LDS plan in l2hmc_abi.hip and every traj_ek*.hip must agree).  Container only.
set -e
src="$(cd "$(dirname "$0")/../l2hmc_amd/csrc" && pwd)"
name=$1; shift
root=/tmp/l2hmc_variants/$name
work=$root/l2hmc_amd/csrc
mkdir -p "$work" "$root/include" "$src/variants"
cp "$src"/../../include/*.h "$root/include/"
cp "$src"/*.hip "$src"/*.hpp "$src"/Makefile "$work"/
make -C "$work" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-return-type $*" >/dev/null
cp "$work/libl2hmc_hip.so" "$src/variants/libl2hmc_hip_$name.so"
echo "built variants/libl2hmc_hip_$name.so"
