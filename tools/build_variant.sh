#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: csrc/variants/libl2hmc_hip_<name>.so = the product objects with ONE translation
# unit rebuilt with the extra flags -- traj_ek1 (the diagonal-Gaussian trajectory kernels: the bench workload) by default,
# L2HMC_VARIANT_TU=split (GEMM engine) / train / ... to pick another.  Container only.
set -e
cd "$(dirname "$0")/../l2hmc_amd/csrc"
name=$1; shift
tu=${L2HMC_VARIANT_TU:-traj_ek1}
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-return-type "$@" -c -o variants/${tu}_$name.o $tu.hip
objs=$(ls *.o | grep -v "^$tu.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libl2hmc_hip_$name.so $objs variants/${tu}_$name.o
rm -f variants/${tu}_$name.o
