#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: csrc/variants/libl2hmc_hip_<name>.so = the product objects with traj_ek1.o
# (the diagonal-Gaussian trajectory kernels: the bench workload) rebuilt with the extra flags.  Container only.
set -e
cd "$(dirname "$0")/../l2hmc_amd/csrc"
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-return-type "$@" -c -o variants/traj_ek1_$name.o traj_ek1.hip
objs=$(ls *.o | grep -v '^traj_ek1.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libl2hmc_hip_$name.so $objs variants/traj_ek1_$name.o
rm -f variants/traj_ek1_$name.o
