#!/bin/bash
# build_from_asm.sh <name> <device.s> [hipcc flags for the host pass...]: csrc/variants/libl2hmc_hip_<name>.so whose train.hip
# DEVICE code is the given (hand-edited) assembly listing -- `hipcc -S --cuda-device-only train.hip` -- so that single
# instructions can be moved / padded without the register allocator reshuffling everything (round-5 diagnosis of the
# train_fast_kernel store that went missing).  Container only.
set -e
name=$1; asm=$(readlink -f $2); shift 2
LL=/opt/rocm/lib/llvm/bin
cd "$(dirname "$0")/../l2hmc_amd/csrc"
mkdir -p variants
T=$(mktemp -d)
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $asm -o $T/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
   -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-return-type "$@" --cuda-host-only \
   -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c -o $T/train_asm.o train.hip
objs=$(ls *.o | grep -v "^train.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libl2hmc_hip_$name.so $objs $T/train_asm.o
rm -rf $T
echo variants/libl2hmc_hip_$name.so
