#!/bin/bash
# usage: asm_build.sh <src.hip> <device.s (edited)> <out.o>
set -e
LL=/opt/rocm/lib/llvm/bin
SRC=$1; S=$2; OUT=$3; T=$(mktemp -d)
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $S -o $T/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $SRC -o $OUT
rm -rf $T
