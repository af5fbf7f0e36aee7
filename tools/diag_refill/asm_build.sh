#!/bin/bash
# Build a library from a HIP source whose DEVICE code is taken from a (hand-patched) assembly file:
#   asm_build.sh src.hip dev.s out.so
# (hipcc -S --cuda-device-only src.hip -o dev.s gives the unpatched assembly.)
set -e
SRC=$1; ASM=$2; OUT=$3
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$ASM" -o $T/dev.o
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -O3 -std=c++17 -shared -fPIC -o "$OUT" "$SRC" 2>&1 | grep -v "argument unused" || true
rm -rf $T
ls -la "$OUT"
