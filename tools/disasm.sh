#!/bin/bash
# disassemble one object of the library for gfx950: tools/disasm.sh pointwise -> /tmp/pointwise.s
llvm=/opt/rocm/lib/llvm/bin; o=/root/repo/p2p_bridge_amd/csrc/build/$1.o
$llvm/llvm-objcopy --dump-section .hip_fatbin=/tmp/$1.fat $o && $llvm/clang-offload-bundler --type=o --unbundle --input=/tmp/$1.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/$1.co && $llvm/llvm-objdump -d /tmp/$1.co > /tmp/$1.s && grep -c . /tmp/$1.s
