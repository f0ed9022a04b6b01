#!/bin/bash
# usage: scripts/kernel_isa.sh <mangled-name-substring> [out.s] — disassembly of one kernel of the built library + instruction histogram
LIB=${PT_LIB:-pathtracer_amd/csrc/build/pt_kernels.hip.o}
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB 2>/dev/null
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
SYM=$($L/llvm-readelf -s $T/k.co | grep FUNC | awk "{print \$8}" | grep -- "$1" | head -1)
echo "kernel: $(echo $SYM | c++filt | cut -c1-160)"
$L/llvm-objdump -d --mcpu=gfx950 --disassemble-symbols=$SYM $T/k.co > ${2:-$T/k.s}
grep -E "^\s+[a-z_0-9]+ " ${2:-$T/k.s} | awk '{print $1}' | sort | uniq -c | sort -rn | head -${TOP:-45}
echo "total: $(grep -cE "^\s+[a-z_0-9]+ " ${2:-$T/k.s})"
rm -rf $T
