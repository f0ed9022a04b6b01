#!/bin/bash
# usage: scripts/kernel_regs.sh [pattern] — VGPRs / SGPRs / LDS / scratch of every kernel in the built library
# (from the code object's metadata notes; no GPU needed)
LIB=${PT_LIB:-pathtracer_amd/csrc/build/pt_kernels.hip.o}
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $LIB 2>/dev/null
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
$L/llvm-readelf --notes $T/k.co | python3 -c "
import sys, re
txt = sys.stdin.read()
pat = sys.argv[1] if len(sys.argv) > 1 else ''
for m in re.finditer(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    lds, name, scratch, sg, vg = m.groups()
    if pat in name: print(f'{int(vg):4d} vgpr {int(sg):4d} sgpr {int(lds):6d} lds {int(scratch):5d} scratch  {name[:150]}')
" "$1"
rm -rf $T
