#!/usr/bin/env python3
"""Instruction mix of one kernel: static (disassembly of the built code object, by category) beside the dynamic
per-wave mix the SQ counters give (profiles/pmc.json, scripts/pmc_digest.py).

    python scripts/isa_mix.py <mangled-name-substring> <pmc workload key> <kernel kind in pmc.json> [digest.txt instantiation]
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "pathtracer_amd", "csrc", "build", "pt_kernels.hip.o")

CATS = [
    ("f64 arithmetic (add / mul / fma / fmac / rcp / rsq / sqrt / div helpers / ldexp / frexp)",
     r"v_(add|mul|fma|fmac|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp_mant|frexp_exp_i32|trig_preop|fract|rndne|floor|trunc|ceil)_f64"),
    ("f64 compares / min / max / class", r"v_(cmp\w*|min|max|cmp_class)_f64|v_cmp_class_f64"),
    ("f32 box tests (fma / mul / min / max / min3 / max3 / rcp / compares)", r"v_\w+_f32"),
    ("conversions", r"v_cvt_"),
    ("integer: Philox (mul / mad_u64 / xor), slot and pixel maps, addresses, bit operations",
     r"v_(mad_u64_u32|mul_lo_u32|mul_hi_u32|xor_b32|add_u32|sub_u32|subrev_u32|add3_u32|add_co_u32|addc_co_u32|lshl_add_u64|lshl_add_u32|lshlrev_b32|lshrrev_b32|lshlrev_b64|lshrrev_b64|lshl_or_b32|and_b32|or_b32|and_or_b32|or3_b32|bfe_u32|bfi_b32|not_b32|bitop3_b32|ashrrev_i32|min_u32|max_u32|min_i32|max_i32|mbcnt\w*|bcnt\w*|ffbh\w*|alignbit_b32|perm_b32|mad_u32_u24|mul_u32_u24|xad_u32|add_lshl_u32|lshl_or_b32|bfrev_b32)"),
    ("integer compares", r"v_cmp\w*_(u32|i32|u64|i64)"),
    ("moves / selects / lane operations (v_mov, v_cndmask, readlane, writelane, readfirstlane, swap)",
     r"v_(mov_b32|mov_b64|cndmask_b32|readlane_b32|writelane_b32|readfirstlane_b32|swap_b32|accvgpr\w*|nop)"),
    ("LDS", r"ds_"),
    ("vector memory", r"(global|flat|buffer|scratch)_"),
    ("scalar ALU / control", r"s_(?!load|buffer_load|waitcnt|nop|cbranch|branch|barrier|endpgm)"),
    ("scalar memory", r"s_(load|buffer_load)"),
    ("branches / waits / barriers", r"s_(waitcnt|nop|cbranch\w*|branch|barrier|endpgm)"),
]


def disassemble(pattern):
    with tempfile.TemporaryDirectory() as t:
        subprocess.check_call([f"{L}/llvm-objcopy", f"--dump-section=.hip_fatbin={t}/fat.bin", OBJ], stderr=subprocess.DEVNULL)
        subprocess.check_call([f"{L}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={t}/fat.bin",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={t}/k.co"])
        syms = subprocess.check_output([f"{L}/llvm-readelf", "-s", f"{t}/k.co"], text=True)
        sym = next(l.split()[7] for l in syms.splitlines() if "FUNC" in l and pattern in l)
        asm = subprocess.check_output([f"{L}/llvm-objdump", "-d", "--mcpu=gfx950", f"--disassemble-symbols={sym}", f"{t}/k.co"], text=True)
        name = subprocess.check_output(["c++filt", sym], text=True).strip()
    ops = [m.group(1) for m in re.finditer(r"^\s+([a-z_0-9]+) ", asm, re.M)]
    return name, ops


def main():
    pattern, key, kind = sys.argv[1], sys.argv[2], sys.argv[3]
    name, ops = disassemble(pattern)
    by_cat, by_op = collections.Counter(), collections.defaultdict(collections.Counter)
    for op in ops:
        for cat, rx in CATS:
            if re.match(rx, op):
                break
        else:
            cat = "unclassified"
        by_cat[cat] += 1
        by_op[cat][op] += 1
    print(f"# instruction mix of {name[:110]}")
    print(f"\n## static: {len(ops)} instructions in the code object ({sum(1 for o in ops if o.startswith('v_'))} vector ALU)")
    for cat, _ in CATS + [("unclassified", "")]:
        if by_cat[cat]:
            top = ", ".join(f"{o} {n}" for o, n in by_op[cat].most_common(6))
            print(f"{by_cat[cat]:6d}  {cat}\n        {top}")
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc.json"))).get(key, {}).get(kind)
    if pmc:
        print(f"\n## dynamic, per launch of the kernel kind '{kind}' of {key} (SQ counters, all instantiations of the kind together;")
        print("## the per-instantiation figures are in the pmc_*_digest.txt beside this file)")
        for k in ("valu_insts_per_wave", "lane_utilisation", "valu_busy", "waves_per_simd", "f64_insts_per_launch", "other_insts_per_launch",
                  "valu_mix_per_wave", "valu_mix_share", "non_valu_per_wave"):
            if k in pmc:
                print(f"{k}: {pmc[k]}")


if __name__ == "__main__":
    main()
