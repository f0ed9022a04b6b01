#!/usr/bin/env python3
"""Per-kernel digest of gpurun_out/pmc_<tag>/summary.txt (written by scripts/pmc_passes.sh)."""
import re, sys
txt = open(sys.argv[1]).read()
for b in re.split(r'\n(?!   )', txt):
    lines = b.strip().split('\n'); name = lines[0]
    d = {}
    for l in lines[1:]:
        w = l.split()
        if len(w) == 3 and w[1] == "total":
            d[w[0]] = float(w[2])
    if 'SQ_WAVES' not in d or d['SQ_WAVES'] < 100: continue
    cyc = d['GRBM_GUI_ACTIVE'] / 8
    hit = d.get('TCC_HIT_sum', 0) / max(1, d.get('TCC_HIT_sum', 0) + d.get('TCC_MISS_sum', 0))
    print('%-34s waves %.3g valu/wave %4.0f lane_util %.2f valu_busy %.2f occ %.2f L2hit %.2f fetchGB %.3g writeGB %.3g Mcycles %.3g' % (
        name[-34:], d['SQ_WAVES'], d['SQ_INSTS_VALU'] / d['SQ_WAVES'],
        d['SQ_THREAD_CYCLES_VALU'] / (d['SQ_ACTIVE_INST_VALU'] * 64), d['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc),
        d['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc), hit, d['FETCH_SIZE'] / 1e6, d['WRITE_SIZE'] / 1e6, cyc / 1e6))
