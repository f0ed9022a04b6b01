#!/usr/bin/env python3
"""Sums every counter of the scripts/pmc_passes.sh passes per kernel -> <dir>/summary.txt."""
import collections
import csv
import glob
import re
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"ptd::(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:32s} total {v:.6g}\n")
print(open(out + "/summary.txt").read())
