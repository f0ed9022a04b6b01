#!/bin/bash
# usage: scripts/diag/path_cut_pmc.sh <scene> — vector instructions per wave of the two-bounce kernel cut after each section
# (diagnostic build alt_pprof.so, see scripts/diag/path_profile.py)
S=${1:-CornellBox}
export PT_LIB_PATH=$PWD/pathtracer_amd/lib/alt_pprof.so
cd /tmp && export TMPDIR=/tmp
for cut in ${CUTS:-5 6 1 2 3 4 0}; do
  rm -rf /tmp/pc$cut
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d /tmp/pc$cut -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/diag/path_cut_once.py $S $cut > /tmp/pc$cut.log 2>&1 || { tail -5 /tmp/pc$cut.log; exit 1; }
  python3 - $cut <<'PY'
import csv, glob, sys, collections
cut = sys.argv[1]
f = glob.glob(f"/tmp/pc{cut}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_path" not in k: continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(cut, k, {c: int(x) for c, x in v.items()}, "valu/wave", round(v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1), 1))
PY
done
