#!/bin/bash
# usage: scripts/diag/lds_pmc.sh <scene> — LDS activity / bank-conflict counters of one 64-frame render
S=${1:-CornellBox}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u | tr '\n' ' '; echo
rm -rf /tmp/lds1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES -d /tmp/lds1 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/diag/lds_once.py $S > /tmp/lds1.log 2>&1 || { tail -5 /tmp/lds1.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/lds1/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_path" not in k and "k_extend" not in k: continue
    acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(k, {c: int(x) for c, x in v.items()})
PY
