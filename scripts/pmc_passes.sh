#!/bin/bash
# usage: scripts/pmc_passes.sh <tag> <bench args...>   — one rocprofv3 --pmc run per counter group
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $grp -d $OUT/p$i -o p --output-format csv -- python3 $ROOT/bench.py "$@" > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i ($grp) rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $OUT/p$i.log; fi
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVES', 'TCC_HIT_sum', 'FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_LDS', 'TA_BUSY_avr'):
            cnt[(k, r['Counter_Name'])] += 1
with open(out + '/summary.txt', 'w') as fo:
    for k, d in agg.items():
        if 'ptd' not in k: continue
        fo.write(k + '\n')
        for c, v in sorted(d.items()):
            n = max(cnt.get((k, c), 0), 1)
            fo.write(f'   {c:32s} total {v:.6g}\n')
print(open(out + '/summary.txt').read())
PY
