#!/bin/bash
# usage: scripts/pmc_collect.sh <tag> <bench.py args...>
# One rocprofv3 --pmc run per counter group (gfx950 holds 8 SQ / 4 TCC / 4 TCP / 2 TA / 2 GRBM counters per
# pass; FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), each with a time limit of its own; the script ends at
# the first pass that fails.  Output: gpurun_out/pmc_<tag>/p<i>/..., digested by scripts/pmc_digest.py.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $grp -d $OUT/p$i -o p --output-format csv -- python3 $ROOT/bench.py --configs none --cpu-seconds 0 --no-cull-steps 0 --no-roofline "$@" > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i ($grp) rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $OUT/p$i.log; exit $rc; fi
done
