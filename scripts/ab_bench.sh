#!/bin/bash
# usage: scripts/ab_bench.sh "<bench args>" lib1.so lib2.so ... — the same bench line with each library build
ARGS=$1; shift
for lib in "$@"; do
  PT_LIB_PATH=$lib timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --no-cull-steps 0 $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'));r=d.get('roofline') or {'kernel_ms':{}};print('$(basename $lib)', d['value'], d['ms_per_step'], {k:v for k,v in r['kernel_ms'].items() if v})"
done
