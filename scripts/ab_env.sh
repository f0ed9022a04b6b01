#!/bin/bash
# usage: scripts/ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV3=c" ... — the same bench line under each environment
ARGS=$1; shift
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --no-cull-steps 0 $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "[$e] FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'));r=d.get('roofline') or {'kernel_ms':{}};print('[$e]', d['value'], d['ms_per_step'], {k:v for k,v in r['kernel_ms'].items() if v})"
done
