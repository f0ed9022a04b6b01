#!/bin/bash
# usage: scripts/ab_env_lib.sh "<bench args>" "lib.so ENV=.. ENV2=.." ... — the same bench line per (library, environment) pair
ARGS=$1; shift
for e in "$@"; do
  set -- $e; lib=$1; shift
  env PT_LIB_PATH=$lib "$@" timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --no-cull-steps 0 --no-roofline $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "[$e] FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('[$e]', d['value'], d['ms_per_step'])"
done
