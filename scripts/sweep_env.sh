#!/bin/bash
# usage: scripts/sweep_env.sh "VAR=val VAR2=val" ... — bench both scenes once per environment setting
run() {  # $1 label, rest: bench args
  local label=$1; shift
  timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none "$@" > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$label FAILED"; tail -3 /tmp/sweep.err; exit 1; }
  python -c "
import json;d=json.load(open('/tmp/sweep.json'));r=d['roofline'];print('$label', d['value'], d['ms_per_step'], r['kernel_ms'])"
}
H="--workload heightfield --spp 16 --batch 16 --steps 5 --warmup 1"
for cfg in "$@"; do
  ( export $cfg
    run "cornell [$cfg]"
    run "heightf [$cfg]" $H ) || exit 1
done
