#!/bin/bash
# usage: scripts/sweep_hf.sh "VAR=val ..." ... — the 10M-triangle bench once per environment setting
H="--workload heightfield --spp 16 --batch 16 --steps 10 --warmup 2 --cpu-seconds 0 --configs none"
for cfg in "$@"; do
  ( export $cfg
    timeout -k 10 300 python bench.py $H > /tmp/sweep.json 2> /tmp/sweep.err || { echo "[$cfg] FAILED"; tail -3 /tmp/sweep.err; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/sweep.json'));r=d['roofline'];print('[$cfg]', d['value'], d['ms_per_step'], r['kernel_ms'])" ) || exit 1
done
