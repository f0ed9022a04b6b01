#!/bin/bash
# usage: scripts/sweep_cornell.sh "VAR=val ..." ... — the headline bench once per environment setting
for cfg in "$@"; do
  ( export $cfg
    timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --steps 50 > /tmp/sweep.json 2> /tmp/sweep.err || { echo "[$cfg] FAILED"; tail -3 /tmp/sweep.err; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/sweep.json'));r=d['roofline'];print('[$cfg]', d['value'], d['ms_per_step'], r['kernel_ms'])" ) || exit 1
done
