#!/bin/bash
# usage: scripts/ab_pipeline.sh — one lane vs two lanes on the 10 M-triangle scene at several batch sizes
for b in 8 16 32 64; do
  for p in 0 1; do
    PT_PIPELINE=$p timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --no-roofline --workload heightfield --spp 256 --batch $b --steps 2 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
    python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('batch $b lanes', $p + 1, d['value'], d['ms_per_step'])"
  done
done
