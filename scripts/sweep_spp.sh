#!/bin/bash
# usage: scripts/sweep_spp.sh — both benches against the batch size (frames per step): kernel time against batch
# size separates a fixed tail from per-ray cost
for SPP in 8 16; do
  for PIPE in 0 1; do
  PT_PIPELINE=$PIPE timeout -k 10 300 python bench.py --workload heightfield --spp $SPP --steps 4 --warmup 1 --cpu-seconds 0 --configs none > /tmp/s.json 2> /tmp/s.err || { tail -3 /tmp/s.err; exit 1; }
  python -c "
import json;d=json.load(open('/tmp/s.json'));r=d['roofline'];print('hf spp $SPP pipe $PIPE', d['value'], d['ms_per_step'], {k:round(v/4,3) for k,v in r['kernel_ms'].items()})"
  done
done
for SPP in 8 16 32 64; do
  timeout -k 10 300 python bench.py --spp $SPP --steps 20 --warmup 2 --cpu-seconds 0 --configs none > /tmp/s.json 2> /tmp/s.err || { tail -3 /tmp/s.err; exit 1; }
  python -c "
import json;d=json.load(open('/tmp/s.json'));r=d['roofline'];print('cornell spp $SPP', d['value'], d['ms_per_step'], {k:round(v/20,3) for k,v in r['kernel_ms'].items()})"
done
