#!/bin/bash
for b in 32 64 32 64 48; do
  timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --workload heightfield --spp 256 --batch $b --steps 3 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "[$b] FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'));r=d['roofline'];print('[batch $b]', d['value'], d['ms_per_step'], {k:v for k,v in r['kernel_ms'].items() if v})"
done
