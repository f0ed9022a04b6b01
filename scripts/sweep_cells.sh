#!/bin/bash
# usage: scripts/sweep_cells.sh — secondary-ray scheduling (while-while vs vote) over scene sizes
for cells in 8 16 64 256 1024; do
  for cfg in "0 64" "1 256"; do
    set -- $cfg
    export PT_TRAV_VOTE=$1 PT_WAVE_RAYS=$2
    timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none --workload heightfield --width 1920 --height 1080 --spp 64 --cells $cells --steps 10 > /tmp/sweep.json 2> /tmp/sweep.err || { echo "FAILED"; tail -3 /tmp/sweep.err; exit 1; }
    python -c "
import json;d=json.load(open('/tmp/sweep.json'));r=d['roofline'];print('cells $cells vote=$1 wr=$2', d['config']['triangles'], d['value'], r['kernel_ms'], r['nodes_per_ray'], r['tris_per_ray'])"
  done
done
