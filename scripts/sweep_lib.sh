#!/bin/bash
# usage: scripts/sweep_lib.sh <scene: cornell|hf> — the bench once per tuning build in pathtracer_amd/lib/exp
# (PT_LIB_PATH) and once with the default library; one line per build.  Stops at the first failure.
SCENE=${1:-hf}
if [ "$SCENE" = hf ]; then ARGS="--workload heightfield --spp 16 --batch 16 --steps 10 --warmup 2"; else ARGS=""; fi
for LIB in default pathtracer_amd/lib/exp/*.so; do
  if [ "$LIB" = default ]; then unset PT_LIB_PATH; else export PT_LIB_PATH=$PWD/$LIB; fi
  timeout -k 10 300 python bench.py $ARGS --cpu-seconds 0 --configs none > gpurun_out/sweep_lib.json 2> gpurun_out/sweep_lib.err || { tail -5 gpurun_out/sweep_lib.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/sweep_lib.json'));r=d['roofline'];print('$LIB', d['value'],d['ms_per_step'],r['kernel_ms'])"
done
