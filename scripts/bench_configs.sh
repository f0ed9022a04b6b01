#!/bin/bash
# usage: scripts/bench_configs.sh — the NEE stage of every BASELINE workload through bench.py (one line each; the
# plain configs are timed by the default bench.py run itself, under "configs").
run() {
  local label=$1; shift
  timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none "$@" > gpurun_out/cfg_$label.json 2> gpurun_out/cfg_$label.err || { echo "$label FAILED"; tail -3 gpurun_out/cfg_$label.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/cfg_$label.json'));r=d['roofline'];print('$label', d['value'], d['ms_per_step'], d['mrays_per_s'], r['kernel'], r['kernel_ms'])"
}
run cornell_nee --nee --steps 20 --warmup 2 --spp 32 || exit 1
run glossy_nee --workload glossy_planes --nee --steps 20 --warmup 2 --spp 32 || exit 1
run test4k_tex_nee --workload Test_textured --nee --steps 10 --warmup 2 --spp 8 || exit 1
run hf_nee --workload heightfield --nee --steps 4 --warmup 1 --spp 8 || exit 1
