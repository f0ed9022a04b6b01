#!/bin/bash
# usage: scripts/gpu_check.sh <tag>  — GPU tests, then both benches; prints a short summary.
# Every GPU step runs under its own time limit and the script ends at the first failing step.
TAG=$1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/test_$TAG.log 2>&1
rc=$?; echo "tests exit=$rc"; tail -3 gpurun_out/test_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/bench_$TAG.json'));r=d['roofline'];print('cornell', d['value'],d['ms_per_step'],r['kernel_ms'],r['kernel'],r['achieved'])"
timeout -k 10 600 python bench.py --workload heightfield --spp 16 --batch 16 --steps 6 --warmup 1 --cpu-seconds 0 --configs none > gpurun_out/bench_hf_$TAG.json 2> gpurun_out/bench_hf_$TAG.err || { tail -5 gpurun_out/bench_hf_$TAG.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/bench_hf_$TAG.json'));r=d['roofline'];print('heightfield', d['value'],d['ms_per_step'],r['kernel_ms'],r['kernel'],r['achieved'])"
