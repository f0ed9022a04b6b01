#!/bin/bash
# usage (on the GPU box): scripts/profile_all.sh — the counter passes and kernel traces of the four bench workloads and one
# default bench run, all into gpurun_out/; scripts/profile_ingest.sh profiles/<round> follows on the CPU side.
scripts/profile_workload.sh cornell --steps 2 --warmup 1 -- --steps 5 --warmup 1 > gpurun_out/prof_cornell.log 2>&1 &&
scripts/profile_workload.sh glossy --workload glossy_planes --steps 2 --warmup 1 -- --workload glossy_planes --steps 4 --warmup 1 > gpurun_out/prof_glossy.log 2>&1 &&
scripts/profile_workload.sh textured --workload Test_textured --steps 2 --warmup 1 -- --workload Test_textured --steps 4 --warmup 1 > gpurun_out/prof_textured.log 2>&1 &&
scripts/profile_workload.sh heightfield --workload heightfield --spp 128 --steps 1 --warmup 1 -- --workload heightfield --spp 256 --steps 2 --warmup 1 > gpurun_out/prof_heightfield.log 2>&1 &&
timeout -k 10 400 python bench.py > gpurun_out/bench_default_run.json 2> gpurun_out/bench_default_run.err
echo rc=$?
