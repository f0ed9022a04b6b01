#!/bin/bash
# usage: scripts/profile_ingest.sh <round dir, e.g. profiles/r02> — after scripts/profile_workload.sh ran for the four
# workloads (tags cornell / glossy / textured / heightfield): digest the counter passes into profiles/pmc.json and
# copy the per-kernel summaries into the round's directory.
R=${1:-profiles/r02}
mkdir -p $R
for p in "cornell CornellBox" "glossy glossy_planes" "textured Test_textured" "heightfield heightfield"; do
  set -- $p
  KEY=$(python3 -c "
import bench
wl = bench.WORKLOADS['$2']
print(bench.pmc_key(wl, 2, False))")
  python3 scripts/pmc_digest.py $KEY gpurun_out/pmc_$1 profiles/pmc.json > $R/pmc_$1_digest.txt || exit 1
  cp gpurun_out/kt_$1/kt_kernel_stats.csv $R/$1_kernel_stats.csv
  cp gpurun_out/kt_$1.json $R/$1_bench_under_kernel_trace.json
  echo "$1 -> $KEY"
done
