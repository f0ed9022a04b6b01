#!/bin/bash
# usage: scripts/profile_round.sh <tag> — rocprofv3 kernel stats of the two bench commands, then the PMC passes.
# Outputs under gpurun_out/prof_<tag>/ and gpurun_out/pmc_<tag>_{cornell,hf}/ (copy what is judged into profiles/).
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HF="--workload heightfield --spp 16 --batch 16 --steps 5 --warmup 1 --cpu-seconds 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT -o cornell --output-format csv -- python3 $ROOT/bench.py --cpu-seconds 0 > $OUT/cornell_bench.json 2> $OUT/cornell.err || { tail -5 $OUT/cornell.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT -o hf --output-format csv -- python3 $ROOT/bench.py $HF > $OUT/hf_bench.json 2> $OUT/hf.err || { tail -5 $OUT/hf.err; exit 1; }
rm -f $OUT/*kernel_trace.csv   # per-dispatch trace: large, the stats file is what is kept
ls $OUT
cd $ROOT
bash scripts/pmc_passes.sh ${TAG}_cornell --steps 2 --warmup 0 --cpu-seconds 0 > $OUT/pmc_cornell.log 2>&1 || exit 1
bash scripts/pmc_passes.sh ${TAG}_hf --workload heightfield --spp 16 --batch 16 --steps 1 --warmup 0 --cpu-seconds 0 > $OUT/pmc_hf.log 2>&1 || exit 1
python3 scripts/pmc_traffic.py CornellBox_1920x1080_spp64_b2 gpurun_out/pmc_${TAG}_cornell $OUT/traffic.json > /dev/null
python3 scripts/pmc_traffic.py heightfield_3840x2160_spp16_b2 gpurun_out/pmc_${TAG}_hf $OUT/traffic.json > /dev/null
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_cornell/summary.txt > $OUT/pmc_cornell_digest.txt
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_hf/summary.txt > $OUT/pmc_hf_digest.txt
cat $OUT/pmc_cornell_digest.txt $OUT/pmc_hf_digest.txt
