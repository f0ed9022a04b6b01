#!/bin/bash
# usage: scripts/profile_workload.sh <tag> <bench.py args for the counter passes> -- <bench.py args for the kernel trace>
# The evidence behind one workload's `roofline`: the counter passes (scripts/pmc_collect.sh) and one
# `rocprofv3 --kernel-trace --stats` run of bench.py itself, whose per-kernel average durations must agree with
# the HIP-event timings bench.py reports.  Everything lands in gpurun_out/; scripts/pmc_digest.py and a copy into
# profiles/<round>/ follow on the CPU side.
set -u
TAG=$1; shift
PMC_ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do PMC_ARGS+=("$1"); shift; done
[ $# -gt 0 ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
"$ROOT/scripts/pmc_collect.sh" "$TAG" "${PMC_ARGS[@]}" || exit $?
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/kt_$TAG" -o kt --output-format csv -- \
    python3 "$ROOT/bench.py" --configs none --cpu-seconds 0 --no-cull-steps 0 "$@" > "$ROOT/gpurun_out/kt_$TAG.json" 2> "$ROOT/gpurun_out/kt_$TAG.err"
rc=$?
echo "kernel trace rc=$rc"; tail -c 600 "$ROOT/gpurun_out/kt_$TAG.json"
exit $rc
