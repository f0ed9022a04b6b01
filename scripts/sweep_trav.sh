#!/bin/bash
# usage: scripts/sweep_trav.sh "<P_vote P_wr S_vote S_wr>" ... — bench both scenes per knob setting
run() {  # $1 label, rest: bench args
  local label=$1; shift
  timeout -k 10 300 python bench.py --cpu-seconds 0 "$@" > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$label FAILED"; tail -3 /tmp/sweep.err; exit 1; }
  python -c "
import json;d=json.load(open('/tmp/sweep.json'));r=d['roofline'];print('$label', d['value'], r['kernel_ms'])"
}
C=""
H="--scene heightfield --width 3840 --height 2160 --spp 2 --steps 5 --warmup 1"
for cfg in "$@"; do
  set -- $cfg
  export PT_TRAV_VOTE_PRIMARY=$1 PT_WAVE_RAYS_PRIMARY=$2 PT_TRAV_VOTE=$3 PT_WAVE_RAYS=$4
  run "cornell P(vote=$1,wr=$2) S(vote=$3,wr=$4)" $C
  run "heightf P(vote=$1,wr=$2) S(vote=$3,wr=$4)" $H
done
