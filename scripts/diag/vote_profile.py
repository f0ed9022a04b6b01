"""Where the lanes of the vote-scheduled traversal loop go (diagnostic build only):

    scripts/build_variant.sh vprof -DPT_VOTE_PROFILE
    PT_LIB_PATH=pathtracer_amd/lib/alt_vprof.so python scripts/diag/vote_profile.py [cells] [nee]

Counters (summed over all waves of the extend / shadow launches of one render): node trips and lanes at a node
in them, second node steps and lanes still at a node for them, triangle trips with the largest and the summed leaf
sizes, retire trips.
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pathtracer_amd as pt
from pathtracer_amd import _lib as L
from pathtracer_amd.synthetic import heightfield_box

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
nee = len(sys.argv) > 2 and sys.argv[2] == "nee"
fn = L.lib.pt_debug_vote_profile
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
scene = pt.Scene(heightfield_box(cells=cells), 0)
w, h, spp = (int(x) for x in os.environ.get("VP_SIZE", "3840,2160,64").split(","))
cfg = pt.Config(width=w, height=h, samples_pr_pixel=spp, bounces=2, seed=1, nee=nee, spp_per_batch=spp)
scene.render(cfg)
out = (C.c_ulonglong * 16)()
assert fn(out, 1) == 0
scene.render(cfg)
assert fn(out, 1) == 0
v = list(out)
r = {
    "node_trips": v[0], "node_lanes_per_trip": v[1] / max(v[0], 1), "node_trip_lanes_at_tri": v[8] / max(v[0], 1),
    "node_trip_lanes_at_retire": v[9] / max(v[0], 1),
    "second_steps": v[2], "second_step_lanes": v[3] / max(v[2], 1),
    "tri_trips": v[4], "tri_lanes_per_trip": v[5] / max(v[4], 1), "tri_max_leaf_per_trip": v[6] / max(v[4], 1),
    "tri_tests_per_trip": v[7] / max(v[4], 1), "tri_slot_utilisation": v[7] / max(64 * v[6], 1),
    "tri_trip_lanes_at_node": v[10] / max(v[4], 1), "tri_trip_lanes_at_retire": v[11] / max(v[4], 1),
    "idle_lanes_node_trips": 64 - (v[1] + v[8] + v[9]) / max(v[0], 1), "idle_lanes_tri_trips": 64 - (v[5] + v[10] + v[11]) / max(v[4], 1),
    "node_steps": v[1] + v[3], "node_steps_all_children_missed": v[14], "children_pushed": v[15],
    "rays": v[13],
    "retire_trips": v[12], "retire_lanes_per_trip": v[13] / max(v[12], 1),
}
print(json.dumps(r, indent=1))
