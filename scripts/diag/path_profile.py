"""Time of the sections of the two-bounce path kernel (diagnostic build only):

    scripts/build_variant.sh pprof -DPT_PATH_PROFILE
    PT_LIB_PATH=pathtracer_amd/lib/alt_pprof.so python scripts/diag/path_profile.py [scene]

The build's path kernels stop after section `cut` (a run-time switch, so the code and its registers are those of the
complete kernel): 1 = ray generation + primary traversal, 2 = + first vertex, 3 = + second traversal, 4 = + second
vertex, 0 = everything (+ compaction and push, and the launches of the last bounce that then have work).  The
differences between the render times are the sections' costs.  (Clock stamps inside the kernel were tried first and
proved useless: the compiler moves the work across them.)
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtracer_amd as pt
from pathtracer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "CornellBox"
fn = L.lib.pt_debug_path_cut
fn.restype = C.c_int
fn.argtypes = [C.c_int]
scene = pt.Scene(pt.scene_data(name), 0)
cfg = pt.Config(width=1920, height=1080, bounces=2, seed=1, spp_per_batch=64)
acc = torch.zeros((cfg.h, cfg.w, 3), dtype=torch.float64, device="cuda")
ms = {}
for cut in (1, 2, 3, 4, 0):
    assert fn(cut) == 0
    scene.render_device(cfg, acc.data_ptr(), 0, 64)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(8):
        scene.render_device(cfg, acc.data_ptr(), 64 * k, 64)
    b.record()
    torch.cuda.synchronize()
    ms[cut] = a.elapsed_time(b) / 8
names = {1: "generate + primary traversal (+ accumulate, launches)", 2: "first vertex", 3: "second traversal", 4: "second vertex",
         0: "compaction, push, last bounce"}
prev, out = 0.0, {}
for cut in (1, 2, 3, 4, 0):
    out[names[cut]] = round(ms[cut] - prev, 4)
    prev = ms[cut]
print(json.dumps({"scene": name, "ms_per_64_frames_up_to_cut": {str(k): round(v, 4) for k, v in ms.items()}, "section_ms": out}, indent=1))
