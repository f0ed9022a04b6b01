"""One render of 64 frames with the path kernels cut after section argv[2] (see path_profile.py); run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU` to count the instructions of a section."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtracer_amd as pt
from pathtracer_amd import _lib as L

name, cut = sys.argv[1], int(sys.argv[2])
fn = L.lib.pt_debug_path_cut
fn.restype = C.c_int
fn.argtypes = [C.c_int]
if name == "heightfield":
    from pathtracer_amd.synthetic import heightfield_box
    scene = pt.Scene(heightfield_box(), 0)
    cfg = pt.Config(width=3840, height=2160, bounces=2, seed=1, spp_per_batch=64)
elif name == "Test_textured":
    scene = pt.Scene(pt.scene_data("Test", textures=True), 0)
    cfg = pt.Config(width=3840, height=2160, bounces=2, seed=1, spp_per_batch=16)
else:
    scene = pt.Scene(pt.scene_data(name), 0)
    cfg = pt.Config(width=1920, height=1080, bounces=2, seed=1, spp_per_batch=64)
acc = torch.zeros((cfg.h, cfg.w, 3), dtype=torch.float64, device="cuda")
assert fn(cut) == 0
scene.render_device(cfg, acc.data_ptr(), 0, cfg.spp_per_batch or 64)
torch.cuda.synchronize()
