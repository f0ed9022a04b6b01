import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtracer_amd as pt
scene = pt.Scene(pt.scene_data(sys.argv[1]), 0)
cfg = pt.Config(width=1920, height=1080, bounces=2, seed=1, spp_per_batch=64)
acc = torch.zeros((cfg.h, cfg.w, 3), dtype=torch.float64, device="cuda")
scene.render_device(cfg, acc.data_ptr(), 0, 64)
torch.cuda.synchronize()
