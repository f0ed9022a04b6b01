"""Largest tile (1920 wide) one pt_render_device_noalloc call accepts, without and with NEE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pathtracer_amd as pt
scene = pt.Scene(pt.scene_data("CornellBox"), 0)
for nee in (False, True):
    lo, hi = 1, 1080
    while lo < hi:
        mid = (lo + hi + 1) // 2
        cfg = pt.Config(width=1920, height=1080, bounces=2, seed=1, nee=nee, screen_cull=False) if False else pt.Config(width=1920, height=1080, bounces=2, seed=1, nee=nee)
        acc = torch.zeros((1080, 1920, 3), dtype=torch.float64, device="cuda")
        try:
            scene.render_device_noalloc(cfg, acc.data_ptr(), 0, 1, tile=(0, 0, 1920, mid))
            torch.cuda.synchronize()
            lo = mid
        except pt.PtError:
            hi = mid - 1
    print("nee" if nee else "plain", "rows", lo, "tile pixels", 1920 * lo)
