import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import pathtracer_amd as pt
from pathtracer_amd.synthetic import heightfield_box
for cells in (40, 300):
    data = heightfield_box(cells=cells)
    a = pt.Scene(data, 0, options=dict(packet_primary=0))
    b = pt.Scene(data, 0, options=dict(packet_primary=1))
    for cfg in (pt.Config(width=211, height=97, samples_pr_pixel=64, bounces=2, seed=3, spp_per_batch=64),
                pt.Config(width=160, height=90, samples_pr_pixel=5, bounces=2, seed=4, nee=True),
                pt.Config(dim=96, samples_pr_pixel=128, bounces=0, seed=5)):
        ia, ib = a.render(cfg), b.render(cfg)
        print(cells, cfg.w, cfg.samples_pr_pixel, ia.tobytes() == ib.tobytes(), float(ia.mean()), flush=True)
        assert ia.tobytes() == ib.tobytes()
print("ok")
