#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry pt_render: CornellBox 1080p, 8 spp per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracer_amd as pt

scene = pt.Scene(pt.scene_data("CornellBox"), device=0)
cfg = pt.Config(width=1920, height=1080, bounces=2, seed=1, samples_pr_pixel=8, spp_per_batch=8)
acc = np.zeros((1080, 1920, 3))
for i in range(3):
    scene.render(cfg, acc, sample_begin=8 * i, sample_count=8)
t0 = time.perf_counter()
n = 20
for i in range(n):
    scene.render(cfg, acc, sample_begin=8 * (3 + i), sample_count=8)
dt = time.perf_counter() - t0
print(f"pt_render (host accumulation buffer, H2D + D2H per call): {1920 * 1080 * 8 * n / dt / 1e6:.1f} Msamples/s, {dt / n * 1e3:.3f} ms per call")
