#!/usr/bin/env python3
"""What the host boundary costs per frame-loop iteration (DESIGN.md "Host boundary"): CornellBox 1080p, the
reference's frame loop shape (main.rs:400-436: one frame of 1 spp per iteration, displayed every iteration).

  kernel only      Scene.render_device into a device buffer, 1 spp per call (no host traffic)
  pt_accum         pt_accum_render, 1 spp per call, sums resident on the device (the entry INTEGRATION.md gives
                   the window loop); with and without the 8-bit resolve + 8 MB download per displayed frame
  pt_render        the host-buffer entry: 50 MB up and down per call

    python scripts/pcie_rate.py [spp per call] > profiles/<round>/pcie_rate.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pathtracer_amd as pt

W, H = 1920, 1080
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 200


def rate(fn, n=N, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(warm + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"ms_per_call": round(dt / n * 1e3, 4), "Msamples_per_s": round(W * H * SPP * n / dt / 1e6, 1)}


def main():
    scene = pt.Scene(pt.scene_data("CornellBox"), device=0)
    cfg = pt.Config(width=W, height=H, bounces=2, seed=1, samples_pr_pixel=SPP, spp_per_batch=SPP)
    out = {"workload": f"CornellBox {W}x{H}, bounces 2, {SPP} spp per call, {N} calls"}

    dev_acc = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    out["kernel_only_render_device"] = rate(lambda i: scene.render_device(cfg, dev_acc.data_ptr(), i * SPP, SPP, stream=stream))

    acc = pt.Accum(scene, cfg)
    out["pt_accum_render"] = rate(lambda i: acc.render(SPP))

    def frame_and_display(i):
        acc.render(SPP)
        acc.resolve_u32()          # main.rs:456-467: the window's 0x00RRGGBB buffer, resolved on the device
    out["pt_accum_render_plus_resolve_u32_every_frame"] = rate(frame_and_display)

    host = np.zeros((H, W, 3))
    out["pt_render_host_buffer"] = rate(lambda i: scene.render(cfg, host, sample_begin=i * SPP, sample_count=SPP), n=40)

    k = out["kernel_only_render_device"]["ms_per_call"]
    out["pt_accum_over_kernel_only"] = round(out["pt_accum_render"]["ms_per_call"] / k, 3)
    out["pt_accum_with_display_over_kernel_only"] = round(out["pt_accum_render_plus_resolve_u32_every_frame"]["ms_per_call"] / k, 3)
    out["pt_render_over_kernel_only"] = round(out["pt_render_host_buffer"]["ms_per_call"] / k, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
