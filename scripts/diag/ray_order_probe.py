#!/usr/bin/env python3
"""What would sorting the secondary rays buy the deep-tree extend kernel?  

The bounce-1 rays of the 10 M-triangle heightfield at 4K (primary hit -> Lambertian sample, through the parity entry
points of the C ABI) are traced by pt_trace_closest - the vote-scheduled k_extend - in four orders, kernel time from
the HIP-event timers: as the render pushes them (pixel order), shuffled, sorted by direction octant, and sorted by
(origin cell, octant).  An upper bound: the sort itself is free here.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from pathtracer_amd.synthetic import heightfield_box  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
w, h = (3840, 2160) if len(sys.argv) <= 2 else (int(sys.argv[2]), int(sys.argv[3]))
data = heightfield_box(cells)
scene = pt.Scene(data, device=0)
cfg = pt.Config(width=w, height=h, seed=1)
rays = scene.generate_rays(cfg, 0)
tri, t = scene.trace_closest(rays)
hit = tri != pt.PT_MISS
pixel = np.arange(rays.shape[0], dtype=np.uint32)[hit]
f, wi, pdf, em, pos = scene.sample_brdf(rays[hit], tri[hit], t[hit], pixel, cfg, 0, 0)
sec = pt.make_rays(pos, wi)
n = sec.shape[0]
octant = (wi[:, 0] > 0).astype(np.int64) | ((wi[:, 1] > 0).astype(np.int64) << 1) | ((wi[:, 2] > 0).astype(np.int64) << 2)
lo, hi = pos.min(axis=0), pos.max(axis=0)
cell = np.minimum(((pos - lo) / np.maximum(hi - lo, 1e-9) * 64).astype(np.int64), 63)
cell_key = (cell[:, 0] << 12) | (cell[:, 2] << 6) | cell[:, 1]
rng = np.random.default_rng(0)
orders = {
    "pixel_order": np.arange(n),
    "shuffled": rng.permutation(n),
    "by_octant": np.argsort(octant, kind="stable"),
    "by_cell_then_octant": np.argsort((cell_key << 3) | octant, kind="stable"),
    "by_octant_then_cell": np.argsort((octant << 18) | cell_key, kind="stable"),
}
out = {"rays": int(n), "triangles": data.n_tris, "image": [w, h]}
ref = None
for name, order in orders.items():
    batch = np.ascontiguousarray(sec[order])
    scene.trace_closest(batch)                      # warm
    scene.reset_stats()
    scene.set_profiling(1)
    tri2, t2 = scene.trace_closest(batch)
    st = scene.stats()
    scene.set_profiling(0)
    back = np.empty_like(order)
    back[order] = np.arange(n)
    if ref is None:
        ref = (tri2[back], t2[back])
    else:
        assert np.array_equal(tri2[back], ref[0]) and np.array_equal(t2[back], ref[1])
    ms = st["kernel"]["extend"]["ms"]
    out[name] = {"extend_ms": round(ms, 3), "Mrays_per_s": round(n / ms / 1e3, 1)}
print(json.dumps(out))
