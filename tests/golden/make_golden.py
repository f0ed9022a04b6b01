#!/usr/bin/env python3
"""Generates tests/golden/*.npz — run from the repo root: python tests/golden/make_golden.py

The Rust reference cannot be built or run here (no rustc/cargo), and its own tests
hold only two golden values (microfacet.rs:396-409, vec3.rs:408-424; both in
test_oracle_golden.py).  The vectors below are therefore produced by an INDEPENDENT second
restatement of the reference arithmetic, written in numpy (elementwise IEEE f64, no FMA), of
  * scene.rs:37-57 Camera::ray for centre-of-pixel rays (jitter 0), and
  * triangle_mesh.rs:101-126 Triangle::intersect by brute force over all triangles in id order
    (lowest id wins bit-identical t),
so that the C++ oracle (and through it the HIP kernels) is checked against something that shares
no code with it.  K3 of SURVEY.md §8c: primary hits on CornellBox.obj at 128x128.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def normalize(a):
    l = np.sqrt(a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1] + a[..., 2] * a[..., 2])
    return a / l[..., None]


def camera_rays(eye, view, up, camera_const, w, h, dx=0.0, dy=0.0):
    """main.rs:412-417 + scene.rs:37-45 for every pixel, row-major."""
    eye, view, up = (np.asarray(v, np.float64) for v in (eye, view, up))
    a0 = -normalize(cross(view, up))
    a1 = -normalize(cross(a0, view))
    view_dir = view * camera_const
    i = np.arange(w * h)
    x = (i % w).astype(np.float64)
    y = (i // w).astype(np.float64)
    aspect = np.float64(w) / np.float64(h)
    ndc_x = (x + 0.5) / np.float64(w)
    ndc_y = (y + 0.5) / np.float64(h)
    ss_x = (2.0 * ndc_x - 1.0 + dx) * aspect
    ss_y = 1.0 - 2.0 * ndc_y + dy
    d = normalize((view_dir[None, :] + a0[None, :] * ss_x[:, None]) + a1[None, :] * ss_y[:, None])
    o = np.broadcast_to(eye, d.shape).copy()
    return o, d


def closest_hits(verts, tris, o, d, t_min=1e-5, t_max=np.finfo(np.float64).max):
    """triangle_mesh.rs:101-126 against every triangle, in id order."""
    n = o.shape[0]
    best_t = np.full(n, t_max)
    best = np.full(n, 0xFFFFFFFF, np.uint32)
    p = verts[tris].astype(np.float64)
    with np.errstate(all="ignore"):
        for k in range(p.shape[0]):
            v0, v1, v2 = p[k, 0], p[k, 1], p[k, 2]
            e0 = v1 - v0
            e1 = v0 - v2
            normal = cross(e0, e1)
            dn = dot(d, normal[None, :])
            ok = ~(np.abs(dn) < 1.0e-12)
            vo = v0[None, :] - o
            t = dot(vo, normal[None, :]) / dn
            ok &= ~((t < t_min) | (t > best_t))
            tmp = cross(vo, d)
            denom = 1.0 / dn
            v = dot(tmp, e1[None, :]) * denom
            w = dot(tmp, e0[None, :]) * denom
            ok &= ~((v < 0.0) | (w < 0.0) | (v + w > 1.0))
            ok &= (best == 0xFFFFFFFF) | (t < best_t)  # lowest id keeps a bit-identical t
            best[ok] = k
            best_t[ok] = t[ok]
    return best, best_t


def main():
    import pathtracer_amd as pt
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, dim in (("CornellBox", 128), ("Test", 96), ("glossy_planes", 96)):
        data = pt.scene_data(name)
        cam = data.camera
        o, d = camera_rays(cam.eye, cam.view_direction, cam.up_direction, cam.camera_const, dim, dim)
        tri, t = closest_hits(data.verts, data.tris, o, d)
        np.savez_compressed(os.path.join(out_dir, f"primary_hits_{name}_{dim}.npz"),
                            dim=np.int32(dim), direction=d, tri=tri, t=t)
        print(name, dim, "hit fraction", float((tri != 0xFFFFFFFF).mean()))


if __name__ == "__main__":
    main()
