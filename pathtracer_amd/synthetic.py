"""Synthetic large scene of BASELINE.json configs[4] / SURVEY.md §8d config 5.

The Test.obj box (walls, ceiling, light; camera of scene.rs:184-186) whose floor is replaced by a
displaced heightfield of ``cells x cells`` quads (2236 -> 9,999,392 triangles).  Heights are seeded
value noise (amplitude ~40 units), vertices are rounded to f32, everything is Lambertian with
albedo 0.5 except the emissive quad (Ka 20).  Deterministic (seed 12345); nothing is shipped as a
file.
"""
from __future__ import annotations

import numpy as np

from .scene import CAMERAS, MATERIAL_DTYPE, SceneData, scene_data


def value_noise(n: int, seed: int, octaves: int = 4, base_cells: int = 8) -> np.ndarray:
    """(n, n) array in [-1, 1]: sum of smoothstep-interpolated random lattices."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, n), dtype=np.float64)
    amp, total = 1.0, 0.0
    u = np.linspace(0.0, 1.0, n)
    for o in range(octaves):
        cells = base_cells << o
        lat = rng.random((cells + 1, cells + 1)) * 2.0 - 1.0
        x = u * cells
        i = np.minimum(x.astype(np.int64), cells - 1)
        f = x - i
        f = f * f * (3.0 - 2.0 * f)
        a = lat[i][:, i] * (1 - f)[None, :] + lat[i][:, i + 1] * f[None, :]
        b = lat[i + 1][:, i] * (1 - f)[None, :] + lat[i + 1][:, i + 1] * f[None, :]
        out += amp * (a * (1 - f)[:, None] + b * f[:, None])
        total += amp
        amp *= 0.5
    return out / total


def heightfield_box(cells: int = 2236, seed: int = 12345, amplitude: float = 40.0) -> SceneData:
    base = scene_data("Test")
    keep = np.ones(base.n_tris, bool)
    keep[2:4] = False  # the two floor triangles of the first model (Test.obj faces 3, 4)
    n = cells + 1
    h = value_noise(n, seed) * amplitude + amplitude  # >= 0: stays above y = 0
    xs = np.linspace(0.0, 552.8, n)
    zs = np.linspace(0.0, 559.2, n)
    hv = np.empty((n, n, 3), dtype=np.float32)
    hv[..., 0] = xs[:, None]
    hv[..., 1] = h
    hv[..., 2] = zs[None, :]
    idx = np.arange(n * n, dtype=np.uint32).reshape(n, n)
    v00, v10, v01, v11 = idx[:-1, :-1], idx[1:, :-1], idx[:-1, 1:], idx[1:, 1:]
    # wind so the CCW normal points up into the box (the reference scatters about the unflipped normal)
    t0 = np.stack([v00, v01, v11], axis=-1).reshape(-1, 3)
    t1 = np.stack([v00, v11, v10], axis=-1).reshape(-1, 3)
    htris = np.concatenate([t0, t1]) + np.uint32(base.verts.shape[0])
    verts = np.concatenate([base.verts, hv.reshape(-1, 3)])
    tris = np.concatenate([base.tris[keep], htris])
    mats = np.zeros(2, dtype=MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"][0] = 0.5
    mats["color"][1] = 1.0           # light: Kd 1, Ka 20 (Test.mtl)
    mats["emission"][1] = 20.0
    base_mat = base.mats[base.tri_material[keep]]
    tri_mat = np.concatenate([(base_mat["emission"].sum(axis=1) > 0).astype(np.uint32),
                              np.zeros(htris.shape[0], np.uint32)])
    return SceneData(verts, tris, tri_mat, mats, camera=CAMERAS["Test"], background=(0.0, 0.0, 0.0),
                     model_names=["box", "heightfield"], model_first_tri=[0, int(keep.sum())])


def tilted_mirror(normal=(0.17846472, 0.96215658, -0.2059249), roughness: float = 0.0) -> SceneData:
    """A 2 x 2 Glossy quad (Beckmann width max(roughness, 1e-5), microfacet.rs:125) with a normal that is
    NOT axis-aligned, under a large emissive quad, seen from along its normal.

    Why it exists: the reference accumulates NaN samples (main.rs:434-436) when `Beckmann::D` takes the
    acos of a dot product that rounds past 1 (microfacet.rs:161-173).  That needs a half vector within
    ~1e-8 rad of a normal whose components are not 0 / +-1: on glossy_planes.obj every plane is
    axis-aligned, h.n is then a single component of a normalised vector and can never exceed 1 (67 M oracle
    samples: no non-finite value), so the mode only shows on a tilted surface like this one.  With the
    default normal the oracle yields 11 NaN pixels at 256 x 256 x 64 spp, seed 1 (tests pin that)."""
    from .scene import Camera
    from . import _lib as L

    n = np.array(normal, np.float64)
    n /= np.linalg.norm(n)
    a = np.cross(n, [0.0, 0.0, 1.0])
    a /= np.linalg.norm(a)
    b = np.cross(n, a)
    quad = [-a - b, a - b, a + b, -a + b]
    s = 30.0
    sky = [(-s, 6.0, -s), (s, 6.0, -s), (s, 6.0, s), (-s, 6.0, s)]
    verts = np.array(quad + sky, np.float32)
    tris = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]], np.uint32)
    mats = np.zeros(2, dtype=MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"][0] = 0.9
    mats["roughness"][0] = roughness
    mats["kind"][0] = L.PT_MAT_GLOSSY
    mats["kind"][1] = L.PT_MAT_LAMBERT
    mats["emission"][1] = 1.0
    eye = n * 3.0 + a * 0.4
    view = -eye / np.linalg.norm(eye)
    cam = Camera(tuple(eye), tuple(view), (0.0, 1.0, 0.0), 2.0)
    return SceneData(verts, tris, [0, 0, 1, 1], mats, camera=cam, model_names=["mirror", "sky"], model_first_tri=[0, 2])
