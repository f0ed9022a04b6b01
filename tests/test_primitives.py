"""SURVEY.md §8f N4 — analytic Sphere / Plane (geometry/mod.rs:20-158) and the sphere as an area light
(lights.rs:37-78).  No reference scene instantiates them and the reference has no tests for them:
parity is unpinned; the oracle is checked against closed-form answers, the GPU against the oracle.
"""
import numpy as np
import pytest

import pathtracer_amd as pt
import oracle_lib as O
from test_specular import quad


def prim(kind, material, p, n=(0, 0, 0), radius=0.0):
    r = np.zeros(1, pt.PRIM_DTYPE)
    r["kind"], r["material"], r["p"], r["n"], r["radius"] = kind, material, p, n, radius
    return r


def lit_floor(rho=0.6, L=5.0, R=0.5, h=2.0, with_blocker=False):
    """A Lambertian floor (y = 0, CCW normal up) under an emissive sphere of radiance L, radius R,
    centre at height h; the camera looks straight down at the point below the sphere through a
    narrow field of view.  Radiance leaving that point = rho * L * R^2 / h^2."""
    v, tris, mat = [], [], []
    tris += quad(v, [-50, 0, -50], [-50, 0, 50], [50, 0, 50], [50, 0, -50]); mat += [0, 0]
    if with_blocker:  # a small black quad between floor and light, off the camera's line of sight
        tris += quad(v, [-0.2, 1.0, -0.2], [-0.2, 1.0, 0.2], [0.2, 1.0, 0.2], [0.2, 1.0, -0.2]); mat += [2, 2]
    mats = np.zeros(3, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"][0] = rho
    mats["emission"][1] = L
    prims = prim(pt.PT_PRIM_SPHERE, 1, (0, h, 0), radius=R)
    cam = pt.Camera((3.0, 3.0, 0.0), (-1.0, -1.0, 0.0), (0.0, 1.0, 0.0), 400.0)   # sees (0,0,0) obliquely, past the sphere
    return pt.SceneData(np.array(v, np.float32), np.array(tris, np.uint32), np.array(mat, np.uint32), mats, camera=cam,
                        prims=prims)


def mixed_scene():
    """Triangles (CornellBox) plus a mirror sphere, a glass sphere, a Lambertian sphere and a plane."""
    data = pt.scene_data("CornellBox")
    mats = np.concatenate([data.mats, np.zeros(4, pt.MATERIAL_DTYPE)])
    k = len(data.mats)
    mats["tex"][k:] = -1
    mats["kind"][k], mats["kind"][k + 1], mats["ior"][k + 1] = pt.PT_MAT_MIRROR, pt.PT_MAT_TRANSPARENT, 1.5
    mats["color"][k + 2] = (0.2, 0.5, 0.8)
    mats["color"][k + 3] = (0.7, 0.7, 0.2)
    lo, hi = data.verts.min(axis=0), data.verts.max(axis=0)
    c, e = (lo + hi) / 2, (hi - lo)
    prims = np.concatenate([
        prim(pt.PT_PRIM_SPHERE, k, (c[0] - 0.25 * e[0], lo[1] + 0.2 * e[1], c[2]), radius=0.18 * e[1]),
        prim(pt.PT_PRIM_SPHERE, k + 1, (c[0] + 0.25 * e[0], lo[1] + 0.25 * e[1], c[2] - 0.1 * e[2]), radius=0.15 * e[1]),
        prim(pt.PT_PRIM_SPHERE, k + 2, (c[0], lo[1] + 0.6 * e[1], c[2] + 0.2 * e[2]), radius=0.1 * e[1]),
        prim(pt.PT_PRIM_PLANE, k + 3, (0, lo[1] + 0.05 * e[1], 0), n=(0.0, 1.0, 0.0)),
    ])
    out = pt.SceneData(data.verts, data.tris, data.tri_material, mats, camera=data.camera, prims=prims)
    return out


def test_sphere_and_plane_closed_form_oracle():
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    prims = np.concatenate([prim(pt.PT_PRIM_SPHERE, 0, (0, 0, 0), radius=1.0),
                            prim(pt.PT_PRIM_PLANE, 0, (0, -2, 0), n=(0, 1, 0))])
    data = pt.SceneData(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), np.zeros(0, np.uint32), mats, prims=prims)
    oracle = O.OracleScene(data)
    o = np.array([[0, 0, -5.0], [0, 0, 0.0], [0, 0.6, -5.0], [0, 3.0, -5.0], [3, 3.0, -5.0], [5, 5, 5.0]])
    d = np.array([[0, 0, 1.0], [0, 0, 1.0], [0, 0, 1.0], [0, 0, 1.0], [0, -0.6, 0.8], [0, 1.0, 0.0]])
    tri, t = oracle.trace_closest(pt.make_rays(o, d))
    S, P = pt.PT_PRIM_BASE, pt.PT_PRIM_BASE + 1
    assert tri.tolist() == [S, S, S, pt.PT_MISS, P, pt.PT_MISS]
    assert t[0] == 4.0 and t[1] == 1.0                       # outside: near root; centre: far root (near < t_min)
    assert np.isclose(t[2], 5 - 0.8, rtol=1e-15)             # chord at height 0.6
    assert np.isclose(t[4], 5 / 0.6, rtol=1e-15)             # plane y = -2 from y = 3 along (0,-.6,.8)
    rays = pt.make_rays(o[:1], d[:1])
    rays["t_max"] = 3.9                                      # strict t < t_max on spheres, t <= t_max on planes
    assert oracle.trace_closest(rays)[0][0] == pt.PT_MISS
    rays = pt.make_rays(o[4:5], d[4:5])
    rays["t_max"] = t[4]
    assert oracle.trace_closest(rays)[0][0] == P


@pytest.mark.parametrize("nee", [False, True])
def test_sphere_light_irradiance_oracle(nee):
    rho, L, R, h = 0.6, 5.0, 0.5, 2.0
    oracle = O.OracleScene(lit_floor(rho, L, R, h))
    n = 4096
    img = oracle.render(pt.Config(dim=8, samples_pr_pixel=n, bounces=1, seed=3, nee=nee), threads=16) / n
    expect = rho * L * R * R / (h * h)
    assert abs(img.mean() / expect - 1) < (0.01 if nee else 0.04)      # sigma ~ 0.25 % with NEE, ~ 1 % without
    assert np.allclose(img[..., 0], img[..., 1]) and np.allclose(img[..., 0], img[..., 2])


def test_prims_scene_validation():
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["emission"] = 1.0
    empty = dict(verts=np.zeros((0, 3), np.float32), tris=np.zeros((0, 3), np.uint32), tri_material=np.zeros(0, np.uint32))
    with pytest.raises(pt.PtError, match="emissive plane"):
        pt.Scene(pt.SceneData(mats=mats, prims=prim(pt.PT_PRIM_PLANE, 0, (0, 0, 0), n=(0, 1, 0)), **empty))
    with pytest.raises(pt.PtError, match="radius"):
        pt.Scene(pt.SceneData(mats=mats, prims=prim(pt.PT_PRIM_SPHERE, 0, (0, 0, 0), radius=0.0), **empty))
    with pytest.raises(pt.PtError, match="material"):
        pt.Scene(pt.SceneData(mats=mats, prims=prim(pt.PT_PRIM_SPHERE, 3, (0, 0, 0), radius=1.0), **empty))


# ---------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_prims_hits_bit_exact(gpu, monkeypatch):
    from test_gpu_parity import bits, random_rays, FORCE_VOTE
    data = mixed_scene()
    oracle = O.OracleScene(data)
    scenes = [pt.Scene(data, gpu)]
    for k, v in FORCE_VOTE.items():
        monkeypatch.setenv(k, v)
    scenes.append(pt.Scene(data, gpu))                    # the vote-scheduled traversal sees the primitives too
    for k in FORCE_VOTE:
        monkeypatch.delenv(k)
    rays = random_rays(data, 30000, seed=2)
    rtri, rt = oracle.trace_closest(rays, use_bvh=False)
    assert (rtri >= pt.PT_PRIM_BASE).sum() > 3000 and (rtri < data.n_tris).sum() > 3000
    for s in scenes:
        tri, t = s.trace_closest(rays)
        assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    prim_rays = oracle.generate_rays(pt.Config(dim=128, seed=1), 0)
    tri, t = scenes[0].trace_closest(prim_rays)
    rtri, rt = oracle.trace_closest(prim_rays)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    # primitives alone, no triangle and no tree
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    only = pt.SceneData(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), np.zeros(0, np.uint32), mats,
                        prims=np.concatenate([prim(pt.PT_PRIM_SPHERE, 0, (0, 0, 0), radius=1.0),
                                              prim(pt.PT_PRIM_PLANE, 0, (0, -2, 0), n=(0, 1, 0))]))
    rng = np.random.default_rng(0)
    o = rng.uniform(-4, 4, (5000, 3))
    d = rng.normal(size=(5000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = pt.make_rays(o, d)
    tri, t = pt.Scene(only, gpu).trace_closest(rays)
    rtri, rt = O.OracleScene(only).trace_closest(rays)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    assert set(np.unique(tri).tolist()) == {pt.PT_PRIM_BASE, pt.PT_PRIM_BASE + 1, pt.PT_MISS}


@pytest.mark.gpu
def test_gpu_prims_render_matches_oracle(gpu, monkeypatch):
    data = mixed_scene()
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    for nee in (False, True):
        cfg = pt.Config(dim=96, samples_pr_pixel=4, bounces=4, seed=7, nee=nee)
        img, ref = scene.render(cfg), oracle.render(cfg)
        assert np.isfinite(img).all() and ref.mean() > 0.01
        assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 2e-3
    monkeypatch.setenv("PT_FUSE", "0")
    unfused = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_FUSE")
    cfg = pt.Config(width=70, height=45, samples_pr_pixel=3, bounces=3, seed=1, nee=True)
    assert scene.render(cfg).tobytes() == unfused.render(cfg).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("blocker", [False, True])
def test_gpu_sphere_light_matches_oracle_and_closed_form(gpu, blocker):
    rho, L, R, h = 0.6, 5.0, 0.5, 2.0
    data = lit_floor(rho, L, R, h, with_blocker=blocker)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    cfg = pt.Config(dim=8, samples_pr_pixel=64, bounces=1, seed=3, nee=True)
    img, ref = scene.render(cfg) / 64, oracle.render(cfg) / 64
    assert np.allclose(img, ref, rtol=1e-9, atol=1e-12)
    if not blocker:
        big = scene.render(pt.Config(dim=8, samples_pr_pixel=4096, bounces=1, seed=5, nee=True)) / 4096
        assert abs(big.mean() / (rho * L * R * R / (h * h)) - 1) < 1e-2        # sigma ~ 0.25 %
    else:
        assert img.mean() < 0.9 * rho * L * R * R / (h * h)       # the blocker's shadow: any-hit rays are stopped by it
