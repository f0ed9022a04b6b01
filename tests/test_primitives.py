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
    mats["tex"] = 0                                           # a textured primitive needs the texture to exist
    with pytest.raises(pt.PtError, match="texture index"):
        pt.Scene(pt.SceneData(mats=mats, prims=prim(pt.PT_PRIM_SPHERE, 0, (0, 0, 0), radius=1.0), **empty))


def test_sphere_surface_derivatives_known_answers():
    # geometry/mod.rs:76-132.  On a sphere n = (p - c) / R, so the Weingarten derivatives the reference builds
    # from the fundamental forms must come out as dn/du = dp/du / R and dn/dv = dp/dv / R; dp/du, dp/dv are
    # tangent, of lengths 2 pi sqrt(x^2 + y^2) and pi R.
    rng = np.random.default_rng(3)
    c, R = np.array([0.3, -1.2, 2.0]), 1.7
    for _ in range(200):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        p = c + R * n
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if d @ n > -0.2:
            d = -d if (-d) @ n < -0.2 else -n
        o = p - 3.0 * d
        s = O.sphere_surface(c, R, o, d, 3.0)
        hl = s["position"] - c
        assert np.allclose(s["normal"], hl / np.linalg.norm(hl), rtol=0, atol=1e-15)
        assert abs(s["dpdu"] @ s["normal"]) < 1e-12 and abs(s["dpdv"] @ s["normal"]) < 1e-12
        assert np.isclose(np.linalg.norm(s["dpdu"]), 2 * np.pi * np.hypot(hl[0], hl[1]), rtol=1e-13)
        assert np.isclose(np.linalg.norm(s["dpdv"]), np.pi * R, rtol=1e-12)
        assert np.allclose(s["dndu"], s["dpdu"] / R, rtol=1e-9, atol=1e-11)
        assert np.allclose(s["dndv"], s["dpdv"] / R, rtol=1e-9, atol=1e-11)
    # ray differentials (surface.rs:59-109): a ray offset by a small tangent step e moves the hit by e, and the
    # (u, v) derivatives solve dpdx = dpdu dudx + dpdv dvdx
    o, d = np.array([0.3, -1.2, -4.0]) + np.array([0.4, 0.3, 0.0]), np.array([0.0, 0.0, 1.0])
    hl = np.array([0.4, 0.3, -np.sqrt(R * R - 0.25)])
    t = (c + hl - o)[2]
    e = 1e-3
    s = O.sphere_surface(c, R, o, d, t, rx_origin=o + [e, 0, 0], ry_origin=o + [0, e, 0], rx_direction=d, ry_direction=d)
    dpdx = s["dpdu"] * s["dudx"] + s["dpdv"] * s["dvdx"]
    dpdy = s["dpdu"] * s["dudy"] + s["dpdv"] * s["dvdy"]
    # the offset rays meet the tangent plane at position + (e - (e . n) n / (d . n) d): check its tangent part
    nrm = s["normal"]
    for dp, step in ((dpdx, np.array([e, 0, 0])), (dpdy, np.array([0, e, 0]))):
        hit = step - (step @ nrm) / (d @ nrm) * d
        assert np.allclose(dp, hit, rtol=0, atol=1e-12)
    # a ray without differentials (Ray::new, main.rs:314): all four derivatives are zero (surface.rs:72-80)
    s0 = O.sphere_surface(c, R, o, d, t)
    assert s0["dudx"] == s0["dvdx"] == s0["dudy"] == s0["dvdy"] == 0.0


def textured_prims_scene():
    """The textured Cornell box of BASELINE configs[3] plus a sphere and a plane wearing its texture, a
    mirror sphere and a glass sphere (whose spawned rays need dndu / dndv)."""
    data = pt.scene_data("Test", textures=True)
    tex_mat = int(np.flatnonzero(data.mats["tex"] >= 0)[0])
    mats = np.concatenate([data.mats, np.zeros(2, pt.MATERIAL_DTYPE)])
    k = len(data.mats)
    mats["tex"][k:] = -1
    mats["kind"][k], mats["kind"][k + 1], mats["ior"][k + 1] = pt.PT_MAT_MIRROR, pt.PT_MAT_TRANSPARENT, 1.5
    lo, hi = data.verts.min(axis=0), data.verts.max(axis=0)
    c, e = (lo + hi) / 2, (hi - lo)
    prims = np.concatenate([
        prim(pt.PT_PRIM_SPHERE, tex_mat, (c[0], lo[1] + 0.55 * e[1], c[2] + 0.1 * e[2]), radius=0.16 * e[1]),
        prim(pt.PT_PRIM_SPHERE, k, (c[0] - 0.28 * e[0], lo[1] + 0.2 * e[1], c[2]), radius=0.17 * e[1]),
        prim(pt.PT_PRIM_SPHERE, k + 1, (c[0] + 0.28 * e[0], lo[1] + 0.22 * e[1], c[2] + 0.15 * e[2]), radius=0.15 * e[1]),
        prim(pt.PT_PRIM_PLANE, tex_mat, (0, lo[1] + 0.03 * e[1], 0), n=(0.0, 1.0, 0.0)),
    ])
    return pt.SceneData(data.verts, data.tris, data.tri_material, mats, tri_uvs=data.tri_uvs, textures=data.textures,
                        camera=data.camera, prims=prims)


def lone_textured_sphere(distance):
    """One sphere wearing a noise texture under a white sky, seen from `distance`: no triangle at all."""
    rng = np.random.default_rng(11)
    tex = rng.integers(0, 256, (64, 64, 3)).astype(np.uint8)
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["tex"] = 0
    cam = pt.Camera((0.0, 0.0, -distance), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 2.0)
    data = pt.SceneData(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), np.zeros(0, np.uint32), mats,
                        textures=[tex], camera=cam, prims=prim(pt.PT_PRIM_SPHERE, 0, (0, 0, 0), radius=1.0))
    data.background = (1.0, 1.0, 1.0)
    return data


def test_textured_sphere_level_of_detail_oracle():
    # A textured sphere looks its texture up at uv (0, 0) (Surface::default; Sphere::intersect never sets uv) at
    # the level its own derivatives give.  Without ray differentials that is level 0 everywhere: one colour.
    # With them the level rises with the footprint: from far away the lookup is higher up the pyramid, where a
    # noise texture has averaged towards grey.
    # (The level-0 colour may be negative: at uv (0, 0) the reference's triangle filter, mipmap.rs, reads texel
    # -1 as texel 0 with weight 1.5 and -0.5; kept as it is.)
    cfg = dict(dim=48, samples_pr_pixel=1, bounces=1, seed=1)
    on_sphere = lambda img: np.abs(img - 1.0).max(axis=2) > 1e-9      # the sky is exactly 1
    flat = O.OracleScene(lone_textured_sphere(4.0)).render(pt.Config(use_diffs=False, **cfg))
    on = on_sphere(flat)
    assert on.sum() > 100 and len(np.unique(flat[on].round(12), axis=0)) == 1
    near = O.OracleScene(lone_textured_sphere(4.0)).render(pt.Config(use_diffs=True, **cfg))
    far = O.OracleScene(lone_textured_sphere(60.0)).render(pt.Config(use_diffs=True, **cfg))
    assert np.isfinite(near).all() and np.isfinite(far).all()
    assert len(np.unique(near[on].round(12), axis=0)) > 100           # the level varies over the sphere
    top = far[on_sphere(far)]
    assert len(top) >= 1 and np.abs(top - 0.5).max() < 0.05           # near the top of the pyramid: noise averaged to grey


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


@pytest.mark.gpu
def test_gpu_textured_and_curved_prims_match_oracle(gpu):
    # geometry/mod.rs:76-145 on the device: texture level of detail from the sphere's dpdu / dpdv, and, under
    # PT_FLAG_PROPAGATE_DIFFS, the dndx / dndy terms of reflect_diff / refract_diff (main.rs:210-221, 268-275)
    # at the mirror and glass spheres.  Tolerance: acos / sin / atan-free, but ocml and glibc differ in the last
    # place, and a texture level boundary can then fall differently: 99.5 % of values within 1e-8 relative.
    data = textured_prims_scene()
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    for kw in (dict(bounces=0), dict(bounces=3), dict(bounces=4, propagate_diffs=True, diff_size=0.02),
               dict(bounces=3, propagate_diffs=True, diff_size=0.02, nee=True), dict(bounces=2, use_diffs=False)):
        cfg = pt.Config(dim=80, samples_pr_pixel=3, seed=5, drop_nonfinite=True, **kw)
        img, ref = scene.render(cfg), oracle.render(cfg)
        assert np.isfinite(img).all() and ref.mean() > 0.01
        assert (np.abs(img - ref) > 1e-8 * np.maximum(1.0, np.abs(ref))).mean() < 5e-3, kw
    for dist in (4.0, 20.0):                                   # primitives alone: the level of detail over one sphere
        lone = lone_textured_sphere(dist)
        cfg = pt.Config(dim=96, samples_pr_pixel=2, bounces=2, seed=2)
        img, ref = pt.Scene(lone, gpu).render(cfg), O.OracleScene(lone).render(cfg)
        assert (np.abs(img - ref) > 1e-8 * np.maximum(1.0, np.abs(ref))).mean() < 5e-3, dist
    # the propagated differentials matter: the image with them differs from the one without
    with_d = oracle.render(pt.Config(dim=80, samples_pr_pixel=3, seed=5, drop_nonfinite=True, bounces=4,
                                     propagate_diffs=True, diff_size=0.02))
    without = oracle.render(pt.Config(dim=80, samples_pr_pixel=3, seed=5, drop_nonfinite=True, bounces=4))
    assert np.abs(with_d - without).max() > 1e-3
