"""PT_FLAG_REFERENCE_LIGHTS — the reference's dormant direct-light path as written (lights.rs:42-62 AreaLight::sample,
:70-78 Sphere::sample_surface, :110-137 EstimateDirect).

The reference never populates Scene.lights (scene.rs:189) and EstimateDirect has no caller, so the reference holds no
vector for this path: parity unpinned.  The oracle restates the three functions literally and is checked here against
what that arithmetic converges to, worked out independently by quadrature; the GPU is checked against the oracle.

What it converges to.  A point is drawn uniformly on the light's sphere (area measure) but the pdf handed back is the
constant 1 / (4 pi) — a pdf over the unit sphere of directions u, whatever the radius.  With irradiance = I / |d|^2 and
a Lambertian BRDF rho / pi the estimate at a surface point x is

    E[Ld] = (1 / 4 pi) Int_{S^2} [u . (-w) >= 0] (rho / pi) |w . n| (I / |d|^2) / (1 / 4 pi) du,
    d = c + R u - x,  w = d / |d|

i.e. the part of the sphere that faces x, weighted by 4 rho |cos| I / |d|^2 — by area, where a Lambertian sphere of
radiance L weights it by L R^2 cos_l (cos_l at the light).  Its bias against the naive loop (the same scene with the
sphere as an emitter of radiance L, main.rs:283-318), measured below:

  * unoccluded, sphere wholly above the receiver's horizon: none, for L = 2 I / R^2.  Both integrals are
    2 rho cos(theta) I / h^2 = rho L R^2 cos(theta) / h^2 at every radius (the part of a uniformly charged shell that
    faces a point exerts half the shell's force on it), so there the estimator is an unbiased one of that emitter.
  * partly occluded: biased.  The missing cos_l makes the rim of the sphere count as much as its centre, so a blocker
    under the light removes too little: the floor test below keeps 44 % of the open value where the emitter keeps 24 %.
  * |w . n| (lights.rs:132) also counts light arriving from below the receiver's horizon, and the light sphere itself is
    neither seen by camera rays nor hit by BRDF-sampled rays, nor does it shadow anything.
"""
import numpy as np
import pytest

import pathtracer_amd as pt
import oracle_lib as O
from test_specular import quad


def floor_under_light(rho=0.6, I=3.0, R=0.5, h=2.0, with_blocker=False):
    """A Lambertian floor (y = 0) under AreaLight(centre (0, h, 0), radius R, intensity I); the camera looks at the
    origin through a narrow field of view."""
    v, tris, mat = [], [], []
    tris += quad(v, [-50, 0, -50], [-50, 0, 50], [50, 0, 50], [50, 0, -50]); mat += [0, 0]
    if with_blocker:  # a black quad between floor and light, off the camera's line of sight
        tris += quad(v, [-0.2, 1.0, -0.2], [-0.2, 1.0, 0.2], [0.2, 1.0, 0.2], [0.2, 1.0, -0.2]); mat += [1, 1]
    mats = np.zeros(2, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"][0] = rho
    cam = pt.Camera((3.0, 3.0, 0.0), (-1.0, -1.0, 0.0), (0.0, 1.0, 0.0), 400.0)
    return pt.SceneData(np.array(v, np.float32), np.array(tris, np.uint32), np.array(mat, np.uint32), mats, camera=cam,
                        area_light=((0.0, h, 0.0), R, (I, I, I)))


def expected_direct(rho, I, R, h, n=1500):
    """The integral of the module docstring at x = origin, n = +y, by the midpoint rule in (z, phi)."""
    z = 1.0 - 2.0 * (np.arange(n) + 0.5) / n
    phi = 2.0 * np.pi * (np.arange(n) + 0.5) / n
    zz, pp = np.meshgrid(z, phi, indexing="ij")
    rr = np.sqrt(1.0 - zz * zz)
    u = np.stack([rr * np.cos(pp), rr * np.sin(pp), zz], axis=-1)
    d = np.array([0.0, h, 0.0]) + R * u
    d2 = (d * d).sum(axis=-1)
    w = d / np.sqrt(d2)[..., None]
    front = (u * -w).sum(axis=-1) >= 0.0
    return float((front * 4.0 * rho * np.abs(w[..., 1]) * I / d2).mean())


def test_reference_lights_converge_to_the_stated_integral_oracle():
    rho, I, h = 0.6, 3.0, 2.0
    n = 2048
    for R in (0.5, 0.01):
        oracle = O.OracleScene(floor_under_light(rho, I, R, h))
        cfg = pt.Config(dim=8, samples_pr_pixel=n, bounces=1, seed=3, reference_lights=True)
        img = oracle.render(cfg, threads=16) / n
        expect = expected_direct(rho, I, R, h)
        assert abs(img.mean() / expect - 1) < 0.01, (R, img.mean(), expect)     # sigma ~ 0.3 %
        assert np.allclose(img[..., 0], img[..., 1]) and np.allclose(img[..., 0], img[..., 2])
    assert abs(expected_direct(rho, I, 0.01, h) / (2 * rho * I / (h * h)) - 1) < 1e-3   # the small-light limit
    # the flag without a light, and a light without the flag, change nothing
    plain = O.OracleScene(floor_under_light(rho, I, 0.5, h))
    assert not plain.render(pt.Config(dim=8, samples_pr_pixel=4, bounces=1, seed=3)).any()


def test_reference_lights_are_not_geometry_and_cast_shadows_oracle():
    rho, I, R, h = 0.6, 3.0, 0.5, 2.0
    data = floor_under_light(rho, I, R, h)
    oracle = O.OracleScene(data)
    # a ray through the light's sphere sees nothing: Scene.lights is not Scene.objects (scene.rs:14-15)
    tri, _ = oracle.trace_closest(pt.make_rays(np.array([[0.0, h, -5.0]]), np.array([[0.0, 0.0, 1.0]])))
    assert tri[0] == pt.PT_MISS
    n = 512
    cfg = pt.Config(dim=8, samples_pr_pixel=n, bounces=1, seed=5, reference_lights=True)
    open_ = oracle.render(cfg, threads=16).mean() / n
    shadowed = O.OracleScene(floor_under_light(rho, I, R, h, with_blocker=True)).render(cfg, threads=16).mean() / n
    assert 0.05 * open_ < shadowed < 0.9 * open_          # the blocker hides part of the sphere from the origin


def test_reference_lights_bias_against_the_naive_loop_oracle():
    # The same floor under (a) the AreaLight of intensity I and (b) an emissive sphere of radiance L = 2 I / R^2 that
    # rays can hit (lights.rs vs the naive loop / physically weighted NEE of main.rs:283-318).
    from test_primitives import lit_floor
    rho, L, R, h = 0.6, 5.0, 0.5, 2.0
    I = L * R * R / 2
    n = 2048
    ref_cfg = pt.Config(dim=8, samples_pr_pixel=n, bounces=1, seed=5, reference_lights=True)
    nee_cfg = pt.Config(dim=8, samples_pr_pixel=n, bounces=1, seed=5, nee=True)
    naive_cfg = pt.Config(dim=8, samples_pr_pixel=4 * n, bounces=1, seed=5)
    mean = lambda data, cfg: O.OracleScene(data).render(cfg, threads=16).mean() / cfg.samples_pr_pixel
    closed = rho * L * R * R / (h * h)
    open_ref, open_nee, open_naive = (mean(floor_under_light(rho, I, R, h), ref_cfg), mean(lit_floor(rho, L, R, h), nee_cfg),
                                      mean(lit_floor(rho, L, R, h), naive_cfg))
    assert abs(open_ref / closed - 1) < 0.01 and abs(open_nee / closed - 1) < 0.01 and abs(open_naive / closed - 1) < 0.04
    for radius in (0.25, 1.0, 1.5):                       # ... at every radius, by quadrature
        assert abs(expected_direct(rho, L * radius * radius / 2, radius, h) / (rho * L * radius * radius / (h * h)) - 1) < 1e-3
    shadow_ref = mean(floor_under_light(rho, I, R, h, with_blocker=True), ref_cfg)
    shadow_nee = mean(lit_floor(rho, L, R, h, with_blocker=True), nee_cfg)
    assert 0.40 < shadow_ref / open_ref < 0.49            # measured 0.444
    assert 0.21 < shadow_nee / open_nee < 0.27            # measured 0.239: the emitter loses its bright centre


def lit_box(name, **kw):
    """A reference scene with an AreaLight hung inside it, below the ceiling."""
    data = pt.scene_data(name, **kw)
    lo, hi = data.verts.min(axis=0), data.verts.max(axis=0)
    c, e = (lo + hi) / 2, hi - lo
    data.area_light = ((c[0], lo[1] + 0.8 * e[1], c[2]), 0.08 * e[1], (0.3 * e[1] ** 2, 0.25 * e[1] ** 2, 0.2 * e[1] ** 2))
    return data


# ---------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("CornellBox", {}), ("Test", {"textures": True}), ("glossy_planes", {"glossy": True})])
def test_gpu_reference_lights_match_oracle(gpu, name, kw, monkeypatch):
    data = lit_box(name, **kw)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    for bounces in (1, 3):
        cfg = pt.Config(dim=96, samples_pr_pixel=3, bounces=bounces, seed=7, reference_lights=True, drop_nonfinite=True)
        img, ref = scene.render(cfg), oracle.render(cfg)
        off = oracle.render(pt.Config(dim=96, samples_pr_pixel=3, bounces=bounces, seed=7, drop_nonfinite=True))
        assert np.isfinite(img).all() and (ref - off).mean() > 1e-3            # the light adds to the image
        assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 5e-3, (name, bounces)
    # the kernel-per-stage pipeline and the fused one add the same contributions in the same order
    monkeypatch.setenv("PT_FUSE", "0")
    unfused = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_FUSE")
    cfg = pt.Config(width=70, height=45, samples_pr_pixel=3, bounces=3, seed=1, reference_lights=True, drop_nonfinite=True)
    assert scene.render(cfg).tobytes() == unfused.render(cfg).tobytes()
    with pytest.raises(pt.PtError, match="exclude"):
        scene.render(pt.Config(dim=16, samples_pr_pixel=1, reference_lights=True, nee=True))


@pytest.mark.gpu
def test_gpu_reference_lights_closed_form_and_shadow(gpu):
    rho, I, R, h = 0.6, 3.0, 0.5, 2.0
    n = 4096
    cfg = pt.Config(dim=8, samples_pr_pixel=n, bounces=1, seed=11, reference_lights=True)
    img = pt.Scene(floor_under_light(rho, I, R, h), gpu).render(cfg) / n
    assert abs(img.mean() / expected_direct(rho, I, R, h) - 1) < 0.01            # sigma ~ 0.2 %
    data = floor_under_light(rho, I, R, h, with_blocker=True)
    cfg = pt.Config(dim=8, samples_pr_pixel=64, bounces=1, seed=3, reference_lights=True)
    a, b = pt.Scene(data, gpu).render(cfg), O.OracleScene(data).render(cfg)
    assert np.allclose(a, b, rtol=1e-9, atol=1e-12) and a.mean() < 0.9 * 64 * expected_direct(rho, I, R, h)
    # without the flag the light is ignored; with the flag and no light nothing changes either
    assert not pt.Scene(data, gpu).render(pt.Config(dim=8, samples_pr_pixel=2, bounces=1, seed=3)).any()
    bare = pt.scene_data("CornellBox")
    s = pt.Scene(bare, gpu)
    c0 = pt.Config(dim=32, samples_pr_pixel=2, bounces=2, seed=3)
    c1 = pt.Config(dim=32, samples_pr_pixel=2, bounces=2, seed=3, reference_lights=True)
    assert s.render(c0).tobytes() == s.render(c1).tobytes()
