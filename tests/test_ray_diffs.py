"""SURVEY.md §8f N3, last item — ray-differential propagation of the legacy integrator
(`reflect_diff` / `refract_diff`, main.rs:192-281, commented out in the reference) as
PT_FLAG_PROPAGATE_DIFFS: the ray spawned at a vertex carries differential rays, so a texture seen
after a bounce is filtered like one seen directly.  No reference vectors exist (the code never
compiled there): parity unpinned; the oracle is checked against closed forms, the GPU against the
oracle.
"""
import numpy as np
import pytest

import pathtracer_amd as pt
import oracle_lib as O
from test_specular import quad


def unit(v):
    v = np.asarray(v, float)
    return v / np.linalg.norm(v)


def test_spawned_differentials_known_answers_oracle():
    n = unit([0.1, 0.2, 1.0])
    pos, dpdx, dpdy = np.array([1.0, 2.0, 3.0]), np.array([1e-3, 2e-3, 0.0]), np.array([0.0, -1e-3, 3e-3])
    d = unit([0.3, -0.2, -1.0])
    rx, ry = unit(d + [2e-4, 0, 0]), unit(d + [0, 2e-4, 0])
    lin_reflect = lambda v: v - 2 * np.dot(v, n) * n
    # mirror: the differential directions are the mirror images of the incident ones (the transform
    # is linear), origins are the hit point moved by dpdx / dpdy
    out = O.spawn_diffs(pos, n, dpdx, dpdy, d, rx, ry, lin_reflect(d), pt.PT_MAT_MIRROR, 1)
    assert np.allclose(out[0], pos + dpdx, rtol=0, atol=0) and np.allclose(out[1], pos + dpdy, rtol=0, atol=0)
    assert np.abs(out[2] - lin_reflect(rx)).max() < 1e-15 and np.abs(out[3] - lin_reflect(ry)).max() < 1e-15
    # seen from inside (normal flipped, main.rs:260-265) the answer is the same
    out_b = O.spawn_diffs(pos, -n, dpdx, dpdy, d, rx, ry, lin_reflect(d), pt.PT_MAT_MIRROR, 1)
    assert np.abs(out_b[2:] - out[2:]).max() < 1e-15
    # refraction: main.rs:216-221 as written, which is pbrt-v3's transmitted differential.  It is
    # not the first-order expansion of Vec3::refract (that one reads wi - eta dwodx + dmudx n; pbrt
    # corrected it later) - the quirk is the reference's, so it is kept and pinned here.
    for ior, nn in ((1.5, n), (1.5, -n), (1.1, n)):
        cos_theta = np.dot(-d, nn)
        eta = ior if cos_theta < 0 else 1 / ior
        face = -nn if cos_theta < 0 else nn
        wi = O.refract(d, face, eta)
        out = O.spawn_diffs(pos, nn, dpdx, dpdy, d, rx, ry, wi, pt.PT_MAT_TRANSPARENT, 2, ior=ior)
        for k, r in ((2, rx), (3, ry)):
            dwo = -r + d
            dmu = (eta - eta * eta * np.dot(d, face) / np.dot(wi, face)) * np.dot(dwo, face)
            assert np.abs(out[k] - (wi + eta * dwo - dmu * face)).max() < 1e-15
            exact = O.refract(r, face, eta)
            assert np.abs((wi - eta * dwo + dmu * face) - exact).max() < 5e-8   # second order in |r - d| = 2e-4
            assert 1e-5 < np.abs(out[k] - exact).max() < 1e-3                    # the quirk, same magnitude
    # Lambertian / Glossy: a cone of half-width diff_size around the sampled direction
    wi = unit([0.5, 0.4, 0.3])
    out = O.spawn_diffs(pos, n, dpdx, dpdy, d, rx, ry, wi, pt.PT_MAT_LAMBERT, 0, diff_size=0.0)
    assert np.abs(out[2] - wi).max() < 1e-15 and np.abs(out[3] - wi).max() < 1e-15
    out = O.spawn_diffs(pos, n, dpdx, dpdy, d, rx, ry, wi, pt.PT_MAT_LAMBERT, 0, diff_size=0.1)
    c = 1 / np.sqrt(1.01)
    assert np.isclose(out[2] @ wi, c, rtol=1e-14) and np.isclose(out[3] @ wi, c, rtol=1e-14)
    assert abs((out[2] - c * wi) @ (out[3] - c * wi)) < 1e-16     # the two tangents are orthogonal
    # roughdiff scales the cone by roughness / 0.8 on Glossy only (main.rs:249-253)
    g = O.spawn_diffs(pos, n, dpdx, dpdy, d, rx, ry, wi, pt.PT_MAT_GLOSSY, 0, roughness=0.4, diff_size=0.1,
                      flags=pt.PT_FLAG_ROUGHDIFF)
    assert np.isclose(g[2] @ wi, 1 / np.sqrt(1 + 0.05 ** 2), rtol=1e-14)
    l = O.spawn_diffs(pos, n, dpdx, dpdy, d, rx, ry, wi, pt.PT_MAT_LAMBERT, 0, roughness=0.4, diff_size=0.1,
                      flags=pt.PT_FLAG_ROUGHDIFF)
    assert np.isclose(l[2] @ wi, c, rtol=1e-14)


def noise_texture(n=128, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, n, 3), dtype=np.uint8)   # white noise: every mip level differs


def wall_uv():
    return [[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]]


def mirrored_wall_scene(with_mirror):
    """Eye at z = 5 looks down -z at a huge mirror in the plane z = 0, which shows a textured wall
    at z = 12 behind the eye.  The wall's geometric normal is +z, away from the mirror, so what the
    wall scatters (the reference samples around the unflipped normal, main.rs:310) always escapes to
    the white background: a pixel is the filtered texture colour itself.  Without the mirror the wall
    stands at its mirror image z = -12 and is seen directly."""
    v, tris, mat, uvs = [], [], [], []
    z = 12.0 if with_mirror else -12.0
    tris += quad(v, [-8, -8, z], [8, -8, z], [8, 8, z], [-8, 8, z]); mat += [0, 0]; uvs += wall_uv()
    if with_mirror:
        tris += quad(v, [-400, -400, 0], [400, -400, 0], [400, 400, 0], [-400, 400, 0]); mat += [1, 1]; uvs += wall_uv()
    mats = np.zeros(2, pt.MATERIAL_DTYPE)
    mats["tex"] = [0, -1]
    mats["color"][0] = 1.0
    mats["kind"][1] = pt.PT_MAT_MIRROR
    cam = pt.Camera((0.5, -0.25, 5.0), (0.12, 0.07, -1.0), (0.0, 1.0, 0.0), 3.0)
    return pt.SceneData(np.array(v, np.float32), np.array(tris, np.uint32), np.array(mat, np.uint32), mats,
                        tri_uvs=np.array(uvs, np.float32), textures=[noise_texture()], camera=cam,
                        background=(1.0, 1.0, 1.0))


def test_texture_seen_in_a_mirror_is_filtered_like_the_unfolded_view_oracle():
    folded, unfolded = O.OracleScene(mirrored_wall_scene(True)), O.OracleScene(mirrored_wall_scene(False))
    cfg = dict(dim=40, samples_pr_pixel=2, bounces=2, seed=3)
    direct = unfolded.render(pt.Config(**cfg))
    seen = folded.render(pt.Config(**cfg, propagate_diffs=True))
    sharp = folded.render(pt.Config(**cfg))
    on_wall = (direct != 2.0).any(axis=2)                 # 2 spp of background = 2
    assert on_wall.mean() > 0.5
    # the mirror unfolds exactly: same footprint, same mip level, same colour
    assert np.abs(seen - direct)[on_wall].max() < 1e-9
    # the live render() (no propagation) looks the finest level up instead: noisy, not the same image
    assert np.abs(sharp - direct)[on_wall].max() > 0.2
    assert sharp[on_wall].std() > 2 * seen[on_wall].std()
    # a scene without textures ignores the flag bit for bit
    plain = O.OracleScene(pt.scene_data("CornellBox"))
    c = dict(dim=24, samples_pr_pixel=2, seed=5)
    assert plain.render(pt.Config(**c, propagate_diffs=True)).tobytes() == plain.render(pt.Config(**c)).tobytes()


# ---------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_mirror_unfolding_matches_oracle(gpu):
    data = mirrored_wall_scene(True)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    cfg = pt.Config(dim=64, samples_pr_pixel=3, bounces=2, seed=3, propagate_diffs=True)
    img, ref = scene.render(cfg), oracle.render(cfg)
    # mirror + trilinear lookup: the only transcendental is the log2 of the level selection
    assert np.abs(img - ref).max() < 1e-11
    off = scene.render(pt.Config(dim=64, samples_pr_pixel=3, bounces=2, seed=3))
    assert np.abs(off - img).max() > 0.2
    assert off.tobytes() == oracle.render(pt.Config(dim=64, samples_pr_pixel=3, bounces=2, seed=3)).tobytes() or \
        np.abs(off - oracle.render(pt.Config(dim=64, samples_pr_pixel=3, bounces=2, seed=3))).max() < 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("nee", [False, True])
def test_gpu_propagated_differentials_match_oracle_on_test_scene(gpu, nee):
    # Test.obj with its textured back wall, two mirror walls (illum 3) and Lambertian rest: every
    # spawn rule except refraction
    data = pt.scene_data("Test", specular=True, textures=True)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    for diff_size, rough in ((0.0, False), (0.05, True)):
        cfg = pt.Config(dim=80, samples_pr_pixel=3, bounces=4, seed=6, nee=nee, propagate_diffs=True,
                        diff_size=diff_size, roughdiff=rough)
        img, ref = scene.render(cfg), oracle.render(cfg)
        assert np.isfinite(img).all() and ref.mean() > 0.01
        assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 2e-3
    base = oracle.render(pt.Config(dim=80, samples_pr_pixel=3, bounces=4, seed=6, nee=nee))
    assert np.abs(base - ref).max() > 1e-3                  # propagation does change the image
    # without USE_DIFFS there is nothing to propagate: bit-identical to the plain render
    a = scene.render(pt.Config(dim=48, samples_pr_pixel=2, seed=2, use_diffs=False, propagate_diffs=True, nee=nee))
    b = scene.render(pt.Config(dim=48, samples_pr_pixel=2, seed=2, use_diffs=False, nee=nee))
    assert a.tobytes() == b.tobytes()


@pytest.mark.gpu
def test_gpu_refracted_and_glossy_differentials_match_oracle(gpu):
    # a glass pane and a glossy pane in front of the textured wall: refract_diff and the roughdiff
    # spread, GPU against the oracle
    v, tris, mat, uvs = [], [], [], []
    tris += quad(v, [-8, -8, -12], [8, -8, -12], [8, 8, -12], [-8, 8, -12]); mat += [0, 0]; uvs += wall_uv()
    tris += quad(v, [-3, -3, 0], [0, -3, 0.4], [0, 3, 0.4], [-3, 3, 0]); mat += [1, 1]; uvs += wall_uv()     # glass, tilted
    tris += quad(v, [0, -3, -1.0], [3, -3, 0.5], [3, 3, 0.5], [0, 3, -1.0]); mat += [2, 2]; uvs += wall_uv()  # glossy, tilted
    mats = np.zeros(3, pt.MATERIAL_DTYPE)
    mats["tex"] = [0, -1, -1]
    mats["color"][0] = 1.0
    mats["kind"][1], mats["ior"][1] = pt.PT_MAT_TRANSPARENT, 1.5
    mats["kind"][2], mats["roughness"][2], mats["color"][2] = pt.PT_MAT_GLOSSY, 0.3, 0.9
    cam = pt.Camera((0.2, 0.1, 5.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 1.5)
    data = pt.SceneData(np.array(v, np.float32), np.array(tris, np.uint32), np.array(mat, np.uint32), mats,
                        tri_uvs=np.array(uvs, np.float32), textures=[noise_texture()], camera=cam,
                        background=(1.0, 1.0, 1.0))
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    cfg = pt.Config(dim=72, samples_pr_pixel=4, bounces=3, seed=9, propagate_diffs=True, diff_size=0.02, roughdiff=True,
                    drop_nonfinite=True)
    img, ref = scene.render(cfg), oracle.render(cfg)
    assert np.isfinite(img).all()
    assert (np.abs(img - ref) > 1e-8 * np.maximum(1.0, np.abs(ref))).mean() < 5e-3
    plain = oracle.render(pt.Config(dim=72, samples_pr_pixel=4, bounces=3, seed=9, drop_nonfinite=True))
    assert np.abs(plain - ref).max() > 1e-2
