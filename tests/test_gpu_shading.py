"""GPU parity of the shade-side entry points: _Material::sample_brdf (Lambertian, Glossy),
mip-mapped texture lookup, glossy / textured / NEE renders — HIP path vs the CPU oracle.

Floating-point bar: sin/cos/log/exp/acos/tan come from ROCm's ocml on the GPU and glibc on the
CPU (both <= 1-2 ulp), everything else is IEEE-exact and uncontracted on both sides, so
per-sample quantities agree to ~1e-12 relative except where the reference's own formulas amplify
an ulp (Beckmann D near grazing half-vectors, tiny roughness); image-level checks use the
tolerance written in each test.
"""
import numpy as np
import pytest

import pathtracer_amd as pt
import oracle_lib as O
from test_gpu_parity import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trio(gpu):
    cache = {}

    def get(name, **kw):
        key = (name, tuple(sorted(kw.items())))
        if key not in cache:
            data = pt.scene_data(name, **kw)
            cache[key] = (data, pt.Scene(data, device=gpu), O.OracleScene(data))
        return cache[key]
    return get


def primary_hits(scene, oracle, cfg, sample):
    rays = oracle.generate_rays(cfg, sample)
    tri, t = oracle.trace_closest(rays)
    hit = tri != pt.PT_MISS
    pixel = np.arange(rays.shape[0], dtype=np.uint32)[hit]
    return rays[hit], tri[hit], t[hit], pixel


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def test_sample_brdf_lambertian(trio):
    # pt_sample_brdf's Lambertian branch is the code the renders run: lambertian_dir over the triangle's stored
    # basis (TriBasis) — and over rotate_to where a scene has no table (eye_tables / lds_scene off changes nothing
    # there; a scene above 65 536 triangles has none: covered by the heightfield renders).
    data, scene, oracle = trio("CornellBox")
    cfg = pt.Config(dim=96, seed=5)
    rays, tri, t, pixel = primary_hits(scene, oracle, cfg, 3)
    for bounce in (0, 1):
        g = scene.sample_brdf(rays, tri, t, pixel, cfg, 3, bounce)
        r = oracle.sample_brdf(rays, tri, t, pixel, cfg, 3, bounce)
        f, wi, pdf, em, pos = g
        rf, rwi, rpdf, rem, rpos = r
        assert np.array_equal(bits(f), bits(rf))          # rho / pi: exact
        assert np.array_equal(bits(em), bits(rem))
        assert np.array_equal(bits(pos), bits(rpos))      # o + d t: exact
        assert np.abs(wi - rwi).max() < 1e-14             # sincos ulp
        assert rel_err(pdf, rpdf).max() < 1e-11
        assert (em.sum(axis=1) > 0).any()                 # some rays see the light
        # The one declared deviation: the renders multiply the throughput by rho where
        # main.rs:311 evaluates f |wi . n| / pdf.  On the device's own f, wi, pdf the quotient is within 4 ulp of
        # rho wherever wi is not in the tangent plane (xi1 >= 2^-52: every sample here).
        normal = np.cross(data.verts[data.tris[tri, 1]].astype(np.float64) - data.verts[data.tris[tri, 0]],
                          data.verts[data.tris[tri, 0]].astype(np.float64) - data.verts[data.tris[tri, 2]])
        normal = -normal / np.linalg.norm(normal, axis=1, keepdims=True)
        cosn = np.abs((wi * normal).sum(axis=1))
        rho = data.mats[data.tri_material[tri]]["color"].astype(np.float64)
        quot = f * cosn[:, None] / pdf[:, None]
        lit = rho > 0
        assert (np.abs(quot - rho)[lit] <= 4 * np.spacing(rho[lit])).all()


def test_sample_brdf_glossy(trio):
    data, scene, oracle = trio("glossy_planes")
    cfg = pt.Config(dim=96, seed=5)
    rays, tri, t, pixel = primary_hits(scene, oracle, cfg, 0)
    kinds = data.mats[data.tri_material[tri]]["kind"]
    rough = data.mats[data.tri_material[tri]]["roughness"]
    assert (kinds == pt.PT_MAT_GLOSSY).sum() > 1000
    f, wi, pdf, em, pos = scene.sample_brdf(rays, tri, t, pixel, cfg, 0, 0)
    rf, rwi, rpdf, rem, rpos = oracle.sample_brdf(rays, tri, t, pixel, cfg, 0, 0)
    assert np.array_equal(bits(pos), bits(rpos)) and np.array_equal(bits(em), bits(rem))
    assert np.abs(wi - rwi).max() < 1e-12
    # D = exp(-tan^2/alpha^2)/(pi alpha^2 cos^4) via acos+tan amplifies an ulp of cos(theta_h) by
    # ~1/(alpha^2 theta): loose tolerance for the mirror-like strips, tight for the rough ones
    rough_ok = (kinds == pt.PT_MAT_GLOSSY) & (rough >= 0.05)
    fin = np.isfinite(rpdf) & np.isfinite(rf).all(axis=1)
    assert fin.mean() > 0.99
    assert (np.isfinite(pdf) == np.isfinite(rpdf)).all()
    m = rough_ok & fin
    assert rel_err(pdf[m], rpdf[m]).max() < 1e-9
    assert (np.abs(f[m] - rf[m]).max(axis=1) <= 1e-9 * np.maximum(np.abs(rf[m]).max(axis=1), 1e-300)).all()
    m2 = (kinds == pt.PT_MAT_GLOSSY) & fin
    assert np.median(rel_err(pdf[m2], rpdf[m2])) < 1e-9
    lam = kinds == pt.PT_MAT_LAMBERT
    assert rel_err(pdf[lam], rpdf[lam]).max() < 1e-11


def test_texture_lookup_matches_oracle(trio):
    data, scene, oracle = trio("Test", textures=True)
    rng = np.random.default_rng(0)
    n = 20000
    uv = rng.random((n, 2)) * 1.2 - 0.1                    # includes out-of-range uvs (clamp / extrapolation quirk)
    width = 10.0 ** rng.uniform(-9, 0.5, n)
    duvdx = np.stack([width, rng.random(n) * width], 1) * rng.choice([-1, 1], (n, 2))
    duvdy = np.stack([rng.random(n) * width, rng.random(n) * width], 1)
    duvdx[:100] = 0.0
    duvdy[:100] = 0.0
    got = scene.texture_lookup(0, uv, duvdx, duvdy)
    ref = oracle.texture_lookup(0, uv, duvdx, duvdy)
    # level selection uses log2 (ocml vs glibc); the blend weight differs by an ulp at most
    assert (np.abs(got - ref) <= 1e-12 * np.maximum(1.0, np.abs(ref))).all()
    assert np.array_equal(bits(got[:100]), bits(ref[:100]))   # level-0 bilinear path is exact
    inside = (uv >= 0.01).all(axis=1) & (uv <= 1.0).all(axis=1)   # negative ds extrapolates wildly (quirk)
    assert got[inside].min() >= -1e-12 and got[inside].max() <= 1.0 + 1e-12


def test_render_textured_matches_oracle(trio):
    data, scene, oracle = trio("Test", textures=True)
    cfg = pt.Config(dim=96, samples_pr_pixel=3, seed=3)
    img, ref = scene.render(cfg), oracle.render(cfg)
    bad = np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))
    assert bad.mean() < 2e-3
    plain = trio("Test")[2].render(cfg)
    assert np.abs(ref - plain).max() > 0.01                 # the texture really changes the image
    cfg_nd = pt.Config(dim=96, samples_pr_pixel=3, seed=3, use_diffs=False)
    img2, ref2 = scene.render(cfg_nd), oracle.render(cfg_nd)
    assert (np.abs(img2 - ref2) > 1e-9 * np.maximum(1.0, np.abs(ref2))).mean() < 2e-3


def test_render_glossy_matches_oracle(trio):
    data, scene, oracle = trio("glossy_planes")
    cfg = pt.Config(dim=96, samples_pr_pixel=4, seed=8)
    img, ref = scene.render(cfg), oracle.render(cfg)
    # NaN/inf samples are reference behaviour for alpha = 1e-5 (acos of a dot product rounding past 1);
    # both sides must produce them in (almost) the same pixels
    nf_g, nf_r = ~np.isfinite(img), ~np.isfinite(ref)
    assert (nf_g != nf_r).mean() < 1e-3
    ok = ~(nf_g | nf_r)
    err = np.abs(img[ok] - ref[ok])
    scale = np.maximum(1.0, np.abs(ref[ok]))
    assert (err > 1e-6 * scale).mean() < 5e-3
    assert abs(img[ok].mean() - ref[ok].mean()) < 1e-3 * max(1.0, abs(ref[ok].mean()))
    # DROP_NONFINITE deviates from the reference on purpose: the image becomes finite
    cfg_d = pt.Config(dim=96, samples_pr_pixel=4, seed=8, drop_nonfinite=True)
    assert np.isfinite(scene.render(cfg_d)).all()


def test_nan_accumulating_mode_on_a_tilted_near_mirror(gpu):
    # The reference ACCUMULATES non-finite samples (main.rs:434-436): Beckmann::D takes acos of h.n
    # (microfacet.rs:161-173), which is NaN once the dot product rounds past 1.  That cannot happen on
    # glossy_planes.obj (axis-aligned planes: h.n is one component of a normalised vector), so this test uses a
    # tilted alpha = 1e-5 quad on which the ORACLE ALONE was found to produce NaNs (synthetic.tilted_mirror;
    # pinned: 256 x 256 x 64 spp, seed 1 -> 11 NaN pixels, tests/test_oracle_golden.py).
    from pathtracer_amd.synthetic import tilted_mirror
    from conftest import record_parity
    data = tilted_mirror()
    scene, oracle = pt.Scene(data, device=gpu), O.OracleScene(data)
    spp = 64
    cfg = pt.Config(dim=256, samples_pr_pixel=spp, seed=1, bounces=2)
    img, ref = scene.render(cfg), oracle.render(cfg)
    nf_g, nf_r = ~np.isfinite(img).all(axis=2), ~np.isfinite(ref).all(axis=2)
    both, either = int((nf_g & nf_r).sum()), int((nf_g | nf_r).sum())
    ok = ~(nf_g | nf_r)
    rel = np.abs(img[ok] - ref[ok]) / np.maximum(1.0, np.abs(ref[ok]))
    record_parity("nan_mode_tilted_mirror_256x256_64spp", {
        "nonfinite_pixels_oracle": int(nf_r.sum()), "nonfinite_pixels_gpu": int(nf_g.sum()),
        "nonfinite_pixels_both": both, "nonfinite_pixels_either": either,
        "fraction_over_1e-6_of_finite": float((rel > 1e-6).mean()), "max_rel_err_finite": float(rel.max())})
    assert nf_r.sum() > 0, "the oracle must yield NaN pixels here (pinned)"
    assert nf_g.sum() > 0, "the HIP path must accumulate NaN samples like the reference"
    # a NaN hinges on the last bit of a dot product that follows a sincos (ocml on the GPU, glibc on the CPU):
    # stated bound, at least half of the non-finite pixels of either side are non-finite on both
    assert 2 * both >= either, (both, either)
    assert (rel > 1e-6).mean() < 5e-3
    # the exact culls keep non-finite throughputs (NaN * 0 != 0): culls on and off give the same bytes, NaNs included
    plain = pt.Scene(data, device=gpu, options={"light_cull": 0, "screen_cull": 0})
    img0 = plain.render(cfg)
    assert img0.tobytes() == img.tobytes()
    assert scene.stats()["rays"] < plain.stats()["rays"]          # ... and the cull did drop rays
    # PT_FLAG_DROP_NONFINITE is the opt-in deviation: finite, and equal to the reference mode where that is finite
    dropped = scene.render(pt.Config(dim=256, samples_pr_pixel=spp, seed=1, bounces=2, drop_nonfinite=True))
    assert np.isfinite(dropped).all()
    assert np.array_equal(bits(dropped[~nf_g]), bits(img[~nf_g]))


def test_textured_scene_without_uvs_uses_the_default_uvs(gpu):
    # triangle_mesh.rs:92-94: a mesh without texture coordinates gives every triangle the uvs (0,0), (1,0), (1,1).
    # pt_load_obj always materialises uvs once any model has `vt`, so the branch is reached by dropping them:
    # Test.obj with floor.png bound and tri_uvs = NULL (pt_shade.hip.h textured_color, sc.uvs == nullptr).
    import copy
    data = copy.copy(pt.scene_data("Test", textures=True))
    with_uvs = pt.Scene(data, device=gpu)
    data.tri_uvs = None
    scene, oracle = pt.Scene(data, device=gpu), O.OracleScene(data)
    for cfg in (pt.Config(dim=96, samples_pr_pixel=3, seed=3), pt.Config(dim=64, samples_pr_pixel=2, seed=5, use_diffs=False, nee=True)):
        img, ref = scene.render(cfg), oracle.render(cfg)
        assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 2e-3
        assert np.abs(img - with_uvs.render(cfg)).max() > 0.01          # the default uvs really change the image
    # per sample: the textured Lambertian colour f = texel / pi at primary hits, GPU vs oracle
    cfg = pt.Config(dim=96, seed=5)
    rays, tri, t, pixel = primary_hits(scene, oracle, cfg, 1)
    tex = data.mats[data.tri_material[tri]]["tex"] >= 0
    assert tex.sum() > 500
    f, wi, pdf, em, pos = scene.sample_brdf(rays[tex], tri[tex], t[tex], pixel[tex], cfg, 1, 0)
    rf, rwi, rpdf, rem, rpos = oracle.sample_brdf(rays[tex], tri[tex], t[tex], pixel[tex], cfg, 1, 0)
    assert (np.abs(f - rf) <= 1e-12 * np.maximum(1.0, np.abs(rf))).all()
    assert np.unique(np.round(f, 6), axis=0).shape[0] > 50              # a texture, not one colour


def test_render_nee_matches_oracle_and_converges(trio):
    data, scene, oracle = trio("CornellBox")
    cfg = pt.Config(dim=64, samples_pr_pixel=4, seed=6, nee=True)
    img, ref = scene.render(cfg), oracle.render(cfg)
    assert np.isfinite(img).all()
    bad = np.abs(img - ref) > 1e-8 * np.maximum(1.0, np.abs(ref))
    assert bad.mean() < 2e-3
    st0 = scene.stats()
    assert st0["shadow_rays"] > 0
    # converged NEE image == converged naive image, at GPU sample counts
    naive = scene.render(pt.Config(dim=32, samples_pr_pixel=4096, seed=1)) / 4096
    nee = scene.render(pt.Config(dim=32, samples_pr_pixel=1024, seed=2, nee=True)) / 1024
    assert abs(nee.mean() - naive.mean()) < 0.01 * naive.mean()
    bn = naive.reshape(8, 4, 8, 4, 3).mean(axis=(1, 3))
    be = nee.reshape(8, 4, 8, 4, 3).mean(axis=(1, 3))
    assert np.abs(be - bn).max() < 0.08 * bn.max()
    # variance: NEE is what it is for — lower per-pixel noise at equal sample count
    a = scene.render(pt.Config(dim=32, samples_pr_pixel=64, seed=3)) / 64
    b = scene.render(pt.Config(dim=32, samples_pr_pixel=64, seed=3, nee=True)) / 64
    assert np.abs(b - naive).mean() < np.abs(a - naive).mean()


def test_nee_glossy_and_textured_run(trio):
    for name, kw in (("glossy_planes", {}), ("Test", {"textures": True})):
        data, scene, oracle = trio(name, **kw)
        cfg = pt.Config(dim=48, samples_pr_pixel=2, seed=4, nee=True, drop_nonfinite=True)
        img, ref = scene.render(cfg), oracle.render(cfg)
        assert np.isfinite(img).all()
        err = np.abs(img - ref) > 1e-6 * np.maximum(1.0, np.abs(ref))
        assert err.mean() < 1e-2


def test_stats_counters(trio):
    data, scene, oracle = trio("CornellBox")
    scene.reset_stats()
    scene.set_profiling(2)
    cfg = pt.Config(dim=64, samples_pr_pixel=2, bounces=2)
    scene.render(cfg)
    st = scene.stats()
    scene.set_profiling(0)
    assert st["samples"] == 64 * 64 * 2
    assert st["samples"] <= st["rays"] <= 3 * st["samples"]
    assert st["nodes_visited"] >= 2 * st["rays"] and st["tris_tested"] > st["rays"] // 2
    assert 0 < st["lit_samples"] < st["samples"] // 2        # most samples gather nothing and are never stored
    k = st["kernel"]
    assert k["extend"]["launches"] == 3 and k["shade"]["launches"] == 3 and k["generate"]["launches"] == 0  # fused into bounce-0 extend/shade
    assert k["extend"]["ms"] > 0 and k["path"]["launches"] == 0   # level 2 counts in the extend kernel: unfused
    # level 1 (timers only) runs the fused extend+shade kernel on this shallow tree
    scene.reset_stats()
    scene.set_profiling(1)
    scene.render(cfg)
    st1 = scene.stats()
    scene.set_profiling(0)
    k = st1["kernel"]
    # ... and, the scene living in LDS, bounces 0 and 1 in ONE launch (k_path F01), then the sparse last bounce
    assert k["path"]["launches"] == 2 and k["extend"]["launches"] == 0 and k["shade"]["launches"] == 0
    assert st1["rays"] == st["rays"]
    # fuse01 = 0: one launch per bounce; same rays, same image
    plain = pt.Scene(data, device=scene.device, options={"fuse01": 0})
    plain.set_profiling(1)
    img_plain = plain.render(cfg)
    st2 = plain.stats()
    assert st2["kernel"]["path"]["launches"] == 3 and st2["rays"] == st["rays"]
    assert img_plain.tobytes() == scene.render(cfg).tobytes()


def test_progressive_checkpoint_resume_is_bit_identical(trio, tmp_path):
    # N1: a render interrupted after 5 of 12 frames and resumed from its checkpoint equals the
    # uninterrupted one bit for bit
    data, scene, oracle = trio("CornellBox")
    cfg = pt.Config(dim=72, samples_pr_pixel=12, seed=21)
    whole = scene.render(cfg)
    path = str(tmp_path / "cornell.ckpt")
    part = pt.Config(dim=72, samples_pr_pixel=5, seed=21)
    pt.render_progressive(scene, part, path, every=2)          # "crashes" after frames 0..4
    _, frames, _ = pt.load_checkpoint(path)
    assert frames == 5
    resumed = pt.render_progressive(scene, cfg, path, every=4)
    assert resumed.tobytes() == whole.tobytes()
    with pytest.raises(ValueError):
        pt.render_progressive(scene, pt.Config(dim=72, samples_pr_pixel=12, seed=22), path)
