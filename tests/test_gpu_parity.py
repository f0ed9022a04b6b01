"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle.

Bar (prompt §3 / BASELINE.json north_star): primary-ray generation, hit triangle ids and hit
distances are bit-exact; images agree within a stated tolerance because sin/cos come from
different math libraries on the two sides (glibc vs ROCm ocml).
"""
import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import _lib as L
import oracle_lib as O
from conftest import record_parity

pytestmark = pytest.mark.gpu


def parity_numbers(a, b, tol, spp):
    """What a GPU-vs-oracle image comparison achieved (a, b: accumulated f64 sums of `spp` frames)."""
    fin = np.isfinite(a) & np.isfinite(b)
    rel = np.abs(a - b)[fin] / np.maximum(1.0, np.abs(b[fin]))
    qa = pt.resolve_rgb8(np.nan_to_num(a), spp).astype(np.float64)
    qb = pt.resolve_rgb8(np.nan_to_num(b), spp).astype(np.float64)
    mse = ((qa - qb) ** 2).mean()
    return {
        "values": int(a.size), "spp": int(spp), "tolerance": tol,
        "rel_rmse": float(np.sqrt(((a - b)[fin] ** 2).mean()) / max(np.sqrt((b[fin] ** 2).mean()), 1e-300)),
        "max_rel_err": float(rel.max()) if rel.size else 0.0,
        "fraction_over_tolerance": float((rel > tol).mean()) if rel.size else 0.0,
        "psnr_8bit_db": None if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse)),
        "identical_8bit": bool(mse == 0),
        "mean_rel_diff": float(abs(a[fin].mean() - b[fin].mean()) / max(abs(b[fin].mean()), 1e-300)),
        "nonfinite_gpu": int((~np.isfinite(a)).sum()), "nonfinite_oracle": int((~np.isfinite(b)).sum()),
        "nonfinite_mask_mismatch": int((np.isfinite(a) != np.isfinite(b)).sum()),
    }

SCENES = ["CornellBox", "Test", "glossy_planes", "cornell_box"]


@pytest.fixture(scope="module")
def scenes(gpu):
    out = {}
    for name in SCENES:
        data = pt.scene_data(name)
        out[name] = (data, pt.Scene(data, device=gpu), O.OracleScene(data))
    return out


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("dims", [(128, 128), (192, 108)])
def test_generate_rays_bit_exact(scenes, name, dims):
    data, scene, oracle = scenes[name]
    cfg = pt.Config(width=dims[0], height=dims[1], seed=3)
    for sample in (0, 5):
        got = scene.generate_rays(cfg, sample)
        ref = oracle.generate_rays(cfg, sample)
        assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("name", SCENES)
def test_primary_hits_bit_exact(scenes, name):
    data, scene, oracle = scenes[name]
    cfg = pt.Config(dim=128, seed=11)
    rays = oracle.generate_rays(cfg, 1)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=False)
    assert np.array_equal(tri, rtri)
    assert np.array_equal(bits(t), bits(rt))
    if name != "cornell_box":
        assert (tri != pt.PT_MISS).mean() > 0.3


def random_rays(data, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = data.verts.min(axis=0).astype(np.float64), data.verts.max(axis=0).astype(np.float64)
    ext = np.maximum(hi - lo, 1e-3)
    o = lo - 0.25 * ext + rng.random((n, 3)) * 1.5 * ext
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a share of axis-aligned and zero-component directions (slab test edge cases)
    k = n // 8
    d[:k] = 0.0
    d[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    d[k:2 * k, rng.integers(0, 3)] = 0.0
    d[k:2 * k] /= np.maximum(np.linalg.norm(d[k:2 * k], axis=1, keepdims=True), 1e-30)
    # a share of rays starting exactly on vertices / triangle centroids (secondary-ray-like origins)
    tris = data.verts[data.tris].astype(np.float64)
    pick = rng.integers(0, tris.shape[0], k)
    o[2 * k:3 * k] = tris[pick].mean(axis=1)
    o[3 * k:4 * k] = tris[pick, 0]
    return pt.make_rays(o, d)


@pytest.mark.parametrize("name", SCENES)
def test_random_rays_bit_exact(scenes, name):
    data, scene, oracle = scenes[name]
    rays = random_rays(data, 20000, seed=5)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=False)
    assert np.array_equal(tri, rtri)
    assert np.array_equal(bits(t), bits(rt))


def test_tmin_tmax_respected(scenes):
    data, scene, oracle = scenes["CornellBox"]
    rays = random_rays(data, 4000, seed=9)
    rays["t_min"] = 0.3
    rays["t_max"] = 0.9
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    hit = tri != pt.PT_MISS
    assert hit.any() and (~hit).any()
    assert np.all(t[hit] >= 0.3) and np.all(t[hit] <= 0.9)
    assert np.all(t[~hit] == 0.9)  # a miss leaves t_max untouched (scene.rs:61 &mut Ray)


def test_exact_ties_lowest_id_wins(gpu):
    # two coincident triangles with different ids and one shared-edge pair: bit-identical t
    verts = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]], np.float32)
    tris = np.array([[0, 1, 2], [0, 1, 2], [1, 3, 2], [0, 1, 2]], np.uint32)
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    data = pt.SceneData(verts, tris, np.zeros(4, np.uint32), mats)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    o = np.array([[0.2, 0.2, 0.0], [0.5, 0.5, 0.0], [0.9, 0.9, 0.0]])
    d = np.array([[0, 0, 1.0]] * 3)
    rays = pt.make_rays(o, d)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    assert tri[0] == 0 and tri[2] == 2


@pytest.mark.parametrize("name", ["CornellBox", "Test"])
def test_render_matches_oracle(scenes, name):
    data, scene, oracle = scenes[name]
    cfg = pt.Config(dim=96, samples_pr_pixel=4, bounces=2, seed=2)
    img = scene.render(cfg)
    ref = oracle.render(cfg)
    assert np.isfinite(img).all()
    # tolerance: sin/cos differ by <= 1-2 ulp between ocml and glibc; a path whose hit flips
    # across a triangle edge is rare.  Require 99.9 % of values within 1e-9 relative and a
    # near-identical mean.
    bad = np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))
    assert bad.mean() < 1e-3
    assert abs(img.mean() - ref.mean()) <= 1e-6 * max(1.0, ref.mean())
    assert ref.mean() > 0.01


def test_render_batching_and_tiles_are_bitwise_invariant(scenes):
    data, scene, _ = scenes["CornellBox"]
    cfg = pt.Config(dim=64, samples_pr_pixel=6, seed=4)
    whole = scene.render(cfg)
    cfg_b = pt.Config(dim=64, samples_pr_pixel=6, seed=4, spp_per_batch=1)
    one = scene.render(cfg_b)
    assert whole.tobytes() == one.tobytes()
    acc = np.zeros_like(whole)
    for tile in [(0, 0, 40, 64), (40, 0, 64, 30), (40, 30, 64, 64)]:
        scene.render(cfg, acc, tile=tile)
    assert acc.tobytes() == whole.tobytes()
    # sample ranges: 0..3 then 3..6 accumulate to the same sums in the same order
    acc2 = scene.render(cfg, None, sample_begin=0, sample_count=3)
    scene.render(cfg, acc2, sample_begin=3, sample_count=3)
    assert acc2.tobytes() == whole.tobytes()


def test_k1_cornell_box_obj_is_black(scenes):
    # K1: models/cornell_box.obj has no mtllib -> black non-emissive quad -> zeros
    data, scene, oracle = scenes["cornell_box"]
    cfg = pt.Config(dim=64, samples_pr_pixel=3)
    img = scene.render(cfg)
    assert not img.any()
    assert not oracle.render(cfg).any()


def furnace(rho, emission):
    # closed box, every surface Lambertian rho with emission E, camera inside
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = np.array([[q[0], q[1], q[2]] for q in quads] + [[q[0], q[2], q[3]] for q in quads], np.uint32)
    # the reference scatters about the unflipped CCW normal (triangle_mesh.rs:141): wind every
    # triangle so that normal points into the box, else paths leave through the walls
    p = v[tris].astype(np.float64)
    ccw = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    outward = np.einsum("ij,ij->i", ccw, p.mean(axis=1)) > 0
    tris[outward] = tris[outward][:, [0, 2, 1]]
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["color"], mats["emission"], mats["tex"] = rho, emission, -1
    cam = pt.Camera((0.1, -0.2, 0.05), (0.3, 0.2, 1.0), (0, 1, 0), 1.0)
    return pt.SceneData(v, tris, np.zeros(len(tris), np.uint32), mats, camera=cam)


def test_k2_furnace(gpu):
    # K2: every sample = E * sum_{k=0..B} rho^k (main.rs:292-315 with f*cos/pdf == rho)
    rho, E, B = 0.5, 2.0, 3
    data = furnace(rho, E)
    scene = pt.Scene(data, gpu)
    cfg = pt.Config(dim=48, samples_pr_pixel=2, bounces=B)
    img = scene.render(cfg) / 2
    expect = E * sum(rho ** k for k in range(B + 1))
    assert np.allclose(img, expect, rtol=1e-12, atol=0)


# ---- large synthetic scene (SURVEY config 5 generator) --------------------------------------------
def test_heightfield_hits_bit_exact(gpu):
    from pathtracer_amd.synthetic import heightfield_box
    data = heightfield_box(cells=200)          # 80,010 triangles, deep BVH, real leaves
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    nodes, order, depth = scene.bvh()
    assert depth <= 32 and sorted(order.tolist()) == list(range(data.n_tris))
    cfg = pt.Config(dim=160, seed=2)
    rays = scene.generate_rays(cfg, 0)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=True)   # oracle BVH == oracle brute force (CPU test)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    rays = random_rays(data, 30000, seed=3)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=True)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    sub = slice(0, 300)                                   # and a brute-force slice, no BVH on either side
    btri, bt = oracle.trace_closest(rays[sub], use_bvh=False)
    assert np.array_equal(tri[sub], btri) and np.array_equal(bits(t[sub]), bits(bt))
    img = scene.render(pt.Config(dim=64, samples_pr_pixel=3, seed=1))
    ref = oracle.render(pt.Config(dim=64, samples_pr_pixel=3, seed=1), use_bvh=True)
    assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 2e-3


def test_caller_supplied_bvh_and_validation(gpu):
    data = pt.scene_data("CornellBox")
    nodes, order, depth = pt.bvh_build(data.verts, data.tris, max_leaf=2)
    data.bvh_nodes, data.bvh_tri_order = nodes, order
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(pt.scene_data("CornellBox"))
    rays = random_rays(data, 5000, seed=1)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    bad = pt.scene_data("CornellBox")
    bad.bvh_nodes, bad.bvh_tri_order = nodes.copy(), order.copy()
    bad.bvh_nodes["child"][0, 0] = 10_000              # out-of-range child must be refused, not traversed
    with pytest.raises(pt.PtError):
        pt.Scene(bad, gpu)
    bad.bvh_nodes = nodes.copy()
    bad.bvh_tri_order = np.zeros_like(order)           # duplicate triangle ids
    with pytest.raises(pt.PtError):
        pt.Scene(bad, gpu)


# ---- traversal scheduling: both schedulers (vote-scheduled ranges / one ray per lane) on every scene -----
FORCE_VOTE = dict(PT_TRAV_VOTE="1", PT_TRAV_VOTE_PRIMARY="1", PT_TRAV_VOTE_SHADOW="1", PT_WAVE_RAYS="128")
FORCE_LANE = dict(PT_TRAV_VOTE="0", PT_TRAV_VOTE_PRIMARY="0", PT_TRAV_VOTE_SHADOW="0")


def test_scene_options_select_the_same_variants_as_the_environment(gpu, monkeypatch):
    # pt_scene_create_ex: the per-scene API for what the PT_* environment switches do for sweeps.  Each variant
    # renders the same bytes as the default scene, and takes the path the matching environment switch takes
    # (kernel launch counts by kind are the fingerprint of the path).
    data = pt.scene_data("Test")
    default = pt.Scene(data, gpu, options={})
    cfg = pt.Config(width=160, height=90, samples_pr_pixel=4, bounces=2, seed=6, nee=True)

    def fingerprint(scene):
        scene.reset_stats()
        scene.set_profiling(1)
        img = scene.render(cfg)
        st = scene.stats()
        scene.set_profiling(0)
        return img.tobytes(), tuple(st["kernel"][k]["launches"] for k in L.KERNEL_NAMES)

    want, base_launches = fingerprint(default)
    variants = (
        (dict(fuse=0), dict(PT_FUSE="0")),
        (dict(fuse_shadow=0), dict(PT_FUSE_SHADOW="0")),
        (dict(vote_primary=1, vote_secondary=1, vote_shadow=1, wave_rays=128), FORCE_VOTE),
        (dict(lds_scene=0), dict(PT_LDS_SCENE="0")),
        (dict(fuse01=0), dict(PT_FUSE01="0")),
        (dict(eye_tables=0), dict(PT_EYE_TABLES="0")),
        (dict(light_cull=0, screen_cull=0, persistent=0), dict(PT_LIGHT_CULL="0", PT_SCREEN_CULL="0", PT_PERSISTENT="0")),
        (dict(pipeline=1, bvh_max_leaf=4, bvh_threads=2), dict(PT_PIPELINE="1", PT_BVH_LEAF="4", PT_BVH_THREADS="2")),
    )
    differ = 0
    for options, env in variants:
        got, launches = fingerprint(pt.Scene(data, gpu, options=options))
        assert got == want, options
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        by_env, env_launches = fingerprint(pt.Scene(data, gpu))           # pt_scene_create: options from the environment
        assert fingerprint(pt.Scene(data, gpu, options={}))[1] == base_launches   # ..._ex never reads it
        for k in env:
            monkeypatch.delenv(k)
        assert by_env == want and env_launches == launches, options
        differ += launches != base_launches
    assert differ >= 2                                       # unfused and vote-scheduled paths launch other kernels


@pytest.mark.parametrize("name", ["CornellBox", "heightfield40"])
def test_frame_grouping_is_bitwise_invariant(gpu, name):
    # pt_scene_options::group_frames: which slot a (frame, pixel) pair takes - a wave of primary rays holds one
    # pixel in 64 frames, or 64 / G pixels in G frames, instead of an 8x8 block in one frame - is invisible in the
    # image: frame counts that divide by 64, by smaller powers of two and by none, tiles with pixels outside whole
    # 8x8 blocks, sample ranges, two lanes, a row partition, NEE.
    from pathtracer_amd.synthetic import heightfield_box
    data = heightfield_box(cells=40) if name == "heightfield40" else pt.scene_data(name)
    plain = pt.Scene(data, gpu, options=dict(group_frames=0))
    for k, extra in ((6, {}), (3, {}), (6, dict(pipeline=1)), (1, dict(fuse=0))):
        grouped = pt.Scene(data, gpu, options=dict(group_frames=k, **extra))
        for kw, spp, tile, rows in ((dict(width=211, height=97), 64, None, None),
                                    (dict(width=83, height=47, nee=True), 24, None, None),
                                    (dict(width=200, height=120), 7, None, None),
                                    (dict(width=200, height=120), 48, (13, 9, 171, 100), None),
                                    (dict(width=96, height=200), 16, None, (8, 3, 1))):
            cfg = pt.Config(samples_pr_pixel=spp, bounces=2, seed=9, drop_nonfinite=True, **kw)
            a = plain.render(cfg, tile=tile, rows=rows)
            b = grouped.render(cfg, tile=tile, rows=rows)
            assert a.any() and a.tobytes() == b.tobytes(), (k, extra, kw, spp)
            c = grouped.render(cfg, None, sample_begin=0, sample_count=spp // 2, tile=tile, rows=rows)
            grouped.render(cfg, c, sample_begin=spp // 2, sample_count=spp - spp // 2, tile=tile, rows=rows)
            assert c.tobytes() == a.tobytes()


@pytest.mark.parametrize("wave_rays", ["64", "128", "1024"])
def test_schedulers_agree_bitwise(gpu, monkeypatch, wave_rays):
    from pathtracer_amd.synthetic import heightfield_box
    datas = {name: pt.scene_data(name) for name in ("CornellBox", "Test", "glossy_planes")}
    datas["heightfield40"] = heightfield_box(cells=40)
    for name, data in datas.items():
        made = {}
        for label, env in (("vote", dict(FORCE_VOTE, PT_WAVE_RAYS=wave_rays)), ("lane", FORCE_LANE)):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            made[label] = pt.Scene(data, gpu)          # the scheduling is resolved at scene creation
            for k in env:
                monkeypatch.delenv(k)
        oracle = O.OracleScene(data)
        for n in (1, 63, 65, 4099):                    # ragged ranges: below, across and above one wave's range
            rays = random_rays(data, 4099, seed=n)[:n]
            rtri, rt = oracle.trace_closest(rays, use_bvh=True)
            for scene in made.values():
                tri, t = scene.trace_closest(rays)
                assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt)), (name, n)
        for nee in (False, True):
            cfg = pt.Config(width=57, height=41, samples_pr_pixel=3, bounces=3, seed=8, nee=nee)
            a, b = made["vote"].render(cfg), made["lane"].render(cfg)
            assert a.tobytes() == b.tobytes(), (name, nee)
            assert a.any()


@pytest.mark.parametrize("nee", [False, True])
def test_two_lane_pipeline_is_bitwise_invariant(gpu, monkeypatch, nee):
    # batches of >= 2 frames and >= 2^20 paths per half run as two lanes on two streams; the image
    # must not depend on it (same per-path arithmetic, same accumulation order)
    data = pt.scene_data("CornellBox")
    monkeypatch.setenv("PT_PIPELINE", "0")
    one = pt.Scene(data, gpu)
    monkeypatch.setenv("PT_PIPELINE", "1")
    two = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_PIPELINE")
    cfg = pt.Config(width=1024, height=1024, samples_pr_pixel=5, bounces=2, seed=6, nee=nee, spp_per_batch=3)
    a, b = one.render(cfg), two.render(cfg)          # batches of 3 (lanes of 2 + 1) and 2 (1 + 1) frames
    assert a.tobytes() == b.tobytes()
    small = pt.Config(dim=64, samples_pr_pixel=4, seed=6, nee=nee)   # too small to split: one lane either way
    assert one.render(small).tobytes() == two.render(small).tobytes()
    oracle = O.OracleScene(data)
    cfg1 = pt.Config(width=1024, height=1024, samples_pr_pixel=2, bounces=2, seed=6, nee=nee)
    ref = oracle.render(cfg1, threads=16)
    img = two.render(cfg1)
    assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 1e-3


def test_accum_handle_frame_loop_equals_one_render(scenes):
    # the reference's window loop (main.rs:400-452): one frame per iteration, the image shown after each.
    # Through the device-resident handle N single-frame calls leave the very sums of one N-frame pt_render,
    # and the device-side resolve equals the host's (main.rs:456-467) byte for byte, packed words included.
    data, scene, oracle = scenes["CornellBox"]
    cfg = pt.Config(width=320, height=180, bounces=2, seed=17)
    acc = pt.Accum(scene, cfg)
    n = 7
    for it in range(1, n + 1):
        acc.render(1)
        if it in (1, 4):
            shown = acc.resolve_rgb8()
            assert shown.tobytes() == pt.resolve_rgb8(scene.render(cfg, sample_begin=0, sample_count=it), it).tobytes()
    whole = scene.render(cfg, sample_begin=0, sample_count=n)
    sums = acc.download()
    assert sums.tobytes() == whole.tobytes()
    rgb = acc.resolve_rgb8()
    assert rgb.tobytes() == pt.resolve_rgb8(whole, n).tobytes() and rgb.any()
    u32 = acc.resolve_u32()
    r, g, b = (rgb[..., k].astype(np.uint32) for k in range(3))
    assert np.array_equal(u32, (r << 16) | (g << 8) | b)           # main.rs:463-465
    # resume from a checkpoint of the sums: upload, go on, same bits as the uninterrupted loop
    other = pt.Accum(scene, cfg)
    other.upload(scene.render(cfg, sample_begin=0, sample_count=3), iterations=3)
    other.render(4)
    assert other.download().tobytes() == whole.tobytes()
    other.upload(None)
    assert not other.download().any()
    with pytest.raises(pt.PtError):                                # a handle renders its own image size only
        L = pt.Config(width=64, height=64)
        import ctypes as C
        from pathtracer_amd import _lib
        p = L.params(0, 1)
        _lib.check(_lib.lib.pt_accum_render(other._h, C.byref(p)))
    # NaN / negative / > 1 sums resolve as the host does (`as u8` saturates, NaN -> 0)
    odd = np.zeros((cfg.h, cfg.w, 3)); odd[0, 0] = [np.nan, -3.0, 9.0]; odd[0, 1] = [0.5, 1.0, 0.999999]
    other.upload(odd, iterations=1)
    assert other.resolve_rgb8().tobytes() == pt.resolve_rgb8(odd, 1).tobytes()


def test_two_streams_share_one_scene(gpu):
    # pt_render_device is asynchronous on the caller's stream, and every call on a scene uses the scene's
    # one workspace: calls issued alternately on two streams must still not overlap on the device (each
    # call waits for the event the previous one recorded).  Same bits as the one-stream renders.
    import torch
    torch.cuda.set_device(gpu)
    scene = pt.Scene(pt.scene_data("CornellBox"), gpu)
    cfg_a = pt.Config(width=1024, height=1024, bounces=2, seed=21)
    cfg_b = pt.Config(width=768, height=512, bounces=3, seed=22, nee=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    acc_a = torch.zeros((cfg_a.h, cfg_a.w, 3), dtype=torch.float64, device="cuda")
    acc_b = torch.zeros((cfg_b.h, cfg_b.w, 3), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for i in range(6):
        scene.render_device(cfg_a, acc_a.data_ptr(), 3 * i, 3, stream=s1.cuda_stream)
        scene.render_device(cfg_b, acc_b.data_ptr(), 3 * i, 3, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    ref_a = scene.render(cfg_a, sample_begin=0, sample_count=18)
    ref_b = scene.render(cfg_b, sample_begin=0, sample_count=18)
    assert acc_a.cpu().numpy().tobytes() == ref_a.tobytes()
    assert acc_b.cpu().numpy().tobytes() == ref_b.tobytes()


@pytest.mark.parametrize("name", ["CornellBox", "Test", "glossy_planes", "heightfield40"])
def test_fused_path_kernel_is_bitwise_identical(gpu, monkeypatch, name):
    # extend+shade in one kernel (every bounce of a shallow tree, the primary bounce of a deep one)
    # against the two-kernel form: same arithmetic per path, so the same image bit for bit; with
    # textures and NEE as well
    if name == "heightfield40":
        from pathtracer_amd.synthetic import heightfield_box
        data = heightfield_box(cells=40)
    else:
        data = pt.scene_data(name, textures=True) if name == "Test" else pt.scene_data(name)
    monkeypatch.setenv("PT_FUSE", "0")
    plain = pt.Scene(data, gpu)
    monkeypatch.setenv("PT_FUSE", "1")
    fused = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_FUSE")
    for nee in (False, True):
        for bounces in (0, 1, 3):
            cfg = pt.Config(width=83, height=47, samples_pr_pixel=3, bounces=bounces, seed=12, nee=nee, drop_nonfinite=True)
            a, b = plain.render(cfg), fused.render(cfg)
            assert a.tobytes() == b.tobytes(), (name, nee, bounces)


@pytest.mark.parametrize("name", ["CornellBox", "Test", "glossy_planes", "tilted_mirror", "sparse_hits"])
def test_two_bounce_kernel_is_bitwise_identical(gpu, name):
    # k_path F01: the primary launch of a scene that lives in LDS traces and shades the path's second vertex too
    # (no record written and read back between bounces 0 and 1) against one launch per bounce: same arithmetic per
    # path, additions in the same order, so the same bytes - textures, Glossy with NaN-producing samples, NEE with
    # its in-kernel shadow rays, Russian roulette, more bounces than two, tiles, a row partition, odd sizes; and the
    # ray statistics agree.  With fewer than two bounces the kernel does not apply and the paths coincide.
    if name == "tilted_mirror":
        from pathtracer_amd.synthetic import tilted_mirror
        data = tilted_mirror()
    elif name == "sparse_hits":      # most primary rays miss: the probe leaves the kernel off, the option forces it
        import copy
        data = copy.copy(pt.scene_data("CornellBox"))
        c = data.camera
        data.camera = pt.Camera((0.5, 0.5, -6.0), c.view_direction, c.up_direction, 1.0)
        data.background = (0.05, 0.1, 0.2)
    else:
        data = pt.scene_data(name, textures=name == "Test", specular=name == "Test")
    one = pt.Scene(data, gpu, options=dict(fuse01=0))
    two = pt.Scene(data, gpu, options=dict(fuse01=1))
    auto = pt.Scene(data, gpu, options={})
    cfgs = [pt.Config(width=83, height=47, samples_pr_pixel=3, bounces=b, seed=12, nee=nee)
            for nee in (False, True) for b in (0, 1, 2, 4)]
    cfgs += [pt.Config(dim=64, samples_pr_pixel=2, bounces=6, seed=9, nee=True, russian_roulette=True),
             pt.Config(dim=256, samples_pr_pixel=64, seed=1, bounces=2)]
    for cfg in cfgs:
        one.reset_stats(); two.reset_stats()
        a, b = one.render(cfg), two.render(cfg)
        assert a.tobytes() == b.tobytes(), (name, cfg)
        sa, sb = one.stats(), two.stats()
        assert (sa["rays"], sa["shadow_rays"], sa["samples"]) == (sb["rays"], sb["shadow_rays"], sb["samples"]), (name, cfg)
        # the rays of bounce 1 never exist as queue records in the two-bounce launch
        assert sa["unqueued_rays"] == 0 and (sb["unqueued_rays"] > 0) == (cfg.bounces >= 2)
        assert auto.render(cfg).tobytes() == a.tobytes()
    cfg = pt.Config(width=211, height=97, samples_pr_pixel=5, bounces=3, seed=4, nee=True, spp_per_batch=2)
    whole = one.render(cfg)
    acc = np.zeros_like(whole)
    for x0, x1 in ((0, 70), (70, 211)):
        two.render(cfg, acc, tile=(x0, 0, x1, 97))
    assert acc.tobytes() == whole.tobytes()
    acc = np.zeros_like(whole)
    for index in range(3):
        two.render(cfg, acc, rows=(8, 3, index))
    assert acc.tobytes() == whole.tobytes()
    if name == "tilted_mirror":
        assert not np.isfinite(one.render(cfgs[-1])).all()       # the NaN pixels are there, on both paths alike
    # the launch counts tell which path ran
    for sc, want in ((one, 3), (two, 2)):
        sc.reset_stats(); sc.set_profiling(1)
        sc.render(pt.Config(dim=64, samples_pr_pixel=2, bounces=2, seed=1))
        assert sc.stats()["kernel"]["path"]["launches"] == want
        sc.set_profiling(0)
    auto.reset_stats(); auto.set_profiling(1)
    auto.render(pt.Config(dim=64, samples_pr_pixel=2, bounces=2, seed=1))
    assert auto.stats()["kernel"]["path"]["launches"] == (3 if name in ("sparse_hits", "tilted_mirror") else 2)   # the probe
    auto.set_profiling(0)


@pytest.mark.parametrize("cells", [40, 300])
def test_packet_traversal_of_primary_rays_is_bitwise_identical(gpu, cells):
    # deep trees: the 64 primary rays of a wave walk the four-wide tree together (traverse_packet4: wave-uniform node
    # index and stack, a node opened if any lane hits its box) against one traversal per lane: every lane runs the
    # same slab and triangle tests on a superset of the leaves, so hit ids / distances and the image are the same
    # bits - frame counts that put one pixel in 64 frames per wave, a few pixels per wave, an 8x8 block per wave;
    # off-screen lanes, partial waves, NEE, a row partition, the top of the tree in LDS or not.
    from pathtracer_amd.synthetic import heightfield_box
    data = heightfield_box(cells=cells)
    lane = pt.Scene(data, gpu, options=dict(packet_primary=0))
    for extra in ({}, dict(lds_top=0), dict(group_frames=0), dict(group_frames=3)):
        packet = pt.Scene(data, gpu, options=dict(packet_primary=1, **extra))
        for cfg in (pt.Config(width=211, height=97, samples_pr_pixel=64, bounces=2, seed=3, spp_per_batch=64),
                    pt.Config(width=160, height=90, samples_pr_pixel=5, bounces=2, seed=4, nee=True),
                    pt.Config(dim=96, samples_pr_pixel=128, bounces=0, seed=5)):
            lane.reset_stats(); packet.reset_stats()
            a, b = lane.render(cfg), packet.render(cfg)
            assert a.any() and a.tobytes() == b.tobytes(), (cells, extra, cfg)
            assert lane.stats()["rays"] == packet.stats()["rays"]
        cfg = pt.Config(width=211, height=97, samples_pr_pixel=8, bounces=1, seed=6)
        whole = lane.render(cfg)
        acc = np.zeros_like(whole)
        for index in range(3):
            packet.render(cfg, acc, rows=(8, 3, index))
        assert acc.tobytes() == whole.tobytes()


def test_baseline_full_size_cornellbox_1080p(scenes):
    # BASELINE.json configs[1] geometry at its full image size: primary hits bit-exact against the
    # oracle on all 2,073,600 rays, and the size-independent properties of the accumulation
    data, scene, oracle = scenes["CornellBox"]
    cfg = pt.Config(width=1920, height=1080, samples_pr_pixel=8, bounces=2, seed=1, spp_per_batch=8)
    rays = scene.generate_rays(cfg, 3)
    assert rays.tobytes() == oracle.generate_rays(cfg, 3).tobytes()
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=True)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))
    whole = scene.render(cfg)
    # linearity in the sample range (frames are keyed by index and added in index order) ...
    acc = scene.render(cfg, None, sample_begin=0, sample_count=3)
    scene.render(cfg, acc, sample_begin=3, sample_count=5)
    assert acc.tobytes() == whole.tobytes()
    # ... tile decomposition (row bands as the multi-GPU `tiles` partitioning cuts them) ...
    acc = np.zeros_like(whole)
    for y0, y1 in ((0, 135), (135, 540), (540, 1080)):
        scene.render(cfg, acc, tile=(0, y0, 1920, y1))
    assert acc.tobytes() == whole.tobytes()
    # ... idempotence of a repeated render, and energy: the mean radiance of the full-size image
    # equals the oracle's on a strided subset of the same frames' pixels (same RNG streams)
    assert scene.render(cfg).tobytes() == whole.tobytes()
    ys, xs = slice(0, 1080, 36), slice(0, 1920, 1)           # 30 full rows
    ref = np.zeros_like(whole)
    for y in range(0, 1080, 36):
        oracle.render(cfg, ref, tile=(0, y, 1920, y + 1), threads=16)
    sub, rsub = whole[ys, xs], ref[ys, xs]
    assert (np.abs(sub - rsub) > 1e-9 * np.maximum(1.0, np.abs(rsub))).mean() < 1e-3
    assert abs(sub.mean() - rsub.mean()) <= 1e-6 * rsub.mean()


def test_config5_full_size_heightfield_10m(gpu, monkeypatch):
    # SURVEY config 5 at its full size (9,999,402 triangles, 3840x2160): no CPU oracle can render it
    # in test time, so: a brute-force oracle slice of its primary rays (no tree on the CPU side),
    # and the size-independent properties — scheduler invariance (vote-scheduled persistent waves vs
    # one ray per lane, fused vs two-kernel, one vs two lanes), sample-range linearity, tiling.
    from pathtracer_amd.synthetic import heightfield_box
    data = heightfield_box()
    assert data.n_tris == 9_999_402
    scene = pt.Scene(data, gpu)
    nodes, order, depth = scene.bvh()
    assert depth <= 30 and len(order) == data.n_tris
    cfg = pt.Config(width=3840, height=2160, samples_pr_pixel=2, bounces=2, seed=1, spp_per_batch=2)
    rays = scene.generate_rays(cfg, 0)
    tri, t = scene.trace_closest(rays)
    hit = np.flatnonzero(tri != pt.PT_MISS)
    assert 0.3 < len(hit) / len(tri) < 1.0                 # the box is open towards the camera
    rng = np.random.default_rng(5)
    pick = np.concatenate([rng.choice(hit, 12, replace=False), rng.choice(np.flatnonzero(tri == pt.PT_MISS), 4, replace=False)])
    monkeypatch.setenv("ORC_NO_BVH", "1")                  # 10 M x 16 brute-force triangle tests, no CPU tree
    brute = O.OracleScene(data)
    monkeypatch.delenv("ORC_NO_BVH")
    btri, bt = brute.trace_closest(rays[pick], use_bvh=False)
    del brute
    assert np.array_equal(tri[pick], btri) and np.array_equal(bits(t[pick]), bits(bt))
    whole = scene.render(cfg)
    assert np.isfinite(whole).all() and whole.mean() > 0.01
    # ... and 24 full rows of the image against the oracle walking its own tree (median split, f64 boxes): every
    # bounce of every sample of those pixels, the numbers recorded with the other full-size comparisons
    oracle = O.OracleScene(data)
    rows = list(range(40, 2160, 90))
    ref = np.zeros_like(whole)
    for y in rows:
        oracle.render(cfg, ref, tile=(0, y, 3840, y + 1), threads=16, use_bvh=True)
    del oracle
    a, b = whole[rows], ref[rows]
    nums = parity_numbers(a, b, 1e-9, cfg.samples_pr_pixel)
    record_parity("config5_heightfield_10M_3840x2160_2spp", nums)
    assert b.mean() > 0.01 and nums["fraction_over_tolerance"] < 2e-3, nums
    acc = scene.render(cfg, None, sample_begin=0, sample_count=1)
    scene.render(cfg, acc, sample_begin=1, sample_count=1)
    assert acc.tobytes() == whole.tobytes()
    acc = np.zeros_like(whole)
    for y0, y1 in ((0, 700), (700, 2160)):
        scene.render(cfg, acc, tile=(0, y0, 3840, y1))
    assert acc.tobytes() == whole.tobytes()
    data.bvh_nodes, data.bvh_tri_order = nodes, order      # same tree, no second 10 M build
    for env in (dict(PT_TRAV_VOTE="0", PT_TRAV_VOTE_SHADOW="0", PT_FUSE="0", PT_PIPELINE="0"),
                dict(PT_PERSISTENT="0", PT_WAVE_RAYS="1024", PT_LIGHT_CULL="0")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = pt.Scene(data, gpu)
        for k in env:
            monkeypatch.delenv(k)
        assert other.render(cfg).tobytes() == whole.tobytes(), env
        del other


@pytest.mark.parametrize("name,kw,width,height,spp,tol,frac", [
    ("glossy_planes", dict(glossy=True), 1920, 1080, 4, 1e-6, 5e-3),      # BASELINE configs[2]
    ("Test", dict(textures=True), 3840, 2160, 2, 1e-9, 2e-3),             # BASELINE configs[3]
])
def test_baseline_full_size_glossy_and_textured(gpu, name, kw, width, height, spp, tol, frac):
    # configs[2] and configs[3] at their full image sizes: a strided set of full rows against the
    # oracle (same frames, same RNG streams) and the size-independent properties of the render
    data = pt.scene_data(name, **kw)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    cfg = pt.Config(width=width, height=height, samples_pr_pixel=spp, bounces=2, seed=2, drop_nonfinite=True)
    whole = scene.render(cfg)
    assert np.isfinite(whole).all() and whole.mean() > 1e-3
    rows = range(7, height, height // 24)
    ref = np.zeros_like(whole)
    for y in rows:
        oracle.render(cfg, ref, tile=(0, y, width, y + 1), threads=16)
    sub, rsub = whole[list(rows)], ref[list(rows)]
    record_parity(f"full_size_{name}_{width}x{height}_{spp}spp", parity_numbers(sub, rsub, tol, spp))
    assert (np.abs(sub - rsub) > tol * np.maximum(1.0, np.abs(rsub))).mean() < frac
    assert abs(sub.mean() - rsub.mean()) <= 1e-3 * rsub.mean()
    acc = scene.render(cfg, None, sample_begin=0, sample_count=1)
    scene.render(cfg, acc, sample_begin=1, sample_count=spp - 1)
    assert acc.tobytes() == whole.tobytes()
    acc = np.zeros_like(whole)
    for x0, x1 in ((0, width // 3), (width // 3, width)):
        scene.render(cfg, acc, tile=(x0, 0, x1, height))
    assert acc.tobytes() == whole.tobytes()
    nee = scene.render(pt.Config(width=width, height=height, samples_pr_pixel=1, bounces=2, seed=2, nee=True, drop_nonfinite=True))
    assert np.isfinite(nee).all() and nee.mean() > 1e-3


@pytest.mark.parametrize("name", ["CornellBox", "Test", "glossy_planes", "heightfield40"])
def test_eye_tables_are_bitwise_identical(gpu, monkeypatch, name):
    # primary rays traced through the eye-relative tables (hoisted origin-dependent arithmetic) against
    # the general traversal: the same bits, images included
    if name == "heightfield40":
        from pathtracer_amd.synthetic import heightfield_box
        data = heightfield_box(cells=40)
    else:
        data = pt.scene_data(name, textures=True) if name == "Test" else pt.scene_data(name)
    monkeypatch.setenv("PT_EYE_TABLES", "0")
    general = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_EYE_TABLES")
    tables = pt.Scene(data, gpu)
    monkeypatch.setenv("PT_LDS_SCENE", "0")          # tables read from global memory, not from the LDS copy
    from_global = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_LDS_SCENE")
    for cfg in (pt.Config(width=211, height=97, samples_pr_pixel=3, bounces=2, seed=5, drop_nonfinite=True),
                pt.Config(dim=64, samples_pr_pixel=2, bounces=0, seed=1, nee=True, drop_nonfinite=True)):
        assert general.render(cfg).tobytes() == tables.render(cfg).tobytes(), (name, cfg)
        assert from_global.render(cfg).tobytes() == tables.render(cfg).tobytes(), (name, cfg)


@pytest.mark.parametrize("name", ["CornellBox", "Test", "glossy_planes", "heightfield40"])
def test_last_bounce_emitter_cull_is_bit_identical(gpu, monkeypatch, name):
    # a ray spawned at the second-to-last vertex that passes no emitter's triangle test is dropped at
    # its producer instead of being queued and traced (black background): same bits, fewer rays.
    # glossy_planes' alpha = 1e-5 strips make NaN / inf throughputs, which the cull must keep.
    if name == "heightfield40":
        from pathtracer_amd.synthetic import heightfield_box
        data = heightfield_box(cells=40)
    else:
        data = pt.scene_data(name, textures=True, specular=True) if name == "Test" else pt.scene_data(name, glossy=name == "glossy_planes")
    monkeypatch.setenv("PT_LIGHT_CULL", "0")
    full = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_LIGHT_CULL")
    cull = pt.Scene(data, gpu)
    for cfg in (pt.Config(width=160, height=90, samples_pr_pixel=4, bounces=2, seed=5),
                pt.Config(dim=72, samples_pr_pixel=3, bounces=1, seed=2, nee=True),
                pt.Config(dim=64, samples_pr_pixel=2, bounces=5, seed=9, nee=True, russian_roulette=True),
                pt.Config(dim=64, samples_pr_pixel=2, bounces=0, seed=1)):
        full.reset_stats(); cull.reset_stats()
        a, b = full.render(cfg), cull.render(cfg)
        assert a.tobytes() == b.tobytes(), (name, cfg)
        if cfg.bounces:
            assert cull.stats()["rays"] < full.stats()["rays"]
    # the unfused two-kernel form culls the same way
    monkeypatch.setenv("PT_FUSE", "0")
    unfused = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_FUSE")
    cfg = pt.Config(dim=80, samples_pr_pixel=3, bounces=3, seed=4)
    assert unfused.render(cfg).tobytes() == full.render(cfg).tobytes()
    # a scene with a bright background cannot cull: every miss adds radiance
    import copy
    lit = copy.copy(data)
    lit.background = (0.25, 0.5, 1.0)
    sl, ol = pt.Scene(lit, gpu), O.OracleScene(lit)
    cfg = pt.Config(dim=48, samples_pr_pixel=2, bounces=2, seed=3, drop_nonfinite=True)
    img, ref = sl.render(cfg), ol.render(pt.Config(dim=48, samples_pr_pixel=2, bounces=2, seed=3, drop_nonfinite=True))
    assert (np.abs(img - ref) > 1e-6 * np.maximum(1.0, np.abs(ref))).mean() < 1e-2


@pytest.mark.parametrize("name", ["CornellBox", "Test", "glossy_planes", "heightfield40"])
def test_screen_bounds_cull_is_bit_identical(gpu, monkeypatch, name):
    # primary rays through pixels outside the projection of the scene's root box are not generated:
    # same image, whether the eye is outside the box (CornellBox: 44 % of the columns are culled),
    # inside it, or the image is rendered in tiles / at odd sizes
    if name == "heightfield40":
        from pathtracer_amd.synthetic import heightfield_box
        data = heightfield_box(cells=40)
    else:
        data = pt.scene_data(name)
    monkeypatch.setenv("PT_SCREEN_CULL", "0")
    full = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_SCREEN_CULL")
    cull = pt.Scene(data, gpu)
    for cfg in (pt.Config(width=320, height=180, samples_pr_pixel=3, bounces=2, seed=5, drop_nonfinite=True),
                pt.Config(width=211, height=97, samples_pr_pixel=2, bounces=1, seed=2, nee=True, drop_nonfinite=True),
                pt.Config(dim=64, samples_pr_pixel=2, bounces=0, seed=1)):
        a = full.render(cfg)
        assert a.tobytes() == cull.render(cfg).tobytes(), (name, cfg)
        acc = np.zeros_like(a)
        for x0, x1 in ((0, cfg.w // 3), (cfg.w // 3, cfg.w)):
            cull.render(cfg, acc, tile=(x0, 0, x1, cfg.h))
        assert acc.tobytes() == a.tobytes()
    # a camera that sees the scene at the very edge of a wide image, and one looking away from it
    import copy
    for view in ((0.9, 0.1, -1.0), (0.0, 0.0, 1.0)):
        moved = copy.copy(data)
        c = data.camera
        moved.camera = pt.Camera(c.eye, view, c.up_direction, c.camera_const)
        a, b = pt.Scene(moved, gpu), O.OracleScene(moved)
        cfg = pt.Config(width=160, height=48, samples_pr_pixel=2, bounces=1, seed=3, drop_nonfinite=True)
        img, ref = a.render(cfg), b.render(cfg)
        assert (np.abs(img - ref) > 1e-6 * np.maximum(1.0, np.abs(ref))).mean() < 1e-2


def test_thin_and_tiny_tiles_assemble_bitwise(scenes):
    # tiles without a single whole 8x8 block (one pixel high, one pixel wide, 3x5, 9x7): the frame's
    # slot order degenerates to its strips; the image assembled from them equals the one-call image
    data, scene, oracle = scenes["CornellBox"]
    cfg = pt.Config(width=37, height=23, samples_pr_pixel=3, bounces=2, seed=7, nee=True)
    whole = scene.render(cfg)
    for tw, th in ((37, 1), (1, 23), (3, 5), (9, 7), (16, 8)):
        acc = np.zeros_like(whole)
        for y0 in range(0, 23, th):
            for x0 in range(0, 37, tw):
                scene.render(cfg, acc, tile=(x0, y0, min(x0 + tw, 37), min(y0 + th, 23)))
        assert acc.tobytes() == whole.tobytes(), (tw, th)
    ref = oracle.render(cfg)
    assert (np.abs(whole - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 5e-3


@pytest.mark.parametrize("n_tris", [31, 32, 33, 64])
def test_lds_scene_boundary_sizes(gpu, monkeypatch, n_tris):
    # the fused kernel keeps scenes of up to 32 nodes / triangles in LDS: sizes on both sides of the
    # limit, LDS copy against global memory (bitwise) and against the oracle
    rng = np.random.default_rng(n_tris)
    c = rng.uniform(-1.0, 1.0, (n_tris, 1, 3)) * [1.0, 1.0, 0.3] + [0.0, 0.0, 3.0]
    verts = (c + rng.normal(0.0, 0.35, (n_tris, 3, 3))).reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * n_tris, dtype=np.uint32).reshape(-1, 3)
    mats = np.zeros(3, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"][0], mats["color"][1] = [0.7, 0.6, 0.5], [0.2, 0.5, 0.8]
    mats["emission"][2] = [5.0, 4.0, 3.0]
    tri_mat = (np.arange(n_tris) % 3).astype(np.uint32)
    cam = pt.Camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 1.2)
    data = pt.SceneData(verts, tris, tri_mat, mats, camera=cam)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    monkeypatch.setenv("PT_LDS_SCENE", "0")
    from_global = pt.Scene(data, gpu)
    monkeypatch.delenv("PT_LDS_SCENE")
    for cfg in (pt.Config(width=96, height=64, samples_pr_pixel=4, bounces=3, seed=1),
                pt.Config(width=96, height=64, samples_pr_pixel=3, bounces=2, seed=2, nee=True)):
        img = scene.render(cfg)
        assert img.tobytes() == from_global.render(cfg).tobytes()
        ref = oracle.render(cfg)
        assert ref.mean() > 1e-3
        assert (np.abs(img - ref) > 1e-9 * np.maximum(1.0, np.abs(ref))).mean() < 5e-3
    rays = oracle.generate_rays(pt.Config(width=96, height=64), 0)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=False)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))


def test_config1_as_named_full_image(scenes):
    # BASELINE.json configs[0] as named: CornellBox 512 x 512, 16 frames x 1 spp, bounces 2 (the reference's own
    # CPU-runnable case: --dim 512, main.rs:346-347 defaults).  The WHOLE image, HIP path vs oracle (4.2 M samples),
    # frame by frame through the frame loop's shape (one frame per call, main.rs:400-436) and in one call.
    data, scene, oracle = scenes["CornellBox"]
    spp = 16
    cfg = pt.Config(dim=512, samples_pr_pixel=spp, bounces=2, seed=1)
    img = scene.render(cfg)
    ref = oracle.render(cfg, threads=16, use_bvh=True)
    nums = parity_numbers(img, ref, 1e-9, spp)
    record_parity("config1_CornellBox_512x512_16spp_full_image", nums)
    assert ref.mean() > 0.05 * spp / 16 and np.isfinite(img).all()
    assert nums["rel_rmse"] <= 1e-9, nums
    assert nums["fraction_over_tolerance"] < 1e-3, nums
    assert nums["identical_8bit"], nums
    loop = np.zeros_like(img)
    for frame in range(spp):                                  # pixels[i] += next[i], one frame per call
        scene.render(cfg, loop, sample_begin=frame, sample_count=1)
    assert loop.tobytes() == img.tobytes()
    # primary hits of the whole 512 x 512 frame 0: ids and distances bit-exact
    rays = oracle.generate_rays(cfg, 0)
    tri, t = scene.trace_closest(rays)
    rtri, rt = oracle.trace_closest(rays, use_bvh=False)
    assert np.array_equal(tri, rtri) and np.array_equal(bits(t), bits(rt))


def test_config2_converged_image_psnr(scenes):
    # BASELINE.json configs[1] as named: CornellBox 1920x1080 at 1024 spp.  The converged image under
    # identical per-pixel RNG streams against the oracle on a strided set of full rows: relative RMSE
    # and the PSNR of the 8-bit resolve (main.rs:456-467, no gamma).  Stated tolerance: relative RMSE
    # <= 1e-9 on the f64 sums, PSNR >= 80 dB (in fact every 8-bit value is equal).
    data, scene, oracle = scenes["CornellBox"]
    spp = 1024
    cfg = pt.Config(width=1920, height=1080, samples_pr_pixel=spp, bounces=2, seed=1)
    img = scene.render(cfg)
    rows = list(range(11, 1080, 54))                       # 20 rows x 1920 px x 1024 spp = 39 M oracle samples
    ref = np.zeros_like(img)
    for y in rows:
        oracle.render(cfg, ref, tile=(0, y, 1920, y + 1), threads=16)
    a, b = img[rows], ref[rows]
    record_parity("config2_CornellBox_1920x1080_1024spp", parity_numbers(a, b, 1e-9, spp))
    assert b.mean() > 1.0                                   # sums of 1024 samples
    rel_rmse = np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())
    assert rel_rmse <= 1e-9, rel_rmse
    qa, qb = pt.resolve_rgb8(a, spp).astype(np.float64), pt.resolve_rgb8(b, spp).astype(np.float64)
    mse = ((qa - qb) ** 2).mean()
    psnr = np.inf if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert psnr >= 80.0, psnr
    assert np.array_equal(qa, qb)


def test_config3_converged_glossy_psnr(gpu):
    # BASELINE.json configs[2]: glossy_planes 1080p at 256 spp (Beckmann strips down to alpha = 1e-5).
    # Non-finite samples are dropped on both sides (the reference would accumulate them); the sets can
    # differ where acos of a dot product rounds past 1 on one side only, hence a PSNR bound instead of
    # equality.  Stated tolerance: PSNR >= 50 dB on the 8-bit resolve, mean radiance within 1e-4.
    data = pt.scene_data("glossy_planes", glossy=True)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    spp = 256
    cfg = pt.Config(width=1920, height=1080, samples_pr_pixel=spp, bounces=2, seed=4, drop_nonfinite=True)
    img = scene.render(cfg)
    rows = list(range(17, 1080, 90))                       # 12 rows x 1920 px x 256 spp
    ref = np.zeros_like(img)
    for y in rows:
        oracle.render(cfg, ref, tile=(0, y, 1920, y + 1), threads=16)
    a, b = img[rows], ref[rows]
    record_parity("config3_glossy_planes_1920x1080_256spp_drop_nonfinite", parity_numbers(a, b, 1e-6, spp))
    assert np.isfinite(a).all() and np.isfinite(b).all() and b.mean() > 1.0
    assert abs(a.mean() - b.mean()) <= 1e-4 * b.mean()
    qa, qb = pt.resolve_rgb8(a, spp).astype(np.float64), pt.resolve_rgb8(b, spp).astype(np.float64)
    mse = ((qa - qb) ** 2).mean()
    psnr = np.inf if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert psnr >= 50.0, psnr


def test_config3_glossy_full_size_nonfinite_samples_as_the_reference(gpu):
    # BASELINE.json configs[2] at its full size in the reference's own mode: non-finite samples are ACCUMULATED, as
    # main.rs:434-436 does, not dropped.  On THIS scene there are none, and there cannot be: Beckmann::D's acos
    # (microfacet.rs:161-173) is NaN only when h.n rounds past 1, and every plane of glossy_planes.obj is
    # axis-aligned, so h.n is one component of a normalised vector (<= 1 exactly; 67 M oracle samples and the
    # whole 1080p device image hold no non-finite value).  The test therefore pins that the two modes coincide at
    # full size; the NaN-accumulating mode itself is tested where NaNs are known to exist:
    # test_gpu_shading.test_nan_accumulating_mode_on_a_tilted_near_mirror.
    data = pt.scene_data("glossy_planes", glossy=True)
    scene, oracle = pt.Scene(data, gpu), O.OracleScene(data)
    spp = 32
    cfg = pt.Config(width=1920, height=1080, samples_pr_pixel=spp, bounces=2, seed=4)
    img = scene.render(cfg)
    # 24 rows at a fixed stride, plus up to 16 of the rows where the device image holds a non-finite pixel
    # (they are rare at this size; where there are none the recorded count says so)
    bad_rows = np.flatnonzero(~np.isfinite(img).all(axis=(1, 2)))
    rows = sorted(set(range(5, 1080, 45)) | set(int(y) for y in bad_rows[:16]))
    ref = np.zeros_like(img)
    for y in rows:
        oracle.render(cfg, ref, tile=(0, y, 1920, y + 1), threads=16)
    a, b = img[rows], ref[rows]
    nums = parity_numbers(a, b, 1e-6, spp)
    nums["rows_compared"] = len(rows)
    nums["nonfinite_values_in_whole_gpu_image"] = int((~np.isfinite(img)).sum())
    record_parity("config3_glossy_planes_1920x1080_32spp_nonfinite_accumulated", nums)
    assert nums["nonfinite_values_in_whole_gpu_image"] == 0 and nums["nonfinite_oracle"] == 0, nums
    assert nums["fraction_over_tolerance"] < 5e-3, nums
