"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/pt.h declares, the OBJ/MTL loader, the BVH builder, PNG decode, resolve/PPM, host RNG —
and the compute entry points fail loudly without a GPU (no CPU fallback).  No compute on the GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import _lib as L
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    lib = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in L.lib.pt_version()


def test_struct_layouts_match_the_header():
    assert C.sizeof(L.pt_bvh_node) == 64
    assert C.sizeof(L.pt_ray) == 64
    assert C.sizeof(L.pt_material) == 40
    assert C.sizeof(L.pt_render_params) == 80
    assert C.sizeof(L.pt_camera) == 80
    assert C.sizeof(L.pt_scene_options) == 32
    assert C.sizeof(L.pt_area_light) == 56


def test_scene_options_defaults_environment_and_validation(monkeypatch):
    o = L.pt_scene_options()
    L.lib.pt_scene_options_init(C.byref(o))
    flags = [n for n, t in L.pt_scene_options._fields_ if t is C.c_int8]
    assert o.struct_size == 32 and all(getattr(o, n) == -1 for n in flags)
    assert (o.wave_rays, o.bvh_max_leaf, o.bvh_threads) == (0, 0, 0)
    # the environment is read by pt_scene_options_from_env alone (the sweeps' hook, what pt_scene_create uses)
    for k, v in (("PT_FUSE", "0"), ("PT_TRAV_VOTE", "1"), ("PT_WAVE_RAYS", "128"), ("PT_BVH_LEAF", "4"), ("PT_LDS_SCENE", "0")):
        monkeypatch.setenv(k, v)
    L.lib.pt_scene_options_from_env(C.byref(o))
    assert (o.fuse, o.vote_secondary, o.wave_rays, o.bvh_max_leaf, o.lds_scene) == (0, 1, 128, 4, 0)
    assert o.vote_primary == -1 and o.pipeline == -1
    monkeypatch.setenv("PT_WAVE_RAYS", "100")               # not a power of two: the sweep falls back to the default
    L.lib.pt_scene_options_from_env(C.byref(o))
    assert o.wave_rays == 0
    # bad options are refused before any device is touched
    data = pt.scene_data("CornellBox")
    for bad, why in ((dict(wave_rays=100), "wave_rays"), (dict(bvh_max_leaf=9), "bvh_max_leaf"), (dict(bvh_threads=65), "bvh_threads")):
        with pytest.raises(pt.PtError, match=why):
            pt.Scene(data, options=bad)
    with pytest.raises(ValueError):
        pt.Scene(data, options=dict(no_such_option=1))
    d, h = data.desc(), C.c_void_p()
    o.struct_size = 12
    assert L.lib.pt_scene_create_ex(C.byref(d), 0, C.byref(o), C.byref(h)) == L.PT_ERR_INVALID
    assert b"struct_size" in L.lib.pt_last_error()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_a_gpu():
    with pytest.raises(pt.PtError) as e:
        pt.Scene(pt.scene_data("CornellBox"))
    assert e.value.code == L.PT_ERR_DEVICE and "HIP" in str(e.value)


# ---- obj.rs:13-107 / tobj semantics -----------------------------------------------------------------
def test_load_cornellbox():
    d = pt.scene_data("CornellBox")
    assert d.model_names == ["Floor", "Light", "Ceiling", "BackWall", "RightWall", "LeftWall"]
    assert d.model_first_tri == [0, 2, 4, 6, 8, 10] and d.n_tris == 12
    assert d.verts.shape == (24, 3) and d.tri_uvs is None
    # per-model re-indexing: first quad is (0,1,2),(0,2,3) of its own four vertices
    assert d.tris[:2].tolist() == [[0, 1, 2], [0, 2, 3]] and d.tris[2:4].tolist() == [[4, 5, 6], [4, 6, 7]]
    assert d.verts[d.tris[2]].tolist() == [[np.float32(0.66), np.float32(0.99), np.float32(0.33)],
                                           [np.float32(0.66), np.float32(0.99), np.float32(0.66)],
                                           [np.float32(0.33), np.float32(0.99), np.float32(0.66)]]
    m = d.mats[d.tri_material]
    assert np.allclose(m["color"][0], 0.9) and not m["emission"][0].any()                # white, illum 0
    assert m["emission"][2].tolist() == [np.float32(36.8), np.float32(31.2), np.float32(16.0)]  # Ka of `light`
    assert not m["color"][2].any()
    assert m["color"][8].tolist() == [0.0, np.float32(0.9), 0.0]                          # green right wall
    assert m["color"][10].tolist() == [np.float32(0.9), 0.0, 0.0]                         # red left wall
    assert (d.mats["kind"] == pt.PT_MAT_LAMBERT).all() and (d.mats["tex"] == -1).all()


def test_load_cornell_box_quad_has_no_material():
    # F5: no mtllib -> usemtl resolves to None -> black non-emissive default (obj.rs:55-58, 74-76)
    d = pt.scene_data("cornell_box")
    assert d.model_names == ["quad"] and d.n_tris == 2
    assert d.tris.tolist() == [[0, 1, 2], [0, 2, 3]]                  # negative indices, quad -> 2 triangles
    assert d.verts.tolist() == [[1, 0, 1], [0, 0, 1], [0, 1, 1], [1, 1, 1]]
    m = d.mats[d.tri_material]
    assert not m["color"].any() and not m["emission"].any()


def test_load_test_obj():
    d = pt.scene_data("Test")
    assert d.model_names == ["ceiling_Box.004", "leftWall_Box.003", "rightWall_Box.002", "backWall_Box.001", "light_Box"]
    assert d.model_first_tri == [0, 4, 6, 8, 10] and d.n_tris == 12
    m = d.mats[d.tri_material]
    assert np.allclose(m["color"][0], 0.5)                       # white Kd 0.5
    assert m["emission"][10].tolist() == [20.0, 20.0, 20.0]      # light Ka 20, negative face indices
    assert np.allclose(d.verts[d.tris[10]][:, 1], 548.7, atol=1e-3)
    # the back wall carries vt; other models get the default uvs (triangle_mesh.rs:93)
    assert d.tri_uvs is not None
    assert d.tri_uvs[0].tolist() == [[0, 0], [1, 0], [1, 1]]
    assert np.allclose(d.tri_uvs[8], [[0.9999, 0.0001], [0.0001, 0.0001], [0.0001, 0.9999]])
    assert (d.mats["tex"] == -1).all()                           # textures are not wired by default (F4)
    dt = pt.scene_data("Test", textures=True)
    tex_mat = dt.mats[dt.tri_material[8]]
    assert tex_mat["tex"] == 0 and len(dt.textures) == 1 and dt.textures[0].shape == (1024, 1024, 3)


def test_load_glossy_planes():
    d = pt.scene_data("glossy_planes")
    assert len(d.model_names) == 9 and d.n_tris == 18
    assert d.model_names[0] == "Floor_Plane" and d.model_names[-1] == "Floor_Plane"
    kinds = d.mats[d.tri_material]["kind"]
    assert kinds[:2].tolist() == [0, 0] and kinds[-2:].tolist() == [0, 0] and (kinds[2:16] == pt.PT_MAT_GLOSSY).all()
    rough = d.mats[d.tri_material]["roughness"][2:16:2]
    assert np.allclose(rough, [0.3, 0.1, 0.05, 0.01, 0.005, 0.005, 0.0])
    floor = d.mats[d.tri_material[0]]
    assert floor["emission"].tolist() == [1.0, 1.0, 1.0] and floor["color"].tolist() == [1.0, 1.0, 1.0]
    # reference behaviour (glossy wiring off): everything Lambertian
    assert (pt.scene_data("glossy_planes", glossy=False).mats["kind"] == 0).all()
    # absolute map_Kd path from the author's machine is remapped by basename (debug.png)
    dt = pt.scene_data("glossy_planes", textures=True)
    assert len(dt.textures) == 1 and dt.mats[dt.tri_material[0]]["tex"] == 0


def test_load_errors_are_status_codes_not_panics(tmp_path):
    with pytest.raises(pt.PtError) as e:
        pt.load_obj(str(tmp_path / "missing.obj"))
    assert e.value.code == L.PT_ERR_IO
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nf 1 2 9\n")
    with pytest.raises(pt.PtError):
        pt.load_obj(str(bad))
    poly = tmp_path / "poly.obj"
    poly.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0.5 1.5 0\nv 0 1 0\nf 1 2 3 4 5\nl 1 2\n")
    d = pt.load_obj(str(poly))
    assert d.model_names == ["unnamed_object"] and d.tris.tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4]]


# ---- bvh.rs:17-21 stand-in ----------------------------------------------------------------------------
def check_bvh(verts, tris, nodes, order, max_leaf):
    nt = tris.shape[0]
    assert sorted(order.tolist()) == list(range(nt))
    p = verts[tris[order]]
    lo, hi = p.min(axis=1), p.max(axis=1)
    seen = np.zeros(nt, bool)

    def walk(ref, depth):
        """returns (lo, hi) of the subtree"""
        if ref < 0:
            code = ~ref
            first, cnt = code >> 3, (code & 7) + 1
            assert 1 <= cnt <= max_leaf and first + cnt <= nt
            assert not seen[first:first + cnt].any()
            seen[first:first + cnt] = True
            return lo[first:first + cnt].min(axis=0), hi[first:first + cnt].max(axis=0), depth
        n = nodes[ref]
        l0 = h0 = None
        d = depth
        for c in range(2):
            ch = int(n["child"][c])
            if ch == L.PT_BVH_EMPTY:
                continue
            clo, chi, dd = walk(ch, depth + 1)
            b = n["bounds"][c * 6:c * 6 + 6]
            assert (b[:3] < clo).all() and (b[3:] > chi).all()   # padded strictly outward
            l0 = clo if l0 is None else np.minimum(l0, clo)
            h0 = chi if h0 is None else np.maximum(h0, chi)
            d = max(d, dd)
        return l0, h0, d

    _, _, depth = walk(0, 1)
    assert seen.all()
    return depth


def test_bvh_build_small_scenes():
    for name in ("CornellBox", "Test", "glossy_planes", "cornell_box"):
        d = pt.scene_data(name)
        for max_leaf in (1, 4, 8):
            nodes, order, depth = pt.bvh_build(d.verts, d.tris, max_leaf)
            assert check_bvh(d.verts, d.tris, nodes, order, max_leaf) <= 32


def test_bvh_build_random_and_degenerate():
    rng = np.random.default_rng(0)
    verts = rng.random((3000, 3)).astype(np.float32) * 100
    tris = rng.integers(0, 3000, (5000, 3)).astype(np.uint32)
    nodes, order, depth = pt.bvh_build(verts, tris, 4)
    assert check_bvh(verts, tris, nodes, order, 4) <= 32 and depth <= 30
    # all triangles identical: SAH has nothing to split on -> median fallback must terminate
    tris2 = np.tile(np.array([[0, 1, 2]], np.uint32), (257, 1))
    nodes, order, depth = pt.bvh_build(verts, tris2, 4)
    assert check_bvh(verts, tris2, nodes, order, 4) <= 32
    # very skewed distribution (exponentially spaced) stays within the depth budget
    x = (2.0 ** np.arange(0, 60, 0.01)).astype(np.float32)
    v = np.stack([x, np.zeros_like(x), np.zeros_like(x)], 1)
    v = np.concatenate([v, v + [0, 1, 0], v + [0, 0, 1]]).astype(np.float32)
    n = x.shape[0]
    t = np.stack([np.arange(n), np.arange(n) + n, np.arange(n) + 2 * n], 1).astype(np.uint32)
    nodes, order, depth = pt.bvh_build(v, t, 2)
    assert check_bvh(v, t, nodes, order, 2) <= 32
    with pytest.raises(pt.PtError):
        pt.bvh_build(verts, np.array([[0, 1, 3000]], np.uint32), 4)
    bad = verts.copy()
    bad[0, 0] = np.nan
    with pytest.raises(pt.PtError):
        pt.bvh_build(bad, tris, 4)
    nodes, order, depth = pt.bvh_build(verts, np.zeros((0, 3), np.uint32), 4)
    assert nodes.shape[0] == 0


# ---- mipmap.rs:59-74 decode, main.rs:456-467 resolve, mipmap.rs:25-42 PPM --------------------------------
@pytest.mark.parametrize("name", ["floor.png", "debug.png"])
def test_png_decode_matches_pil(name):
    from PIL import Image
    path = os.path.join(pt.MODELS_DIR, name)
    got = pt.decode_png(path)
    ref = np.asarray(Image.open(path).convert("RGBA"))[..., :3]   # to_rgb() drops alpha (mipmap.rs:62)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_png_decode_all_filters_and_types(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    for mode, arr in (("RGB", rng.integers(0, 256, (16, 32, 3), np.uint8)),
                      ("L", rng.integers(0, 256, (8, 8), np.uint8)),
                      ("RGBA", rng.integers(0, 256, (8, 16, 4), np.uint8))):
        p = str(tmp_path / f"{mode}.png")
        Image.fromarray(arr, mode).save(p)
        got = pt.decode_png(p)
        ref = np.asarray(Image.open(p).convert("RGBA"))[..., :3]
        assert np.array_equal(got, ref)
    with pytest.raises(pt.PtError):
        pt.decode_png(os.path.join(pt.MODELS_DIR, "Test.obj"))


def test_resolve_and_ppm_match_the_oracle(tmp_path):
    rng = np.random.default_rng(0)
    acc = rng.normal(2.0, 3.0, (5, 7, 3))
    acc[0, 0] = [np.nan, np.inf, -np.inf]
    got = pt.resolve_rgb8(acc, 4)
    ref = np.zeros_like(got)
    O.lib.orc_resolve_rgb8(acc.ctypes.data_as(O._dp), 35, 4, ref.ctypes.data_as(O._P(C.c_uint8)))
    assert np.array_equal(got, ref) and got[0, 0].tolist() == [0, 255, 0]
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    pt.save_ppm(a, got)
    O.lib.orc_write_ppm(os.fsencode(b), 7, 5, ref.ctypes.data_as(O._P(C.c_uint8)))
    assert open(a).read() == open(b).read() and open(a).read().startswith("P3\n7 5\n255\n")


def test_host_rng_matches_the_oracle():
    rng = np.random.default_rng(0)
    for _ in range(200):
        args = (int(rng.integers(0, 2**63)), int(rng.integers(0, 2**32)), int(rng.integers(0, 2**40)),
                int(rng.integers(0, 62)), int(rng.integers(0, 3)))
        assert pt.rng_uniform2(*args) == O.rng_uniform2(*args)


def test_checkpoint_roundtrip_and_damage(tmp_path):
    # the running sums + frame count on disk, bit for bit, and damage is detected
    rng = np.random.default_rng(3)
    acc = rng.random((37, 53, 3)) * 1e3
    acc[0, 0] = [np.inf, -0.0, 5e-324]
    path = str(tmp_path / "run.ckpt")
    pt.save_checkpoint(path, acc, frames=1234567890123, seed=2**63 + 5)
    back, frames, seed = pt.load_checkpoint(path)
    assert back.tobytes() == acc.tobytes() and frames == 1234567890123 and seed == 2**63 + 5
    assert not (tmp_path / "run.ckpt.tmp").exists()
    raw = bytearray(open(path, "rb").read())
    assert len(raw) == 48 + acc.nbytes
    flipped = bytearray(raw)
    flipped[48 + 1000] ^= 1
    open(path, "wb").write(flipped)
    with pytest.raises(pt.PtError, match="checksum"):
        pt.load_checkpoint(path)
    open(path, "wb").write(raw[:-8])
    with pytest.raises(pt.PtError, match="truncated"):
        pt.load_checkpoint(path)
    open(path, "wb").write(b"P3\n1 1\n255\n0 0 0\n" + bytes(64))
    with pytest.raises(pt.PtError, match="not a checkpoint"):
        pt.load_checkpoint(path)
    with pytest.raises(pt.PtError):
        pt.load_checkpoint(str(tmp_path / "missing.ckpt"))


def test_parallel_bvh_build_is_independent_of_thread_count():
    # the 10 M-triangle scene is built by worker threads; the tree must be the same
    # array whatever the thread count (serial std::partition vs. the stable parallel partition,
    # subtree splicing)
    from pathtracer_amd.synthetic import heightfield_box
    data = heightfield_box(cells=260)               # 135,210 triangles: above the parallel threshold
    out = []
    for threads in (1, 2, 7):
        nodes, order, depth = pt.bvh_build(data.verts, data.tris, threads=threads)
        out.append((nodes.tobytes(), order.tobytes(), depth))
    assert out[0] == out[1] == out[2]
    assert sorted(np.frombuffer(out[0][1], np.uint32).tolist()) == list(range(data.n_tris))


# ---- caller-supplied trees (pt_bvh_validate): what the traversal kernels assume is checked on the host ----
def _chain_tree(n_tris):
    """A degenerate left-deep tree over n_tris triangles: inner node i = {leaf i, node i + 1}."""
    nodes = np.zeros(n_tris - 1, pt.NODE_DTYPE)
    nodes["bounds"][:, 0:3], nodes["bounds"][:, 3:6] = -1.0, 1.0
    nodes["bounds"][:, 6:9], nodes["bounds"][:, 9:12] = -1.0, 1.0
    for i in range(n_tris - 1):
        nodes["child"][i, 0] = ~(i << 3)                       # leaf: one triangle, slot i
        nodes["child"][i, 1] = i + 1 if i + 2 < n_tris else ~((i + 1) << 3)
    return nodes, np.arange(n_tris, dtype=np.uint32)


def _validate(nodes, order, n_tris):
    depth = C.c_uint32()
    rc = L.lib.pt_bvh_validate(C.cast(nodes.ctypes.data, C.POINTER(L.pt_bvh_node)), len(nodes),
                               order.ctypes.data_as(C.POINTER(C.c_uint32)), n_tris, C.byref(depth))
    return rc, depth.value, (L.lib.pt_last_error() or b"").decode()


def test_tree_deeper_than_the_traversal_stack_is_refused():
    # the kernels' LDS stack has 32 entries per lane and a push beyond it would be dropped (geometry
    # silently missing): such a tree must never reach them
    nodes, order = _chain_tree(33)                             # 32 inner levels: fits exactly
    rc, depth, _ = _validate(nodes, order, 33)
    assert rc == L.PT_OK and depth == 32
    nodes, order = _chain_tree(34)                             # 33 inner levels
    rc, _, why = _validate(nodes, order, 34)
    assert rc == L.PT_ERR_UNSUPPORTED and "deeper" in why
    # the library's own builder stays inside the bound whatever the input (here: 5000 coincident triangles)
    verts = np.zeros((3, 3), np.float32); verts[1, 0] = verts[2, 1] = 1.0
    tris = np.tile(np.array([[0, 1, 2]], np.uint32), (5000, 1))
    nodes, order, depth = pt.bvh_build(verts, tris, max_leaf=1)
    assert depth <= 32 and _validate(nodes, order, 5000)[0] == L.PT_OK


def test_validate_refuses_malformed_trees():
    data = pt.scene_data("CornellBox")
    nodes, order, _ = pt.bvh_build(data.verts, data.tris, max_leaf=2)
    assert _validate(nodes, order, data.n_tris)[0] == L.PT_OK
    bad = nodes.copy(); bad["child"][0, 0] = 10_000
    assert _validate(bad, order, data.n_tris)[0] == L.PT_ERR_INVALID
    bad = nodes.copy(); bad["child"][0, 1] = 0                 # cycle back to the root
    assert _validate(bad, order, data.n_tris)[0] == L.PT_ERR_INVALID
    assert _validate(nodes, np.zeros_like(order), data.n_tris)[0] == L.PT_ERR_INVALID      # not a permutation
    assert _validate(nodes[:1], order, data.n_tris)[0] == L.PT_ERR_INVALID                 # leaves do not cover
