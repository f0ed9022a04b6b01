"""Host-side mirror of the reference's scene/driver interface over the C ABI (include/pt.h).

Names follow the reference: ``Camera`` (scene.rs:20-35), ``Scene`` with ``cornell_box()`` /
``glossy_planes()`` constructors (scene.rs:70-104, 140-193), ``Scene.trace_closest``
(scene.rs:61), ``load_obj`` (obj.rs:13), ``Config`` (main.rs:320-379) and the frame driver
``render`` (main.rs:400-436).  All radiance arithmetic runs in the HIP library; numpy is only
used to carry buffers.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

MODELS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models")

RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("t_min", "<f8"), ("t_max", "<f8")])
assert RAY_DTYPE.itemsize == C.sizeof(L.pt_ray)
NODE_DTYPE = np.dtype([("bounds", "<f4", 12), ("child", "<i4", 2), ("pad", "<u4", 2)])
MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("emission", "<f4", 3), ("roughness", "<f4"),
                           ("kind", "<i4"), ("tex", "<i4"), ("ior", "<f4")])
assert MATERIAL_DTYPE.itemsize == C.sizeof(L.pt_material)
PRIM_DTYPE = np.dtype([("kind", "<i4"), ("material", "<u4"), ("p", "<f8", 3), ("n", "<f8", 3), ("radius", "<f8")])
assert PRIM_DTYPE.itemsize == C.sizeof(L.pt_primitive)


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class Camera:
    """scene.rs:20-35."""
    eye: Sequence[float]
    view_direction: Sequence[float]
    up_direction: Sequence[float]
    camera_const: float

    def to_c(self) -> L.pt_camera:
        c = L.pt_camera()
        c.eye[:] = [float(x) for x in self.eye]
        c.view[:] = [float(x) for x in self.view_direction]
        c.up[:] = [float(x) for x in self.up_direction]
        c.camera_const = float(self.camera_const)
        return c


@dataclass
class Config:
    """main.rs:320-379 CONFIG (flag set and defaults) plus the partitioning knobs of the C ABI."""
    dim: int = 512                 # --dim
    samples_pr_pixel: int = 8      # --samples (the reference parses and ignores it, F10)
    bounces: int = 2               # -b
    use_diffs: bool = True         # --no-diffs clears it
    width: Optional[int] = None    # non-square extension (F7); default dim x dim
    height: Optional[int] = None
    seed: int = 1
    nee: bool = False
    drop_nonfinite: bool = False
    russian_roulette: bool = False  # legacy integrator main.rs:85-93
    propagate_diffs: bool = False   # legacy reflect_diff / refract_diff, main.rs:192-281 (§8f N3)
    diff_size: float = 0.0          # --diff (main.rs:348)
    roughdiff: bool = False         # -p (main.rs:351)
    reference_lights: bool = False  # lights.rs:42-78,110-137 as written, against SceneData.area_light
    spp_per_batch: int = 0

    @property
    def w(self) -> int:
        return self.width if self.width is not None else self.dim

    @property
    def h(self) -> int:
        return self.height if self.height is not None else self.dim

    def flags(self) -> int:
        f = 0
        if self.use_diffs:
            f |= L.PT_FLAG_USE_DIFFS
        if self.nee:
            f |= L.PT_FLAG_NEE
        if self.drop_nonfinite:
            f |= L.PT_FLAG_DROP_NONFINITE
        if self.russian_roulette:
            f |= L.PT_FLAG_RUSSIAN_ROULETTE
        if self.propagate_diffs:
            f |= L.PT_FLAG_PROPAGATE_DIFFS
        if self.roughdiff:
            f |= L.PT_FLAG_ROUGHDIFF
        if self.reference_lights:
            f |= L.PT_FLAG_REFERENCE_LIGHTS
        return f

    def params(self, sample_begin: int, sample_count: int, tile=None, rows=None) -> L.pt_render_params:
        """``rows`` = (row_group, row_parts, row_index): the interleaved row partition of pt_render_params."""
        x0, y0, x1, y1 = tile if tile is not None else (0, 0, self.w, self.h)
        p = L.pt_render_params()
        p.width, p.height = self.w, self.h
        p.x0, p.y0, p.x1, p.y1 = x0, y0, x1, y1
        p.sample_begin, p.sample_count = sample_begin, sample_count
        p.max_bounces = self.bounces
        p.flags = self.flags()
        p.spp_per_batch = self.spp_per_batch
        p.seed = self.seed
        p.diff_size = self.diff_size
        if rows is not None:
            p.row_group, p.row_parts, p.row_index = rows
        return p


class SceneData:
    """A flat scene description (pt_scene_desc) held in numpy arrays.

    Built by ``load_obj`` or by hand; consumed by ``Scene`` (the HIP product) and, in tests, by the
    CPU oracle, which takes the very same struct.
    """

    def __init__(self, verts, tris, tri_material, mats, tri_uvs=None, textures=(), camera: Camera = None,
                 background=(0.0, 0.0, 0.0), model_names=(), model_first_tri=(), prims=None, area_light=None):
        self.verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self.tris = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 3)
        self.tri_material = np.ascontiguousarray(tri_material, dtype=np.uint32).reshape(-1)
        self.mats = np.ascontiguousarray(mats, dtype=MATERIAL_DTYPE).reshape(-1)
        self.tri_uvs = None if tri_uvs is None else np.ascontiguousarray(tri_uvs, dtype=np.float32).reshape(-1, 3, 2)
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]  # (h, w, 3) each
        self.camera = camera or Camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 1.0)
        self.background = tuple(float(x) for x in background)
        self.model_names = list(model_names)
        self.model_first_tri = list(model_first_tri)
        self.bvh_nodes = None
        self.bvh_tri_order = None
        # analytic spheres / planes (geometry/mod.rs:20-29), PRIM_DTYPE records; hit ids PT_PRIM_BASE + k
        self.prims = None if prims is None else np.ascontiguousarray(prims, dtype=PRIM_DTYPE).reshape(-1)
        # lights.rs:37-40 AreaLight: (center, radius, intensity), the light of PT_FLAG_REFERENCE_LIGHTS
        self.area_light = area_light

    @property
    def n_tris(self) -> int:
        return int(self.tris.shape[0])

    def desc(self) -> L.pt_scene_desc:
        d = L.pt_scene_desc()
        d.verts = _ptr(self.verts, C.c_float)
        d.n_verts = self.verts.shape[0]
        d.tris = _ptr(self.tris, C.c_uint32)
        d.n_tris = self.tris.shape[0]
        d.tri_uvs = _ptr(self.tri_uvs, C.c_float) if self.tri_uvs is not None else None
        d.tri_material = _ptr(self.tri_material, C.c_uint32)
        d.mats = C.cast(self.mats.ctypes.data, C.POINTER(L.pt_material))
        d.n_mats = self.mats.shape[0]
        self._tex_structs = (L.pt_texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            self._tex_structs[i].height, self._tex_structs[i].width = t.shape[0], t.shape[1]
            self._tex_structs[i].rgb8 = _ptr(t, C.c_uint8)
        d.texs = self._tex_structs if self.textures else None
        d.n_texs = len(self.textures)
        d.camera = self.camera.to_c()
        d.background[:] = self.background
        if self.bvh_nodes is not None:
            d.bvh_nodes = C.cast(self.bvh_nodes.ctypes.data, C.POINTER(L.pt_bvh_node))
            d.n_bvh_nodes = self.bvh_nodes.shape[0]
            d.bvh_tri_order = _ptr(self.bvh_tri_order, C.c_uint32)
        if self.prims is not None and len(self.prims):
            d.prims = C.cast(self.prims.ctypes.data, C.POINTER(L.pt_primitive))
            d.n_prims = self.prims.shape[0]
        if self.area_light is not None:
            center, radius, intensity = self.area_light
            self._area_light = L.pt_area_light()
            self._area_light.center[:] = [float(x) for x in center]
            self._area_light.radius = float(radius)
            self._area_light.intensity[:] = [float(x) for x in intensity]
            d.area_lights = C.pointer(self._area_light)
            d.n_area_lights = 1
        return d


def load_obj(obj_name: str, glossy: bool = False, textures: bool = False, specular: bool = False,
             ggx: bool = False) -> SceneData:
    """obj.rs:13-107 ``load_obj``; ``glossy`` / ``textures`` / ``specular`` enable the wiring the
    reference has commented out (obj.rs:30-53; illum 3 = Mirror, 4 = Transparent)."""
    flags = ((L.PT_LOAD_GLOSSY if glossy else 0) | (L.PT_LOAD_TEXTURES if textures else 0)
             | (L.PT_LOAD_SPECULAR if specular else 0) | (L.PT_LOAD_GGX if ggx else 0))
    h = C.c_void_p()
    L.check(L.lib.pt_load_obj(os.fsencode(obj_name), flags, C.byref(h)))
    try:
        d = L.lib.pt_model_desc(h).contents
        nv, nt, nm = d.n_verts, d.n_tris, d.n_mats
        verts = np.ctypeslib.as_array(d.verts, shape=(nv * 3,)).copy() if nv else np.zeros(0, np.float32)
        tris = np.ctypeslib.as_array(d.tris, shape=(nt * 3,)).copy() if nt else np.zeros(0, np.uint32)
        tmat = np.ctypeslib.as_array(d.tri_material, shape=(nt,)).copy() if nt else np.zeros(0, np.uint32)
        uvs = np.ctypeslib.as_array(d.tri_uvs, shape=(nt * 6,)).copy() if (nt and d.tri_uvs) else None
        mats = np.frombuffer(C.string_at(d.mats, nm * C.sizeof(L.pt_material)), dtype=MATERIAL_DTYPE).copy()
        texs = []
        for i in range(d.n_texs):
            t = d.texs[i]
            texs.append(np.ctypeslib.as_array(t.rgb8, shape=(t.height, t.width, 3)).copy())
        n_models = L.lib.pt_model_count(h)
        names = [L.lib.pt_model_name(h, i).decode() for i in range(n_models)]
        first = [L.lib.pt_model_first_tri(h, i) for i in range(n_models)]
    finally:
        L.lib.pt_model_destroy(h)
    return SceneData(verts, tris, tmat, mats, uvs, texs, model_names=names, model_first_tri=first)


def bvh_build(verts: np.ndarray, tris: np.ndarray, max_leaf: int = 4, threads: int = 0):
    """bvh.rs:17-21 stand-in: host-side binned-SAH build. Returns (nodes, tri_order, depth)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, dtype=np.uint32).reshape(-1, 3)
    nt = tris.shape[0]
    nodes = np.zeros(max(1, nt), dtype=NODE_DTYPE)
    order = np.zeros(max(1, nt), dtype=np.uint32)
    n_nodes, depth = C.c_uint32(), C.c_uint32()
    L.check(L.lib.pt_bvh_build(_ptr(verts, C.c_float), verts.shape[0], _ptr(tris, C.c_uint32), nt, max_leaf,
                               C.cast(nodes.ctypes.data, C.POINTER(L.pt_bvh_node)), nodes.shape[0],
                               C.byref(n_nodes), _ptr(order, C.c_uint32), C.byref(depth), threads))
    return nodes[: n_nodes.value].copy(), order[:nt].copy(), depth.value


def make_rays(origins, directions, t_min: float = 1e-5, t_max: float = np.finfo(np.float64).max) -> np.ndarray:
    """geometry/ray.rs:16-28 Ray::new for a batch."""
    o = np.asarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float64).reshape(-1, 3)
    r = np.zeros(o.shape[0], dtype=RAY_DTYPE)
    r["o"], r["d"], r["t_min"], r["t_max"] = o, d, t_min, t_max
    return r


class Scene:
    """scene.rs:13-18: the device-resident scene (BVH + triangles + materials + textures + camera)."""

    def __init__(self, data: SceneData, device: int = 0, options: Optional[dict] = None):
        """``options``: pt_scene_options fields by name (the rest stay at their defaults; the environment is not
        read).  None = pt_scene_create, whose options come from the PT_* environment of tuning sweeps."""
        self.data = data
        self.device = device
        self._h = C.c_void_p()
        d = data.desc()
        if options is None:
            L.check(L.lib.pt_scene_create(C.byref(d), device, C.byref(self._h)))
        else:
            o = L.pt_scene_options()
            L.lib.pt_scene_options_init(C.byref(o))
            for k, v in options.items():
                if k not in dict(o._fields_) or k == "struct_size":
                    raise ValueError(f"unknown scene option {k!r}")
                setattr(o, k, int(v))
            L.check(L.lib.pt_scene_create_ex(C.byref(d), device, C.byref(o), C.byref(self._h)))

    # -- reference constructors ----------------------------------------------------------------
    @classmethod
    def cornell_box(cls, device: int = 0, textures: bool = False) -> "Scene":
        """scene.rs:140-193: ./models/Test.obj, eye (275,275,-600) looking down +z, camera_const 2."""
        return cls(scene_data("Test", textures=textures), device)

    @classmethod
    def glossy_planes(cls, device: int = 0, textures: bool = False) -> "Scene":
        """scene.rs:70-104 (commented constructor)."""
        return cls(scene_data("glossy_planes", textures=textures), device)

    @classmethod
    def from_obj(cls, path: str, camera: Camera, background=(0, 0, 0), glossy=False, textures=False,
                 device: int = 0) -> "Scene":
        data = load_obj(path, glossy=glossy, textures=textures)
        data.camera, data.background = camera, tuple(background)
        return cls(data, device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib.pt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene.rs:61 ---------------------------------------------------------------------------
    def trace_closest(self, rays: np.ndarray):
        """Returns (tri_id uint32 [PT_MISS = none], t float64)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        n = rays.shape[0]
        tri = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float64)
        L.check(L.lib.pt_trace_closest(self._h, C.cast(rays.ctypes.data, C.POINTER(L.pt_ray)), n,
                                       _ptr(tri, C.c_uint32), _ptr(t, C.c_double)))
        return tri, t

    def generate_rays(self, cfg: Config, sample: int, tile=None) -> np.ndarray:
        p = cfg.params(sample, 1, tile)
        n = (p.x1 - p.x0) * (p.y1 - p.y0)
        rays = np.zeros(n, dtype=RAY_DTYPE)
        L.check(L.lib.pt_generate_rays(self._h, C.byref(p), sample, C.cast(rays.ctypes.data, C.POINTER(L.pt_ray))))
        return rays

    def sample_brdf(self, rays, tri_id, t, pixel, cfg: Config, sample: int, bounce: int):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        n = rays.shape[0]
        tri_id = np.ascontiguousarray(tri_id, dtype=np.uint32)
        t = np.ascontiguousarray(t, dtype=np.float64)
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        f, wi, em, pos = (np.empty((n, 3)) for _ in range(4))
        pdf = np.empty(n)
        dp = lambda a: _ptr(a, C.c_double)
        L.check(L.lib.pt_sample_brdf(self._h, C.cast(rays.ctypes.data, C.POINTER(L.pt_ray)), _ptr(tri_id, C.c_uint32),
                                     dp(t), _ptr(pixel, C.c_uint32), n, sample, bounce, cfg.seed, cfg.flags(),
                                     cfg.w, cfg.h, dp(f), dp(wi), dp(pdf), dp(em), dp(pos)))
        return f, wi, pdf, em, pos

    def texture_lookup(self, tex: int, uv, duvdx, duvdy):
        uv, duvdx, duvdy = (np.ascontiguousarray(a, dtype=np.float64).reshape(-1, 2) for a in (uv, duvdx, duvdy))
        n = uv.shape[0]
        rgb = np.empty((n, 3))
        dp = lambda a: _ptr(a, C.c_double)
        L.check(L.lib.pt_texture_lookup(self._h, tex, dp(uv), dp(duvdx), dp(duvdy), n, dp(rgb)))
        return rgb

    # -- main.rs:400-436 -------------------------------------------------------------------------
    def render(self, cfg: Config, accum: Optional[np.ndarray] = None, sample_begin: int = 0,
               sample_count: Optional[int] = None, tile=None, rows=None) -> np.ndarray:
        """Adds ``sample_count`` frames to ``accum`` (h, w, 3 float64, ``pixels[i] += next[i]``)."""
        if accum is None:
            accum = np.zeros((cfg.h, cfg.w, 3), dtype=np.float64)
        assert accum.dtype == np.float64 and accum.flags.c_contiguous and accum.size == cfg.w * cfg.h * 3
        n = cfg.samples_pr_pixel if sample_count is None else sample_count
        p = cfg.params(sample_begin, n, tile, rows)
        L.check(L.lib.pt_render(self._h, C.byref(p), _ptr(accum, C.c_double)))
        return accum

    def render_device(self, cfg: Config, accum_ptr: int, sample_begin: int, sample_count: int, tile=None,
                      stream: int = 0, rows=None) -> None:
        """Same, into a device buffer (e.g. ``torch.Tensor.data_ptr()``) on ``stream``."""
        p = cfg.params(sample_begin, sample_count, tile, rows)
        L.check(L.lib.pt_render_device(self._h, C.byref(p), C.c_void_p(accum_ptr), C.c_void_p(stream)))

    def render_device_noalloc(self, cfg: Config, accum_ptr: int, sample_begin: int, sample_count: int, tile=None,
                              stream: int = 0) -> None:
        """``render_device`` over the library's static workspace (pt_render_device_noalloc): no device
        allocation, bounded tile size."""
        p = cfg.params(sample_begin, sample_count, tile)
        L.check(L.lib.pt_render_device_noalloc(self._h, C.byref(p), C.c_void_p(accum_ptr), C.c_void_p(stream)))

    def set_profiling(self, level: int) -> None:
        L.check(L.lib.pt_set_profiling(self._h, level))

    def reset_stats(self) -> None:
        L.check(L.lib.pt_reset_stats(self._h))

    def stats(self) -> dict:
        s = L.pt_stats()
        L.check(L.lib.pt_get_stats(self._h, C.byref(s)))
        out = {k: int(getattr(s, k)) for k in ("samples", "rays", "shadow_rays", "nodes_visited", "tris_tested", "lit_samples",
                                               "scene_bytes", "tree_width", "stack_need", "two_bounce", "packet_primary", "unqueued_rays")}
        out["kernel"] = {L.KERNEL_NAMES[i]: {"launches": int(s.kernel[i].launches), "ms": float(s.kernel[i].ms)}
                         for i in range(L.PT_K_COUNT)}
        return out

    def bvh(self):
        nodes_p, order_p = C.POINTER(L.pt_bvh_node)(), C.POINTER(C.c_uint32)()
        n, depth = C.c_uint32(), C.c_uint32()
        L.check(L.lib.pt_scene_get_bvh(self._h, C.byref(nodes_p), C.byref(n), C.byref(order_p), C.byref(depth)))
        nodes = np.frombuffer(C.string_at(nodes_p, n.value * C.sizeof(L.pt_bvh_node)), dtype=NODE_DTYPE).copy()
        order = np.ctypeslib.as_array(order_p, shape=(self.data.n_tris,)).copy() if self.data.n_tris else np.zeros(0, np.uint32)
        return nodes, order, depth.value


class Accum:
    """main.rs:396-400 ``pixels`` + ``iterations``: the running sum of a progressive render, resident on the
    scene's device (pt_accum).  ``render`` adds frames without a host round trip; ``resolve_rgb8`` /
    ``resolve_u32`` bring back only the 8-bit image (main.rs:456-467), ``download`` the f64 sums."""

    def __init__(self, scene: "Scene", cfg: Config):
        self.scene, self.cfg = scene, cfg
        self.iterations = 0
        self._h = C.c_void_p()
        L.check(L.lib.pt_accum_create(scene._h, cfg.w, cfg.h, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            L.lib.pt_accum_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, sample_count: int = 1, tile=None) -> None:
        """The next ``sample_count`` iterations of the frame loop (main.rs:400-436)."""
        p = self.cfg.params(self.iterations, sample_count, tile)
        L.check(L.lib.pt_accum_render(self._h, C.byref(p)))
        self.iterations += sample_count

    def resolve_rgb8(self) -> np.ndarray:
        out = np.empty((self.cfg.h, self.cfg.w, 3), np.uint8)
        L.check(L.lib.pt_accum_resolve_rgb8(self._h, max(1, self.iterations), _ptr(out, C.c_uint8)))
        return out

    def resolve_u32(self) -> np.ndarray:
        out = np.empty((self.cfg.h, self.cfg.w), np.uint32)
        L.check(L.lib.pt_accum_resolve_u32(self._h, max(1, self.iterations), _ptr(out, C.c_uint32)))
        return out

    def download(self) -> np.ndarray:
        out = np.empty((self.cfg.h, self.cfg.w, 3), np.float64)
        L.check(L.lib.pt_accum_download(self._h, _ptr(out, C.c_double)))
        return out

    def upload(self, accum: Optional[np.ndarray], iterations: int = 0) -> None:
        if accum is None:
            L.check(L.lib.pt_accum_upload(self._h, None))
        else:
            accum = np.ascontiguousarray(accum, dtype=np.float64)
            assert accum.size == self.cfg.w * self.cfg.h * 3
            L.check(L.lib.pt_accum_upload(self._h, _ptr(accum, C.c_double)))
        self.iterations = iterations

    def device(self):
        """(device pointer of the sums, hipStream_t of the handle) as integers."""
        p, st = C.c_void_p(), C.c_void_p()
        L.check(L.lib.pt_accum_device(self._h, C.byref(p), C.byref(st)))
        return p.value, st.value or 0


# ---- scene presets (cameras: SURVEY.md F6) ------------------------------------------------------
CAMERAS = {
    # scene.rs:184-186
    "Test": Camera((275.0, 275.0, -600.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 2.0),
    # the Test camera scaled by 1/556 for the unit-cube CornellBox.obj (no camera exists in the reference)
    "CornellBox": Camera((0.5, 0.5, -1.08), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 2.0),
    # scene.rs:71-73 (commented): eye (0,3,2), direction normalize(0,-1,1), camera_const 1
    "glossy_planes": Camera((0.0, 3.0, 2.0), (0.0 / math.sqrt(2.0), -1.0 / math.sqrt(2.0), 1.0 / math.sqrt(2.0)),
                            (0.0, 1.0, 0.0), 1.0),
    # cornell_box.obj is one black quad at z = 1 (F5): any camera that sees it
    "cornell_box": Camera((0.5, 0.5, -1.08), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 2.0),
}


def scene_data(name: str, glossy: Optional[bool] = None, textures: bool = False, models_dir: str = MODELS_DIR,
               specular: bool = False, ggx: bool = False) -> SceneData:
    """Loads models/<name>.obj with the camera of SURVEY.md F6 and a black background (scene.rs:177)."""
    if glossy is None:
        glossy = name == "glossy_planes"
    data = load_obj(os.path.join(models_dir, name + ".obj"), glossy=glossy, textures=textures, specular=specular, ggx=ggx)
    data.camera = CAMERAS[name]
    data.background = (0.0, 0.0, 0.0)
    return data


def resolve_rgb8(accum: np.ndarray, iterations: int) -> np.ndarray:
    """main.rs:456-467 / mipmap.rs:21-23: clamp(sum / iterations) * 255 + 0.5, truncated; no gamma."""
    accum = np.ascontiguousarray(accum, dtype=np.float64)
    out = np.empty(accum.shape, dtype=np.uint8)
    L.check(L.lib.pt_resolve_rgb8(_ptr(accum, C.c_double), accum.size // 3, iterations, _ptr(out, C.c_uint8)))
    return out


def save_ppm(path: str, rgb8: np.ndarray) -> None:
    """mipmap.rs:25-42 Image::save_ppm (ASCII P3, one pixel per line)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = rgb8.shape[0], rgb8.shape[1]
    L.check(L.lib.pt_write_ppm(os.fsencode(path), w, h, _ptr(rgb8, C.c_uint8)))


def save_checkpoint(path: str, accum: np.ndarray, frames: int, seed: int) -> None:
    """the running f64 sums plus how many frames they hold (pt_checkpoint_save)."""
    accum = np.ascontiguousarray(accum, dtype=np.float64)
    info = L.pt_checkpoint_info(accum.shape[1], accum.shape[0], frames, seed)
    L.check(L.lib.pt_checkpoint_save(os.fsencode(path), C.byref(info), _ptr(accum, C.c_double)))


def load_checkpoint(path: str):
    """-> (accum[h, w, 3] f64, frames, seed); raises PtError on a damaged file."""
    info = L.pt_checkpoint_info()
    L.check(L.lib.pt_checkpoint_load(os.fsencode(path), C.byref(info), None, 0))
    accum = np.empty((info.height, info.width, 3), np.float64)
    L.check(L.lib.pt_checkpoint_load(os.fsencode(path), C.byref(info), _ptr(accum, C.c_double),
                                     info.width * info.height))
    return accum, int(info.frames), int(info.seed)


def render_progressive(scene: "Scene", cfg: "Config", path: str, every: int = 64) -> np.ndarray:
    """Renders cfg.samples_pr_pixel frames, checkpointing to `path` every `every` frames and
    resuming from `path` if it exists (same size and seed required).  The result is bit-identical
    to an uninterrupted Scene.render(cfg): frames are keyed by (seed, frame index) and added in
    index order."""
    done, accum = 0, np.zeros((cfg.h, cfg.w, 3), np.float64)
    if os.path.exists(path):
        accum, done, seed = load_checkpoint(path)
        if accum.shape != (cfg.h, cfg.w, 3) or seed != cfg.seed:
            raise ValueError(f"{path}: checkpoint of a different render ({accum.shape[1]}x{accum.shape[0]}, seed {seed})")
    total = cfg.samples_pr_pixel
    while done < total:
        n = min(every, total - done)
        scene.render(cfg, accum, sample_begin=done, sample_count=n)
        done += n
        save_checkpoint(path, accum, done, cfg.seed)
    return accum


def decode_png(path: str) -> np.ndarray:
    w, h = C.c_uint32(), C.c_uint32()
    p = C.POINTER(C.c_uint8)()
    L.check(L.lib.pt_decode_png(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        L.lib.pt_free(p)


def rng_uniform2(seed: int, pixel: int, sample: int, bounce: int, stream: int):
    out = (C.c_double * 2)()
    L.lib.pt_rng_uniform2(seed, pixel, sample, bounce, stream, out)
    return out[0], out[1]
