"""ctypes binding of include/pt.h (libpt_hip.so).

The shared library is the product: there is no Python or CPU implementation behind these calls.
If the in-tree library is missing the import fails loudly (build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C pathtracer_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_LIB_PATH") or os.path.join(_HERE, "lib", "libpt_hip.so")  # PT_LIB_PATH: tuning builds

PT_MISS = 0xFFFFFFFF
PT_PRIM_BASE = 0xF0000000
PT_PRIM_SPHERE, PT_PRIM_PLANE = 0, 1
PT_OK, PT_ERR_INVALID, PT_ERR_NOMEM, PT_ERR_DEVICE, PT_ERR_IO, PT_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
PT_MAT_LAMBERT, PT_MAT_GLOSSY, PT_MAT_MIRROR, PT_MAT_TRANSPARENT, PT_MAT_GLOSSY_GGX = 0, 1, 2, 3, 4
PT_BVH_EMPTY = 0x7FFFFFFF
PT_FLAG_USE_DIFFS, PT_FLAG_NEE, PT_FLAG_DROP_NONFINITE, PT_FLAG_RUSSIAN_ROULETTE = 1, 2, 4, 8
PT_FLAG_PROPAGATE_DIFFS, PT_FLAG_ROUGHDIFF, PT_FLAG_REFERENCE_LIGHTS = 16, 32, 64
PT_LOAD_GLOSSY, PT_LOAD_TEXTURES, PT_LOAD_SPECULAR, PT_LOAD_GGX = 1, 2, 4, 8
PT_K_GENERATE, PT_K_EXTEND, PT_K_SHADE, PT_K_SHADOW, PT_K_ACCUMULATE, PT_K_PATH, PT_K_COUNT = 0, 1, 2, 3, 4, 5, 6
KERNEL_NAMES = ("generate", "extend", "shade", "shadow", "accumulate", "path")


class pt_material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("emission", C.c_float * 3), ("roughness", C.c_float),
                ("kind", C.c_int32), ("tex", C.c_int32), ("ior", C.c_float)]


class pt_texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb8", C.POINTER(C.c_uint8))]


class pt_camera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("view", C.c_double * 3), ("up", C.c_double * 3),
                ("camera_const", C.c_double)]


class pt_bvh_node(C.Structure):
    _fields_ = [("bounds", C.c_float * 12), ("child", C.c_int32 * 2), ("pad", C.c_uint32 * 2)]


class pt_primitive(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_uint32), ("p", C.c_double * 3), ("n", C.c_double * 3),
                ("radius", C.c_double)]


class pt_area_light(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("intensity", C.c_double * 3)]


class pt_scene_desc(C.Structure):
    _fields_ = [("verts", C.POINTER(C.c_float)), ("n_verts", C.c_uint32),
                ("tris", C.POINTER(C.c_uint32)), ("n_tris", C.c_uint32),
                ("tri_uvs", C.POINTER(C.c_float)),
                ("tri_material", C.POINTER(C.c_uint32)),
                ("mats", C.POINTER(pt_material)), ("n_mats", C.c_uint32),
                ("texs", C.POINTER(pt_texture)), ("n_texs", C.c_uint32),
                ("camera", pt_camera), ("background", C.c_double * 3),
                ("bvh_nodes", C.POINTER(pt_bvh_node)), ("n_bvh_nodes", C.c_uint32),
                ("bvh_tri_order", C.POINTER(C.c_uint32)),
                ("prims", C.POINTER(pt_primitive)), ("n_prims", C.c_uint32),
                ("area_lights", C.POINTER(pt_area_light)), ("n_area_lights", C.c_uint32)]


class pt_scene_options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int8) for n in (
        "vote_primary", "vote_secondary", "vote_shadow", "fuse", "fuse_shadow", "light_cull", "screen_cull", "pipeline",
        "persistent", "eye_tables", "lds_scene", "group_frames")] + [
        ("wave_rays", C.c_uint32), ("bvh_max_leaf", C.c_uint32), ("bvh_threads", C.c_uint32),
        ("lds_top", C.c_int8), ("fuse01", C.c_int8), ("packet_primary", C.c_int8), ("reserved8", C.c_int8 * 1)]


class pt_ray(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3), ("t_min", C.c_double), ("t_max", C.c_double)]


class pt_render_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
                ("sample_begin", C.c_uint64), ("sample_count", C.c_uint32),
                ("max_bounces", C.c_uint32), ("flags", C.c_uint32), ("spp_per_batch", C.c_uint32),
                ("seed", C.c_uint64), ("diff_size", C.c_double),
                ("row_group", C.c_uint32), ("row_parts", C.c_uint32), ("row_index", C.c_uint32), ("reserved", C.c_uint32)]


class pt_checkpoint_info(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("frames", C.c_uint64), ("seed", C.c_uint64)]


class pt_kernel_stat(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("ms", C.c_double)]


class pt_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("kernel", pt_kernel_stat * PT_K_COUNT), ("lit_samples", C.c_uint64),
                ("scene_bytes", C.c_uint64), ("tree_width", C.c_uint32), ("stack_need", C.c_uint32),
                ("two_bounce", C.c_uint32), ("packet_primary", C.c_uint32), ("unqueued_rays", C.c_uint64)]


# every symbol include/pt.h declares: name -> (restype, argtypes)
_P = C.POINTER
_vp = C.c_void_p
PROTOTYPES = {
    "pt_scene_create": (C.c_int, [_P(pt_scene_desc), C.c_int, _P(_vp)]),
    "pt_scene_create_ex": (C.c_int, [_P(pt_scene_desc), C.c_int, _P(pt_scene_options), _P(_vp)]),
    "pt_scene_options_init": (None, [_P(pt_scene_options)]),
    "pt_scene_options_from_env": (None, [_P(pt_scene_options)]),
    "pt_scene_destroy": (None, [_vp]),
    "pt_scene_get_bvh": (C.c_int, [_vp, _P(_P(pt_bvh_node)), _P(C.c_uint32), _P(_P(C.c_uint32)), _P(C.c_uint32)]),
    "pt_bvh_build": (C.c_int, [_P(C.c_float), C.c_uint32, _P(C.c_uint32), C.c_uint32, C.c_uint32,
                               _P(pt_bvh_node), C.c_uint32, _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32), C.c_uint32]),
    "pt_bvh_validate": (C.c_int, [_P(pt_bvh_node), C.c_uint32, _P(C.c_uint32), C.c_uint32, _P(C.c_uint32)]),
    "pt_trace_closest": (C.c_int, [_vp, _P(pt_ray), C.c_size_t, _P(C.c_uint32), _P(C.c_double)]),
    "pt_generate_rays": (C.c_int, [_vp, _P(pt_render_params), C.c_uint64, _P(pt_ray)]),
    "pt_sample_brdf": (C.c_int, [_vp, _P(pt_ray), _P(C.c_uint32), _P(C.c_double), _P(C.c_uint32), C.c_size_t,
                                 C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                 _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "pt_texture_lookup": (C.c_int, [_vp, C.c_uint32, _P(C.c_double), _P(C.c_double), _P(C.c_double), C.c_size_t,
                                    _P(C.c_double)]),
    "pt_render": (C.c_int, [_vp, _P(pt_render_params), _P(C.c_double)]),
    "pt_render_device": (C.c_int, [_vp, _P(pt_render_params), _vp, _vp]),
    "pt_accum_create": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _P(_vp)]),
    "pt_accum_destroy": (None, [_vp]),
    "pt_accum_render": (C.c_int, [_vp, _P(pt_render_params)]),
    "pt_accum_resolve_rgb8": (C.c_int, [_vp, C.c_uint64, _P(C.c_uint8)]),
    "pt_accum_resolve_u32": (C.c_int, [_vp, C.c_uint64, _P(C.c_uint32)]),
    "pt_accum_download": (C.c_int, [_vp, _P(C.c_double)]),
    "pt_accum_upload": (C.c_int, [_vp, _P(C.c_double)]),
    "pt_accum_device": (C.c_int, [_vp, _P(_vp), _P(_vp)]),
    "pt_render_device_noalloc": (C.c_int, [_vp, _P(pt_render_params), _vp, _vp]),
    "pt_resolve_rgb8": (C.c_int, [_P(C.c_double), C.c_size_t, C.c_uint64, _P(C.c_uint8)]),
    "pt_write_ppm": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _P(C.c_uint8)]),
    "pt_checkpoint_save": (C.c_int, [C.c_char_p, _P(pt_checkpoint_info), _P(C.c_double)]),
    "pt_checkpoint_load": (C.c_int, [C.c_char_p, _P(pt_checkpoint_info), _P(C.c_double), C.c_size_t]),
    "pt_set_profiling": (C.c_int, [_vp, C.c_int]),
    "pt_get_stats": (C.c_int, [_vp, _P(pt_stats)]),
    "pt_reset_stats": (C.c_int, [_vp]),
    "pt_load_obj": (C.c_int, [C.c_char_p, C.c_uint32, _P(_vp)]),
    "pt_model_desc": (_P(pt_scene_desc), [_vp]),
    "pt_model_count": (C.c_uint32, [_vp]),
    "pt_model_name": (C.c_char_p, [_vp, C.c_uint32]),
    "pt_model_first_tri": (C.c_uint32, [_vp, C.c_uint32]),
    "pt_model_destroy": (None, [_vp]),
    "pt_decode_png": (C.c_int, [C.c_char_p, _P(C.c_uint32), _P(C.c_uint32), _P(_P(C.c_uint8))]),
    "pt_free": (None, [_vp]),
    "pt_rng_uniform2": (None, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, _P(C.c_double)]),
    "pt_last_error": (C.c_char_p, []),
    "pt_version": (C.c_char_p, []),
}


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pt error {code}: {message}")
        self.code = code


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's); if this library pulled in /opt/rocm's copy and torch later loaded its
    own by path, two runtimes would fight over the GPU and the second to initialise sees no
    device.  Loading torch's copy first (without importing torch) makes the dynamic loader reuse
    it for libpt_hip.so and for a later ``import torch`` alike, so device pointers and streams can
    be shared with torch tensors / torch.distributed."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _load() -> C.CDLL:
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is the only implementation of pathtracer_amd "
            "(no CPU fallback). Build it with `make -C pathtracer_amd/csrc` or __graft_entry__.build().")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != PT_OK:
        raise PtError(rc, (lib.pt_last_error() or b"").decode("utf-8", "replace"))
