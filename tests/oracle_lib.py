"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so) — test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import _lib as L
from pathtracer_amd.scene import RAY_DTYPE, Config, SceneData

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libpt_oracle.so")


def _load():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return C.CDLL(ORACLE_SO)


lib = _load()
_P = C.POINTER
_dp = _P(C.c_double)
_vp = C.c_void_p
lib.orc_scene_create.argtypes = [_P(L.pt_scene_desc), _P(_vp)]
lib.orc_scene_destroy.argtypes = [_vp]
lib.orc_scene_destroy.restype = None
lib.orc_trace_closest.argtypes = [_vp, _P(L.pt_ray), C.c_size_t, _P(C.c_uint32), _dp, C.c_int]
lib.orc_generate_rays.argtypes = [_vp, _P(L.pt_render_params), C.c_uint64, _P(L.pt_ray)]
lib.orc_render.argtypes = [_vp, _P(L.pt_render_params), _dp, C.c_int, C.c_int]
lib.orc_sample_brdf.argtypes = [_vp, _P(L.pt_ray), _P(C.c_uint32), _dp, _P(C.c_uint32), C.c_size_t, C.c_uint64,
                                C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp, _dp, _dp, _dp, _dp]
lib.orc_texture_lookup.argtypes = [_vp, C.c_uint32, _dp, _dp, _dp, C.c_size_t, _dp]
lib.orc_texture_levels.argtypes = [_vp, C.c_uint32]
lib.orc_texture_texel.argtypes = [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _dp]
lib.orc_camera_ray.argtypes = [_P(L.pt_camera), C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _dp]
lib.orc_camera_ray.restype = None
lib.orc_philox4x32_10.argtypes = [_P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint32)]
lib.orc_philox4x32_10.restype = None
lib.orc_rng_uniform2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, _dp]
lib.orc_rng_uniform2.restype = None
lib.orc_lambertian_sample.argtypes = [_dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
lib.orc_lambertian_sample.restype = None
lib.orc_glossy_sample.argtypes = [_dp, C.c_double, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
lib.orc_glossy_sample.restype = None
lib.orc_beckmann_D.argtypes = [C.c_double, _dp, _dp]
lib.orc_beckmann_D.restype = C.c_double
lib.orc_beckmann_G1.argtypes = [C.c_double, _dp, _dp, _dp]
lib.orc_beckmann_G1.restype = C.c_double
lib.orc_ggx_D.argtypes = [C.c_double, _dp, _dp]
lib.orc_ggx_D.restype = C.c_double
lib.orc_ggx_G1.argtypes = [C.c_double, _dp, _dp, _dp]
lib.orc_ggx_G1.restype = C.c_double
lib.orc_ggx_sample_normal.argtypes = [C.c_double, _dp, C.c_double, C.c_double, _dp]
lib.orc_ggx_sample_normal.restype = None
lib.orc_conductor.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, _dp]
lib.orc_conductor.restype = None
lib.orc_fresnel_dielectric.argtypes = [C.c_double, C.c_double]
lib.orc_fresnel_dielectric.restype = C.c_double
lib.orc_spawn_diffs.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint32, _dp]
lib.orc_spawn_diffs.restype = None
lib.orc_sphere_surface.argtypes = [_dp, C.c_double, _dp, C.c_double, _dp]
lib.orc_sphere_surface.restype = None
lib.orc_refract.argtypes = [_dp, _dp, C.c_double, _dp]
lib.orc_refract.restype = None
lib.orc_normalize.argtypes = [_dp, _dp]
lib.orc_normalize.restype = None
lib.orc_rotate_to.argtypes = [_dp, _dp, _dp]
lib.orc_rotate_to.restype = None
lib.orc_sample_sphere.argtypes = [C.c_double, C.c_double, _dp]
lib.orc_sample_sphere.restype = None
lib.orc_f64_to_8bit_color.argtypes = [C.c_double]
lib.orc_f64_to_8bit_color.restype = C.c_uint8
lib.orc_resolve_rgb8.argtypes = [_dp, C.c_size_t, C.c_uint64, _P(C.c_uint8)]
lib.orc_write_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _P(C.c_uint8)]
lib.orc_hardware_threads.restype = C.c_int


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class OracleScene:
    def __init__(self, data: SceneData):
        self.data = data
        self._h = _vp()
        d = data.desc()
        rc = lib.orc_scene_create(C.byref(d), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"orc_scene_create failed: {rc}")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib.orc_scene_destroy(self._h)
            self._h = _vp()

    def trace_closest(self, rays, use_bvh=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        n = rays.shape[0]
        tri = np.empty(n, np.uint32)
        t = np.empty(n, np.float64)
        lib.orc_trace_closest(self._h, C.cast(rays.ctypes.data, _P(L.pt_ray)), n, _ptr(tri, C.c_uint32), _ptr(t, C.c_double), int(use_bvh))
        return tri, t

    def generate_rays(self, cfg: Config, sample: int, tile=None):
        p = cfg.params(sample, 1, tile)
        n = (p.x1 - p.x0) * (p.y1 - p.y0)
        rays = np.zeros(n, dtype=RAY_DTYPE)
        lib.orc_generate_rays(self._h, C.byref(p), sample, C.cast(rays.ctypes.data, _P(L.pt_ray)))
        return rays

    def render(self, cfg: Config, accum=None, sample_begin=0, sample_count=None, tile=None, threads=8, use_bvh=False,
               rows=None):
        if accum is None:
            accum = np.zeros((cfg.h, cfg.w, 3), np.float64)
        n = cfg.samples_pr_pixel if sample_count is None else sample_count
        p = cfg.params(sample_begin, n, tile, rows)
        lib.orc_render(self._h, C.byref(p), _ptr(accum, C.c_double), threads, int(use_bvh))
        return accum

    def sample_brdf(self, rays, tri_id, t, pixel, cfg: Config, sample: int, bounce: int):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        n = rays.shape[0]
        tri_id = np.ascontiguousarray(tri_id, np.uint32)
        t = np.ascontiguousarray(t, np.float64)
        pixel = np.ascontiguousarray(pixel, np.uint32)
        f, wi, em, pos = (np.empty((n, 3)) for _ in range(4))
        pdf = np.empty(n)
        dp = lambda a: _ptr(a, C.c_double)
        rc = lib.orc_sample_brdf(self._h, C.cast(rays.ctypes.data, _P(L.pt_ray)), _ptr(tri_id, C.c_uint32), dp(t),
                                 _ptr(pixel, C.c_uint32), n, sample, bounce, cfg.seed, cfg.flags(), cfg.w, cfg.h,
                                 dp(f), dp(wi), dp(pdf), dp(em), dp(pos))
        assert rc == 0
        return f, wi, pdf, em, pos

    def texture_lookup(self, tex, uv, duvdx, duvdy):
        uv, duvdx, duvdy = (np.ascontiguousarray(a, np.float64).reshape(-1, 2) for a in (uv, duvdx, duvdy))
        n = uv.shape[0]
        rgb = np.empty((n, 3))
        dp = lambda a: _ptr(a, C.c_double)
        rc = lib.orc_texture_lookup(self._h, tex, dp(uv), dp(duvdx), dp(duvdy), n, dp(rgb))
        assert rc == 0
        return rgb

    def texture_levels(self, tex):
        return lib.orc_texture_levels(self._h, tex)

    def texture_texel(self, tex, level, index):
        out = (C.c_double * 3)()
        assert lib.orc_texture_texel(self._h, tex, level, index, out) == 0
        return np.array(out[:])


def camera_ray(cam, x, y, pdx, pdy, use_diffs=True):
    out = (C.c_double * 18)()
    c = cam.to_c()
    lib.orc_camera_ray(C.byref(c), x, y, pdx, pdy, int(use_diffs), out)
    return np.array(out[:]).reshape(6, 3)


def philox(key, ctr):
    k = (C.c_uint32 * 2)(*key)
    c = (C.c_uint32 * 4)(*ctr)
    o = (C.c_uint32 * 4)()
    lib.orc_philox4x32_10(k, c, o)
    return list(o)


def rng_uniform2(seed, pixel, sample, bounce, stream):
    o = (C.c_double * 2)()
    lib.orc_rng_uniform2(seed, pixel, sample, bounce, stream, o)
    return o[0], o[1]


def lambertian_sample(color, n, xi1, xi2):
    f, wi, pdf = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    lib.orc_lambertian_sample(_d3(color), _d3(n), xi1, xi2, f, wi, C.byref(pdf))
    return np.array(f[:]), np.array(wi[:]), pdf.value


def glossy_sample(color, roughness, direction, n, xi1, xi2):
    f, wi, pdf = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    lib.orc_glossy_sample(_d3(color), roughness, _d3(direction), _d3(n), xi1, xi2, f, wi, C.byref(pdf))
    return np.array(f[:]), np.array(wi[:]), pdf.value


def beckmann_D(roughness, m, n):
    return lib.orc_beckmann_D(roughness, _d3(m), _d3(n))


def beckmann_G1(roughness, w, m, n):
    return lib.orc_beckmann_G1(roughness, _d3(w), _d3(m), _d3(n))


def conductor(cos_theta_i, ior_i=32.0, ior_t=1.0, absorption=0.1):
    o = (C.c_double * 3)()
    lib.orc_conductor(cos_theta_i, ior_i, ior_t, absorption, o)
    return np.array(o[:])


def vec3_fn(fn, *args):
    o = (C.c_double * 3)()
    fn(*args, o)
    return np.array(o[:])


def normalize(v):
    return vec3_fn(lib.orc_normalize, _d3(v))


def refract(v, n, eta):
    return vec3_fn(lib.orc_refract, _d3(v), _d3(n), eta)


def rotate_to(v, n):
    return vec3_fn(lib.orc_rotate_to, _d3(v), _d3(n))


def sample_sphere(r1, r2):
    return vec3_fn(lib.orc_sample_sphere, r1, r2)


def spawn_diffs(position, normal, dpdx, dpdy, inc_dir, inc_rx_dir, inc_ry_dir, wi, kind, lobe, roughness=0.0, ior=1.0,
                diff_size=0.0, flags=0):
    """main.rs:192-281 reflect_diff / refract_diff: (rx_origin, ry_origin, rx_direction, ry_direction)."""
    vin = np.ascontiguousarray(np.concatenate([position, normal, dpdx, dpdy, inc_dir, inc_rx_dir, inc_ry_dir]), np.float64)
    out = (C.c_double * 12)()
    lib.orc_spawn_diffs(vin.ctypes.data_as(_dp), _d3(wi), kind, lobe, roughness, ior, diff_size, flags, out)
    return np.array(out[:]).reshape(4, 3)


def sphere_surface(centre, radius, origin, direction, t, rx_origin=None, ry_origin=None, rx_direction=None,
                   ry_direction=None):
    """geometry/mod.rs:73-146: the Surface of a sphere hit, as a dict of its fields."""
    z = np.zeros(3)
    ray = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64) if v is not None else z for v in
                                               (origin, direction, rx_origin, ry_origin, rx_direction, ry_direction)]))
    out = (C.c_double * 22)()
    lib.orc_sphere_surface(_d3(centre), float(radius), ray.ctypes.data_as(_dp), float(t), out)
    a = np.array(out[:])
    names = ("position", "normal", "dpdu", "dpdv", "dndu", "dndv")
    r = {n: a[3 * i:3 * i + 3] for i, n in enumerate(names)}
    r.update(dudx=a[18], dvdx=a[19], dudy=a[20], dvdy=a[21])
    return r


def ggx_D(roughness, m, n):
    return lib.orc_ggx_D(roughness, _d3(m), _d3(n))


def ggx_G1(roughness, w, m, n):
    return lib.orc_ggx_G1(roughness, _d3(w), _d3(m), _d3(n))


def ggx_sample_normal(roughness, n, xi1, xi2):
    h = (C.c_double * 3)()
    lib.orc_ggx_sample_normal(roughness, _d3(n), xi1, xi2, h)
    return np.array(h[:])
