"""pathtracer_amd — MI355X (gfx950) implementation of emimvi/pathtracer's radiance inner loop.

The package is a thin host-side mirror of the reference's scene/driver interface; the work is done
by hand-written HIP kernels behind the C ABI of include/pt.h (pathtracer_amd/lib/libpt_hip.so).
Importing fails if that library has not been built: there is no CPU fallback.
"""
from ._lib import (PT_MISS, PT_PRIM_BASE, PT_PRIM_SPHERE, PT_PRIM_PLANE, PT_MAT_GLOSSY, PT_MAT_LAMBERT, PT_MAT_MIRROR, PT_MAT_TRANSPARENT, PT_MAT_GLOSSY_GGX, PT_FLAG_PROPAGATE_DIFFS, PT_FLAG_ROUGHDIFF, PT_FLAG_REFERENCE_LIGHTS, PtError, LIB_PATH)  # noqa: F401
from .scene import (Accum, Camera, Config, Scene, SceneData, load_obj, bvh_build, make_rays, scene_data,  # noqa: F401
                    resolve_rgb8, save_ppm, decode_png, save_checkpoint, load_checkpoint, render_progressive, rng_uniform2, RAY_DTYPE, NODE_DTYPE, MATERIAL_DTYPE, PRIM_DTYPE,
                    CAMERAS, MODELS_DIR)
