#!/usr/bin/env python3
"""Run with a library built with -DPT_DEBUG_CHECKS (scripts/build_variant.sh dbg -DPT_DEBUG_CHECKS;
PT_LIB_PATH=pathtracer_amd/lib/alt_dbg.so): renders scenes whose trees stress the traversal stacks - the reference's
scenes, heightfields of several depths, a caller-supplied degenerate tree of the greatest depth pt_bvh_validate lets
through - closest-hit, any-hit (NEE) and vote-scheduled, and reads the count of pushes that found a stack full.
Prints one line per case; exit status 1 if any count is not zero."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pathtracer_amd as pt
from pathtracer_amd import _lib as L
from pathtracer_amd.synthetic import heightfield_box

fn = getattr(L.lib, "pt_debug_stack_overflows", None)
if fn is None:
    sys.exit("this library was not built with -DPT_DEBUG_CHECKS")
fn.argtypes = [C.POINTER(C.c_uint)]


def overflows():
    n = C.c_uint()
    assert fn(C.byref(n)) == 0
    return n.value


def chain_scene(n_tris):
    """n_tris random triangles under a left-deep (degenerate) caller-supplied tree: depth n_tris - 1."""
    rng = np.random.default_rng(1)
    v = rng.uniform(-1, 1, (n_tris * 3, 3)).astype(np.float32)
    t = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    mats = np.zeros(1, pt.MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"] = 0.7
    mats["emission"] = 1.0
    data = pt.SceneData(v, t, np.zeros(n_tris, np.uint32), mats, camera=pt.Camera((0, 0, -4), (0, 0, 1), (0, 1, 0), 2.0))
    nodes = np.zeros(n_tris - 1, pt.NODE_DTYPE)
    for i in range(n_tris - 1):
        rest = v[3 * (i + 1):].reshape(-1, 3)
        own = v[3 * i:3 * i + 3]
        nodes["bounds"][i, 0:3], nodes["bounds"][i, 3:6] = own.min(0), own.max(0)
        nodes["bounds"][i, 6:9], nodes["bounds"][i, 9:12] = rest.min(0), rest.max(0)
        nodes["child"][i, 0] = ~(i << 3)
        nodes["child"][i, 1] = i + 1 if i + 2 < n_tris else ~((i + 1) << 3)
    data.bvh_nodes, data.bvh_tri_order = nodes, np.arange(n_tris, dtype=np.uint32)
    return data


cases = [(name, pt.scene_data(name)) for name in ("CornellBox", "Test", "glossy_planes")]
cases += [(f"heightfield cells={c}", heightfield_box(cells=c)) for c in (8, 40, 300, 1200)]
cases += [("chain tree, 33 triangles (depth 32)", chain_scene(33))]
bad = 0
for name, data in cases:
    for options in ({}, dict(vote_primary=1, vote_secondary=1, vote_shadow=1, fuse=0), dict(vote_secondary=0, vote_shadow=0)):
        scene = pt.Scene(data, 0, options=options)
        before = overflows()
        for nee in (False, True):
            scene.render(pt.Config(width=320, height=200, samples_pr_pixel=4, bounces=3, seed=3, nee=nee, drop_nonfinite=True))
        st = scene.stats()
        n = overflows() - before
        bad += n
        print(f"{name:40s} {str(options):70s} depth {scene.bvh()[2]:2d} tree width {st['tree_width']} stack need {st['stack_need']:2d} overflows {n}")
sys.exit(1 if bad else 0)
