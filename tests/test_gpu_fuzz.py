"""Random triangle soups through every variant of the render: the exact culls, the scene-in-LDS kernels, the
two-bounce kernel, the vote-scheduled and packet traversals must give the bytes of the plainest pipeline, and the
plainest pipeline the oracle's image.  (Tilted, overlapping, partly degenerate triangles and emitters in odd places:
what the hand-made scenes of the other tests do not have.)"""
import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import _lib as L
from pathtracer_amd.scene import Camera, MATERIAL_DTYPE, SceneData
import oracle_lib as O

pytestmark = pytest.mark.gpu


def soup(n_tris, seed, glossy=False):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1.0, 1.0, (n_tris, 1, 3))
    size = rng.uniform(0.05, 0.9, (n_tris, 1, 1)) * min(1.0, (60.0 / n_tris) ** (1.0 / 3.0))   # about the same cover at any count
    verts = (centres + rng.normal(size=(n_tris, 3, 3)) * size).astype(np.float32)
    if n_tris > 4:
        verts[3, 2] = verts[3, 1]                      # a degenerate triangle
        verts[4] = verts[2]                            # a duplicate (exact ties: the lowest id wins)
    tris = np.arange(n_tris * 3, dtype=np.uint32).reshape(n_tris, 3)
    mats = np.zeros(4, dtype=MATERIAL_DTYPE)
    mats["tex"] = -1
    mats["color"][0] = (0.7, 0.6, 0.5)
    mats["color"][1] = (0.2, 0.8, 0.3)
    mats["color"][2] = 1.0
    mats["emission"][2] = (6.0, 5.0, 4.0)
    mats["color"][3] = 0.9
    if glossy:
        mats["kind"][3] = L.PT_MAT_GLOSSY
        mats["roughness"][3] = 0.2
    tri_mat = rng.integers(0, 2, n_tris).astype(np.uint32)
    tri_mat[rng.choice(n_tris, size=max(1, min(8, n_tris // 8)), replace=False)] = 2   # a few emitters, anywhere
    if glossy:
        tri_mat[rng.choice(n_tris, size=max(1, n_tris // 4), replace=False)] = 3
    cam = Camera((0.2, 0.1, 3.5), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 1.2)
    return SceneData(verts.reshape(-1, 3), tris, tri_mat, mats, camera=cam)


VARIANTS = [
    dict(light_cull=0, screen_cull=0, lds_scene=0, fuse01=0, eye_tables=0),   # the plainest pipeline
    dict(),                                                                    # all defaults
    dict(lds_scene=0),
    dict(fuse01=0),
    dict(light_cull=0),
    dict(fuse=0),
    dict(vote_secondary=1, vote_primary=1, vote_shadow=1),
    dict(packet_primary=0, lds_top=0, group_frames=0),
]


@pytest.mark.parametrize("n_tris,seed", [(6, 1), (20, 2), (31, 3), (32, 4), (33, 5), (90, 6), (400, 7), (5000, 8)])
def test_random_soup_every_variant_gives_the_same_bytes(gpu, n_tris, seed):
    data = soup(n_tris, seed, glossy=seed % 2 == 0)
    cfgs = [pt.Config(width=96, height=54, samples_pr_pixel=6, bounces=2, seed=seed, drop_nonfinite=True),
            pt.Config(dim=48, samples_pr_pixel=4, bounces=3, seed=seed + 10, nee=True, drop_nonfinite=True),
            pt.Config(dim=40, samples_pr_pixel=3, bounces=1, seed=seed + 20, drop_nonfinite=True)]
    if n_tris >= 5000:  # deep tree: 64 frames in flight, so that waves hold one pixel in 64 frames and walk the tree as packets
        cfgs.append(pt.Config(dim=32, samples_pr_pixel=64, bounces=2, seed=seed + 30, spp_per_batch=64, drop_nonfinite=True))
    ref = None
    for opts in VARIANTS:
        scene = pt.Scene(data, gpu, options=opts)
        imgs = [scene.render(c) for c in cfgs]
        if ref is None:
            ref = imgs
            assert any(i.any() for i in imgs)
        else:
            for a, b, c in zip(ref, imgs, cfgs):
                assert a.tobytes() == b.tobytes(), (n_tris, seed, opts, c.bounces, c.nee)
    # ... and the oracle's image (sin / cos come from two libraries: 1e-9 relative on all but a sliver of the values)
    if n_tris <= 400:
        oracle = O.OracleScene(data)
        for img, c in zip(ref, cfgs):
            want = oracle.render(c)
            bad = np.abs(img - want) > 1e-9 * np.maximum(1.0, np.abs(want))
            assert bad.mean() < 2e-3, (n_tris, seed, float(bad.mean()))
