"""pt_render_device_noalloc (include/pt.h): the render over the library's static workspace, through
the C ABI, against pt_render_device — the entry point it is a variant of — bit for bit.

Runs last (file name): a fault here must not hide the other tests from `pytest -x`.
"""
import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import _lib as L

pytestmark = pytest.mark.gpu


def _both(scene, cfg, spp, torch, tile=None, start=None):
    """(pt_render_device, pt_render_device_noalloc) images of the same call, as numpy arrays."""
    out = []
    for fn in (scene.render_device, scene.render_device_noalloc):
        acc = torch.zeros((cfg.h, cfg.w, 3), dtype=torch.float64, device="cuda") if start is None else start.clone()
        fn(cfg, acc.data_ptr(), 0, spp, tile=tile, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out.append(acc.cpu().numpy())
    return out


def test_noalloc_two_calls_in_a_row_second_smaller(gpu):
    import torch
    torch.cuda.set_device(gpu)
    scene = pt.Scene(pt.scene_data("CornellBox"), device=gpu)
    # a large call (its 6 frames do not fit the static workspace at once: it is cut into batches), then a
    # smaller one on the same GPU, which finds the first call's bytes in the array
    big = pt.Config(width=640, height=360, bounces=2, seed=5)
    ref, got = _both(scene, big, 6, torch)
    assert ref.any() and ref.tobytes() == got.tobytes()
    small = pt.Config(width=96, height=64, bounces=2, seed=9)
    ref, got = _both(scene, small, 3, torch)
    assert ref.any() and ref.tobytes() == got.tobytes()
    # and the large one again, after the small one
    ref2, got2 = _both(scene, big, 6, torch)
    assert got2.tobytes() == ref2.tobytes()


def test_noalloc_allocates_nothing_and_adds_to_the_buffer(gpu):
    import torch
    torch.cuda.set_device(gpu)
    scene = pt.Scene(pt.scene_data("CornellBox"), device=gpu)
    cfg = pt.Config(width=320, height=200, bounces=2, seed=3, nee=True)
    start = torch.full((cfg.h, cfg.w, 3), 0.25, dtype=torch.float64, device="cuda")
    ref, got = _both(scene, cfg, 4, torch, start=start)       # first call on the device: looks the array up
    assert ref.tobytes() == got.tobytes() and (ref >= 0.25).all() and (ref > 0.25).any()   # pixels[i] += next[i]
    acc = start.clone()
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(gpu)[0]
    for i in range(5):
        scene.render_device_noalloc(cfg, acc.data_ptr(), 4 * i, 4, stream=stream)
        assert torch.cuda.mem_get_info(gpu)[0] == free0       # hipMemGetInfo: no device memory taken or returned
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(gpu)[0] == free0
    ref = start.clone()
    scene.render_device(cfg, ref.data_ptr(), 0, 20, stream=stream)
    torch.cuda.synchronize()
    assert acc.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes()


@pytest.mark.parametrize("name", ["Test", "glossy_planes", "heightfield40"])
def test_noalloc_every_kernel_family(gpu, name):
    # textures + NEE (fused path kernel with its shadow rays), Glossy, and a deep tree (separate extend /
    # shade kernels, vote-scheduled persistent waves, the shadow queue): the same bits as pt_render_device
    import torch
    torch.cuda.set_device(gpu)
    if name == "heightfield40":
        from pathtracer_amd.synthetic import heightfield_box
        data = heightfield_box(cells=40)
    else:
        data = pt.scene_data(name, textures=name == "Test")
    scene = pt.Scene(data, device=gpu)
    for nee in (False, True):
        cfg = pt.Config(width=211, height=97, bounces=3, seed=7, nee=nee, drop_nonfinite=True)
        ref, got = _both(scene, cfg, 5, torch)
        assert ref.any() and ref.tobytes() == got.tobytes(), (name, nee)
        ref, got = _both(scene, cfg, 2, torch, tile=(13, 8, 150, 61))
        assert ref.tobytes() == got.tobytes(), (name, nee, "tile")


def test_noalloc_rejects_what_does_not_fit(gpu):
    import torch
    torch.cuda.set_device(gpu)
    scene = pt.Scene(pt.scene_data("CornellBox"), device=gpu)
    cfg = pt.Config(width=1920, height=1080, bounces=2, seed=1)
    acc = torch.zeros((cfg.h, cfg.w, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(pt.PtError) as e:                      # one 1080p frame needs ~400 MB of queues
        scene.render_device_noalloc(cfg, acc.data_ptr(), 0, 1)
    assert e.value.code == L.PT_ERR_UNSUPPORTED and "PT_NOALLOC_SCRATCH_BYTES" in str(e.value)
    torch.cuda.synchronize()
    assert not acc.any().item()                               # nothing was enqueued
    # the same image in tiles that fit: identical to one pt_render_device call
    for y0 in range(0, 1080, 270):
        scene.render_device_noalloc(cfg, acc.data_ptr(), 0, 2, tile=(0, y0, 1920, y0 + 270))
    ref = torch.zeros_like(acc)
    scene.render_device(cfg, ref.data_ptr(), 0, 2)
    torch.cuda.synchronize()
    assert acc.cpu().numpy().tobytes() == ref.cpu().numpy().tobytes()


def test_noalloc_two_scenes_on_two_streams_are_ordered(gpu):
    # every call on a GPU shares that GPU's one static array, whatever scene or stream it comes from: the
    # library orders them (per-device event), so two scenes alternating on two streams with no synchronisation
    # in between give the images of the same calls made one at a time
    import torch
    torch.cuda.set_device(gpu)
    a = pt.Scene(pt.scene_data("CornellBox"), device=gpu)
    b = pt.Scene(pt.scene_data("glossy_planes"), device=gpu)
    cfg_a = pt.Config(width=480, height=270, bounces=2, seed=4)
    cfg_b = pt.Config(width=384, height=216, bounces=3, seed=6, nee=True, drop_nonfinite=True)
    ref_a = torch.zeros((cfg_a.h, cfg_a.w, 3), dtype=torch.float64, device="cuda")
    ref_b = torch.zeros((cfg_b.h, cfg_b.w, 3), dtype=torch.float64, device="cuda")
    n_rounds, spp = 6, 3
    for i in range(n_rounds):
        a.render_device(cfg_a, ref_a.data_ptr(), spp * i, spp)
        b.render_device(cfg_b, ref_b.data_ptr(), spp * i, spp)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    acc_a, acc_b = torch.zeros_like(ref_a), torch.zeros_like(ref_b)
    torch.cuda.synchronize()
    for i in range(n_rounds):                                  # enqueue only: nothing waits on the host in between
        a.render_device_noalloc(cfg_a, acc_a.data_ptr(), spp * i, spp, stream=s1.cuda_stream)
        b.render_device_noalloc(cfg_b, acc_b.data_ptr(), spp * i, spp, stream=s2.cuda_stream)
    torch.cuda.synchronize()
    assert ref_a.any().item() and ref_b.any().item()
    assert acc_a.cpu().numpy().tobytes() == ref_a.cpu().numpy().tobytes()
    assert acc_b.cpu().numpy().tobytes() == ref_b.cpu().numpy().tobytes()
    # one scene on two streams as well
    acc = torch.zeros_like(ref_a)
    torch.cuda.synchronize()
    for i in range(n_rounds):
        a.render_device_noalloc(cfg_a, acc.data_ptr(), spp * i, spp, stream=(s1 if i % 2 else s2).cuda_stream)
    torch.cuda.synchronize()
    assert acc.cpu().numpy().tobytes() == ref_a.cpu().numpy().tobytes()
