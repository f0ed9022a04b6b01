"""The multi-GPU path on the GPU: two ranks on one device over gloo (RCCL refuses two ranks on one GPU;
the collective's backend is the only difference from the 8-GPU job), through Scene.render_device on
torch's stream — the code bench.py times — against the one-rank image."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import pathtracer_amd as pt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, name, w, h, spp, out_path):
    import torch
    import torch.distributed as dist
    from pathtracer_amd.dist import render_partitioned
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    scene = pt.Scene(pt.scene_data(name, textures=name == "Test"), device=0)
    cfg = pt.Config(width=w, height=h, bounces=2, seed=13, nee=name == "Test")
    stream = torch.cuda.current_stream().cuda_stream

    def render_fn(accum, tile, s0, n, rows):
        scene.render_device(cfg, accum.data_ptr(), s0, n, tile=tile, stream=stream, rows=rows)

    acc = render_partitioned(render_fn, w, h, 3, spp, mode=mode, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out_path, acc.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,name", [("tiles", "CornellBox"), ("tiles", "Test"), ("bands", "CornellBox"), ("samples", "CornellBox")])
def test_two_ranks_on_one_gpu_equal_one_rank(gpu, tmp_path, mode, name):
    import torch.multiprocessing as mp
    w, h, spp, world = 333, 250, 6, 2          # 250 rows: 31 whole groups of 8 and a short one
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, name, w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    scene = pt.Scene(pt.scene_data(name, textures=name == "Test"), device=gpu)
    cfg = pt.Config(width=w, height=h, bounces=2, seed=13, nee=name == "Test")
    if mode == "samples":                      # ordered exchange: S0 + S1, each the rank's frames in frame order
        ref = scene.render(cfg, sample_begin=3, sample_count=3) + scene.render(cfg, sample_begin=6, sample_count=3)
    else:                                      # disjoint pixels: the one-rank image, bit for bit
        ref = scene.render(cfg, sample_begin=3, sample_count=spp)
    assert ref.any() and got.tobytes() == ref.tobytes()


def test_row_partition_renders_cover_the_image_once(gpu):
    # pt_render_params::row_group / row_parts / row_index on the device: the parts' renders add up to the
    # whole image bit for bit, whatever the group size, with tiles, NEE and the deep-tree kernels
    from pathtracer_amd.synthetic import heightfield_box
    for data, nee in ((pt.scene_data("CornellBox"), False), (heightfield_box(cells=40), True)):
        scene = pt.Scene(data, device=gpu)
        cfg = pt.Config(width=211, height=97, bounces=2, seed=3, nee=nee)
        whole = scene.render(cfg, sample_count=3)
        for group, parts, tile in ((8, 2, None), (8, 3, None), (1, 4, None), (5, 2, (3, 2, 200, 91)), (8, 16, None)):
            acc = np.zeros_like(whole)
            for index in range(parts):
                scene.render(cfg, acc, sample_count=3, tile=tile, rows=(group, parts, index))
            ref = whole if tile is None else scene.render(cfg, sample_count=3, tile=tile)
            assert acc.tobytes() == ref.tobytes(), (group, parts, tile)
    with pytest.raises(pt.PtError):
        scene.render(cfg, sample_count=1, rows=(0, 2, 0))
    with pytest.raises(pt.PtError):
        scene.render(cfg, sample_count=1, rows=(8, 2, 2))
    with pytest.raises(pt.PtError):                            # a group beyond the image (a wrapped product otherwise)
        scene.render(cfg, sample_count=1, rows=(0xFFFFFFFF, 2, 0))
    p = cfg.params(0, 1)
    p.reserved = 7                                              # a struct that was not zero-initialised
    acc = np.zeros((cfg.h, cfg.w, 3))
    from pathtracer_amd import _lib as L
    import ctypes as C
    assert L.lib.pt_render(scene._h, C.byref(p), acc.ctypes.data_as(C.POINTER(C.c_double))) == L.PT_ERR_INVALID


def test_bench_starts_its_own_ranks(gpu):
    # `python bench.py --gpus 2` with no launcher in the environment: bench.py spawns the two ranks itself.
    # Default partition: tiles of the ONE image (strong scaling: the step's samples equal the 1-GPU line's),
    # gathered to rank 0 inside every step; the samples partition is timed after it and both are reported.
    env = dict(os.environ, PT_BENCH_BACKEND="gloo", PT_BENCH_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    for partition in (None, "samples"):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--spp", "8", "--batch", "8", "--width", "640", "--height", "360", "--cpu-seconds", "0"]
        if partition:
            cmd += ["--partition", partition, "--other-partition", "none"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["kernel"] == "path"
        assert d["exchange_ms_per_step"] > 0
        if partition is None:
            assert d["scaling"] == "strong" and d["config"]["samples_per_step"] == 640 * 360 * 8
            assert set(d["partitions"]) == {"tiles", "samples"}
            assert d["partitions"]["samples"]["scaling"] == "weak"
            assert d["partitions"]["samples"]["samples_per_step"] == 640 * 360 * 8 * 2
            assert d["partitions"]["tiles"]["value"] == d["value"]
        else:
            assert d["scaling"] == "weak" and d["config"]["samples_per_step"] == 640 * 360 * 8 * 2
            assert set(d["partitions"]) == {"samples"}


def _nccl_worker(rank, port, out_path):
    import torch
    import torch.distributed as dist
    from pathtracer_amd.dist import Partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    scene = pt.Scene(pt.scene_data("CornellBox"), device=0)
    w, h = 333, 250
    cfg = pt.Config(width=w, height=h, bounces=2, seed=13)
    stream = torch.cuda.current_stream().cuda_stream

    def render_fn(accum, tile, s0, n, rows):
        scene.render_device(cfg, accum.data_ptr(), s0, n, tile=tile, stream=stream, rows=rows)

    results = {}
    for label, mode, ordered in (("tiles", "tiles", True), ("bands", "bands", True), ("samples_ordered", "samples", True),
                                 ("samples_reduce", "samples", False)):
        part = Partition(mode, w, h, ordered=ordered)
        assert part.world == 1 and dist.get_backend() == "nccl"
        acc = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
        part.render(render_fn, acc, 3, 4)
        before = acc.clone()
        part.exchange(acc)                    # RCCL gather / reduce on torch's stream, beside libpt_hip.so's kernels
        part.exchange(acc)                    # ... and again over the cached staging buffers
        torch.cuda.synchronize()
        results[label] = bool(torch.equal(acc, before)) and bool(before.any())
    t = torch.tensor([2.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    results["all_reduce"] = float(t.item()) == 2.5
    results["image"] = before.cpu().numpy().tobytes().hex()[:64]
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(results, f)


def test_exchange_under_a_one_rank_nccl_group(gpu, tmp_path):
    # RCCL itself on the box: a world-size-1 nccl process group (two RCCL ranks cannot share one GPU) running
    # Partition.exchange for tiles / bands / samples-ordered / samples-reduce — the collectives bench.py issues on the
    # 8-GPU node, loaded beside libpt_hip.so's HIP runtime, on torch's stream behind the render kernels.  One rank's
    # exchange must hand back exactly the bytes it was given.
    import torch.multiprocessing as mp
    out = str(tmp_path / "nccl.json")
    mp.spawn(_nccl_worker, args=(_free_port(), out), nprocs=1, join=True)
    with open(out) as f:
        res = json.load(f)
    for k in ("tiles", "bands", "samples_ordered", "samples_reduce", "all_reduce"):
        assert res[k] is True, res
