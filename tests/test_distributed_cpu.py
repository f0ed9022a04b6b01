"""world_size-2 (and 3) gloo tests of the multi-GPU partitioning logic on CPU.  The renderer
plugged in here is the CPU oracle (test infrastructure); on GPUs it is Scene.render_device."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pathtracer_amd as pt
from pathtracer_amd.dist import render_partitioned, split_range


def test_split_range_is_a_partition():
    for n in (0, 1, 7, 1080):
        for parts in (1, 2, 3, 8):
            spans = [split_range(n, parts, i) for i in range(parts)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, w, h, spp, out_path):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = O.OracleScene(pt.scene_data("CornellBox"))
    cfg = pt.Config(width=w, height=h, seed=9)

    def render_fn(accum, tile, s0, n):
        a = accum.numpy()
        oracle.render(cfg, a, sample_begin=s0, sample_count=n, tile=tile, threads=1)

    acc = render_partitioned(render_fn, w, h, 2, spp, mode=mode)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["tiles", "samples"])
def test_partitioned_render_equals_single_process(tmp_path, world, mode):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as O
    w, h, spp = 40, 23, 5
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    ref = O.OracleScene(pt.scene_data("CornellBox")).render(pt.Config(width=w, height=h, seed=9), sample_begin=2, sample_count=spp)
    if mode == "tiles":
        assert got.tobytes() == ref.tobytes()          # disjoint pixels: bitwise identical
    else:
        assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)  # sum order of the reduce differs
    assert ref.mean() > 0.01
