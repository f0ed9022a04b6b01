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
from pathtracer_amd.dist import Partition, interleaved_rows, render_partitioned, split_range


def test_split_range_is_a_partition():
    for n in (0, 1, 7, 1080):
        for parts in (1, 2, 3, 8):
            spans = [split_range(n, parts, i) for i in range(parts)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_interleaved_rows_are_a_partition():
    for h in (1, 7, 8, 23, 1080):
        for parts in (1, 2, 3, 8):
            rows = [interleaved_rows(h, parts, i).tolist() for i in range(parts)]
            assert sorted(sum(rows, [])) == list(range(h))
            for i, r in enumerate(rows):
                assert all((y // 8) % parts == i for y in r)
            part = Partition("tiles", 16, h)
            assert part.world == 1 and part.pixels() == 16 * h


def test_row_partition_calls_cover_the_image_once():
    # pt_render_params::row_group / row_parts / row_index through the oracle: the calls of all parts
    # together add exactly what one call adds, for group sizes that do and do not divide the tile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as O
    oracle = O.OracleScene(pt.scene_data("CornellBox"))
    cfg = pt.Config(width=31, height=29, seed=4)
    whole = oracle.render(cfg, sample_count=2, threads=4)
    for group, parts, tile in ((8, 2, None), (8, 3, None), (1, 4, None), (5, 2, (3, 2, 30, 27)), (8, 5, None)):
        acc = np.zeros_like(whole)
        for index in range(parts):
            before = acc.copy()
            oracle.render(cfg, acc, sample_count=2, tile=tile, rows=(group, parts, index), threads=4)
            y0 = tile[1] if tile else 0
            touched = np.flatnonzero((acc != before).any(axis=(1, 2)))
            assert all(((y - y0) // group) % parts == index for y in touched)
        ref = whole if tile is None else oracle.render(cfg, sample_count=2, tile=tile, threads=4)
        assert acc.tobytes() == ref.tobytes(), (group, parts, tile)


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

    def render_fn(accum, tile, s0, n, rows):
        a = accum.numpy()
        oracle.render(cfg, a, sample_begin=s0, sample_count=n, tile=tile, threads=1, rows=rows)

    ordered = mode != "samples-reduce"
    acc = render_partitioned(render_fn, w, h, 2, spp, mode=mode.split("-")[0], ordered=ordered)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["tiles", "bands", "samples", "samples-reduce"])
def test_partitioned_render_equals_single_process(tmp_path, world, mode):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_lib as O
    w, h, spp = 40, 23, 5
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, _free_port(), mode, w, h, spp, out), nprocs=world, join=True)
    got = np.load(out)
    oracle = O.OracleScene(pt.scene_data("CornellBox"))
    cfg = pt.Config(width=w, height=h, seed=9)
    ref = oracle.render(cfg, sample_begin=2, sample_count=spp)
    if mode in ("tiles", "bands"):
        assert got.tobytes() == ref.tobytes()          # disjoint pixels: bitwise identical
    else:
        assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)  # partial sums per rank: not the frame order
    if mode == "samples":
        # the ordered exchange: exactly ((S0 + S1) + S2), S_r = rank r's frames added in frame order
        parts = []
        for r in range(world):
            s0, s1 = split_range(spp, world, r)
            parts.append(oracle.render(cfg, sample_begin=2 + s0, sample_count=s1 - s0))
        total = parts[0]
        for p in parts[1:]:
            total = total + p
        assert got.tobytes() == total.tobytes()
    assert ref.mean() > 0.01
