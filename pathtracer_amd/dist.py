"""Multi-GPU partitioning of the frame loop: one process per GPU, launched by
torch.distributed.run, RCCL over xGMI for the single exchange step.

Samples are independent and accumulation is a sum (main.rs:283, 434-436), so the path shards
with no data-path collective; the only communication is the final hand-over of the accumulation
buffers to rank 0:

  * ``tiles``   — rank r renders rows [y0, y1) of every frame; disjoint pixels, so the result is
                  bitwise identical to a single-GPU render.  One gather to rank 0.
  * ``samples`` — rank r renders the whole image for frames [s0, s1); one sum-reduce to rank 0
                  (f64; the summation order of the collective is not the frame order, so the last
                  bits may differ from a single-GPU render).

``render_fn(accum, tile, sample_begin, sample_count)`` adds frames to ``accum`` (a torch tensor,
h x w x 3 f64, on the rank's device) — ``Scene.render_device`` on GPUs; the CPU tests pass a host
renderer and the gloo backend.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def split_range(n: int, parts: int, index: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of part ``index`` (sizes differ by at most one)."""
    base, extra = divmod(n, parts)
    begin = index * base + min(index, extra)
    return begin, begin + base + (1 if index < extra else 0)


def render_partitioned(render_fn: Callable, width: int, height: int, sample_begin: int, sample_count: int,
                       mode: str = "tiles", device=None, group=None) -> torch.Tensor:
    """Renders ``sample_count`` frames across the ranks of ``group``; returns the full h x w x 3
    accumulation buffer on rank 0 (other ranks get their partial buffer back)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    accum = torch.zeros((height, width, 3), dtype=torch.float64, device=device)
    if mode == "tiles":
        y0, y1 = split_range(height, world, rank)
        if y1 > y0:
            render_fn(accum, (0, y0, width, y1), sample_begin, sample_count)
        if world > 1:
            # equal-size row bands (padded by at most one row) so one gather moves everything
            rows = -(-height // world)
            band = torch.zeros((rows, width, 3), dtype=torch.float64, device=device)
            band[: y1 - y0] = accum[y0:y1]
            bands = [torch.empty_like(band) for _ in range(world)] if rank == 0 else None
            dist.gather(band, bands, dst=0, group=group)
            if rank == 0:
                for r in range(world):
                    a, b = split_range(height, world, r)
                    accum[a:b] = bands[r][: b - a]
    elif mode == "samples":
        s0, s1 = split_range(sample_count, world, rank)
        if s1 > s0:
            render_fn(accum, (0, 0, width, height), sample_begin + s0, s1 - s0)
        if world > 1:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, group=group)
    else:
        raise ValueError(f"unknown partitioning mode {mode!r}")
    return accum
