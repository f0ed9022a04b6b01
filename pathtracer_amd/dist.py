"""Multi-GPU partitioning of the frame loop: one process per GPU, launched by
torch.distributed.run, RCCL over xGMI for the single exchange step.

Samples are independent and accumulation is a sum (main.rs:283, 434-436), so the path shards
with no data-path collective; the only communication is the final hand-over of the accumulation
buffers to rank 0:

  * ``tiles``   — rank r renders the groups of 8 image rows g with g % world == r, every frame
                  (pt_render_params::row_group / row_parts / row_index: ONE render call per rank, the
                  interleaving happens in the kernels' pixel mapping).  Each GPU sees every region of
                  the image, so a cheap ceiling and an expensive floor are shared out evenly; the pixels
                  are disjoint, so the result is bitwise identical to a single-GPU render.  One gather
                  of the ranks' rows to rank 0.
  * ``bands``   — the same with one contiguous band of rows per rank (uniform scenes only).
  * ``samples`` — rank r renders the whole image for frames [s0, s1).  The partial sums are gathered
                  to rank 0 and added in rank order (``ordered=True``, the default: a fixed summation
                  order, so a given world size always gives the same bits — SURVEY §8e.2), or summed by
                  one RCCL reduce (``ordered=False``: the ring's order, last bits may differ run to run).
                  Neither equals the single-GPU frame order bit for bit; f64 drift is ~1e-16 relative.

``render_fn(accum, tile, sample_begin, sample_count, rows)`` adds frames to ``accum`` (a torch tensor,
h x w x 3 f64, on the rank's device), ``rows`` = (row_group, row_parts, row_index) or None —
``Scene.render_device`` on GPUs; the CPU tests pass a host renderer and the gloo backend.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

ROW_GROUP = 8  # rows per interleaved group: the kernels trace primary rays in 8x8-pixel blocks


def split_range(n: int, parts: int, index: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of part ``index`` (sizes differ by at most one)."""
    base, extra = divmod(n, parts)
    begin = index * base + min(index, extra)
    return begin, begin + base + (1 if index < extra else 0)


def interleaved_rows(height: int, parts: int, index: int, group: int = ROW_GROUP) -> torch.Tensor:
    """Image rows of part ``index``: the row groups g = index, index + parts, ... (int64 tensor)."""
    rows = torch.arange(height, dtype=torch.int64)
    return rows[(rows // group) % parts == index]


class Partition:
    """How one render is cut across the ranks of ``group``, and the one exchange that reassembles it."""

    def __init__(self, mode: str, width: int, height: int, group=None, ordered: bool = True,
                 row_group: int = ROW_GROUP):
        if mode not in ("tiles", "bands", "samples"):
            raise ValueError(f"unknown partitioning mode {mode!r}")
        self.mode, self.width, self.height, self.group = mode, width, height, group
        self.ordered, self.row_group = ordered, row_group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # exchange() runs once per image: its row indices and staging buffers are made once per device
        self._rows = {}     # (rank, device) -> int64 row indices on that device
        self._stage = {}    # (name, shape, dtype, device) -> reusable tensor

    # ---- what this rank renders --------------------------------------------------------------------
    def rows_of(self, rank: int) -> torch.Tensor:
        if self.mode == "tiles":
            return interleaved_rows(self.height, self.world, rank, self.row_group)
        if self.mode == "bands":
            return torch.arange(*split_range(self.height, self.world, rank), dtype=torch.int64)
        return torch.arange(self.height, dtype=torch.int64)

    def _rows_on(self, rank: int, device) -> torch.Tensor:
        key = (rank, str(device))
        if key not in self._rows:
            self._rows[key] = self.rows_of(rank).to(device)
        return self._rows[key]

    def _buffer(self, name: str, shape, like: torch.Tensor, device=None) -> torch.Tensor:
        device = like.device if device is None else device
        key = (name, tuple(shape), like.dtype, str(device))
        if key not in self._stage:
            self._stage[key] = torch.zeros(tuple(shape), dtype=like.dtype, device=device)
        return self._stage[key]

    def pixels(self) -> int:
        """Pixels this rank renders per frame."""
        return int(self.rows_of(self.rank).numel()) * self.width

    def frames(self, sample_count: int) -> Tuple[int, int]:
        """[begin, end) within a range of ``sample_count`` frames that this rank renders."""
        return split_range(sample_count, self.world, self.rank) if self.mode == "samples" else (0, sample_count)

    def render(self, render_fn: Callable, accum: torch.Tensor, sample_begin: int, sample_count: int) -> None:
        """This rank's share of frames [sample_begin, sample_begin + sample_count), added to ``accum``."""
        full = (0, 0, self.width, self.height)
        if self.world == 1:
            render_fn(accum, full, sample_begin, sample_count, None)
        elif self.mode == "tiles":
            if self.rank * self.row_group < self.height:
                render_fn(accum, full, sample_begin, sample_count, (self.row_group, self.world, self.rank))
        elif self.mode == "bands":
            y0, y1 = split_range(self.height, self.world, self.rank)
            if y1 > y0:
                render_fn(accum, (0, y0, self.width, y1), sample_begin, sample_count, None)
        else:
            s0, s1 = self.frames(sample_count)
            if s1 > s0:
                render_fn(accum, full, sample_begin + s0, s1 - s0, None)

    # ---- the one exchange --------------------------------------------------------------------------
    def _gather(self, t: torch.Tensor, dst: int = 0):
        """dist.gather of equal-shaped tensors into reusable buffers; gloo moves device tensors through the host."""
        via_host = t.is_cuda and dist.get_backend(self.group) == "gloo"
        src = t.cpu() if via_host else t
        parts = None
        if self.rank == dst:
            parts = [self._buffer(f"part{r}", src.shape, src, src.device) for r in range(self.world)]
        dist.gather(src, parts, dst=dst, group=self.group)
        if parts is not None and via_host:
            parts = [p.to(t.device) for p in parts]
        return parts

    def exchange(self, accum: torch.Tensor) -> torch.Tensor:
        """Hands every rank's share to rank 0; ``accum`` is complete there afterwards.  One call per image: an
        image is done when rank 0 holds it."""
        if self.world == 1 and not dist.is_initialized():
            return accum
        if self.mode in ("tiles", "bands"):
            # every rank sends its own rows, packed and padded to the largest share: one gather (grouped
            # point-to-point sends into rank 0 under RCCL: 7 xGMI links fan in, no ring)
            counts = [int(self.rows_of(r).numel()) for r in range(self.world)]
            mine = self._rows_on(self.rank, accum.device)
            band = self._buffer("band", (max(counts), self.width, 3), accum)
            torch.index_select(accum, 0, mine, out=band[: mine.numel()])
            parts = self._gather(band)
            if self.rank == 0:
                for r in range(1, self.world):
                    accum.index_copy_(0, self._rows_on(r, accum.device), parts[r][: counts[r]])
        elif self.ordered:
            parts = self._gather(accum)
            if self.rank == 0:
                total = parts[0]
                for r in range(1, self.world):   # fixed order: ((S0 + S1) + S2) + ...
                    total = total + parts[r]
                accum.copy_(total)
        else:
            if accum.is_cuda and dist.get_backend(self.group) == "gloo":
                host = accum.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM, group=self.group)
                accum.copy_(host)
            else:
                dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM, group=self.group)
        return accum


def render_partitioned(render_fn: Callable, width: int, height: int, sample_begin: int, sample_count: int,
                       mode: str = "tiles", device=None, group=None, ordered: bool = True,
                       accum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Renders ``sample_count`` frames across the ranks of ``group``; returns the full h x w x 3
    accumulation buffer on rank 0 (other ranks get their partial buffer back)."""
    part = Partition(mode, width, height, group=group, ordered=ordered)
    if accum is None:
        accum = torch.zeros((height, width, 3), dtype=torch.float64, device=device)
    part.render(render_fn, accum, sample_begin, sample_count)
    return part.exchange(accum)
