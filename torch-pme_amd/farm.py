"""Frame farm: independent cells/frames spread over the GPUs of one node.

The reference has no parallelism of any kind (mesh calculators cannot even batch:
``calculators/pme.py:102-105``); frames are independent, so the MI355X layout is one process per GPU, a
contiguous block of frames per rank, no intra-cell decomposition, and ONE small collective per evaluation:
an all-gather of the per-frame energies (``torch.distributed``: backend ``nccl`` = RCCL over xGMI on the GPU
box, ``gloo`` in the CPU tests).  Message size is 4-8 bytes per frame -- latency-bound, far below the per-link
xGMI bandwidth, so no bucketing or ring/tree tuning applies.
"""

from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


def frame_block(n_frames: int, rank: int, world: int) -> range:
    """Contiguous block of frame indices owned by ``rank`` (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def frame_owner(frame: int, n_frames: int, world: int) -> int:
    for r in range(world):
        if frame in frame_block(n_frames, r, world):
            return r
    raise ValueError(f"frame {frame} outside 0..{n_frames - 1}")


def farm_energies(n_frames: int, evaluate: Callable[[int], torch.Tensor], device=None, dtype=torch.float64,
                  group=None) -> torch.Tensor:
    """Evaluate ``evaluate(frame_index) -> scalar energy tensor`` for this rank's block of frames and return
    the energies of ALL frames on every rank (one all-gather).  Without an initialised process group this is
    the single-process loop."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine = frame_block(n_frames, rank, world)
    per_rank = -(-n_frames // world)  # padded block length so that every rank contributes the same count
    local = torch.zeros(per_rank, dtype=dtype, device=device)
    for k, f in enumerate(mine):
        local[k] = evaluate(f).detach().to(dtype).reshape(())
    if not distributed:
        return local[:n_frames]
    gathered = torch.empty(world * per_rank, dtype=dtype, device=device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    out = torch.empty(n_frames, dtype=dtype, device=device)
    for r in range(world):
        blk = frame_block(n_frames, r, world)
        out[blk.start : blk.stop] = gathered[r * per_rank : r * per_rank + len(blk)]
    return out


def split_evenly(items: Sequence, rank: int, world: int) -> list:
    return [items[i] for i in frame_block(len(items), rank, world)]
