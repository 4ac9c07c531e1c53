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


def farm_energies_forces(n_atoms: Sequence[int], evaluate: Callable[[int], tuple], device=None, dtype=torch.float64,
                         group=None):
    """As :func:`farm_energies` for ``evaluate(frame_index) -> (energy, per-atom array (N_f, K))`` -- forces, or potentials +
    forces side by side (the optional gather of per-atom results, 8 frames x 8 000 atoms x 4 values ~ 2 MiB per
    rank at fp64).  ``n_atoms[f]`` is the size of frame ``f`` (known to every rank: it sizes the padded exchange buffer).
    Returns ``(energies (F,), [per-atom array of frame f, ...])`` on every rank; still ONE all-gather per quantity, no
    collective inside a frame."""
    n_frames = len(n_atoms)
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine = frame_block(n_frames, rank, world)
    per_rank = -(-n_frames // world)
    n_max = max(n_atoms) if n_frames else 0
    local_e = torch.zeros(per_rank, dtype=dtype, device=device)
    local_f = None
    for k, f in enumerate(mine):
        e, arr = evaluate(f)
        local_e[k] = e.detach().to(dtype).reshape(())
        arr = arr.detach().to(dtype)
        if arr.shape[0] != n_atoms[f]:
            raise ValueError(f"frame {f}: per-atom array has {arr.shape[0]} rows, `n_atoms` says {n_atoms[f]}")
        if local_f is None:
            local_f = torch.zeros((per_rank, n_max, arr.shape[1]), dtype=dtype, device=device)
        local_f[k, : arr.shape[0]] = arr
    width = torch.tensor([0 if local_f is None else local_f.shape[2]], dtype=torch.int64, device=device)
    if distributed:
        dist.all_reduce(width, op=dist.ReduceOp.MAX, group=group)  # a rank without frames learns the column count
    if local_f is None:
        local_f = torch.zeros((per_rank, n_max, int(width)), dtype=dtype, device=device)
    if not distributed:
        return local_e[:n_frames], [local_f[f, : n_atoms[f]] for f in range(n_frames)]
    all_e = torch.empty(world * per_rank, dtype=dtype, device=device)
    all_f = torch.empty((world * per_rank,) + tuple(local_f.shape[1:]), dtype=dtype, device=device)
    dist.all_gather_into_tensor(all_e, local_e, group=group)
    dist.all_gather_into_tensor(all_f, local_f, group=group)
    energies = torch.empty(n_frames, dtype=dtype, device=device)
    arrays = [None] * n_frames
    for r in range(world):
        blk = frame_block(n_frames, r, world)
        energies[blk.start : blk.stop] = all_e[r * per_rank : r * per_rank + len(blk)]
        for k, f in enumerate(blk):
            arrays[f] = all_f[r * per_rank + k, : n_atoms[f]]
    return energies, arrays


def gather_energy_log(values: torch.Tensor, group=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """ONE all-gather for MANY evaluations: ``values`` (K, F) = this rank's frame energies of K successive evaluations of its
    F frames (``graphed.EnergyLog.values[:K]``, filled on the device by the captured steps, or any tensor of that shape; the
    same K and F on every rank).  Returns (world, K, F): ``out[r, k, f]`` = energy of frame f of rank r's k-th batch, on every
    rank.  This is the exchange of a farm that streams batches of frames through each GPU: nothing is communicated between two
    evaluations, the log (8 bytes per frame and evaluation) crosses xGMI once.  ``out`` (world, K, F), contiguous: a caller that
    exchanges the same log again and again passes its own buffer (no allocation inside a timed region)."""
    if values.dim() != 2:
        raise ValueError(f"expected a (K, F) log, got {tuple(values.shape)}")
    values = values.contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        if out is not None:
            out.copy_(values.unsqueeze(0))
            return out
        return values.unsqueeze(0).clone()
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(values.shape), dtype=values.dtype, device=values.device)
    elif tuple(out.shape) != (world,) + tuple(values.shape) or not out.is_contiguous() or out.dtype != values.dtype:
        raise ValueError(f"`out` must be a contiguous {(world,) + tuple(values.shape)} tensor of {values.dtype}")
    dist.all_gather_into_tensor(out.view(-1), values.view(-1), group=group)
    return out


def split_evenly(items: Sequence, rank: int, world: int) -> list:
    return [items[i] for i in frame_block(len(items), rank, world)]
