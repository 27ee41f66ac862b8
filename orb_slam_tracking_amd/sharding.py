"""Multi-GPU sharding of the hot path: frames (and frame pairs) are independent units, so a batch is
cut into contiguous per-rank blocks, each rank runs its own orbx context on its own GPU, and the only exchange is an
all_gather of the per-frame keypoint counts (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
There is no data-path collective: descriptors never leave the GPU that produced them, because pairs never straddle ranks.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of frames for `rank`.  Blocks are even-sized (except possibly the last) so that the
    consecutive pairs (2k, 2k+1) used by extract+match always live on one rank."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    pairs = (n_frames + 1) // 2
    per = (pairs + world - 1) // world
    lo = min(rank * per * 2, n_frames)
    hi = min((rank + 1) * per * 2, n_frames)
    return lo, hi


def gather_counts(local_counts: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """all_gather of int32 keypoint counts; every rank must pass the same length (pad with -1 if ragged)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if out is None:
            return local_counts.clone()
        out[: local_counts.numel()].copy_(local_counts)
        return out
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * local_counts.numel(), dtype=local_counts.dtype, device=local_counts.device)
    dist.all_gather_into_tensor(out, local_counts.contiguous())
    return out


def gather_counts_ragged(local_counts: torch.Tensor, n_frames: int) -> torch.Tensor:
    """Counts of all `n_frames` frames in global frame order when the shards are ragged (last rank shorter)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    per = max(shard_range(n_frames, world, r)[1] - shard_range(n_frames, world, r)[0] for r in range(world))
    padded = torch.full((per,), -1, dtype=local_counts.dtype, device=local_counts.device)
    lo, hi = shard_range(n_frames, world, rank)
    padded[: hi - lo] = local_counts[: hi - lo]
    allc = gather_counts(padded).view(world, per)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_frames, world, r)
        parts.append(allc[r, : hi - lo])
    return torch.cat(parts)
