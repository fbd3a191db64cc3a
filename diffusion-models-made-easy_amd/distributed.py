"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

Sampling shards the batch across ranks with NO data-path collective (chains are
independent): each rank derives its own Philox stream from (seed, rank) and
only timing / optional result gathering touch the process group.  Training adds exactly
one exchange per step: the mean all-reduce of the flat UNet gradient buffer, issued in
reverse-layer-order buckets so it can overlap the rest of backward."""

from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of `total` items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """distinct, reproducible stream per rank (ranks draw their own t, z, dropout masks)"""
    return (seed * 1000003 + 7919 * rank) & 0x7FFFFFFFFFFFFFFF


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bucket_slices(numel: int, bucket_elems: int) -> List[Tuple[int, int]]:
    """Contiguous slices of a flat buffer, LAST slice first: the parameters at the end of the
    reference's registration order (output conv, middle, up path) get their gradients first."""
    out = []
    end = numel
    while end > 0:
        begin = max(0, end - bucket_elems)
        out.append((begin, end))
        end = begin
    return out


def allreduce_mean_flat(flat_grad: torch.Tensor, bucket_elems: int = 8 << 20, async_op: bool = False):
    """Mean all-reduce of a flat gradient buffer in reverse-order buckets.  With async_op the
    work handles are returned so the caller can overlap them with remaining backward work."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return []
    world = dist.get_world_size()
    handles = []
    for b, e in bucket_slices(flat_grad.numel(), bucket_elems):
        view = flat_grad[b:e]
        view.div_(world)
        h = dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            handles.append(h)
    return handles
