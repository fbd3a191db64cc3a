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


def init_from_env(device_index: Optional[int] = None) -> Tuple[int, int, int]:
    """Join the process group `python -m torch.distributed.run` describes in the environment (no-op for a single process or when
    already initialised).  Backend "nccl" (= RCCL over xGMI) unless DMME_DIST_BACKEND says otherwise (gloo: CPU tests and
    several ranks rehearsed on one GPU).  Returns (rank, local_rank, world)."""
    rank, local, world = env_rank_world()
    if world > 1 and dist.is_available() and not dist.is_initialized():
        backend = os.environ.get("DMME_DIST_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            idx = local if device_index is None else device_index
            torch.cuda.set_device(idx)
            dist.init_process_group("nccl", device_id=torch.device("cuda", idx))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def sync_parameters(model, optimizer=None, src: int = 0) -> bool:
    """Data-parallel start: every rank takes rank `src`'s parameters (and EMA copy), as DistributedDataParallel does when it
    wraps a module - averaged gradients only mean something for identical replicas.  One broadcast of the flat fp32 buffer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return False
    flat = model.flat_parameters()
    with torch.no_grad():
        dist.broadcast(flat, src=src)
    if hasattr(model, "mark_params_updated"):
        model.mark_params_updated()
    ema = optimizer.ema_parameters(model) if optimizer is not None and hasattr(optimizer, "ema_parameters") else None
    if ema is not None:
        with torch.no_grad():
            dist.broadcast(ema, src=src)
    return True


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


class OverlappedGradReducer:
    """Mean all-reduce of a dmme_amd UNet's flat gradient buffer, started bucket by bucket WHILE backward is still running.

    `dmme_unet_backward_buckets` reports each gradient bucket as soon as the launches that write it are enqueued (first the
    up / middle / output parameters - the tail of the flat buffer, ~2/3 of the bytes - then the rest).  For every report an event is
    recorded on the compute stream and the bucket's all-reduce (RCCL; in slices of `bucket_elems` so the ring pipelines) is issued
    on a side stream that waits for that event; the down path's backward keeps the compute stream busy meanwhile.  `finish()` makes
    the compute stream wait for the side stream before the optimiser reads the gradients.  On CPU tensors (gloo tests) there are no
    streams: the collectives run asynchronously and `finish()` waits on their handles."""

    def __init__(self, model, bucket_elems: int = 8 << 20):
        self.model = model
        self.bucket_elems = bucket_elems
        self.handles = []
        self.reported = []
        self._stream = None
        model._bucket_hook = self.bucket_ready  # picked up by UNet._backward_impl

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def bucket_ready(self, offset: int, numel: int, flat_grad: Optional[torch.Tensor] = None):
        self.reported.append((offset, numel))
        if not self.active():
            return
        flat = self.model.flat_grad() if flat_grad is None else flat_grad
        world = dist.get_world_size()
        view = flat[offset : offset + numel]
        if view.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=view.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(view.device))
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ev)
                self._reduce(view, world)
        else:
            self._reduce(view, world)

    def _reduce(self, view: torch.Tensor, world: int):
        view.div_(world)
        for b, e in bucket_slices(view.numel(), self.bucket_elems):
            self.handles.append(dist.all_reduce(view[b:e], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self, flat_grad: Optional[torch.Tensor] = None) -> bool:
        """True when the gradients were reduced here (the caller must then NOT reduce them again).  A step whose backward reported no
        bucket (configuration without a clean split, or a non-bucketed backward) is reduced in one piece now."""
        if not self.active():
            self.reported.clear()
            return False
        flat = self.model.flat_grad() if flat_grad is None else flat_grad
        covered = sum(n for _, n in self.reported)
        if covered == 0:
            self.bucket_ready(0, flat.numel(), flat)
        elif covered != flat.numel():
            raise RuntimeError(f"gradient buckets cover {covered} of {flat.numel()} elements")
        for h in self.handles:
            h.wait()  # CUDA: makes the current stream wait for the collective; gloo: blocks
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self.handles.clear()
        self.reported.clear()
        return True
