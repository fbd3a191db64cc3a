"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

Sampling shards the batch across ranks with NO data-path collective (chains are
independent): each rank derives its own Philox stream from (seed, rank) and
only timing / optional result gathering touch the process group.  Training adds exactly
one exchange per step: the mean of the flat UNet gradient buffer over the ranks, issued in
reverse-layer-order buckets so it can overlap the rest of backward.  Two wire formats
(`exchange=`): "fp32-allreduce" (RCCL ring all-reduce of the fp32 buffer, the divide by the
world size folded into the fused clip + Adam pass) and "bf16-rs-ag" (Bf16ShardExchange: bf16 on
the wire, every rank's shard sent directly to its owner over all xGMI links at once, fp32
accumulation at the owner, all-gather of the rounded means: SURVEY 8e's design for the
16-images-per-GPU reading of north_star)."""

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
            # ONE rule for the device of a rank, here and in trainer.main: local rank modulo the visible devices
            idx = (local % max(1, torch.cuda.device_count())) if device_index is None else device_index
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
    # Adam moments and the step count are assumed identical on every rank (all ranks start fresh or load the same checkpoint):
    # check the one scalar that would show a rank that resumed from something else
    if optimizer is not None and hasattr(optimizer, "_step_count"):
        steps = torch.tensor([int(optimizer._step_count)], dtype=torch.int64, device=flat.device)
        lo, hi = steps.clone(), steps.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if int(lo) != int(hi):
            raise RuntimeError(f"data-parallel start: optimiser step counts differ across ranks ({int(lo)} .. {int(hi)}); load the same checkpoint on every rank")
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


def allreduce_mean_flat(flat_grad: torch.Tensor, bucket_elems: int = 8 << 20, async_op: bool = False, mean: bool = True):
    """All-reduce of a flat gradient buffer in reverse-order buckets.  With async_op the work handles are
    returned so the caller can overlap them with remaining backward work.  mean=False leaves the rank SUMS in the
    buffer: the caller folds 1 / world into the optimiser's fused pass (FusedAdam.grad_scale) instead of sweeping
    the buffer once more here."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return []
    world = dist.get_world_size()
    handles = []
    for b, e in bucket_slices(flat_grad.numel(), bucket_elems):
        view = flat_grad[b:e]
        if mean:
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

    exchange = "fp32-allreduce"

    def __init__(self, model, bucket_elems: int = 8 << 20, fold_mean: bool = True):
        self.model = model
        self.bucket_elems = bucket_elems
        self.fold_mean = fold_mean  # leave rank SUMS in the buffer; `grad_scale()` (1 / world) goes into the fused clip + Adam pass
        self.handles = []
        self.reported = []
        self._stream = None
        self.attach()

    def attach(self):
        """backward reports its gradient buckets to this reducer (UNet._backward_impl picks the hook up)"""
        self.model._bucket_hook = self.bucket_ready

    def detach(self):
        """a step WITHOUT gradient exchange on a model that has a reducer: nothing may be issued from inside its backward"""
        self.model._bucket_hook = None
        for h in self.handles:
            h.wait()
        self.handles.clear()
        self.reported.clear()
        self.model._exchange_in_flight = False

    def abort(self):
        """a backward that raised after handing over its first bucket(s): join what was issued (the collectives themselves cannot be
        recalled - the peers are in them), forget the step's bookkeeping and lift the forward guard, so that the error the caller sees
        is the backward's own and not "exchange in flight" at the next forward"""
        for h in self.handles:
            try:
                h.wait()
            except Exception:  # noqa: BLE001 - the original error is the one to surface
                pass
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self.handles.clear()
        self.reported.clear()
        self.model._exchange_in_flight = False

    def grad_scale(self) -> float:
        """what the optimiser must multiply the exchanged buffer by to obtain the mean gradient"""
        return 1.0 / dist.get_world_size() if (self.fold_mean and self.active()) else 1.0

    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def bucket_ready(self, offset: int, numel: int, flat_grad: Optional[torch.Tensor] = None):
        self.reported.append((offset, numel))
        if not self.active():
            return
        # from the first collective of a step until finish() joined them, no forward of this model may be enqueued: the level
        # engine's persistent launches need every compute unit free of other kernels' workgroups (models/ddpm.py: _forward_impl asks)
        self.model._exchange_in_flight = True
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
        if not self.fold_mean:
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
        self.model._exchange_in_flight = False  # everything the compute stream enqueues from here on is ordered behind the collectives
        return True


class Bf16ShardExchange(OverlappedGradReducer):
    """Gradient mean with bf16 on the wire and fp32 accumulation: reduce-scatter + all-gather in their DIRECT form.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of M bytes moves 2 (w-1)/w M per GPU over ONE outgoing link
    (fp32: 227 MB -> ~1.5 ms), while sending shard j straight to rank j uses all links at once: 2 x M/w per peer (bf16: 2 x 8.1 MB ->
    ~0.11 ms at world 8).  Per sub-bucket (slices of `bucket_elems`, the tail of the flat buffer first, at least four per
    step so the exchange pipelines under the rest of backward):
      pack      fp32 slice -> bf16, padded to world x per            (dmme_grad_pack_bf16)
      all-to-all   rank r sends its copy of shard j to rank j        (torch.distributed: RCCL all_to_all_single)
      reduce    owner: sum of the w received copies in fp32, rank order, x 1/w, ONE rounding to bf16   (dmme_shard_reduce_bf16)
      all-gather   the rounded means                                  (RCCL all_gather_into_tensor)
      unpack    bf16 -> the fp32 gradient buffer                      (dmme_grad_unpack_bf16)
    The owner also takes the ROUNDED mean of its own shard, so every rank's buffer holds identical bits and clip + Adam + EMA stay
    rank-identical with no further communication.  Error: one bf16 rounding per contribution (2^-9 relative each, independent) and one
    of the mean - against 4-5 % per-tensor gradient error of the bf16 backward itself.  On CPU tensors (gloo
    tests) the three conversions run as torch ops with the same roundings."""

    exchange = "bf16-rs-ag"

    def __init__(self, model, bucket_elems: int = 4 << 20):
        super().__init__(model, bucket_elems, fold_mean=False)
        self._bufs = {}

    def grad_scale(self) -> float:
        return 1.0  # the owner's reduce already divided

    def _buffers(self, pad: int, per: int, like: torch.Tensor):
        key = (pad, like.device)
        b = self._bufs.get(key)
        if b is None:
            mk = lambda n: torch.empty(n, dtype=torch.bfloat16, device=like.device)
            b = self._bufs[key] = (mk(pad), mk(pad), mk(per), mk(pad))
        return b

    def _reduce(self, view: torch.Tensor, world: int):
        for b, e in bucket_slices(view.numel(), self.bucket_elems):
            self._exchange(view[b:e], world)

    def _exchange(self, v: torch.Tensor, world: int):
        n = v.numel()
        per = (n + world - 1) // world
        pad = per * world
        send, recv, shard, gathered = self._buffers(pad, per, v)
        if v.is_cuda:
            from . import _lib

            lib, st = _lib.lib(), _lib.stream_ptr()
            _lib.check(lib.dmme_grad_pack_bf16(_lib.ptr(v), n, _lib.ptr(send), pad, st), "dmme_grad_pack_bf16")
            dist.all_to_all_single(recv, send)
            _lib.check(lib.dmme_shard_reduce_bf16(_lib.ptr(recv), world, per, 1.0 / world, _lib.ptr(shard), st), "dmme_shard_reduce_bf16")
            dist.all_gather_into_tensor(gathered, shard)
            _lib.check(lib.dmme_grad_unpack_bf16(_lib.ptr(gathered), n, _lib.ptr(v), st), "dmme_grad_unpack_bf16")
        else:  # gloo rehearsal on CPU tensors: the same roundings and summation order with torch ops
            send.zero_()
            send[:n].copy_(v)
            dist.all_to_all_single(recv, send)
            acc = torch.zeros(per, dtype=torch.float32)
            for j in range(world):
                acc += recv[j * per : (j + 1) * per].to(torch.float32)
            shard.copy_(acc * (1.0 / world))
            dist.all_gather_into_tensor(gathered, shard)
            v.copy_(gathered[:n])


def default_exchange(per_rank_batch: Optional[int] = None, model=None) -> str:
    """`fp32-allreduce` (what the reference's DDP does), except where the step is too short to hide a ring all-reduce of 130 MB over one
    xGMI link (~1.5 ms at world 8) AND the model computes in 16 bits anyway: at <= 32 images per rank (north_star's "batch of 128
    sharded over 8 GPUs" = 16 per rank: ~1.1 ms of backward) a bf16 / fp16 model takes the direct bf16 reduce-scatter + all-gather
    (2 x 8 MB per peer over all seven links, ~0.11 ms).  fp32 and bf16x3 models keep the fp32 all-reduce whatever the batch: their
    parity claim (1e-5) does not survive bf16-rounded gradients.  DMME_EXCHANGE overrides."""
    env = os.environ.get("DMME_EXCHANGE")
    if env:
        return env
    if model is not None:
        from . import _lib

        if getattr(model, "_dtype", None) not in (_lib.BF16, _lib.F16):
            return "fp32-allreduce"
    return "bf16-rs-ag" if (per_rank_batch is not None and per_rank_batch <= 32) else "fp32-allreduce"


def run_exchange(model, per_rank_batch: int, exchange: Optional[str] = None) -> str:
    """the exchange of THIS RUN, decided once - at the first data-parallel step, from that step's batch - and agreed between the ranks
    (rank 0's choice is broadcast): a short last batch must not swap the reducer (new stream, new buffers) and ranks whose local batch
    sizes differ must not pick different collectives (all_reduce against all_to_all would hang)."""
    kind = getattr(model, "_dp_exchange", None)
    if kind is not None and exchange in (None, kind):
        return kind
    kind = exchange or default_exchange(per_rank_batch, model)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        names = ["fp32-allreduce", "bf16-rs-ag"]
        if kind not in names:
            raise ValueError(f"unknown gradient exchange '{kind}' (fp32-allreduce | bf16-rs-ag)")
        code = torch.tensor([names.index(kind)], dtype=torch.int64)
        dev = model.flat_parameters().device
        if dev.type == "cuda" and dist.get_backend() == "nccl":
            code = code.to(dev)
        dist.broadcast(code, src=0)
        kind = names[int(code.item())]
    model._dp_exchange = kind
    return kind


def make_reducer(model, exchange: Optional[str] = None, per_rank_batch: Optional[int] = None):
    """the gradient exchange of a data-parallel run: `exchange` / DMME_EXCHANGE = 'fp32-allreduce' or 'bf16-rs-ag'; unset: by the
    per-rank batch and the model's precision (default_exchange)"""
    kind = exchange or default_exchange(per_rank_batch, model)
    if kind == "bf16-rs-ag":
        return Bf16ShardExchange(model)
    if kind != "fp32-allreduce":
        raise ValueError(f"unknown gradient exchange '{kind}' (fp32-allreduce | bf16-rs-ag)")
    return OverlappedGradReducer(model)
