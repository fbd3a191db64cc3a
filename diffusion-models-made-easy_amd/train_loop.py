"""Plain training loop behind `python -m dmme_amd.trainer fit` (stands in for pl.Trainer.fit on
the keys of configs/ddpm/cifar10.yaml that touch the hot path).  Synthetic CIFAR10-shaped data:
x_0 = 2 U[0,1) - 1 (there is no dataset in the image; the input pipeline is out of scope)."""

from __future__ import annotations

import json
import os
import time

import torch

from . import distributed as D
from .common.noise import gaussian


def synthetic_batch(batch_size: int, device, shape=(3, 32, 32)):
    return torch.rand((batch_size, *shape), device=device) * 2 - 1


def train_step(module, optimizer, scheduler, x0, clip=None, reduce=True, exchange=None):
    """one optimisation step: loss -> HIP backward (data parallel: the gradient exchange of the first bucket runs on a side
    stream under the rest of backward) -> clip+Adam(+EMA) -> LR step.  `reduce=False` leaves the gradient exchange out
    (bench.py times the step without its collective to tell exposed from hidden exchange time).  `exchange`: wire format of
    the gradient mean (distributed.make_reducer: "fp32-allreduce" | "bf16-rs-ag"; default DMME_EXCHANGE or fp32)."""
    model = module.diffusion_model.model
    reducer = getattr(model, "_grad_reducer", None)
    multi = reduce and D.dist.is_available() and D.dist.is_initialized() and D.dist.get_world_size() > 1
    if multi and not getattr(model, "_dp_synced", False):  # first data-parallel step: identical replicas (rank 0's weights)
        D.sync_parameters(model, optimizer)
        model._dp_synced = True
    if multi and not os.environ.get("DMME_NO_OVERLAP"):
        want = D.run_exchange(model, int(x0.shape[0]), exchange)  # decided once per run, rank 0's choice (16-bit models at <= 32 images per rank: bf16)
        if reducer is not None and reducer.exchange != want:
            reducer.detach()
            reducer = None
        if reducer is None:
            reducer = model._grad_reducer = D.make_reducer(model, want)
        reducer.attach()
    elif reducer is not None:
        # no exchange in this step: the hook must be gone BEFORE backward runs, or that backward would issue all-reduces (and divide
        # the gradients) on the side stream with nobody waiting for them
        reducer.detach()
    loss = module.training_step((x0,), 0)
    loss.backward()
    if multi:
        if reducer is not None and getattr(model, "_bucket_hook", None) is not None and reducer.finish():
            if hasattr(optimizer, "grad_scale"):
                optimizer.grad_scale = reducer.grad_scale()
            elif reducer.grad_scale() != 1.0:
                model.flat_grad().mul_(reducer.grad_scale())
        else:
            D.allreduce_mean_flat(model.flat_grad())
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    optimizer.zero_grad()
    return loss


def fit(module, batch_size=128, max_steps=100, clip=None, log_every=50, loader=None, ckpt_path=None, save_path=None):
    dev = next(module.parameters()).device
    module.train()
    opts, scheds = module.configure_optimizers()
    opt = opts[0]
    if clip:
        for g in opt.param_groups:
            g["max_grad_norm"] = float(clip)
    sched = scheds[0]["scheduler"] if scheds else None
    first = 0
    if ckpt_path:
        from .checkpoint import load_checkpoint

        first = int(load_checkpoint(ckpt_path, module, opt, sched).get("global_step", 0))
    model = module.diffusion_model.model
    if D.sync_parameters(model, opt):  # several ranks: start from rank 0's weights (after a resume too)
        model._dp_synced = True
    t0 = time.perf_counter()
    batches = None
    for step in range(first, max_steps):
        if loader is None:
            x0 = synthetic_batch(batch_size, dev)
        else:  # epochs over the HBM-resident set: a fresh permutation each time the loader is exhausted
            try:
                x0 = next(batches)[0]
            except (StopIteration, TypeError):
                batches = iter(loader)
                x0 = next(batches)[0]
        loss = train_step(module, opt, sched, x0, clip)
        if (step + 1) % log_every == 0 or step + 1 == max_steps:
            torch.cuda.synchronize()
            if hasattr(model, "check_engine"):
                model.check_engine()  # a level-engine hand-off that timed out since the last log line: stop, do not train on it
            dt = time.perf_counter() - t0
            print(json.dumps({"step": step + 1, "train/loss": round(float(loss.detach()), 5), "images_per_s": round((step + 1 - first) * batch_size / dt, 1)}), flush=True)
    if save_path and D.env_rank_world()[0] == 0:
        from .checkpoint import save_checkpoint

        save_checkpoint(save_path, module, opt, sched, global_step=max(max_steps, first))
    return module
