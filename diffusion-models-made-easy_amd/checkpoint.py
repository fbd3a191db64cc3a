"""Lightning-shaped checkpoints for the bundled runner.

The reference trains under pytorch_lightning with its EMA callback, whose `.ckpt` files are dicts with
`state_dict` (keys `diffusion_model.model.<unet key>`), `optimizer_states` ([EMAOptimizer.state_dict()]: {"opt", "ema",
"current_step", "decay", ...}, callbacks/ema.py:339-359), `lr_schedulers`, `global_step`, `epoch`.  The same layout is
written and read here, so weights / optimiser moments / EMA copies move between the two code bases with torch.load."""

from __future__ import annotations

from typing import Any, Dict, Optional

import torch


def checkpoint_dict(module, optimizer=None, scheduler=None, global_step: int = 0, epoch: int = 0) -> Dict[str, Any]:
    ckpt = {
        "epoch": epoch,
        "global_step": global_step,
        "pytorch-lightning_version": "1.8.4.post0",  # the version the reference's configs were generated with (configs/ddpm/cifar10.yaml:1)
        "state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()},
        "optimizer_states": [],
        "lr_schedulers": [],
    }
    if optimizer is not None:
        sd = optimizer.state_dict()
        to_cpu = lambda t: t.detach().cpu() if torch.is_tensor(t) else t  # noqa: E731
        sd["opt"]["state"] = {i: {k: to_cpu(v) for k, v in st.items()} for i, st in sd["opt"]["state"].items()}
        sd["ema"] = tuple(to_cpu(t) for t in sd["ema"])
        sd["device"] = str(sd["device"])
        ckpt["optimizer_states"].append(sd)
    if scheduler is not None:
        ckpt["lr_schedulers"].append(scheduler.state_dict())
    return ckpt


def save_checkpoint(path: str, module, optimizer=None, scheduler=None, global_step: int = 0, epoch: int = 0) -> None:
    torch.save(checkpoint_dict(module, optimizer, scheduler, global_step, epoch), path)


def load_checkpoint(path_or_dict, module, optimizer=None, scheduler=None, strict: bool = True) -> Dict[str, Any]:
    """restore weights (+ optimiser moments, EMA copies, LR schedule); returns the checkpoint dict (global_step, epoch, ...)"""
    ckpt = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, str) else path_or_dict
    module.load_state_dict(ckpt["state_dict"], strict=strict)
    if optimizer is not None and ckpt.get("optimizer_states"):
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
    if scheduler is not None and ckpt.get("lr_schedulers"):
        scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return ckpt
