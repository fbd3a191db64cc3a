"""Autograd glue for the training path (the HIP backward is attached here)."""

from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


def unet_apply(model, x: Tensor, c: Tensor) -> Tensor:
    raise NotImplementedError(
        "the HIP backward of the UNet is not built yet: run the model under torch.no_grad() "
        "(sampling / inference) or call .requires_grad_(False) on it"
    )


def mse_loss_apply(eps: Tensor, target: Tensor) -> Tensor:
    """simple_loss (reference: equations/ddpm/losses.py:5-13) through dmme_mse_loss."""
    e = eps.detach().to(torch.float32).contiguous()
    tg = target.detach().to(torch.float32).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=e.device)
    scratch = torch.empty(1024, dtype=torch.float32, device=e.device)
    _lib.check(_lib.lib().dmme_mse_loss(_lib.ptr(e), _lib.ptr(tg), e.numel(), _lib.ptr(loss), _lib.ptr(None), 1.0, _lib.ptr(scratch), _lib.stream_ptr()), "dmme_mse_loss")
    return loss[0]
