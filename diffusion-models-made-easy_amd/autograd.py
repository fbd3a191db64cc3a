"""Autograd glue for the training path: torch.autograd only sees two opaque nodes (the UNet
and the MSE loss); everything inside them is HIP (dmme_unet_forward / dmme_unet_backward /
dmme_mse_loss).  Parameter gradients are accumulated by the library straight into the
model's flat fp32 gradient buffer (every `param.grad` is a view of it)."""

from __future__ import annotations

import torch
from torch import Tensor

from . import _lib


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, anchor: Tensor, model, c: Tensor):
        y, saved = model._forward_impl(x, c, want_ctx=True)
        ctx.model = model
        ctx.saved = saved
        ctx.want_dx = bool(x.requires_grad)
        ctx.x_dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        # parameter gradients: accumulated by the library into model.flat_grad() (param.grad views); the gradient with respect
        # to the input image (guidance, sensitivity tests) is returned to autograd when x required it
        dx = ctx.model._backward_impl(ctx.saved, dy, want_dx=ctx.want_dx)
        ctx.saved = None
        return (dx.to(ctx.x_dtype) if dx is not None else None), None, None, None


def unet_apply(model, x: Tensor, c: Tensor) -> Tensor:
    """eps = UNet(x, c) with a HIP backward; `anchor` only ties the node into the graph."""
    anchor = next(p for p in model.parameters() if p.requires_grad)
    return _UNetFunction.apply(x, anchor, model, c)


class _MSEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eps: Tensor, target: Tensor):
        e = eps.detach().to(torch.float32).contiguous()
        tg = target.detach().to(torch.float32).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=e.device)
        scratch = torch.empty(1024, dtype=torch.float32, device=e.device)
        d_eps = torch.empty_like(e) if eps.requires_grad else None
        _lib.check(_lib.lib().dmme_mse_loss(_lib.ptr(e), _lib.ptr(tg), e.numel(), _lib.ptr(loss), _lib.ptr(d_eps), 1.0, _lib.ptr(scratch), _lib.stream_ptr()), "dmme_mse_loss")
        ctx.d_eps = d_eps
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        d = ctx.d_eps
        ctx.d_eps = None
        return (d * grad_out if d is not None else None), None


def mse_loss_apply(eps: Tensor, target: Tensor) -> Tensor:
    """simple_loss (reference: equations/ddpm/losses.py:5-13): mean((target - eps)^2) over all elements."""
    return _MSEFunction.apply(eps, target)


class _IDDPMLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_out: Tensor, x_t: Tensor, x_0: Tensor, target: Tensor, t: Tensor, coef: Tensor, w_simple: float, w_vlb: float):
        o = model_out.detach().to(torch.float32).contiguous()
        B = o.size(0)
        loss = torch.empty(3, dtype=torch.float32, device=o.device)
        scratch = torch.empty(1024, dtype=torch.float32, device=o.device)
        d_out = torch.empty_like(o) if model_out.requires_grad else None
        _lib.check(
            _lib.lib().dmme_iddpm_loss(_lib.ptr(o), _lib.ptr(x_t), _lib.ptr(x_0), _lib.ptr(target), _lib.ptr(t), _lib.ptr(coef), B, x_t[0].numel(),
                                       w_simple, w_vlb, _lib.ptr(loss), _lib.ptr(d_out), 1.0, _lib.ptr(scratch), _lib.stream_ptr()),
            "dmme_iddpm_loss",
        )
        ctx.d_out = d_out
        ctx.parts = loss
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        d = ctx.d_out
        ctx.d_out = None
        return (d * grad_out if d is not None else None), None, None, None, None, None, None, None


def iddpm_loss_apply(model_out: Tensor, x_t: Tensor, x_0: Tensor, target: Tensor, t: Tensor, coef: Tensor, w_simple: float, w_vlb: float) -> Tensor:
    """w_simple * L_simple + w_vlb * L_vlb (reference: diffusion_models/iddpm.py:92-116, equations/iddpm/losses.py:40-98)."""
    return _IDDPMLossFunction.apply(model_out, x_t, x_0, target, t, coef.to(device=model_out.device, dtype=torch.float32).contiguous(), w_simple, w_vlb)
