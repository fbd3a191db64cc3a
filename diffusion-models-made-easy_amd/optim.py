"""Optimiser side of the training step on the flat parameter buffer.

Replaces `torch.optim.Adam(lr)` + `gradient_clip_val` + the EMA callback of the reference
(lit_modules/ddpm.py:127-141, configs/ddpm/cifar10.yaml:24, callbacks/ema.py:169-176) by one
gradient-norm reduction and ONE fused pass over the flat fp32 buffers (dmme_grad_norm,
dmme_adam_step): clip -> Adam -> EMA.  Works on the parameters of dmme_amd UNets only."""

from __future__ import annotations

import contextlib

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0, ema_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm, ema_decay=ema_decay)
        super().__init__(params, defaults)
        self._step_count = 0  # what the reference's WarmupLR keys on (lr_scheduler/warmup.py:11)
        self._owners = []
        seen = set()
        for group in self.param_groups:
            for p in group["params"]:
                owner = getattr(p, "_dmme_owner", None)
                owner = owner() if owner is not None else None
                if owner is None:
                    raise ValueError("FusedAdam only optimises parameters of dmme_amd.UNet modules (flat-buffer views)")
                if id(owner) not in seen:
                    seen.add(id(owner))
                    self._owners.append(owner)
        self._flat_state = {}
        self.last_grad_norm = None

    def zero_grad(self, set_to_none: bool = False):
        for m in self._owners:
            if m._flat_grad is not None:
                m._flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._step_count += 1
        g0 = self.param_groups[0]
        lr, (b1, b2), eps = float(g0["lr"]), g0["betas"], float(g0["eps"])
        max_norm, decay = float(g0["max_grad_norm"] or 0.0), float(g0["ema_decay"] or 0.0)
        lib = _lib.lib()
        for m in self._owners:
            flat = m.flat_parameters()
            grad = m.flat_grad()
            st = self._flat_state.get(id(m))
            if st is None or st["m"].device != flat.device:
                st = {"m": torch.zeros_like(flat), "v": torch.zeros_like(flat), "norm": torch.zeros(1, device=flat.device),
                      "scratch": torch.empty(1024, device=flat.device), "ema": flat.clone() if decay > 0 else None}
                self._flat_state[id(m)] = st
            norm_ptr = _lib.ptr(None)
            if max_norm > 0:
                _lib.check(lib.dmme_grad_norm(_lib.ptr(grad), grad.numel(), _lib.ptr(st["norm"]), _lib.ptr(st["scratch"]), _lib.stream_ptr()), "dmme_grad_norm")
                norm_ptr = _lib.ptr(st["norm"])
                self.last_grad_norm = st["norm"]
            _lib.check(
                lib.dmme_adam_step(_lib.ptr(flat), _lib.ptr(grad), _lib.ptr(st["m"]), _lib.ptr(st["v"]), _lib.ptr(st["ema"]), flat.numel(), lr, b1, b2, eps,
                                   self._step_count, norm_ptr, max_norm, decay, _lib.stream_ptr()),
                "dmme_adam_step",
            )
            m.mark_params_updated()
        return loss

    def ema_parameters(self, model):
        st = self._flat_state.get(id(model))
        return None if st is None else st["ema"]

    @contextlib.contextmanager
    def swap_ema(self, model):
        """Evaluate / sample with the EMA weights: inside the block the model's flat parameter buffer holds the EMA
        copy, the live weights come back on exit (the reference's EMA callback swaps them the same way around
        validation and image generation, callbacks/ema.py:298-308).  Two flat-buffer copies each way; the packed
        kernel-layout weights are rebuilt on the next forward because the buffer's version changed."""
        ema = self.ema_parameters(model)
        if ema is None:
            raise RuntimeError("swap_ema: no EMA copy yet (ema_decay == 0 or no optimiser step taken)")
        flat = model.flat_parameters()
        with torch.no_grad():
            live = flat.clone()
            flat.copy_(ema)
            model.mark_params_updated()
        try:
            yield model
        finally:
            with torch.no_grad():
                flat.copy_(live)
                model.mark_params_updated()
