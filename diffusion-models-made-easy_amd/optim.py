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
    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0, ema_decay=0.0, amp=True, init_scale=65536.0,
                 growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        """`amp`: dynamic loss scaling whenever the model computes in IEEE half (precision="fp16") - the reference's `precision: 16`
        step under torch.cuda.amp.GradScaler (configs/ddpm/cifar10.yaml:53,66), with GradScaler's default constants; no effect on
        bf16 / fp32 models.  The scaler's state lives on the device (include/dmme_hip.h: dmme_amp_*): no read-back per step."""
        defaults = dict(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm, ema_decay=ema_decay)
        super().__init__(params, defaults)
        self._amp_cfg = dict(init_scale=float(init_scale), growth_factor=float(growth_factor), backoff_factor=float(backoff_factor),
                             growth_interval=int(growth_interval)) if amp else None
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
        for m in self._owners:
            m._amp = self._amp_cfg
        self._flat_state = {}
        # device scalar: norm of the gradient BUFFER of the last step - multiply by `grad_scale` of that step (rank sums under the
        # folded data-parallel mean) and divide by the loss scale (amp) to get the norm of the mean gradient: `grad_norm()`
        self.last_grad_norm = None
        self._last_norm_factor = 1.0
        # multiplies the gradient inside the fused pass: 1 / world when the data-parallel exchange left rank SUMS in the flat buffer
        # (distributed.py folds the mean's divide into this pass instead of a separate sweep over 130 MB); consumed by ONE step
        self.grad_scale = 1.0

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
            amp = m.amp_state(flat.device) if hasattr(m, "amp_state") else None
            if max_norm > 0 or amp is not None:
                _lib.check(lib.dmme_grad_norm(_lib.ptr(grad), grad.numel(), _lib.ptr(st["norm"]), _lib.ptr(st["scratch"]), _lib.stream_ptr()), "dmme_grad_norm")
                norm_ptr = _lib.ptr(st["norm"])
                self.last_grad_norm = st["norm"]
                self._last_norm_factor = float(self.grad_scale)
                self._last_amp = amp
            if amp is not None:  # half precision: unscale + inf / NaN skip + scale update inside the fused pass
                c = self._amp_cfg
                _lib.check(
                    lib.dmme_adam_step_amp(_lib.ptr(flat), _lib.ptr(grad), _lib.ptr(st["m"]), _lib.ptr(st["v"]), _lib.ptr(st["ema"]), flat.numel(), lr, b1, b2, eps,
                                           norm_ptr, max_norm, decay, float(self.grad_scale), _lib.ptr(amp), c["growth_factor"], c["backoff_factor"],
                                           c["growth_interval"], _lib.stream_ptr()),
                    "dmme_adam_step_amp",
                )
                m.mark_params_updated()
                continue
            _lib.check(
                lib.dmme_adam_step(_lib.ptr(flat), _lib.ptr(grad), _lib.ptr(st["m"]), _lib.ptr(st["v"]), _lib.ptr(st["ema"]), flat.numel(), lr, b1, b2, eps,
                                   self._step_count, norm_ptr, max_norm, decay, float(self.grad_scale), _lib.stream_ptr()),
                "dmme_adam_step",
            )
            m.mark_params_updated()
        self.grad_scale = 1.0
        return loss

    def grad_norm(self) -> float:
        """L2 norm of the (mean, unscaled) gradient of the last step - what torch's clip_grad_norm_ would have returned under DDP.
        The flat buffer itself holds rank SUMS under the folded data-parallel mean and S x gradient under loss scaling; this accessor
        undoes both (synchronises: reads device scalars)."""
        if self.last_grad_norm is None:
            return float("nan")
        v = float(self.last_grad_norm.item()) * self._last_norm_factor
        amp = getattr(self, "_last_amp", None)
        if amp is not None:
            v /= float(amp[0].item())  # (the scale AFTER the step's update: off by the growth / backoff factor on the steps that change it)
        return v

    def loss_scale(self, model=None) -> float:
        """current loss scale (1.0 when loss scaling is off); synchronises"""
        m = model if model is not None else self._owners[0]
        amp = m.amp_state() if hasattr(m, "amp_state") else None
        return 1.0 if amp is None else float(amp[0].item())

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

    # ------------------------------------------------------------------ checkpoint layout of the reference's EMAOptimizer
    def _param_slices(self, model):
        """(param, flat offset, numel) for every optimised parameter of `model`, in optimiser order"""
        flat = model.flat_parameters()
        base = flat.data_ptr()
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                owner = p._dmme_owner()
                if owner is model:
                    out.append((p, (p.data_ptr() - base) // 4, p.numel()))
        return out

    def state_dict(self):
        """Same layout as the reference's `EMAOptimizer.state_dict()` (callbacks/ema.py:339-359): {"opt": the wrapped
        torch.optim.Adam's state_dict (per-parameter `step` / `exp_avg` / `exp_avg_sq`, loadable by a stock
        torch.optim.Adam), "ema": tuple of per-parameter EMA tensors, "current_step", "decay", "every_n_steps", "device"}."""
        g0 = self.param_groups[0]
        state, ema, index = {}, [], 0
        device = None
        for m in self._owners:
            st = self._flat_state.get(id(m))
            flat = m.flat_parameters()
            device = flat.device
            for p, off, n in self._param_slices(m):
                if st is not None:
                    state[index] = {"step": torch.tensor(float(self._step_count)), "exp_avg": st["m"][off : off + n].view(p.shape).clone(),
                                    "exp_avg_sq": st["v"][off : off + n].view(p.shape).clone()}
                src = st["ema"] if st is not None and st["ema"] is not None else flat
                ema.append(src[off : off + n].view(p.shape).clone())
                index += 1
        group = {"lr": g0["lr"], "betas": tuple(g0["betas"]), "eps": g0["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(index))}
        if "initial_lr" in g0:
            group["initial_lr"] = g0["initial_lr"]
        out = {"opt": {"state": state, "param_groups": [group]}, "ema": tuple(ema), "current_step": self._step_count,
               "decay": float(g0["ema_decay"] or 0.0), "every_n_steps": 1, "device": device,
               "max_grad_norm": float(g0["max_grad_norm"] or 0.0)}
        # the device-resident loss-scaling state of half-precision models (what GradScaler.state_dict() is to a Lightning checkpoint):
        # scale, growth tracker, the optimiser-step count the fused pass uses for Adam's bias correction (skipped steps do not count,
        # so it differs from `current_step`), found_inf, steps skipped.  Without it a resumed fp16 run restarts at S = 65536 and t = 1
        # beside warm moments: updates damped by sqrt(1 - b2^t) / (1 - b1^t) for thousands of steps.
        amp = [m._amp_state.detach().cpu().tolist() if getattr(m, "_amp_state", None) is not None else None for m in self._owners]
        if any(a is not None for a in amp):
            out["amp_state"] = amp
        return out

    def load_state_dict(self, state_dict):
        """accepts the layout above (written by this class or by the reference's EMAOptimizer around torch.optim.Adam)"""
        opt = state_dict["opt"]
        pg = opt["param_groups"][0]
        for g in self.param_groups:
            g["lr"], g["betas"], g["eps"] = pg["lr"], tuple(pg["betas"]), pg["eps"]
            if "initial_lr" in pg:
                g["initial_lr"] = pg["initial_lr"]
            g["ema_decay"] = state_dict.get("decay", g["ema_decay"])
            if "max_grad_norm" in state_dict:
                g["max_grad_norm"] = state_dict["max_grad_norm"]
        self._step_count = int(state_dict.get("current_step", 0))
        ema_list = list(state_dict.get("ema") or [])
        index = 0
        with torch.no_grad():
            for m in self._owners:
                flat = m.flat_parameters()
                decay = float(self.param_groups[0]["ema_decay"] or 0.0)
                st = {"m": torch.zeros_like(flat), "v": torch.zeros_like(flat), "norm": torch.zeros(1, device=flat.device),
                      "scratch": torch.empty(1024, device=flat.device), "ema": flat.clone() if (decay > 0 or ema_list) else None}
                for p, off, n in self._param_slices(m):
                    ps = opt["state"].get(index)
                    if ps is not None:
                        st["m"][off : off + n].copy_(ps["exp_avg"].reshape(-1))
                        st["v"][off : off + n].copy_(ps["exp_avg_sq"].reshape(-1))
                        self._step_count = max(self._step_count, int(float(ps["step"])))
                    if ema_list:
                        st["ema"][off : off + n].copy_(ema_list[index].reshape(-1))
                    index += 1
                self._flat_state[id(m)] = st
            amp_saved = state_dict.get("amp_state")
            for k, m in enumerate(self._owners):
                amp = m.amp_state(m.flat_parameters().device) if hasattr(m, "amp_state") else None
                if amp is None:
                    continue
                if amp_saved is not None and k < len(amp_saved) and amp_saved[k] is not None:
                    amp.copy_(torch.tensor(amp_saved[k], dtype=torch.float32))
                else:
                    # a checkpoint without the scaler's state (written by a bf16 / fp32 run, or by the reference): keep the initial
                    # scale, but Adam's bias correction continues from the loaded step count instead of restarting at 1
                    amp[2] = float(self._step_count)
