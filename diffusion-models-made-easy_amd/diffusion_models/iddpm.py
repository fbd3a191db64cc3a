"""Improved DDPM (cosine schedule, learned variance, hybrid loss) over the HIP kernels.

Drop-in for the reference's `dmme.diffusion_models.IDDPM` (src/dmme/diffusion_models/iddpm.py:16-164): same
constructor, buffers and methods.  The model returns (N, 2C, H, W) = (eps, v); the variance interpolation, the
reverse update, L_simple, L_vlb (discrete NLL at t == 1, KL elsewhere, stop-gradient on eps) and their gradient
w.r.t. the network output are fused HIP kernels (dmme_iddpm_step, dmme_iddpm_loss)."""

from __future__ import annotations

from collections import namedtuple
from typing import Optional

import torch
from torch import Tensor, nn

from .. import _lib
from ..common.noise import gaussian_like, pad, uniform_int
from ..equations.iddpm import cosine_schedule, interpolate_variance, process_coefficients
from .ddpm import DDPM

NoiseVariance = namedtuple("NoiseVariance", ["noise", "variance"])


class IDDPM(DDPM):
    def __init__(
        self,
        model: nn.Module,
        timesteps: int = 1000,
        loss_type="hybrid",
        gamma=0.001,
        schedule: str = "cosine",
        offset=0.008,
        start: float = 0.0001,
        end: float = 0.02,
    ) -> None:
        super().__init__(model, timesteps, start, end)
        self.loss_type = loss_type
        self.gamma = gamma
        if schedule == "cosine":
            alpha_bar = cosine_schedule(timesteps, offset).reshape(-1, 1, 1, 1)
            # clip to prevent singularities near t = T; the front pad is 1, not 0 (reference :51-52)
            beta = torch.clip(1 - alpha_bar[1:] / alpha_bar[:-1], 0, 0.999)
            beta = pad(beta, value=1)
            alpha = 1 - beta
            self.register_buffer("beta", beta, persistent=False)
            self.register_buffer("alpha", alpha, persistent=False)
            self.register_buffer("alpha_bar", alpha_bar, persistent=False)
            self.register_buffer("_sqrt_alpha_bar", torch.sqrt(alpha_bar).reshape(-1).contiguous(), persistent=False)
            self.register_buffer("_sqrt_one_minus_alpha_bar", torch.sqrt(1 - alpha_bar).reshape(-1).contiguous(), persistent=False)
        elif schedule != "linear":
            raise NotImplementedError
        coef = process_coefficients(self.beta, self.alpha, self.alpha_bar)
        self.register_buffer("_coef", coef.contiguous(), persistent=False)  # device table of the loss kernel
        self._coef_host = coef.tolist()                                      # python floats of the sampler kernel

    # ------------------------------------------------------------------ training
    def training_step(self, x_0: Tensor, t: Optional[Tensor] = None, noise: Optional[Tensor] = None):
        r"""hybrid loss L_simple + gamma L_vlb, or L_vlb alone (reference: diffusion_models/iddpm.py:62-116).
        As in the reference, any other `loss_type` (e.g. "simple") falls through and returns None.
        `t` / `noise` may be injected for parity tests."""
        from ..autograd import iddpm_loss_apply

        B = x_0.size(0)
        if t is None:
            t = uniform_int(1, self.timesteps, B, device=x_0.device)
        if noise is None:
            noise = gaussian_like(x_0)
        x0 = x_0.detach().to(torch.float32).contiguous()
        z = noise.detach().to(torch.float32).contiguous()
        t = t.to(device=x_0.device, dtype=torch.int64).contiguous()
        x_t = torch.empty_like(x0)
        target = torch.empty_like(x0)
        _lib.check(
            _lib.lib().dmme_q_sample(_lib.ptr(x0), _lib.ptr(z), _lib.ptr(self._sqrt_alpha_bar), _lib.ptr(self._sqrt_one_minus_alpha_bar), _lib.ptr(t), B, x0[0].numel(), _lib.ptr(x_t), _lib.ptr(target), _lib.stream_ptr()),
            "dmme_q_sample",
        )
        model_output = self.model(x_t, t)
        if self.loss_type == "vlb":
            return iddpm_loss_apply(model_output, x_t, x0, target, t, self._coef, 0.0, 1.0)
        if self.loss_type == "hybrid":
            return iddpm_loss_apply(model_output, x_t, x0, target, t, self._coef, 1.0, float(self.gamma))
        return None

    # ------------------------------------------------------------------ sampling
    _chain_kind = _lib.CHAIN_IDDPM

    def _chain_tables(self):
        T = self.timesteps
        finite = lambda v: v if v == v and abs(v) != float("inf") else 0.0  # row 0 is never stepped from
        rows = [tuple(finite(v) for v in self._coef_host[t][:4]) for t in range(T + 1)]
        return T, rows, list(range(T + 1))

    def _reverse_update(self, x_t: Tensor, model_output: Tensor, t: int, noise: Optional[Tensor]) -> Tensor:
        if noise is None:
            noise = gaussian_like(x_t)  # drawn even when t == 1, then unused (reference :144-149)
        c = self._coef_host[t]
        B = x_t.size(0)
        _lib.check(
            _lib.lib().dmme_iddpm_step(_lib.ptr(x_t), _lib.ptr(model_output), _lib.ptr(noise), c[0], c[1], c[2], c[3], int(t != 1), B, x_t[0].numel(), _lib.stream_ptr()),
            "dmme_iddpm_step",
        )
        return x_t

    def forward_model(self, x_t: Tensor, t: Tensor, beta_t: Tensor, alpha_bar_t: Tensor, alpha_bar_t_minus_one: Tensor) -> NoiseVariance:
        """model call + variance interpolation as tensors (reference: diffusion_models/iddpm.py:152-164); API parity only --
        sampling_step / training_step use the fused kernels instead."""
        noise_in_x_t, v = self.model(x_t, t).chunk(2, dim=1)
        beta_tilde_t = (1 - alpha_bar_t_minus_one) / (1 - alpha_bar_t) * beta_t
        return NoiseVariance(noise_in_x_t, interpolate_variance(v, beta_t, beta_tilde_t))
