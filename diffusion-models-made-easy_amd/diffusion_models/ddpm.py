"""DDPM training / sampling driver over the HIP kernels.

Drop-in for the reference's `dmme.diffusion_models.DDPM`
(src/dmme/diffusion_models/ddpm.py:15-144): same constructor, buffers
(beta / alpha / alpha_bar, shape (T+1,1,1,1), non-persistent) and methods.  The host
loop stays in Python as in the reference; every per-pixel operation (forward noising,
the reverse update, the MSE loss, the normal draws) is one fused HIP kernel."""

from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib
from ..common.noise import gaussian, gaussian_like, uniform_int
from ..equations.ddpm import linear_schedule, sampling_coefficients


class DDPM(nn.Module):
    beta: Tensor
    alpha: Tensor
    alpha_bar: Tensor

    def __init__(self, model: nn.Module, timesteps: int = 1000, start: float = 0.0001, end: float = 0.02) -> None:
        super().__init__()
        self.model = model
        self.timesteps = timesteps

        beta = linear_schedule(timesteps, start, end).reshape(-1, 1, 1, 1)
        alpha = 1 - beta
        alpha_bar = torch.cumprod(alpha, dim=0)  # alpha[0] = 1
        self.register_buffer("beta", beta, persistent=False)
        self.register_buffer("alpha", alpha, persistent=False)
        self.register_buffer("alpha_bar", alpha_bar, persistent=False)
        # fp32 tables for the forward-noising kernel, evaluated like forward_process does
        # (reference: equations/ddpm/ddpm.py:36-39); device-resident, not part of the state_dict
        self.register_buffer("_sqrt_alpha_bar", torch.sqrt(alpha_bar).reshape(-1).contiguous(), persistent=False)
        self.register_buffer("_sqrt_one_minus_alpha_bar", torch.sqrt(1 - alpha_bar).reshape(-1).contiguous(), persistent=False)
        # host copies of the per-step scalars of the reverse update (python floats)
        self._c1, self._c2, self._sigma = sampling_coefficients(beta, alpha, alpha_bar)
        self._all_t: Optional[Tensor] = None

    # ------------------------------------------------------------------ training
    def training_step(self, x_0: Tensor, t: Optional[Tensor] = None, noise: Optional[Tensor] = None) -> Tensor:
        r"""L_simple for one batch (reference: diffusion_models/ddpm.py:53-81).

        `t` / `noise` may be injected for parity tests; by default t ~ randint(1, T)
        (never T itself, as in the reference) and noise ~ N(0, I)."""
        from ..autograd import mse_loss_apply

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
        noise_in_x_t = self.model(x_t, t)
        return mse_loss_apply(noise_in_x_t, target)

    # ------------------------------------------------------------------ sampling
    def _reverse_update(self, x_t: Tensor, eps: Tensor, t: int, noise: Optional[Tensor]) -> Tensor:
        if noise is None:
            noise = gaussian_like(x_t)  # drawn even when t == 1, then unused (reference :107-110)
        _lib.check(
            _lib.lib().dmme_ddpm_step(_lib.ptr(x_t), _lib.ptr(eps), _lib.ptr(noise), self._c1[t], self._c2[t], self._sigma[t], int(t != 1), x_t.numel(), _lib.stream_ptr()),
            "dmme_ddpm_step",
        )
        return x_t

    def sampling_step(self, x_t: Tensor, t: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        r"""one draw from p_theta(x_{t-1} | x_t) (reference: diffusion_models/ddpm.py:83-111).

        As in the reference only a timestep tensor of shape (1,) is valid (the reference's
        `torch.where(t == 1, ...)` broadcasts t against the last image dimension)."""
        if t.numel() != 1:
            raise RuntimeError(f"sampling_step expects a timestep tensor of shape (1,), got {tuple(t.shape)}")
        step = int(t.reshape(-1)[0].item())
        eps = self.model(x_t, t.reshape(1))
        x = x_t.detach().to(torch.float32).clone()
        return self._reverse_update(x, eps, step, noise)

    @torch.no_grad()
    def generate(self, img_size: Tuple[int, int, int, int]) -> Tensor:
        """run the full T-step chain from pure noise (reference: diffusion_models/ddpm.py:113-133)"""
        dev = self.beta.device
        x_t = gaussian(img_size, device=dev)
        if self._all_t is None or self._all_t.device != dev or self._all_t.numel() != self.timesteps + 1:
            self._all_t = torch.arange(0, self.timesteps + 1, device=dev).unsqueeze(1)
        # small batches are launch/latency bound: replay the forward from a hipGraph (no gain at B >= 128)
        graphed = hasattr(self.model, "graphed_forward") and not self.model.training and int(img_size[0]) <= 64
        t_buf = self._all_t[self.timesteps].clone() if graphed else None
        for t in range(self.timesteps, 0, -1):
            if graphed:  # one hipGraph replay per step instead of ~160 launches (small batches are launch-bound)
                t_buf.copy_(self._all_t[t])
                eps = self.model.graphed_forward(x_t, t_buf)
            else:
                eps = self.model(x_t, self._all_t[t])
            self._reverse_update(x_t, eps, t, None)
        return x_t

    def forward(self, x: Tensor, t: Tensor) -> Tensor:
        return self.model(x, t)
