"""DDIM strided sampler over the HIP kernels.

Drop-in for the reference's `dmme.diffusion_models.DDIM`
(src/dmme/diffusion_models/ddim.py:15-99).  The update is the one the reference ships
(numerically x - sqrt(1 - abar_tau_i) * eps), computed directly so the
`Normal(mean, 0)` ValueError of the reference at tau_{i-1} = 0 cannot occur."""

from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib
from ..common.noise import gaussian
from ..equations.ddim import linear_tau, quadratic_tau
from .ddpm import DDPM


class DDIM(DDPM):
    tau: Tensor

    def __init__(self, model: nn.Module, timesteps: int = 1000, sub_timesteps: int = 50, tau_schedule: str = "quadratic") -> None:
        super().__init__(model, timesteps)  # start/end are not forwarded, as in the reference (:39)
        self.sub_timesteps = sub_timesteps
        kind = tau_schedule.lower()
        if kind == "linear":
            tau = linear_tau(timesteps, sub_timesteps)
        elif kind == "quadratic":
            tau = quadratic_tau(timesteps, sub_timesteps)
        else:
            raise NotImplementedError
        self.register_buffer("tau", tau, persistent=False)
        ab = self.alpha_bar.reshape(-1).to(torch.float32).cpu()
        self._tau_host = [int(v) for v in tau]
        self._s1 = torch.sqrt(1 - ab).tolist()  # sqrt(1 - abar_t), indexed by t
        self._s2 = torch.sqrt(ab).tolist()      # sqrt(abar_t)
        self._tau_dev: Optional[Tensor] = None

    def _ddim_update(self, x: Tensor, eps: Tensor, i: int) -> Tensor:
        ti, tp = self._tau_host[i], self._tau_host[i - 1]
        _lib.check(_lib.lib().dmme_ddim_step(_lib.ptr(x), _lib.ptr(eps), self._s1[ti], self._s2[tp], x.numel(), _lib.stream_ptr()), "dmme_ddim_step")
        return x

    def sampling_step(self, x_tau_i: Tensor, i: Tensor) -> Tensor:
        r"""x_{tau_{i-1}} from x_{tau_i} (reference: diffusion_models/ddim.py:55-77); i has shape (1,)."""
        if i.numel() != 1:
            raise RuntimeError(f"sampling_step expects an index tensor of shape (1,), got {tuple(i.shape)}")
        idx = int(i.reshape(-1)[0].item())
        eps = self.model(x_tau_i, self.tau[idx].reshape(1))
        x = x_tau_i.detach().to(torch.float32).clone()
        return self._ddim_update(x, eps, idx)

    _chain_kind = _lib.CHAIN_DDIM

    def _chain_tables(self):
        S, tau = self.sub_timesteps, self._tau_host
        rows = [(0.0, 1.0, 0.0, 0.0)] + [(self._s1[tau[i]], self._s2[tau[i - 1]], 0.0, 0.0) for i in range(1, S + 1)]
        return S, rows, list(tau)

    def denoise_once(self, x: Tensor, i: int) -> Tensor:
        i = int(i)
        done = self._once_via_runner(x, i)
        if done is not None:
            return done
        eps = self.model(x, self.tau_tensor(i, x.device))
        out = x.detach().to(torch.float32).clone()
        return self._ddim_update(out, eps, i)

    def tau_tensor(self, i: int, device) -> Tensor:
        if self._tau_dev is None or self._tau_dev.device != torch.device(device):
            self._tau_dev = self.tau.to(device).unsqueeze(1)
        return self._tau_dev[i]

    @torch.no_grad()
    def generate(self, img_size: Tuple[int, int, int, int]) -> Tensor:
        """S-step strided chain (reference: diffusion_models/ddim.py:79-99)"""
        dev = self.beta.device
        x = gaussian(img_size, device=dev)
        runner = self._generate_runner(img_size, dev) if len(img_size) == 4 and not self.model.training else None
        if runner is not None:
            runner.x.copy_(x)
            return runner.run(self.sub_timesteps, self.sub_timesteps).clone()
        for i in range(self.sub_timesteps, 0, -1):
            eps = self.model(x, self.tau_tensor(i, dev))
            self._ddim_update(x, eps, i)
        return x
