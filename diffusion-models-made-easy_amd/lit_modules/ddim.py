"""LitDDIM: LitDDPM checkpoints + the strided DDIM sampler
(reference: src/dmme/lit_modules/ddim.py:11-45)."""

from __future__ import annotations

from typing import Optional

from torch import nn

from ..diffusion_models import DDIM
from ..models.ddpm import UNet
from .ddpm import LitDDPM


class LitDDIM(LitDDPM):
    def __init__(
        self,
        lr: float = 2e-4,
        warmup: int = 5000,
        decay: float = 0.9999,
        diffusion_model: Optional[DDIM] = None,
        model: Optional[nn.Module] = None,
        timesteps: int = 1000,
        sample_steps: int = 50,
        tau_schedule: str = "quadratic",
    ):
        if diffusion_model is None:
            if model is None:
                model = UNet()
            diffusion_model = DDIM(model, timesteps, sample_steps, tau_schedule)
        super().__init__(lr, warmup, decay, diffusion_model)
