"""LitIDDPM: the facade `configs/iddpm/cifar10.yaml` instantiates (reference: src/dmme/lit_modules/iddpm.py:11-58)."""

from __future__ import annotations

from typing import Optional

from torch import nn

from ..diffusion_models import IDDPM
from ..models.iddpm import UNet
from .ddpm import LitDDPM


class LitIDDPM(LitDDPM):
    def __init__(
        self,
        lr: float = 0.0002,
        warmup: int = 5000,
        decay: float = 0.9999,
        diffusion_model: Optional[IDDPM] = None,
        model: Optional[nn.Module] = None,
        timesteps: int = 1000,
        loss_type: str = "hybrid",
        gamma: float = 0.001,
        schedule: str = "cosine",
        offset: float = 0.008,
        start: float = 0.0001,
        end: float = 0.02,
    ):
        if diffusion_model is None:
            if model is None:
                model = UNet()
            diffusion_model = IDDPM(model, timesteps, loss_type, gamma, schedule, offset, start, end)
        super().__init__(lr, warmup, decay, diffusion_model)
