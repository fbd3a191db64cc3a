"""LitDDPM: the training / sampling facade `configs/ddpm/cifar10.yaml` instantiates.

Same constructor and methods as the reference's LightningModule
(src/dmme/lit_modules/ddpm.py:21-141).  pytorch_lightning is optional: when it is
importable the class derives from pl.LightningModule, otherwise from nn.Module and the
bundled runner (dmme_amd.trainer) drives it.  FID / Inception metrics are out of scope
(SURVEY 2.1 #6)."""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl

    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module

from ..diffusion_models import DDPM
from ..lr_scheduler import WarmupLR
from ..models.ddpm import UNet


class LitDDPM(_Base):
    def __init__(
        self,
        lr: float = 2e-4,
        warmup: int = 5000,
        decay: float = 0.9999,
        diffusion_model: Optional[DDPM] = None,
        model: Optional[nn.Module] = None,
        timesteps: int = 1000,
    ) -> None:
        super().__init__()
        self.lr = lr
        self.warmup = warmup
        self.decay = decay
        if diffusion_model is None:
            if model is None:
                model = UNet()
            diffusion_model = DDPM(model, timesteps)
        self.diffusion_model = diffusion_model

    def forward(self, x_t: Tensor, t: int):
        r"""denoise once: x_t -> x_{t-1} (reference: lit_modules/ddpm.py:65-79)"""
        if hasattr(self.diffusion_model, "denoise_once") and not isinstance(t, torch.Tensor):
            return self.diffusion_model.denoise_once(x_t, t)  # resident timestep table: no host-to-device copy per step
        timestep = torch.as_tensor(t, device=x_t.device).reshape(1)
        return self.diffusion_model.sampling_step(x_t, timestep)

    def training_step(self, batch, batch_idx):
        r"""L_simple on batch[0] (reference: lit_modules/ddpm.py:81-89)"""
        x_0: Tensor = batch[0]
        loss = self.diffusion_model.training_step(x_0)
        if hasattr(self, "log") and _Base is not nn.Module:
            self.log("train/loss", loss)
        return loss

    def generate(self, img_size):
        return self.diffusion_model.generate(img_size=img_size)

    def configure_optimizers(self):
        """Adam(lr) + per-step linear warm-up (reference: lit_modules/ddpm.py:127-135)"""
        from ..optim import FusedAdam

        optimizer = FusedAdam(self.diffusion_model.parameters(), lr=self.lr, ema_decay=self.decay)
        scheduler = {"scheduler": WarmupLR(optimizer, self.warmup), "interval": "step", "frequency": 1}
        return [optimizer], [scheduler]
