"""Improved-DDPM UNet (noise + variance interpolation coefficient) executed by libdmme_hip.

Drop-in for the reference's `dmme.models.iddpm.UNet` (src/dmme/models/iddpm.py:125-265): same
constructor arguments and defaults (dropout 0.3, attention on depths 2 and 3), same
`forward(x, c)` returning `(N, 2*in_channels, H, W)`, same state_dict (keys `*.conv1.{0,2}`,
`*.norm`, `*.condition.0` with `2*c_out` rows, `*.conv2.{3|2}`, ...).  The scale-shift ResBlock,
the 4-head attention (including the reference's head-merge order, which mixes samples for
N > 1, models/iddpm.py:38-46) and the layer graph live in the C++ plan (csrc/plan.hip,
DMME_ARCH_IDDPM); this class only selects that architecture."""

from __future__ import annotations

from typing import Sequence

from .ddpm import UNet as _PlanUNet


class UNet(_PlanUNet):
    r"""U-Net for predicting noise in images and learning variance (reference: models/iddpm.py:125-149).

    `num_heads` exposes the head count the reference hard-codes in its ResBlock (4, models/iddpm.py:82)."""

    def __init__(
        self,
        in_channels: int = 3,
        pos_dim: int = 128,
        emb_dim: int = 512,
        num_groups: int = 32,
        dropout: float = 0.3,
        channels_per_depth: Sequence[int] = (128, 256, 256, 256),
        num_blocks: int = 2,
        attention_depths: Sequence[int] = (2, 3),
        precision: str = "fp32",
        num_heads: int = 4,
    ):
        super().__init__(in_channels, pos_dim, emb_dim, num_groups, dropout, channels_per_depth, num_blocks, attention_depths,
                         precision=precision, _arch=1, _num_heads=num_heads)
