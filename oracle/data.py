"""CPU restatement of the reference's CIFAR10 sample transform (test infrastructure only).

The reference composes `[*augs, torchvision.transforms.ToTensor(), norm]` (src/dmme/data_modules/cifar10.py:39-44).
torchvision is a third-party dependency that is absent from /root/reference and from this image (unpinned in
setup.py / requirements.txt); its published ToTensor maps an HWC uint8 image to CHW float32 `x / 255`, and
RandomHorizontalFlip mirrors the width axis.  `norm` is the reference's own `(x - 0.5) * 2` (common/norm.py:4-6),
pinned by tests/golden/data.npz (all 256 byte values through the reference's function)."""

from __future__ import annotations

import torch


def norm(x: torch.Tensor) -> torch.Tensor:
    return (x - 0.5) * 2


def image_batch(data_u8: torch.Tensor, idx: torch.Tensor, flip=None) -> torch.Tensor:
    """data_u8 (N, C, H, W) uint8, idx (B,), flip (B,) bool or None -> (B, C, H, W) fp32 in [-1, 1]"""
    x = data_u8[idx]
    if flip is not None:
        x = torch.where(flip.reshape(-1, 1, 1, 1).bool(), x.flip(-1), x)
    return norm(x.to(torch.float32).div(255))
