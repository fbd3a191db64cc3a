"""DDIM sub-sequence tables (reference: equations/ddim/ddim.py:9-34); int64, index i -> tau_i."""

import torch
from torch import Tensor


def linear_tau(timesteps: int, sub_timesteps: int) -> Tensor:
    i = torch.arange(0, sub_timesteps + 1)
    return torch.round((timesteps / sub_timesteps) * i).long()


def quadratic_tau(timesteps: int, sub_timesteps: int) -> Tensor:
    i = torch.arange(0, sub_timesteps + 1)
    return torch.round((timesteps / (sub_timesteps**2)) * i**2).long()
