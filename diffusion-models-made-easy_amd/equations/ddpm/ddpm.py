"""Host-side schedule tables of DDPM (tiny, computed once on the CPU in fp32 exactly as
the reference does; the per-pixel math runs in the HIP kernels)."""

from __future__ import annotations

import torch
from torch import Tensor

from ...common.noise import pad


def linear_schedule(timesteps: int, start: float = 0.0001, end: float = 0.02) -> Tensor:
    r"""beta_t for t = 0..T with beta_0 = 0 (reference: equations/ddpm/ddpm.py:9-21)."""
    return pad(torch.linspace(start, end, timesteps))


def sampling_coefficients(beta: Tensor, alpha: Tensor, alpha_bar: Tensor):
    """fp32 per-timestep scalars consumed by dmme_ddpm_step, evaluated with the same
    fp32 torch ops as reverse_process (reference: equations/ddpm/ddpm.py:65-71):
    1/sqrt(alpha_t), beta_t/sqrt(1-abar_t), sqrt(beta_t)."""
    b, a, ab = (v.reshape(-1).to(torch.float32).cpu() for v in (beta, alpha, alpha_bar))
    inv_sqrt_alpha = 1 / torch.sqrt(a)
    eps_coef = b / torch.sqrt(1 - ab)
    sigma = torch.sqrt(b)
    return inv_sqrt_alpha.tolist(), eps_coef.tolist(), sigma.tolist()
