"""Host-side tables of Improved DDPM (tiny, computed once on the CPU in fp32 with the same torch ops
as the reference; the per-pixel math runs in the HIP kernels dmme_iddpm_step / dmme_iddpm_loss)."""

from __future__ import annotations

import math

import torch
from torch import Tensor


def cosine_schedule(timesteps: int = 4000, offset: float = 0.008) -> Tensor:
    r"""alpha_bar_t = f(t)/f(0), f(t) = cos^2((t/T + s)/(1 + s) * pi/2), t = 0..T
    (reference: equations/iddpm/iddpm.py:6-20)."""

    def f(t):
        return torch.cos((t / timesteps + offset) / (1 + offset) * math.pi / 2) ** 2

    t = torch.arange(0, timesteps + 1)
    zero = torch.tensor([0], dtype=torch.float32)
    return f(t) / f(zero)


def interpolate_variance(v: Tensor, beta_t: Tensor, beta_tilde_t: Tensor) -> Tensor:
    r"""Sigma = exp(v log beta_t + (1 - v) log beta~_t) (reference: equations/iddpm/losses.py:34-37).
    Host/torch form kept for API parity; the sampler and loss kernels evaluate it per pixel."""
    return torch.exp(v * torch.log(beta_t) + (1 - v) * torch.log(beta_tilde_t.clamp(1e-12)))


def process_coefficients(beta: Tensor, alpha: Tensor, alpha_bar: Tensor) -> Tensor:
    """(T+1, 8) fp32 table consumed by dmme_iddpm_step / dmme_iddpm_loss (layout in include/dmme_hip.h), evaluated
    with the reference's fp32 torch expressions: reverse_process (equations/ddpm/ddpm.py:65-71), beta~
    (diffusion_models/iddpm.py:161), interpolate_variance's logs (equations/iddpm/losses.py:34-37) and
    true_reverse_process (equations/iddpm/losses.py:23-31).  Row 0 (t = 0 is never used) is zero."""
    b, a, ab = (v.reshape(-1).to(torch.float32).cpu() for v in (beta, alpha, alpha_bar))
    T1 = b.numel()
    tab = torch.zeros(T1, 8, dtype=torch.float32)
    bt, at, abt, abp = b[1:], a[1:], ab[1:], ab[:-1]
    beta_tilde = (1 - abp) / (1 - abt) * bt
    tab[1:, 0] = 1 / torch.sqrt(at)
    tab[1:, 1] = bt / torch.sqrt(1 - abt)
    tab[1:, 2] = torch.log(bt)
    tab[1:, 3] = torch.log(beta_tilde.clamp(1e-12))
    tab[1:, 4] = torch.sqrt(abp) * bt / (1 - abt)
    tab[1:, 5] = torch.sqrt(at) * (1 - abp) / (1 - abt)
    tab[1:, 6] = torch.sqrt(beta_tilde)
    return tab
