"""Functional CPU restatement of the reference DDPM / DDIM process (test infrastructure only).

Random draws are always *injected* (z, t) so results are comparable bit-for-bit
across implementations; citations are relative to /root/reference.
"""

from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .unet import UNetConfig, unet_forward

Tensor = torch.Tensor


def linear_beta(timesteps: int, start: float = 1e-4, end: float = 0.02) -> Tensor:
    """linear_schedule (src/dmme/equations/ddpm/ddpm.py:9-21) + pad (common/noise.py:19-23):
    beta[0] = 0, beta[1..T] = linspace(start, end, T); the index IS the timestep."""
    return torch.cat([torch.zeros(1), torch.linspace(start, end, timesteps)])


def alpha_tables(beta: Tensor) -> Tuple[Tensor, Tensor]:
    """DDPM.__init__ (src/dmme/diffusion_models/ddpm.py:44-47): alpha = 1-beta, alpha_bar = cumprod."""
    alpha = 1 - beta
    return alpha, torch.cumprod(alpha, dim=0)


def tau_table(timesteps: int, sub_timesteps: int, schedule: str = "quadratic") -> Tensor:
    """linear_tau / quadratic_tau (src/dmme/equations/ddim/ddim.py:9-34); torch.round is
    round-half-to-even; unknown schedule -> NotImplementedError (diffusion_models/ddim.py:50-51)."""
    idx = torch.arange(0, sub_timesteps + 1)
    s = schedule.lower()
    if s == "linear":
        tau = torch.round((timesteps / sub_timesteps) * idx)
    elif s == "quadratic":
        tau = torch.round((timesteps / sub_timesteps**2) * idx**2)
    else:
        raise NotImplementedError
    return tau.long()


def q_sample(x0: Tensor, abar_t: Tensor, z: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """forward_process (equations/ddpm/ddpm.py:24-41) followed by Normal.sample():
    x_t = sqrt(abar) x0 + sqrt(1-abar) z.  Returns (x_t, mean, std)."""
    a = abar_t.reshape(-1, 1, 1, 1)
    mean = torch.sqrt(a) * x0
    std = torch.sqrt(1 - a)
    return mean + std * z, mean, std


def training_loss(
    eps_model: Callable[[Tensor, Tensor], Tensor], x0: Tensor, t: Tensor, z: Tensor, alpha_bar: Tensor
) -> Tensor:
    """DDPM.training_step (diffusion_models/ddpm.py:53-81) with t and z injected.

    The target is re-derived as (x_t - mean)/std (:79), not the drawn z, then
    simple_loss = mse over every element (equations/ddpm/losses.py:13)."""
    x_t, mean, std = q_sample(x0, alpha_bar[t], z)
    x_t = x_t.detach()  # Normal.sample() draws under no_grad: x_t carries no graph back to x_0
    eps = eps_model(x_t, t)
    target = (x_t - mean) / std
    return torch.mean((target - eps) ** 2)


def ddpm_step(x: Tensor, t: int, eps: Tensor, z: Tensor, beta: Tensor, alpha: Tensor, alpha_bar: Tensor) -> Tensor:
    """DDPM.sampling_step (diffusion_models/ddpm.py:94-111) + reverse_process
    (equations/ddpm/ddpm.py:65-71) for a scalar timestep (t.shape == (1,) in the
    reference): sigma^2 = beta_t; the drawn noise is discarded when t == 1."""
    mean = 1 / torch.sqrt(alpha[t]) * (x - beta[t] / torch.sqrt(1 - alpha_bar[t]) * eps)
    if t == 1:
        return mean
    return mean + torch.sqrt(beta[t]) * z


def ddim_step(x: Tensor, tau_i: int, tau_prev: int, eps: Tensor, alpha_bar: Tensor) -> Tensor:
    """DDIM.sampling_step (diffusion_models/ddim.py:65-77) + ddim.reverse_process
    (equations/ddim/ddim.py:52-57): mean of N(sqrt(abar_prev) x0_hat, .) with
    x0_hat = (x - sqrt(1-abar_i) eps)/sqrt(abar_prev) -- as shipped, not paper-DDIM."""
    x0_hat = (x - torch.sqrt(1 - alpha_bar[tau_i]) * eps) / torch.sqrt(alpha_bar[tau_prev])
    return torch.sqrt(alpha_bar[tau_prev]) * x0_hat


def ddpm_generate(
    eps_model: Callable[[Tensor, Tensor], Tensor],
    x_T: Tensor,
    noises: Sequence[Tensor],
    timesteps: int,
    n_steps: Optional[int] = None,
    start: float = 1e-4,
    end: float = 0.02,
) -> List[Tensor]:
    """DDPM.generate (diffusion_models/ddpm.py:113-133) with x_T and the per-step noise
    injected; t runs T, T-1, ...; returns the trajectory (one tensor per step).
    n_steps limits the number of steps taken (fixtures use a short prefix)."""
    beta = linear_beta(timesteps, start, end)
    alpha, abar = alpha_tables(beta)
    x = x_T
    traj = []
    steps = timesteps if n_steps is None else n_steps
    for k in range(steps):
        t = timesteps - k
        eps = eps_model(x, torch.tensor([t]))
        x = ddpm_step(x, t, eps, noises[k], beta, alpha, abar)
        traj.append(x)
    return traj


def ddim_generate(
    eps_model: Callable[[Tensor, Tensor], Tensor],
    x_T: Tensor,
    timesteps: int,
    sub_timesteps: int,
    schedule: str = "quadratic",
) -> List[Tensor]:
    """DDIM.generate (diffusion_models/ddim.py:79-99): i = S..1, model evaluated at tau_i.
    DDIM always uses the default linear beta range (ddim.py:39 drops start/end)."""
    beta = linear_beta(timesteps)
    _, abar = alpha_tables(beta)
    tau = tau_table(timesteps, sub_timesteps, schedule)
    x = x_T
    traj = []
    for i in range(sub_timesteps, 0, -1):
        ti, tp = int(tau[i]), int(tau[i - 1])
        eps = eps_model(x, torch.tensor([ti]))
        x = ddim_step(x, ti, tp, eps, abar)
        traj.append(x)
    return traj
