"""Frozen-stream synthetic data + compact digests shared by the golden generator and tests
(test infrastructure only).  numpy.random.RandomState (MT19937) is a frozen legacy
stream, so a fixture that records only ``seed`` reproduces identical inputs anywhere."""

from __future__ import annotations

import numpy as np
import torch


def normal(seed: int, shape) -> torch.Tensor:
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(size=tuple(shape)).astype(np.float32))


def uniform(seed: int, shape, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    return torch.from_numpy(np.random.RandomState(seed).uniform(lo, hi, size=tuple(shape)).astype(np.float32))


def randint(seed: int, lo: int, hi: int, count: int) -> torch.Tensor:
    """integers in [lo, hi) like torch.randint (the reference's uniform_int, common/noise.py:14-16)."""
    return torch.from_numpy(np.random.RandomState(seed).randint(lo, hi, size=(count,)).astype(np.int64))


N_SAMPLE = 2048


def digest(t: torch.Tensor) -> np.ndarray:
    """[sum, abs-sum (fp64 accumulate), then up to N_SAMPLE strided samples] as float64."""
    flat = t.detach().to(torch.float32).reshape(-1)
    stride = max(1, flat.numel() // N_SAMPLE)
    samp = flat[::stride][:N_SAMPLE].to(torch.float64)
    head = torch.stack([flat.to(torch.float64).sum(), flat.to(torch.float64).abs().sum()])
    return torch.cat([head, samp]).numpy()


def digest_close(t: torch.Tensor, ref: np.ndarray, atol: float, rtol: float = 0.0):
    """Compare a tensor with a stored digest; returns (ok, max_abs_err_on_samples)."""
    d = digest(t)
    n = t.numel()
    samp_err = float(np.max(np.abs(d[2:] - ref[2:]))) if len(d) > 2 else 0.0
    tol = atol + rtol * float(np.max(np.abs(ref[2:]))) if len(ref) > 2 else atol
    # the sums may drift by ~sqrt(n) * atol; allow n * atol / 8 as a loose consistency check
    sum_ok = abs(d[0] - ref[0]) <= max(1e-6, tol * n / 8) and abs(d[1] - ref[1]) <= max(1e-6, tol * n / 8)
    return (samp_err <= tol) and sum_ok, samp_err
