"""Random draws with the reference's names and contracts (src/dmme/common/noise.py:4-23).

On a GPU device the normals come from the library's Philox4x32-10 kernel
(dmme_randn); seed and offset come from torch's CUDA generator, so `torch.manual_seed`
restarts the stream.  Integer timesteps and CPU draws use torch (host plumbing)."""

from __future__ import annotations

import torch

from .. import _lib

def philox_reserve(device: torch.device, numel: int):
    """(seed, quad offset) of a fresh span of the device's Philox stream, taken from torch's own CUDA generator: its seed keys the
    stream and its offset is advanced by what the span consumes, so `torch.manual_seed(s)` restarts the library's draws exactly
    as it restarts torch's (the reference's randn / dropout draws come from that generator)."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed = gen.initial_seed() & 0xFFFFFFFFFFFFFFFF
    off = int(gen.get_offset())
    quads = (int(numel) + 3) // 4
    gen.set_offset(off + 4 * quads)
    return seed, off // 4


def _philox_fill(out: torch.Tensor) -> torch.Tensor:
    seed, off = philox_reserve(out.device, out.numel())
    _lib.check(_lib.lib().dmme_randn(_lib.ptr(out), out.numel(), seed, off, _lib.stream_ptr()), "dmme_randn")
    return out


def gaussian(shape, dtype=None, device=None):
    """standard normal tensor of `shape` (reference: torch.randn)"""
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dev.type == "cuda":
        with torch.cuda.device(dev):
            out = _philox_fill(torch.empty(tuple(shape), dtype=torch.float32, device=dev))
        return out if dtype in (None, torch.float32) else out.to(dtype)
    return torch.randn(shape, dtype=dtype, device=device)


def gaussian_like(x):
    """standard normal tensor shaped like x (reference: torch.randn_like)"""
    return gaussian(x.shape, dtype=x.dtype if x.is_floating_point() else None, device=x.device)


def uniform_int(min, max, count=1, device=None):
    """`count` integers in [min, max) -- high is exclusive, so t = max is never drawn."""
    return torch.randint(min, max, size=(count,), device=device)


def pad(x: torch.Tensor, value: float = 0) -> torch.Tensor:
    r"""prepend one entry so that the tensor index equals the timestep t"""
    head = torch.full_like(x[0:1], value)
    return torch.cat([head, x], dim=0)
