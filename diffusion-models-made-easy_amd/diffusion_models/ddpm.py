"""DDPM training / sampling driver over the HIP kernels.

Drop-in for the reference's `dmme.diffusion_models.DDPM`
(src/dmme/diffusion_models/ddpm.py:15-144): same constructor, buffers
(beta / alpha / alpha_bar, shape (T+1,1,1,1), non-persistent) and methods.  The host
loop stays in Python as in the reference; every per-pixel operation (forward noising,
the reverse update, the MSE loss, the normal draws) is one fused HIP kernel."""

from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib
from ..common.noise import gaussian, gaussian_like, philox_reserve, uniform_int
from ..equations.ddpm import linear_schedule, sampling_coefficients


class ChainRunner:
    """One replayable denoising step on a fixed image buffer (include/dmme_hip.h: dmme_chain_*).

    The loop state (index i, timestep t, Philox seed / offset) and the per-index scalars of the update live on the device;
    `step()` replays ONE captured hipGraph of  time MLP + UNet + noise draw + sampler update + state advance  - no host value
    changes between steps, nothing is copied host-to-device, nothing synchronises.  `x` is updated in place.  Falls back to
    issuing the same launches eagerly when graph capture is unavailable (same results either way)."""

    def __init__(self, process, x: Tensor, use_graph: bool = True):
        model = process.model
        if not (isinstance(x, Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4):
            raise ValueError("ChainRunner needs a contiguous fp32 GPU image batch")
        B, _, H, W = x.shape
        self.process, self.model, self.x = process, model, x
        self.plan = model._plan_for(B, H, W, x.device)
        self.kind = process._chain_kind
        n, rows, ttab = process._chain_tables()
        self.n_steps = n
        self.coef = torch.tensor(rows, dtype=torch.float32).reshape(-1).to(x.device)
        self.ttab = torch.tensor(ttab, dtype=torch.int64).to(x.device)
        self.state = torch.zeros(8, dtype=torch.int64, device=x.device)
        self.out = torch.empty((B, model.out_channels, H, W), dtype=torch.float32, device=x.device)
        self.quads = x.numel() // 4
        self.use_graph = use_graph
        self.graph = None
        self.capture_error = None  # why this runner launches eagerly although a graph was asked for (None: it does not)
        self._wkey = None

    def set(self, i: int, seed: int = 0, offset: int = 0):
        """place the loop at index i (DDPM: t = i) with the Philox stream at (seed, offset in quads)"""
        _lib.check(_lib.lib().dmme_chain_set(_lib.ptr(self.state), int(i), _lib.ptr(self.ttab), seed & 0xFFFFFFFFFFFFFFFF, int(offset), _lib.stream_ptr()), "dmme_chain_set")

    def _launch(self, packed):
        _lib.check(
            _lib.lib().dmme_chain_step(self.plan.h, _lib.ptr(packed), _lib.ptr(self.x), _lib.ptr(self.out), _lib.ptr(self.plan.workspace), self.kind,
                                       _lib.ptr(self.coef), _lib.ptr(self.ttab), _lib.ptr(self.state), _lib.stream_ptr()),
            "dmme_chain_step",
        )
        self.plan.fwd_gen = getattr(self.plan, "fwd_gen", 0) + 1  # the workspace was overwritten (pending backwards must refuse)

    def step(self):
        model = self.model
        if model.training:
            raise RuntimeError("ChainRunner: sampling chains run in eval mode (no dropout masks inside the replayed step)")
        packed = model._packed_for(self.plan)  # re-packs when the parameters changed
        wkey = (self.plan.packed_version, self.plan.packed.data_ptr())
        if not self.use_graph or self.capture_error is not None:
            self._launch(packed)
            return self.x
        if self.graph is None or self._wkey != wkey:
            # capture: one eager step first (kernel attribute setup happens at first launch), on a saved copy of x / the state.
            # A kernel or DMME error in that trial step is a real error and propagates; only a failure of the CAPTURE itself makes
            # this runner (not the model) fall back to eager launches, and the cause is kept in `capture_error`.
            saved_x, saved_state = self.x.clone(), self.state.clone()
            self._launch(packed)
            torch.cuda.synchronize()
            self.plan.check()  # (outside the capture: a level-engine timeout in the trial step is a real error, raised here)
            self.x.copy_(saved_x)
            self.state.copy_(saved_state)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._launch(packed)
            except RuntimeError as exc:  # stream capture unavailable / invalidated: stay eager (same launches)
                self.capture_error = exc
                self.graph = None
                self.x.copy_(saved_x)
                self.state.copy_(saved_state)
                self._launch(packed)
                return self.x
            self.x.copy_(saved_x)
            self.state.copy_(saved_state)
            self.graph, self._wkey = graph, wkey
        # (no synchronisation while the engine's status word is clear: an atomic load of pinned host memory; a hand-off timeout of an
        # earlier replayed step raises here, one step late at most, for callers that never reach `run`'s final check)
        self.plan.check()
        self.graph.replay()
        self.plan.fwd_gen = getattr(self.plan, "fwd_gen", 0) + 1
        return self.x

    def run(self, first: int, count: int):
        """`count` steps from loop index `first` downwards, drawing from torch's CUDA generator like the eager loop"""
        if self.kind == _lib.CHAIN_DDIM:  # the DDIM update draws nothing: torch's generator stays where the eager loop leaves it
            seed, off = 0, 0
        else:
            seed, off = philox_reserve(self.x.device, self.x.numel() * count)
        self.set(first, seed, off)
        for _ in range(count):
            self.step()
        # replayed graphs do not pass through the C entry points that look at the level engine's status word: one synchronisation
        # per chain (the caller reads the images next anyway), then the check - a chain never returns numbers from a launch that
        # gave up on a hand-off (DmmeError instead)
        torch.cuda.current_stream(self.x.device).synchronize()
        self.plan.check()
        return self.x


class DDPM(nn.Module):
    beta: Tensor
    alpha: Tensor
    alpha_bar: Tensor

    def __init__(self, model: nn.Module, timesteps: int = 1000, start: float = 0.0001, end: float = 0.02) -> None:
        super().__init__()
        self.model = model
        self.timesteps = timesteps

        beta = linear_schedule(timesteps, start, end).reshape(-1, 1, 1, 1)
        alpha = 1 - beta
        alpha_bar = torch.cumprod(alpha, dim=0)  # alpha[0] = 1
        self.register_buffer("beta", beta, persistent=False)
        self.register_buffer("alpha", alpha, persistent=False)
        self.register_buffer("alpha_bar", alpha_bar, persistent=False)
        # fp32 tables for the forward-noising kernel, evaluated like forward_process does
        # (reference: equations/ddpm/ddpm.py:36-39); device-resident, not part of the state_dict
        self.register_buffer("_sqrt_alpha_bar", torch.sqrt(alpha_bar).reshape(-1).contiguous(), persistent=False)
        self.register_buffer("_sqrt_one_minus_alpha_bar", torch.sqrt(1 - alpha_bar).reshape(-1).contiguous(), persistent=False)
        # host copies of the per-step scalars of the reverse update (python floats)
        self._c1, self._c2, self._sigma = sampling_coefficients(beta, alpha, alpha_bar)
        self._all_t: Optional[Tensor] = None
        self._runner: Optional[ChainRunner] = None

    # ------------------------------------------------------------------ device-resident loop (replayable step)
    _chain_kind = _lib.CHAIN_DDPM

    def _chain_tables(self):
        """(number of steps, per-index update scalars [n+1][4], timestep at each loop index) of dmme_chain_step"""
        T = self.timesteps
        rows = [(self._c1[t], self._c2[t] if t > 0 else 0.0, self._sigma[t], 0.0) for t in range(T + 1)]  # index 0 is never stepped from
        return T, rows, list(range(T + 1))

    def timestep_tensor(self, t: int, device) -> Tensor:
        """shape-(1,) device view holding t: indexes a resident arange instead of building `torch.tensor([t])` (a host-to-device
        copy per step in the reference, lit_modules/ddpm.py:77)"""
        n = self.timesteps + 1
        if self._all_t is None or self._all_t.device != torch.device(device) or self._all_t.numel() != n:
            self._all_t = torch.arange(0, n, device=device).unsqueeze(1)
        return self._all_t[t]

    def chain_runner(self, x: Tensor, use_graph: bool = True, slot: str = "_runner") -> Optional[ChainRunner]:
        """runner bound to the image buffer `x` (cached per buffer / shape); None when the replayable step does not apply
        (a model that is not a dmme_amd UNet, train mode, an image size that is not a multiple of 4)"""
        model = self.model
        if not hasattr(model, "_plan_for") or model.training or not x.is_cuda or x[0].numel() % 4 or x.dtype != torch.float32 or not x.is_contiguous():
            return None
        r = getattr(self, slot, None)
        key = (x.data_ptr(), tuple(x.shape), model._dtype, use_graph)
        if r is None or r._key != key or r.model is not model:
            r = ChainRunner(self, x, use_graph)
            r._key = key
            setattr(self, slot, r)
        return r

    def _generate_runner(self, img_size, dev) -> Optional["ChainRunner"]:
        """the runner `generate` uses: ONE per (shape, dtype), bound to an internal image buffer that outlives the call - a second
        `generate` of the same shape re-uses the captured graph, the coefficient tables and the buffers instead of building them again"""
        buf = getattr(self, "_gen_buf", None)
        if buf is None or tuple(buf.shape) != tuple(img_size) or buf.device != torch.device(dev):
            buf = self._gen_buf = torch.empty(tuple(img_size), dtype=torch.float32, device=dev)
        return self.chain_runner(buf, slot="_runner")

    def _once_via_runner(self, x_t: Tensor, index: int) -> Optional[Tensor]:
        """single step at loop index `index` through the captured graph, for callers that loop on the host one step at a time
        (LitDDPM.forward <- callbacks/generate.py:82): small batches are bound by the ~160 dependent launches of an eager step"""
        if self.model.training or not x_t.is_cuda or x_t.shape[0] > 64 or torch.is_grad_enabled() and x_t.requires_grad:
            return None
        buf = getattr(self, "_once_buf", None)
        if buf is None or buf.shape != x_t.shape or buf.device != x_t.device:
            buf = self._once_buf = torch.empty(tuple(x_t.shape), dtype=torch.float32, device=x_t.device)
        runner = self.chain_runner(buf, slot="_runner_once")
        if runner is None:
            return None
        buf.copy_(x_t)
        seed, off = philox_reserve(x_t.device, x_t.numel())
        runner.set(index, seed, off)
        with torch.no_grad():
            runner.step()
        out = buf.clone()
        # (per-step callers: `runner.step()` reads the engine's status word in front of every replay without synchronising - a
        # hand-off timeout raises one step late at most - and `model.check_engine()` is the synchronising check at the end of a loop)
        return out

    # ------------------------------------------------------------------ training
    def training_step(self, x_0: Tensor, t: Optional[Tensor] = None, noise: Optional[Tensor] = None) -> Tensor:
        r"""L_simple for one batch (reference: diffusion_models/ddpm.py:53-81).

        `t` / `noise` may be injected for parity tests; by default t ~ randint(1, T)
        (never T itself, as in the reference) and noise ~ N(0, I)."""
        from ..autograd import mse_loss_apply

        B = x_0.size(0)
        if t is None:
            t = uniform_int(1, self.timesteps, B, device=x_0.device)
        if noise is None:
            noise = gaussian_like(x_0)
        x0 = x_0.detach().to(torch.float32).contiguous()
        z = noise.detach().to(torch.float32).contiguous()
        t = t.to(device=x_0.device, dtype=torch.int64).contiguous()
        x_t = torch.empty_like(x0)
        target = torch.empty_like(x0)
        _lib.check(
            _lib.lib().dmme_q_sample(_lib.ptr(x0), _lib.ptr(z), _lib.ptr(self._sqrt_alpha_bar), _lib.ptr(self._sqrt_one_minus_alpha_bar), _lib.ptr(t), B, x0[0].numel(), _lib.ptr(x_t), _lib.ptr(target), _lib.stream_ptr()),
            "dmme_q_sample",
        )
        noise_in_x_t = self.model(x_t, t)
        return mse_loss_apply(noise_in_x_t, target)

    # ------------------------------------------------------------------ sampling
    def _reverse_update(self, x_t: Tensor, eps: Tensor, t: int, noise: Optional[Tensor]) -> Tensor:
        if noise is None:
            noise = gaussian_like(x_t)  # drawn even when t == 1, then unused (reference :107-110)
        _lib.check(
            _lib.lib().dmme_ddpm_step(_lib.ptr(x_t), _lib.ptr(eps), _lib.ptr(noise), self._c1[t], self._c2[t], self._sigma[t], int(t != 1), x_t.numel(), _lib.stream_ptr()),
            "dmme_ddpm_step",
        )
        return x_t

    def sampling_step(self, x_t: Tensor, t: Tensor, noise: Optional[Tensor] = None) -> Tensor:
        r"""one draw from p_theta(x_{t-1} | x_t) (reference: diffusion_models/ddpm.py:83-111).

        As in the reference only a timestep tensor of shape (1,) is valid (the reference's
        `torch.where(t == 1, ...)` broadcasts t against the last image dimension)."""
        if t.numel() != 1:
            raise RuntimeError(f"sampling_step expects a timestep tensor of shape (1,), got {tuple(t.shape)}")
        step = int(t.reshape(-1)[0].item())
        eps = self.model(x_t, t.reshape(1))
        x = x_t.detach().to(torch.float32).clone()
        return self._reverse_update(x, eps, step, noise)

    def denoise_once(self, x_t: Tensor, t: int) -> Tensor:
        """x_{t-1} ~ p_theta(. | x_t) for a host integer t, as a new tensor: what `LitDDPM.forward(x_t, t)` returns
        (reference: lit_modules/ddpm.py:65-79) without the per-step `torch.tensor([t])` upload or a `.item()` read-back"""
        t = int(t)
        out = self._once_via_runner(x_t, t)
        if out is not None:
            return out
        eps = self.model(x_t, self.timestep_tensor(t, x_t.device))
        x = x_t.detach().to(torch.float32).clone()
        return self._reverse_update(x, eps, t, None)

    @torch.no_grad()
    def generate(self, img_size: Tuple[int, int, int, int]) -> Tensor:
        """run the full T-step chain from pure noise (reference: diffusion_models/ddpm.py:113-133)"""
        dev = self.beta.device
        x_t = gaussian(img_size, device=dev)
        runner = self._generate_runner(img_size, dev) if len(img_size) == 4 and not self.model.training else None
        if runner is not None:  # one captured step (UNet + noise + update + t -> t-1) replayed T times
            runner.x.copy_(x_t)
            return runner.run(self.timesteps, self.timesteps).clone()
        for t in range(self.timesteps, 0, -1):
            eps = self.model(x_t, self.timestep_tensor(t, dev))
            self._reverse_update(x_t, eps, t, None)
        return x_t

    def forward(self, x: Tensor, t: Tensor) -> Tensor:
        return self.model(x, t)
