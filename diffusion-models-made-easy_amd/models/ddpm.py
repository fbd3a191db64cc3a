"""UNet noise predictor eps_theta(x_t, t) executed by libdmme_hip on an MI355X.

Drop-in for the reference's `dmme.models.ddpm.UNet` (src/dmme/models/ddpm.py:176-316):
same constructor arguments, same `forward(x, c)`, same 305-entry state_dict (key names,
shapes, order).  The layer graph, the parameter table and the launch sequence live in
the C++ plan (csrc/plan.hip); this class only owns the parameters (fp32 master copy in
one flat buffer; every named nn.Parameter is a view into it) and hands raw pointers to
the C ABI.  There is no PyTorch math on the device and no CPU fallback."""

from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from .. import _lib


class _Scope(nn.Module):
    """Parameter container used to reproduce the reference's dotted state_dict keys."""


def _cfg_struct(in_channels, pos_dim, emb_dim, num_groups, dropout, channels_per_depth, num_blocks, attention_depths, arch=0, num_heads=1):
    cfg = _lib.UNetCfg()
    cfg.arch, cfg.num_heads = int(arch), int(num_heads)
    cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups = in_channels, pos_dim, emb_dim, num_groups
    cfg.dropout = float(dropout)
    if not 1 <= len(channels_per_depth) <= 8 or len(attention_depths) > 8:
        raise ValueError("channels_per_depth must have 1..8 entries and attention_depths at most 8")
    cfg.num_depths = len(channels_per_depth)
    for i, c in enumerate(channels_per_depth):
        cfg.channels_per_depth[i] = int(c)
    cfg.num_blocks = num_blocks
    cfg.num_attention_depths = len(attention_depths)
    for i, d in enumerate(attention_depths):
        cfg.attention_depths[i] = int(d)
    return cfg


class _Plan:
    """Owns one dmme_plan handle plus the device buffers it needs (workspace, packed weights)."""

    def __init__(self, cfg, B, H, W, dtype, device_index):
        self.lib = _lib.lib()
        h = C.c_void_p()
        _lib.check(self.lib.dmme_unet_plan_create(C.byref(cfg), B, H, W, dtype, device_index, C.byref(h)), "dmme_unet_plan_create")
        self.h = h
        self.B, self.H, self.W, self.dtype = B, H, W, dtype
        self.workspace = None
        self.packed = None
        self.packed_version = None

    def param_table(self):
        n = self.lib.dmme_unet_plan_num_params(self.h)
        out = []
        name = C.create_string_buffer(256)
        ndim, off, isb = C.c_int(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(n):
            _lib.check(self.lib.dmme_unet_plan_param_info(self.h, i, name, 256, C.byref(ndim), shape, C.byref(off), C.byref(isb)), "param_info")
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value)), int(off.value), bool(isb.value)))
        return out

    def check(self):
        """raise DmmeError if a level-engine hand-off wait timed out in any launch of this plan since the last check (the outputs
        since are invalid; include/dmme_hip.h: dmme_unet_plan_check).  Call after synchronising with the stream: the entry points
        of the C ABI look at the same word themselves, a replayed hipGraph does not pass through them."""
        _lib.check(self.lib.dmme_unet_plan_check(self.h), "dmme_unet_plan_check")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dmme_unet_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


class UNet(nn.Module):
    r"""U-Net for predicting noise in images (same arguments as the reference, models/ddpm.py:190-200).

    Extra keyword `precision` ("fp32" | "bf16") selects the compute dtype of the HIP
    kernels; parameters are always held in fp32."""

    def __init__(
        self,
        in_channels: int = 3,
        pos_dim: int = 128,
        emb_dim: int = 512,
        num_groups: int = 32,
        dropout: float = 0.1,
        channels_per_depth: Sequence[int] = (128, 256, 256, 256),
        num_blocks: int = 2,
        attention_depths: Sequence[int] = (2,),
        precision: str = "fp32",
        _arch: int = 0,
        _num_heads: int = 1,
    ):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = 2 * in_channels if _arch == 1 else in_channels
        self.dropout = float(dropout)
        self.precision = precision
        self._dtype = _lib.dtype_code(precision)
        self._cfg = _cfg_struct(in_channels, pos_dim, emb_dim, num_groups, dropout, tuple(channels_per_depth), num_blocks, tuple(attention_depths), _arch, _num_heads)
        self._plans: Dict[Tuple, _Plan] = {}
        self._injected_masks: Optional[Tensor] = None
        self._mask_calls = 0
        self._param_epoch = 0  # bumped by optimisers that update the flat buffer through raw pointers
        self._views_version = 0
        self._flat_grad: Optional[Tensor] = None
        self._bucket_hook = None  # callable(offset, numel): set by distributed.OverlappedGradReducer
        # half-precision training: dynamic loss scaling (optim.FusedAdam(amp=...) switches it on; state lives on the device:
        # include/dmme_hip.h dmme_amp_*).  _amp: {"init_scale", "growth_factor", "backoff_factor", "growth_interval"} or None
        self._amp = None
        self._amp_state: Optional[Tensor] = None

        # parameter table from the C++ plan (host only: device = -1)
        table_plan = _Plan(self._cfg, 1, 32, 32, _lib.F32, -1)
        self._table = table_plan.param_table()
        self._ref_numel = int(table_plan.lib.dmme_unet_plan_ref_numel(table_plan.h))
        del table_plan

        flat = torch.zeros(self._ref_numel, dtype=torch.float32)
        self._flat = flat
        self._init_parameters(flat, pos_dim)
        self._register_views(flat)

    # ------------------------------------------------------------------ parameters
    def _init_parameters(self, flat: Tensor, pos_dim: int):
        """torch default initialisation (the reference defines no custom init):
        conv / linear weights kaiming_uniform(a=sqrt 5) => U(+-1/sqrt(fan_in)), biases
        U(+-1/sqrt(fan_in)), GroupNorm affine (1, 0), sinusoid table from the formula."""
        import math

        last_fan_in, prev_was_norm = 1, False
        for name, shape, off, is_buf in self._table:
            n = 1
            for s in shape:
                n *= s
            view = flat[off : off + n].view(shape)
            if is_buf:
                half = pos_dim // 2
                step = math.log(10000) / (half - 1)
                view.copy_(torch.exp(torch.arange(half) * -step).unsqueeze(0))
            elif len(shape) >= 2:  # conv / linear weight
                last_fan_in = n // shape[0]
                nn.init.uniform_(view, -1.0 / math.sqrt(last_fan_in), 1.0 / math.sqrt(last_fan_in))
                prev_was_norm = False
            elif name.endswith(".weight"):  # 1-D weight: GroupNorm gamma
                view.fill_(1.0)
                prev_was_norm = True
            elif prev_was_norm:  # GroupNorm beta
                view.zero_()
                prev_was_norm = False
            else:  # conv / linear bias
                b = 1.0 / math.sqrt(last_fan_in)
                nn.init.uniform_(view, -b, b)

    def _register_views(self, flat: Tensor):
        self._views = []
        for name, shape, off, is_buf in self._table:
            n = 1
            for s in shape:
                n *= s
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Scope())
                mod = mod._modules[p]
            view = flat[off : off + n].view(shape)
            if is_buf:
                mod.register_buffer(parts[-1], view)  # persistent, like the reference (:336)
            else:
                par = nn.Parameter(view)
                par._dmme_owner = weakref.ref(self)
                mod.register_parameter(parts[-1], par)
            self._views.append((mod, parts[-1], off, n, shape, is_buf))

    def _current(self, mod, leaf, is_buf) -> Tensor:
        return mod._buffers[leaf] if is_buf else mod._parameters[leaf]

    def _ensure_flat(self) -> Tensor:
        """Make every named parameter a view of one contiguous fp32 buffer again (after
        .to()/.cuda() moved them tensor by tensor).  Cheap check, rare re-flatten."""
        mod0, leaf0, off0, _, _, isb0 = self._views[0]
        first = self._current(mod0, leaf0, isb0)
        flat = self._flat
        ok = flat.device == first.device and flat.dtype == torch.float32
        if ok:
            # A Parameter rebound with `p.data = view` (what .cuda() / .to() leave behind, below) keeps its OWN version counter:
            # in-place writes through it (load_state_dict, torch.optim steps, p.copy_) do not move flat._version.  The sum of
            # the per-tensor counters is part of every repack / graph key (`_weights_key`).
            base = flat.data_ptr()
            ver = 0
            for mod, leaf, off, n, shape, is_buf in self._views:
                t = self._current(mod, leaf, is_buf)
                if t.data_ptr() != base + 4 * off or t.dtype != torch.float32:
                    ok = False
                    break
                ver += t._version
            self._views_version = ver
        if ok:
            return flat
        new = torch.empty(self._ref_numel, dtype=torch.float32, device=first.device)
        with torch.no_grad():
            for mod, leaf, off, n, shape, is_buf in self._views:
                t = self._current(mod, leaf, is_buf)
                new[off : off + n].copy_(t.detach().reshape(-1).to(torch.float32))
                v = new[off : off + n].view(shape)
                if is_buf:
                    mod._buffers[leaf] = v
                else:
                    t.data = v
        self._flat = new
        self._views_version = sum(self._current(mod, leaf, is_buf)._version for mod, leaf, _, _, _, is_buf in self._views)
        for p in self._plans.values():
            p.packed_version = None
        return new

    def _weights_key(self, flat: Tensor):
        """identifies the parameter VALUES the packed kernel-layout copies / captured graphs were made from: buffer address,
        its version, the versions of every named tensor viewing it, and the epoch raw-pointer optimisers bump"""
        return (flat.data_ptr(), flat._version, self._views_version, self._param_epoch)

    def _apply(self, fn, *a, **kw):
        """.cuda() / .to(device): ONE copy of the flat buffer and a rebind of the 305 named views, instead of nn.Module's
        tensor-by-tensor transfer followed by a re-flatten (2 x 305 small copies: the `copyBuffer` dispatches of the round-1
        profiles).  Conversions that change the dtype take the generic route and are re-flattened to fp32 afterwards."""
        flat = self._ensure_flat()
        with torch.no_grad():
            new = fn(flat)
        if not (isinstance(new, Tensor) and new.dtype == torch.float32 and new.numel() == flat.numel()):
            out = super()._apply(fn, *a, **kw)
            self._ensure_flat()
            return out
        if new is not flat and new.data_ptr() != flat.data_ptr():
            new = new.detach().contiguous()
            grad = self._flat_grad
            with torch.no_grad():
                new_grad = fn(grad).detach().contiguous() if grad is not None else None
            for mod, leaf, off, n, shape, is_buf in self._views:
                v = new[off : off + n].view(shape)
                if is_buf:
                    mod._buffers[leaf] = v
                else:
                    p = mod._parameters[leaf]
                    old_grad = p.grad
                    p.data = v
                    if old_grad is not None:
                        if new_grad is not None and grad is not None and old_grad.data_ptr() == grad.data_ptr() + 4 * off:
                            p.grad = new_grad[off : off + n].view(shape)  # a view of the flat gradient buffer: follows it
                        else:
                            with torch.no_grad():
                                p.grad = fn(old_grad)  # a gradient the user assigned: moved like nn.Module._apply would
            self._flat = new
            self._flat_grad = new_grad
            for pl in self._plans.values():
                pl.packed_version = None
            self._ensure_flat()
        return self

    def flat_parameters(self) -> Tensor:
        """The contiguous fp32 master buffer (reference state_dict order and layouts)."""
        return self._ensure_flat()

    def set_precision(self, precision: str):
        self.precision = precision
        self._dtype = _lib.dtype_code(precision)
        return self

    # ------------------------------------------------------------------ plan / buffers
    def _plan_for(self, B: int, H: int, W: int, device: torch.device) -> _Plan:
        key = (B, H, W, self._dtype, device.index)
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(self._cfg, B, H, W, self._dtype, device.index if device.index is not None else torch.cuda.current_device())
            lib = plan.lib
            plan.workspace = torch.empty(int(lib.dmme_unet_plan_workspace_bytes(plan.h)), dtype=torch.uint8, device=device)
            plan.packed = torch.empty(int(lib.dmme_unet_plan_packed_bytes(plan.h)), dtype=torch.uint8, device=device)
            plan.dropmask_numel = int(lib.dmme_unet_plan_dropmask_numel(plan.h))
            plan.masks = None
            self._plans[key] = plan
        return plan

    def _packed_for(self, plan: _Plan) -> Tensor:
        flat = self._ensure_flat()
        ver = self._weights_key(flat)
        if plan.packed_version != ver:
            _lib.check(plan.lib.dmme_unet_pack_params(plan.h, _lib.ptr(flat), _lib.ptr(plan.packed), _lib.stream_ptr()), "dmme_unet_pack_params")
            plan.packed_version = ver
        return plan.packed

    def inject_dropout_masks(self, masks: Optional[Tensor]):
        """Test hook: use these Dropout2d multipliers (layout of dmme_unet_plan_dropmask_numel)
        instead of drawing them, for train-mode parity with injected randomness."""
        self._injected_masks = masks

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, c: Tensor) -> Tensor:
        r"""Predicts noise from x (reference: models/ddpm.py:281-316).

        x: (N, C, H, W) on an MI355X; c: timesteps of shape (N,) or (1,)."""
        _lib.require_gpu()
        if not x.is_cuda:
            raise _lib.DmmeError("UNet.forward needs a GPU tensor: the HIP denoiser has no CPU fallback")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected input of shape (N, {self.in_channels}, H, W), got {tuple(x.shape)}")
        B, _, H, W = x.shape
        if c.numel() not in (1, B):
            raise RuntimeError(f"timestep tensor of {c.numel()} elements does not broadcast against batch {B}")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..autograd import unet_apply  # training path (HIP backward)

            return unet_apply(self, x, c)
        return self._forward_impl(x, c)

    def amp_state(self, device=None) -> Optional[Tensor]:
        """the device-resident loss-scaling state (8 floats: scale, growth tracker, steps taken, found_inf, steps skipped) when
        dynamic loss scaling is on and the compute dtype is IEEE half; created at first use"""
        if self._amp is None or self._dtype != _lib.F16:
            return None
        dev = device if device is not None else self.flat_parameters().device
        st = self._amp_state
        if st is None or st.device != torch.device(dev):
            st = torch.zeros(8, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().dmme_amp_init(_lib.ptr(st), float(self._amp["init_scale"]), _lib.stream_ptr()), "dmme_amp_init")
            self._amp_state = st
        return st

    def mark_params_updated(self):
        """call after writing the flat parameter buffer through a raw pointer (fused optimiser)"""
        self._param_epoch += 1

    # ------------------------------------------------------------------ gradients
    def flat_grad(self) -> Tensor:
        """fp32 gradient buffer in the flat parameter layout; every param.grad is a view of it.
        (Re)created zeroed when absent, on another device, or after zero_grad(set_to_none=True)."""
        flat = self._ensure_flat()
        g = self._flat_grad
        fresh = g is None or g.device != flat.device or g.numel() != flat.numel()
        if fresh:
            g = torch.zeros_like(flat)
            self._flat_grad = g
        rebind = fresh
        if not rebind:
            for mod, leaf, off, n, shape, is_buf in self._views:
                if not is_buf and mod._parameters[leaf].grad is None:
                    rebind = True
                    break
            if rebind:
                g.zero_()
        if rebind:
            for mod, leaf, off, n, shape, is_buf in self._views:
                if not is_buf:
                    mod._parameters[leaf].grad = g[off : off + n].view(shape)
        return g

    def _backward_impl(self, saved, dy: Tensor, want_dx: bool = False):
        plan, xin, t, masks, gen = saved
        if gen != plan.fwd_gen:
            raise RuntimeError(
                "UNet backward: another forward of the same shape ran after the forward this graph belongs to and overwrote its "
                "saved activations (one workspace per (B, H, W, dtype) plan). Call backward() before the next forward, or "
                "run the extra forward under a different batch size.")
        lib = plan.lib
        dev = xin.device
        dx = torch.empty_like(xin) if want_dx else None
        if getattr(plan, "bws", None) is None:
            plan.bws = torch.empty(int(lib.dmme_unet_plan_bwd_workspace_bytes(plan.h)), dtype=torch.uint8, device=dev)
            plan.packed_bwd = torch.empty(int(lib.dmme_unet_plan_packed_bwd_bytes(plan.h)), dtype=torch.uint8, device=dev)
            plan.packed_bwd_version = None
        flat = self._ensure_flat()
        packed = self._packed_for(plan)
        ver = self._weights_key(flat)
        if plan.packed_bwd_version != ver:
            _lib.check(lib.dmme_unet_pack_params_bwd(plan.h, _lib.ptr(flat), _lib.ptr(plan.packed_bwd), _lib.stream_ptr()), "dmme_unet_pack_params_bwd")
            plan.packed_bwd_version = ver
        g = self.flat_grad()
        d = dy.detach().to(torch.float32).contiguous()
        amp = self.amp_state(dev)
        if amp is not None:  # backward of S * loss: the scale is read on the device, the optimiser pass divides it out again
            if d.data_ptr() == dy.data_ptr():
                d = d.clone()
            _lib.check(lib.dmme_amp_scale(_lib.ptr(d), d.numel(), _lib.ptr(amp), _lib.stream_ptr()), "dmme_amp_scale")
        hook = self._bucket_hook
        if hook is not None:  # two gradient buckets, each handed over as soon as its launches are enqueued (overlapped all-reduce)
            errors = []

            def ready(_user, _bucket, offset, numel):
                try:
                    hook(int(offset), int(numel))
                except Exception as exc:  # noqa: BLE001 - never unwind through the C frames
                    errors.append(exc)

            cb = _lib.BUCKET_FN(ready)
            try:
                _lib.check(
                    lib.dmme_unet_backward_buckets(plan.h, _lib.ptr(packed), _lib.ptr(plan.packed_bwd), _lib.ptr(xin), _lib.ptr(t), int(t.numel()), _lib.ptr(d),
                                                   _lib.ptr(plan.workspace), _lib.ptr(plan.bws), _lib.ptr(masks), _lib.ptr(g), _lib.ptr(dx), _lib.stream_ptr(), cb, None),
                    "dmme_unet_backward_buckets",
                )
                if errors:
                    raise errors[0]
            except Exception:
                # buckets already handed over have collectives on the side stream: join them and lift the forward guard, or every
                # later forward would raise "exchange in flight" instead of this backward's error
                owner = getattr(hook, "__self__", None)
                if owner is not None and hasattr(owner, "abort"):
                    owner.abort()
                else:
                    self._exchange_in_flight = False
                raise
            if dx is not None and amp is not None:
                dx.div_(amp[0])
            return dx
        _lib.check(
            lib.dmme_unet_backward(plan.h, _lib.ptr(packed), _lib.ptr(plan.packed_bwd), _lib.ptr(xin), _lib.ptr(t), int(t.numel()), _lib.ptr(d),
                                   _lib.ptr(plan.workspace), _lib.ptr(plan.bws), _lib.ptr(masks), _lib.ptr(g), _lib.ptr(dx), _lib.stream_ptr()),
            "dmme_unet_backward",
        )
        if dx is not None and amp is not None:
            dx.div_(amp[0])
        return dx

    def _forward_impl(self, x: Tensor, c: Tensor, want_ctx: bool = False):
        if getattr(self, "_exchange_in_flight", False):
            # a gradient exchange of this model is still on its side stream (distributed.OverlappedGradReducer between its first
            # bucket and finish()): a forward enqueued now could put the level engine's persistent, co-residency-dependent launches
            # on the chip together with RCCL's kernels.  train_step joins the exchange before the optimiser; anything else is a bug.
            raise _lib.DmmeError("UNet.forward while a gradient exchange of this model is in flight: call reducer.finish() first")
        B, _, H, W = x.shape
        plan = self._plan_for(B, H, W, x.device)
        packed = self._packed_for(plan)
        xin = x.detach().to(torch.float32).contiguous()
        if c.is_floating_point() and bool((c != c.round()).any()):
            # the reference's sinusoid accepts any real t (models/ddpm.py:347); every caller passes integer timesteps and the C ABI
            # takes int64 - refuse rather than truncate
            raise NotImplementedError("fractional timesteps are not supported by the HIP path (integer t only)")
        t = c.detach().reshape(-1).to(device=x.device, dtype=torch.int64).contiguous()
        y = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        masks = None
        if self.training and self.dropout > 0:
            if self._injected_masks is not None:
                masks = self._injected_masks.to(device=x.device, dtype=torch.float32).contiguous()
                if masks.numel() != plan.dropmask_numel:
                    raise ValueError(f"injected dropout masks have {masks.numel()} elements, plan needs {plan.dropmask_numel}")
            else:
                if plan.masks is None:
                    plan.masks = torch.empty(plan.dropmask_numel, dtype=torch.float32, device=x.device)
                masks = plan.masks
                from ..common.noise import philox_reserve

                self._mask_calls += 1
                seed, off = philox_reserve(x.device, plan.dropmask_numel)  # torch's CUDA generator: manual_seed restarts the draws
                _lib.check(plan.lib.dmme_dropout_masks(plan.h, seed ^ 0x5DEECE66D, off, _lib.ptr(masks), _lib.stream_ptr()), "dmme_dropout_masks")
        # no backward pass will follow (sampling, evaluation): the form that skips tensors only the backward pass reads
        fwd = plan.lib.dmme_unet_forward if want_ctx else plan.lib.dmme_unet_forward_nograd
        _lib.check(
            fwd(plan.h, _lib.ptr(packed), _lib.ptr(xin), _lib.ptr(t), int(t.numel()), _lib.ptr(y), _lib.ptr(plan.workspace), _lib.ptr(masks), _lib.stream_ptr()),
            "dmme_unet_forward",
        )
        self._last_plan = plan
        # every forward of this (B, H, W, dtype) overwrites the activations (plan.workspace) and the drawn masks a pending
        # backward would read: the generation stamp lets that backward refuse instead of producing wrong gradients
        plan.fwd_gen = getattr(plan, "fwd_gen", 0) + 1
        if want_ctx:
            return y, (plan, xin, t, masks, plan.fwd_gen)
        return y

    # ------------------------------------------------------------------ hipGraph replay (sampling loops)
    def graphed_forward(self, x_static: Tensor, t_static: Tensor) -> Tensor:
        """eps for the sampling loops with the ~160 launches of one forward replayed from a hipGraph.

        The caller passes the SAME two tensors every step (it updates them in place: x by the sampler
        kernel, t by a tiny copy); the returned eps tensor is static too.  The graph is captured on first
        use after an eager warm-up (weight packing and kernel attribute setup happen there) and is
        re-captured if the parameters, the precision or the tensors change.  Falls back to the eager
        path when capture is unavailable.  Eval mode only (no dropout masks inside the graph)."""
        if self.training or getattr(self, "_graph_disabled", False):
            return self._forward_impl(x_static, t_static)
        flat = self._ensure_flat()
        key = (x_static.data_ptr(), t_static.data_ptr(), tuple(x_static.shape), self._dtype) + self._weights_key(flat)
        g = getattr(self, "_graph", None)
        if g is None or g[0] != key:
            with torch.no_grad():
                # the eager trial forward (packs weights, sets kernel attributes) is OUTSIDE the fallback: a DmmeError or a kernel
                # fault in it is a real error and propagates (as ChainRunner.step does); only a failure of the CAPTURE itself makes
                # this module stay eager, and the cause is kept in `_graph_error`
                self._forward_impl(x_static, t_static)
                torch.cuda.synchronize()
                self._last_plan.check()
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        y = self._forward_impl(x_static, t_static)
                except RuntimeError as exc:  # stream capture unavailable / invalidated here: the same launches, eagerly
                    if isinstance(exc, _lib.DmmeError):
                        raise
                    self._graph_disabled = True
                    self._graph_error = exc
                    self._graph = None
                    return self._forward_impl(x_static, t_static)
            g = (key, graph, y, self._last_plan)  # the plan whose workspace the replays overwrite
            self._graph = g
        plan = g[3]  # the replay overwrites THAT plan's workspace (not whichever plan ran last): its pending backward must refuse
        # a replay passes through no entry point of the C ABI, so nobody else looks at the level engine's status word: an atomic load
        # of pinned host memory, no synchronisation while the word is clear - a hand-off timeout of an earlier replay is reported
        # at most one step late instead of never; callers that read the result on the host use check_engine()
        plan.check()
        g[1].replay()
        plan.fwd_gen = getattr(plan, "fwd_gen", 0) + 1
        return g[2]

    def check_engine(self):
        """synchronise and raise DmmeError if any level-engine launch of this module's plans gave up on a hand-off (results invalid)"""
        torch.cuda.synchronize()
        for plan in self._plans.values():
            plan.check()

    def debug_activation(self, name: str) -> Tensor:
        """fp32 NCHW copy of the output of module `name` from the last forward (parity tests)."""
        plan = self._last_plan
        cap = plan.B * 4096 * max(plan.H * plan.W, 1)
        buf = torch.empty(cap, dtype=torch.float32, device=plan.workspace.device)
        n = C.c_int64()
        _lib.check(plan.lib.dmme_unet_debug_read(plan.h, _lib.ptr(plan.workspace), name.encode(), _lib.ptr(buf), cap, C.byref(n), _lib.stream_ptr()), "dmme_unet_debug_read")
        return buf[: n.value].clone()
