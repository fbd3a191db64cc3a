"""Functional CPU restatement of the reference DDPM UNet (test infrastructure only).

Everything here is driven by a plain ``dict[str, Tensor]`` with the reference's
state_dict keys; there are no nn.Modules.  Each function cites the reference lines
it restates (paths relative to /root/reference).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass(frozen=True)
class UNetConfig:
    """Constructor arguments of the reference UNet (src/dmme/models/ddpm.py:190-200)."""

    in_channels: int = 3
    pos_dim: int = 128
    emb_dim: int = 512
    num_groups: int = 32
    dropout: float = 0.1
    channels_per_depth: Tuple[int, ...] = (128, 256, 256, 256)
    num_blocks: int = 2
    attention_depths: Tuple[int, ...] = (2,)


# the hyper-parameters the reference's own tests use (tests/test_ddpm.py:8-15)
TINY = UNetConfig(pos_dim=4, emb_dim=8, num_groups=2, channels_per_depth=(4, 8, 16, 32), num_blocks=3)


@dataclass
class Node:
    kind: str  # "res" | "down" | "up"
    prefix: str  # state_dict prefix, e.g. "down_layers.3"
    c_in: int
    c_out: int
    attn: bool = False


@dataclass
class Graph:
    cfg: UNetConfig
    down: List[Node] = field(default_factory=list)
    mid: List[Node] = field(default_factory=list)
    up: List[Node] = field(default_factory=list)
    base: int = 0


def build_graph(cfg: UNetConfig) -> Graph:
    """Layer list of UNet.__init__ (src/dmme/models/ddpm.py:203-279).

    The reference tests ``down_layers[-1] == len(channels) - 1`` (module vs int,
    :242) which can never hold, so the up path always starts without an UpSample and
    every resolution gets its ResBlocks *before* the UpSample; restated as such.
    """
    g = Graph(cfg)
    nb = cfg.num_blocks
    chans = [cfg.channels_per_depth[0]]
    for c in cfg.channels_per_depth:
        chans.extend([c] * nb)
    n_depth = len(cfg.channels_per_depth)
    cut_after = {nb * i for i in range(1, n_depth)}
    g.base = chans[0]

    depth = 1
    for idx in range(len(chans) - 1):
        ci, co = chans[idx], chans[idx + 1]
        g.down.append(Node("res", f"down_layers.{len(g.down)}", ci, co, depth in cfg.attention_depths))
        if (idx + 1) in cut_after:
            g.down.append(Node("down", f"down_layers.{len(g.down)}", co, co))
            depth += 1

    depth = n_depth
    rev = chans[::-1]
    for idx in range(len(rev) - 1):
        ci, co = rev[idx], rev[idx + 1]
        has_attn = depth in cfg.attention_depths
        layer_num = len(chans) - 1 - idx
        g.up.append(Node("res", f"up_layers.{len(g.up)}", 2 * ci, co, has_attn))
        if (layer_num - 1) in cut_after:
            g.up.append(Node("res", f"up_layers.{len(g.up)}", 2 * co, co, has_attn))
            g.up.append(Node("up", f"up_layers.{len(g.up)}", co, co))
            depth -= 1
    g.up.append(Node("res", f"up_layers.{len(g.up)}", 2 * chans[0], chans[0], 1 in cfg.attention_depths))

    top = chans[-1]
    g.mid = [Node("res", "middle_layers.0", top, top, True), Node("res", "middle_layers.1", top, top, False)]
    return g


def _conv2_index(cfg: UNetConfig) -> int:
    # norm_act_drop_conv (models/ddpm.py:25-35): the conv sits at Sequential index 3
    # when a Dropout2d is present (p > 0) and at 2 otherwise.
    return 3 if cfg.dropout > 0 else 2


def param_table(cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, role) for every state_dict entry, in the reference's order.

    role in {"buffer", "conv_w", "conv_b", "lin_w", "lin_b", "gn_w", "gn_b"} with the
    fan-in encoded by the shape.  Order follows nn.Module registration order
    (condition, input_conv, down_layers, up_layers, middle_layers, output_conv --
    models/ddpm.py:211-279).
    """
    g = build_graph(cfg)
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    half = cfg.pos_dim // 2

    def conv(p, ci, co, k):
        out.append((p + ".weight", (co, ci, k, k), "conv_w"))
        out.append((p + ".bias", (co,), "conv_b"))

    def lin(p, ci, co):
        out.append((p + ".weight", (co, ci), "lin_w"))
        out.append((p + ".bias", (co,), "lin_b"))

    def gn(p, c):
        out.append((p + ".weight", (c,), "gn_w"))
        out.append((p + ".bias", (c,), "gn_b"))

    out.append(("condition.0.embeddings", (1, half), "buffer"))
    lin("condition.1", cfg.pos_dim, cfg.emb_dim)
    lin("condition.3", cfg.emb_dim, cfg.emb_dim)
    conv("input_conv", cfg.in_channels, g.base, 3)

    def res(n: Node):
        p = n.prefix
        gn(p + ".conv1.0", n.c_in)
        conv(p + ".conv1.2", n.c_in, n.c_out, 3)
        lin(p + ".condition.0", cfg.emb_dim, n.c_out)
        gn(p + ".conv2.0", n.c_out)
        conv(f"{p}.conv2.{_conv2_index(cfg)}", n.c_out, n.c_out, 3)
        if n.c_in != n.c_out:
            conv(p + ".residual", n.c_in, n.c_out, 1)
        if n.attn:
            gn(p + ".attention.norm", n.c_out)
            conv(p + ".attention.qkv_proj", n.c_out, 3 * n.c_out, 1)
            conv(p + ".attention.proj", n.c_out, n.c_out, 1)

    for seq in (g.down, g.up, g.mid):
        for n in seq:
            if n.kind == "res":
                res(n)
            elif n.kind == "down":
                conv(n.prefix, n.c_in, n.c_out, 3)
            else:
                conv(n.prefix + ".conv", n.c_in, n.c_out, 3)
    gn("output_conv.0", g.base)
    conv("output_conv.2", g.base, cfg.in_channels, 3)
    return out


def sinusoid_freqs(pos_dim: int) -> Tensor:
    """SinusoidalPositionEmbeddings.__init__ (models/ddpm.py:331-334)."""
    half = pos_dim // 2
    step = math.log(10000) / (half - 1)
    return torch.exp(torch.arange(half) * -step).unsqueeze(0)


def make_state_dict(cfg: UNetConfig, seed: int, gn_jitter: bool = True) -> Dict[str, Tensor]:
    """Deterministic synthetic weights from a frozen numpy stream (RandomState/MT19937).

    Magnitudes follow torch's default init (kaiming_uniform(a=sqrt 5) => U(+-1/sqrt(fan_in))
    for conv/linear weights and biases; reference has no custom init, survey 8a-note 13).
    GroupNorm affine is jittered away from (1, 0) so parity tests exercise it.
    The fixtures record only ``seed``; tests/golden/make_golden.py and the tests both
    call this function, so the reference and the build see identical bits.
    """
    rs = np.random.RandomState(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape, role in param_table(cfg):
        if role == "buffer":
            sd[key] = sinusoid_freqs(cfg.pos_dim)
            continue
        if role in ("conv_w", "lin_w"):
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            arr = rs.uniform(-bound, bound, size=shape)
            last_fan_in = fan_in
        elif role in ("conv_b", "lin_b"):
            bound = 1.0 / math.sqrt(last_fan_in)
            arr = rs.uniform(-bound, bound, size=shape)
        elif role == "gn_w":
            arr = 1.0 + (0.2 * rs.standard_normal(size=shape) if gn_jitter else 0.0) * np.ones(shape)
        else:  # gn_b
            arr = (0.1 * rs.standard_normal(size=shape) if gn_jitter else 0.0) * np.ones(shape)
        sd[key] = torch.from_numpy(np.asarray(arr, dtype=np.float32))
    return sd


# --------------------------------------------------------------------------- forward


def time_embedding(sd: Dict[str, Tensor], t: Tensor) -> Tensor:
    """UNet.condition (models/ddpm.py:211-217) incl. the trailing SiLU, and
    SinusoidalPositionEmbeddings.forward (:338-349): sin block first, then cos."""
    arg = t.unsqueeze(1) * sd["condition.0.embeddings"]
    e = torch.cat((arg.sin(), arg.cos()), dim=-1)
    e = F.silu(F.linear(e, sd["condition.1.weight"], sd["condition.1.bias"]))
    e = F.silu(F.linear(e, sd["condition.3.weight"], sd["condition.3.bias"]))
    return e


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def attention_block(sd: Dict[str, Tensor], p: str, x: Tensor, groups: int) -> Tensor:
    """Attention.forward / forward_attention (models/ddpm.py:54-75).

    Single head over H*W tokens; q,k,v are the three channel thirds of the 1x1 qkv
    conv; K is scaled by dim**-0.5 *before* the product (:50,:58).
    """
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".norm", x, groups)
    qkv = F.conv2d(h, sd[p + ".qkv_proj.weight"], sd[p + ".qkv_proj.bias"])
    qkv = qkv.reshape(b, 3 * c, hh * ww).transpose(1, 2)  # b, s, 3c
    q, k, v = qkv[:, :, :c], qkv[:, :, c : 2 * c], qkv[:, :, 2 * c :]
    k = k.transpose(1, 2) * (c**-0.5)
    w = torch.softmax(torch.bmm(q, k), dim=2)
    o = torch.bmm(w, v).transpose(1, 2).reshape(b, c, hh, ww)
    o = F.conv2d(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    return o + x


def res_block(
    sd: Dict[str, Tensor], cfg: UNetConfig, n: Node, x: Tensor, temb: Tensor, drop_mask: Optional[Tensor] = None
) -> Tensor:
    """ResBlock.forward (models/ddpm.py:118-133).

    ``drop_mask`` (B, c_out) holds the Dropout2d multipliers (0 or 1/(1-p)) for the
    conv2 branch (:29, :106); None means eval mode.
    """
    p = n.prefix
    g = cfg.num_groups
    h = F.silu(_gn(sd, p + ".conv1.0", x, g))
    h = F.conv2d(h, sd[p + ".conv1.2.weight"], sd[p + ".conv1.2.bias"], padding=1)
    h = h + F.linear(temb, sd[p + ".condition.0.weight"], sd[p + ".condition.0.bias"])[:, :, None, None]
    h2 = F.silu(_gn(sd, p + ".conv2.0", h, g))
    if drop_mask is not None:
        h2 = h2 * drop_mask[:, :, None, None]
    ck = f"{p}.conv2.{_conv2_index(cfg)}"
    h2 = F.conv2d(h2, sd[ck + ".weight"], sd[ck + ".bias"], padding=1)
    if n.c_in != n.c_out:
        h2 = h2 + F.conv2d(x, sd[p + ".residual.weight"], sd[p + ".residual.bias"])
    else:
        h2 = h2 + x
    if n.attn:
        h2 = attention_block(sd, p + ".attention", h2, g)
    return h2


def res_block_names(cfg: UNetConfig) -> List[str]:
    g = build_graph(cfg)
    return [n.prefix for seq in (g.down, g.mid, g.up) for n in seq if n.kind == "res"]


def unet_forward(
    sd: Dict[str, Tensor],
    cfg: UNetConfig,
    x: Tensor,
    t: Tensor,
    drop_masks: Optional[Dict[str, Tensor]] = None,
    capture: Optional[Dict[str, Tensor]] = None,
) -> Tensor:
    """UNet.forward (models/ddpm.py:281-316).

    x: (B, C, H, W) fp32; t: (B,) or (1,) integer (or float) timesteps.
    drop_masks: per-ResBlock-prefix Dropout2d multipliers for train mode.
    capture: if given, filled with per-module outputs keyed by state_dict prefix.
    """
    g = build_graph(cfg)
    temb = time_embedding(sd, t)
    if capture is not None:
        capture["condition"] = temb

    def keep(name, val):
        if capture is not None:
            capture[name] = val
        return val

    h = keep("input_conv", F.conv2d(x, sd["input_conv.weight"], sd["input_conv.bias"], padding=1))
    skips = [h]
    for n in g.down:
        if n.kind == "res":
            h = res_block(sd, cfg, n, h, temb, None if drop_masks is None else drop_masks[n.prefix])
        else:
            h = F.conv2d(h, sd[n.prefix + ".weight"], sd[n.prefix + ".bias"], stride=2, padding=1)
        skips.append(keep(n.prefix, h))
    for n in g.mid:
        h = keep(n.prefix, res_block(sd, cfg, n, h, temb, None if drop_masks is None else drop_masks[n.prefix]))
    for n in g.up:
        if n.kind == "res":
            h = torch.cat([h, skips.pop()], dim=1)  # x first, skip second (:310)
            h = res_block(sd, cfg, n, h, temb, None if drop_masks is None else drop_masks[n.prefix])
        else:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")  # nn.Upsample default (:161)
            h = F.conv2d(h, sd[n.prefix + ".conv.weight"], sd[n.prefix + ".conv.bias"], padding=1)
        keep(n.prefix, h)
    h = F.silu(_gn(sd, "output_conv.0", h, cfg.num_groups))
    return F.conv2d(h, sd["output_conv.2.weight"], sd["output_conv.2.bias"], padding=1)


def make_drop_masks(cfg: UNetConfig, batch: int, seed: int) -> Dict[str, Tensor]:
    """Dropout2d multipliers (0 or 1/(1-p)) per ResBlock from a frozen numpy stream."""
    rs = np.random.RandomState(seed)
    g = build_graph(cfg)
    masks = {}
    keep_p = 1.0 - cfg.dropout
    for seq in (g.down, g.mid, g.up):
        for n in seq:
            if n.kind == "res":
                m = (rs.uniform(size=(batch, n.c_out)) < keep_p).astype(np.float32) / keep_p
                masks[n.prefix] = torch.from_numpy(m)
    return masks
