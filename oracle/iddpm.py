"""Functional CPU restatement of the reference's Improved-DDPM path (test infrastructure only):
the learned-variance UNet (src/dmme/models/iddpm.py), the IDDPM process
(src/dmme/diffusion_models/iddpm.py) and its equations (src/dmme/equations/iddpm/*.py).

Same conventions as oracle/unet.py and oracle/diffusion.py: a plain ``dict[str, Tensor]`` with
the reference's state_dict keys, injected randomness, one citation per function (paths relative
to /root/reference).  Pinned by tests/golden/iddpm_*.npz, which the reference itself produced
(tests/golden/make_golden.py)."""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import unet as U

Tensor = torch.Tensor


@dataclass(frozen=True)
class IUNetConfig:
    """Constructor arguments of the reference iddpm.UNet (models/iddpm.py:139-149) plus the
    head count its ResBlock hard-codes (``num_heads=4``, :82)."""

    in_channels: int = 3
    pos_dim: int = 128
    emb_dim: int = 512
    num_groups: int = 32
    dropout: float = 0.3
    channels_per_depth: Tuple[int, ...] = (128, 256, 256, 256)
    num_blocks: int = 2
    attention_depths: Tuple[int, ...] = (2, 3)
    num_heads: int = 4


# small enough for fixtures; channels divisible by the 4 heads and the 2 groups
TINY = IUNetConfig(pos_dim=4, emb_dim=8, num_groups=2, channels_per_depth=(4, 8, 16, 32), num_blocks=3)
# attention on every depth that the tiny maps allow: covers S = 256, 64 and 16 with 4 heads
TINY_ATTN = IUNetConfig(pos_dim=8, emb_dim=16, num_groups=2, dropout=0.3, channels_per_depth=(8, 16, 16), num_blocks=1, attention_depths=(2, 3))


def build_graph(cfg: IUNetConfig) -> U.Graph:
    """Layer list of iddpm.UNet.__init__ (models/iddpm.py:152-225): the same walk as the DDPM
    UNet, including the never-true ``down_layers[-1] == len(channels) - 1`` test (:191)."""
    return U.build_graph(cfg)  # duck-typed: only channels_per_depth / num_blocks / attention_depths are read


def _conv2_index(cfg: IUNetConfig) -> int:
    # conv2 = norm_act_drop_conv(...)[1:] (models/iddpm.py:94): slicing an nn.Sequential keeps the
    # original indices as keys, so the conv stays "conv2.3" with a Dropout2d and "conv2.2" without.
    return 3 if cfg.dropout > 0 else 2


def param_table(cfg: IUNetConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, role) for every state_dict entry of iddpm.UNet, in registration order
    (ResBlock: conv1, norm, condition, conv2, residual, attention -- models/iddpm.py:85-104)."""
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

    def res(n: U.Node):
        p = n.prefix
        gn(p + ".conv1.0", n.c_in)
        conv(p + ".conv1.2", n.c_in, n.c_out, 3)  # conv1 is built with p=0.0: no Dropout2d slot (:85)
        gn(p + ".norm", n.c_out)
        lin(p + ".condition.0", cfg.emb_dim, 2 * n.c_out)
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
    conv("output_conv.2", g.base, 2 * cfg.in_channels, 3)  # (eps, v) stacked along channels (:228-230)
    return out


def make_state_dict(cfg: IUNetConfig, seed: int, gn_jitter: bool = True) -> Dict[str, Tensor]:
    """Deterministic synthetic weights, same recipe as oracle.unet.make_state_dict."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, Tensor] = {}
    last_fan_in = 1
    for key, shape, role in param_table(cfg):
        if role == "buffer":
            sd[key] = U.sinusoid_freqs(cfg.pos_dim)
            continue
        if role in ("conv_w", "lin_w"):
            last_fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(last_fan_in)
            arr = rs.uniform(-bound, bound, size=shape)
        elif role in ("conv_b", "lin_b"):
            bound = 1.0 / math.sqrt(last_fan_in)
            arr = rs.uniform(-bound, bound, size=shape)
        elif role == "gn_w":
            arr = 1.0 + (0.2 * rs.standard_normal(size=shape) if gn_jitter else 0.0) * np.ones(shape)
        else:
            arr = (0.1 * rs.standard_normal(size=shape) if gn_jitter else 0.0) * np.ones(shape)
        sd[key] = torch.from_numpy(np.asarray(arr, dtype=np.float32))
    return sd


# --------------------------------------------------------------------------- forward


def multi_head_attention(sd: Dict[str, Tensor], p: str, x: Tensor, groups: int, heads: int) -> Tensor:
    """MultiHeadAttention.forward / forward_attention (models/iddpm.py:35-59), as shipped:

    * heads are split with "b (head c) h w -> (b head) (h w) c" (:38): head h owns the 3C/heads
      consecutive qkv channels [h*3d, (h+1)*3d), chunked *inside that slice* into q, k, v (:39);
    * K is scaled by dim**-0.5 with dim = the full channel count, not the head width (:32,:40);
    * the merge reads the batch axis as "(head b)" (:44-46) although it was built as "(b head)":
      row i = b*heads + head of the attention output lands at batch i % B, head i // B.  For B > 1
      this mixes samples; restated exactly."""
    b, c, hh, ww = x.shape
    d = c // heads
    h = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    qkv = F.conv2d(h, sd[p + ".qkv_proj.weight"], sd[p + ".qkv_proj.bias"])  # b, 3c, h, w
    qkv = qkv.reshape(b, heads, 3 * d, hh * ww).permute(0, 1, 3, 2).reshape(b * heads, hh * ww, 3 * d)
    q, k, v = qkv[:, :, :d], qkv[:, :, d : 2 * d], qkv[:, :, 2 * d :]
    k = k.transpose(1, 2) * (c**-0.5)
    w = torch.softmax(torch.bmm(q, k), dim=2)
    o = torch.bmm(w, v)  # (b*heads, s, d), row index = b*heads + head
    o = o.reshape(heads, b, hh * ww, d)  # ... re-read as (head', b')
    o = o.permute(1, 0, 3, 2).reshape(b, c, hh, ww)
    o = F.conv2d(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    return o + x


def res_block(sd: Dict[str, Tensor], cfg: IUNetConfig, n: U.Node, x: Tensor, temb: Tensor, drop_mask: Optional[Tensor] = None) -> Tensor:
    """iddpm.ResBlock.forward (models/iddpm.py:106-122): scale-shift conditioning.  The time
    projection has 2*c_out outputs chunked as (shift, scale) (:117); h = GN(conv1(x)) * (scale+1) + shift,
    then SiLU -> Dropout2d -> conv (:94,:119)."""
    p = n.prefix
    g = cfg.num_groups
    h = F.silu(F.group_norm(x, g, sd[p + ".conv1.0.weight"], sd[p + ".conv1.0.bias"], eps=1e-5))
    h = F.conv2d(h, sd[p + ".conv1.2.weight"], sd[p + ".conv1.2.bias"], padding=1)
    cond = F.linear(temb, sd[p + ".condition.0.weight"], sd[p + ".condition.0.bias"])[:, :, None, None]
    shift, scale = cond.chunk(2, dim=1)
    h = F.group_norm(h, g, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5) * (scale + 1) + shift
    h = F.silu(h)
    if drop_mask is not None:
        h = h * drop_mask[:, :, None, None]
    ck = f"{p}.conv2.{_conv2_index(cfg)}"
    h = F.conv2d(h, sd[ck + ".weight"], sd[ck + ".bias"], padding=1)
    if n.c_in != n.c_out:
        h = h + F.conv2d(x, sd[p + ".residual.weight"], sd[p + ".residual.bias"])
    else:
        h = h + x
    if n.attn:
        h = multi_head_attention(sd, p + ".attention", h, g, cfg.num_heads)
    return h


def res_block_names(cfg: IUNetConfig) -> List[str]:
    g = build_graph(cfg)
    return [n.prefix for seq in (g.down, g.mid, g.up) for n in seq if n.kind == "res"]


def make_drop_masks(cfg: IUNetConfig, batch: int, seed: int) -> Dict[str, Tensor]:
    return U.make_drop_masks(cfg, batch, seed)


def unet_forward(
    sd: Dict[str, Tensor],
    cfg: IUNetConfig,
    x: Tensor,
    t: Tensor,
    drop_masks: Optional[Dict[str, Tensor]] = None,
    capture: Optional[Dict[str, Tensor]] = None,
) -> Tensor:
    """iddpm.UNet.forward (models/iddpm.py:232-265): returns (B, 2*in_channels, H, W)."""
    g = build_graph(cfg)
    temb = U.time_embedding(sd, t)
    if capture is not None:
        capture["condition"] = temb

    def keep(name, val):
        if capture is not None:
            capture[name] = val
        return val

    def dm(n):
        return None if drop_masks is None else drop_masks[n.prefix]

    h = keep("input_conv", F.conv2d(x, sd["input_conv.weight"], sd["input_conv.bias"], padding=1))
    skips = [h]
    for n in g.down:
        if n.kind == "res":
            h = res_block(sd, cfg, n, h, temb, dm(n))
        else:
            h = F.conv2d(h, sd[n.prefix + ".weight"], sd[n.prefix + ".bias"], stride=2, padding=1)
        skips.append(keep(n.prefix, h))
    for n in g.mid:
        h = keep(n.prefix, res_block(sd, cfg, n, h, temb, dm(n)))
    for n in g.up:
        if n.kind == "res":
            h = torch.cat([h, skips.pop()], dim=1)
            h = res_block(sd, cfg, n, h, temb, dm(n))
        else:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[n.prefix + ".conv.weight"], sd[n.prefix + ".conv.bias"], padding=1)
        keep(n.prefix, h)
    h = F.silu(F.group_norm(h, cfg.num_groups, sd["output_conv.0.weight"], sd["output_conv.0.bias"], eps=1e-5))
    return F.conv2d(h, sd["output_conv.2.weight"], sd["output_conv.2.bias"], padding=1)


# --------------------------------------------------------------------------- process


def cosine_alpha_bar(timesteps: int, offset: float = 0.008) -> Tensor:
    """cosine_schedule (equations/iddpm/iddpm.py:6-20): f(t)/f(0) with
    f(t) = cos((t/T + s)/(1 + s) * pi/2)^2, t = 0..T (int64 arange promoted to fp32 by the division)."""

    def f(t):
        return torch.cos((t / timesteps + offset) / (1 + offset) * math.pi / 2) ** 2

    t = torch.arange(0, timesteps + 1)
    return f(t) / f(torch.tensor([0], dtype=torch.float32))


def schedule_tables(timesteps: int, schedule: str = "cosine", offset: float = 0.008, start: float = 1e-4, end: float = 0.02):
    """IDDPM.__init__ (diffusion_models/iddpm.py:30-60): (beta, alpha, alpha_bar), each (T+1,).
    cosine: beta = clip(1 - abar[1:]/abar[:-1], 0, 0.999) padded in front with **1** (:51-52), so
    alpha[0] = 0; linear: the DDPM tables of the parent constructor; anything else raises."""
    if schedule == "cosine":
        abar = cosine_alpha_bar(timesteps, offset)
        beta = torch.clip(1 - abar[1:] / abar[:-1], 0, 0.999)
        beta = torch.cat([torch.ones(1), beta])
        return beta, 1 - beta, abar
    if schedule != "linear":
        raise NotImplementedError
    from .diffusion import alpha_tables, linear_beta

    beta = linear_beta(timesteps, start, end)
    alpha, abar = alpha_tables(beta)
    return beta, alpha, abar


def interpolate_variance(v: Tensor, beta_t: Tensor, beta_tilde_t: Tensor) -> Tensor:
    """equations/iddpm/losses.py:34-37: exp(v log beta + (1 - v) log max(beta_tilde, 1e-12)); v is the raw
    network output (no squashing)."""
    return torch.exp(v * torch.log(beta_t) + (1 - v) * torch.log(beta_tilde_t.clamp(1e-12)))


def forward_model(out: Tensor, beta_t: Tensor, abar_t: Tensor, abar_prev: Tensor) -> Tuple[Tensor, Tensor]:
    """IDDPM.forward_model after the network call (diffusion_models/iddpm.py:152-164)."""
    eps, v = out.chunk(2, dim=1)
    beta_tilde = (1 - abar_prev) / (1 - abar_t) * beta_t
    return eps, interpolate_variance(v, beta_t, beta_tilde)


def _col(tab: Tensor, t: Tensor) -> Tensor:
    return tab[t].reshape(-1, 1, 1, 1)


def sampling_step(out: Tensor, x_t: Tensor, t: int, z: Tensor, tabs) -> Tensor:
    """IDDPM.sampling_step (diffusion_models/iddpm.py:118-150) for a scalar timestep: mean of
    ddpm.reverse_process, std = sqrt(learned variance); the draw is discarded at t == 1."""
    beta, alpha, abar = tabs
    tt = torch.tensor([t])
    b, a, ab, abp = _col(beta, tt), _col(alpha, tt), _col(abar, tt), _col(abar, tt - 1)
    eps, var = forward_model(out, b, ab, abp)
    mean = 1 / torch.sqrt(a) * (x_t - b / torch.sqrt(1 - ab) * eps)
    if t == 1:
        return mean
    return mean + torch.sqrt(var) * z


def _normal_cdf(x, mu, sigma):
    return 0.5 * (1 + torch.erf((x - mu) / sigma / math.sqrt(2)))  # torch.distributions.Normal.cdf


def loss_vlb(eps: Tensor, var: Tensor, x_t: Tensor, t: Tensor, x_0: Tensor, b, a, ab, abp) -> Tensor:
    """equations/iddpm/losses.py:40-98: per-element L_vlb with stop-gradient on the predicted noise.
    t == 1 rows: discrete NLL of x_0 in bins of +-1/255 (:9-20); other rows: KL(q(x_{t-1}|x_t,x_0) || p_theta)
    (:23-31, torch's kl_normal_normal); mean over all elements."""
    mean = 1 / torch.sqrt(a) * (x_t - b / torch.sqrt(1 - ab) * eps.detach())
    std = torch.sqrt(var)
    rows = []
    m1 = t == 1
    if m1.any():
        mu, sg, x0 = mean[m1], std[m1], x_0[m1]
        hi = torch.where(x0 < 1, _normal_cdf(x0 + 1 / 255, mu, sg), torch.ones_like(x0))
        lo = torch.where(x0 > -1, _normal_cdf(x0 - 1 / 255, mu, sg), torch.zeros_like(x0))
        rows.append(-torch.log((hi - lo).clamp(1e-12)))
    m2 = ~m1
    if m2.any():
        bq, aq, abq, abpq = b[m2], a[m2], ab[m2], abp[m2]
        q_mean = torch.sqrt(abpq) * bq / (1 - abq) * x_0[m2] + torch.sqrt(aq) * (1 - abpq) / (1 - abq) * x_t[m2]
        q_std = torch.sqrt((1 - abpq) / (1 - abq) * bq)
        ratio = (q_std / std[m2]) ** 2
        t1 = ((q_mean - mean[m2]) / std[m2]) ** 2
        rows.append(0.5 * (ratio + t1 - 1 - ratio.log()))
    return (torch.cat(rows, dim=0) if len(rows) > 1 else rows[0]).mean()


def training_loss(
    model: Callable[[Tensor, Tensor], Tensor], x_0: Tensor, t: Tensor, z: Tensor, tabs, loss_type: str = "hybrid", gamma: float = 0.001
):
    """IDDPM.training_step (diffusion_models/iddpm.py:62-116) with t and z injected.  "hybrid":
    L_simple + gamma L_vlb; "vlb": L_vlb; any other loss_type falls off the end of the reference
    function and returns None (restated)."""
    from .diffusion import q_sample

    beta, alpha, abar = tabs
    x_t, q_mean, q_std = q_sample(x_0, abar[t], z)
    x_t = x_t.detach()
    b, a, ab, abp = _col(beta, t), _col(alpha, t), _col(abar, t), _col(abar, t - 1)
    eps, var = forward_model(model(x_t, t), b, ab, abp)
    if loss_type not in ("hybrid", "vlb"):
        return None
    vlb = loss_vlb(eps, var, x_t, t, x_0, b, a, ab, abp)
    if loss_type == "vlb":
        return vlb
    target = (x_t - q_mean) / q_std
    return torch.mean((target - eps) ** 2) + gamma * vlb
