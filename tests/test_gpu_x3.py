"""-m gpu: the accurate mode (precision="bf16x3": fp32 tensors, every convolution product as three bf16 MFMA passes over hi/lo
splits).  north_star asks 1e-3 of the reference for the bf16 path; a single-pass bf16 pipeline cannot reach that (one rounding is
2^-9), this mode does - at a third of the bf16 matrix rate instead of the fp32 MFMA's sixteenth."""

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import unet as O

from tests import gpu_util as G

pytestmark = pytest.mark.gpu

X3_ATOL = 1e-3  # north_star: 1e-3 for the reduced-precision matrix path


def _build(cfg, seed, precision):
    import dmme_amd

    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                        cfg.attention_depths, precision=precision)
    net.load_state_dict(O.make_state_dict(cfg, seed), strict=True)
    return net.cuda().eval()


def test_full_unet_bf16x3_within_1e3_of_reference_golden(golden):
    """the default 32.4 M-parameter UNet against the reference's own output (tests/golden/unet_full.npz: full_y_one / full_y_per)"""
    g = golden("unet_full")
    net = _build(O.UNetConfig(), int(g["full_seed"]), "bf16x3")
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32)).cuda()
    with torch.no_grad():
        y1 = net(x, torch.from_numpy(g["full_t_one"]).cuda()).cpu().numpy()
        y2 = net(x, torch.from_numpy(g["full_t_per"]).cuda()).cpu().numpy()
    e1, e2 = float(np.abs(y1 - g["full_y_one"]).max()), float(np.abs(y2 - g["full_y_per"]).max())
    print(f"bf16x3 full UNet max|err| vs reference: {e1:.3e} (t one), {e2:.3e} (t per image); |y|max {np.abs(g['full_y_one']).max():.3f}")
    assert e1 <= X3_ATOL and e2 <= X3_ATOL
    # and far inside it: the three-pass product drops ~2^-16 per term
    assert e1 <= 2e-4 and e2 <= 2e-4


def test_bf16x3_batch128_rows_match_reference_golden(golden):
    """the benchmark batch: every tile shape B = 128 selects, rows against the golden pair"""
    g = golden("unet_full")
    net = _build(O.UNetConfig(), int(g["full_seed"]), "bf16x3")
    base = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    with torch.no_grad():
        y = net(base.repeat(64, 1, 1, 1).cuda(), torch.from_numpy(g["full_t_one"]).cuda()).cpu()
    rows = y.reshape(64, 2, 3, 32, 32)
    assert torch.equal(rows, rows[:1].expand_as(rows))
    assert float((rows[0] - torch.from_numpy(g["full_y_one"])).abs().max()) <= X3_ATOL


@pytest.mark.parametrize("shape", ["c128_32", "c256_16_cat", "c256_8", "c256_4", "s2", "up", "k1_qkv", "k1_res"])
def test_conv_bf16x3_vs_fp64(shape):
    """single convolutions through dmme_conv2d with dtype DMME_BF16X3 against an fp64 torch convolution of the same fp32 operands:
    <= 3e-5 of the output's max (2^-16 per product, accumulated over K up to 4608), where single-pass bf16 sits at ~4e-3."""
    from dmme_amd import _lib
    import torch.nn.functional as F

    spec = {
        "c128_32": dict(N=8, C=128, Co=128, H=32, k=3),
        "c256_16_cat": dict(N=8, C=256, C2=256, Co=256, H=16, k=3),
        "c256_8": dict(N=16, C=256, Co=256, H=8, k=3),
        "c256_4": dict(N=32, C=256, Co=256, H=4, k=3),
        "s2": dict(N=8, C=128, Co=128, H=32, k=3, stride=2),
        "up": dict(N=8, C=256, Co=256, H=8, k=3, up=True),
        "k1_qkv": dict(N=8, C=256, Co=768, H=16, k=1),
        "k1_res": dict(N=8, C=512, Co=256, H=16, k=1),
    }[shape]
    N, Cc, Co, H, k = spec["N"], spec["C"], spec["Co"], spec["H"], spec["k"]
    x1 = synth.normal(1, (N, Cc, H, H)).cuda()
    x2 = synth.normal(2, (N, spec["C2"], H, H)).cuda() if "C2" in spec else None
    cin = Cc + (spec.get("C2") or 0)
    w = (synth.normal(3, (Co, cin, k, k)) / (cin * k * k) ** 0.5).cuda()
    b = synth.normal(4, (Co,)).cuda()
    got = G.conv2d(_lib.BF16X3, x1, w, b, x2=x2, stride=spec.get("stride", 1), upsample=spec.get("up", False))
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    if spec.get("up"):
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xin.double(), w.double(), b.double(), stride=spec.get("stride", 1), padding=k // 2)
    err = float((got.double() - want).abs().max()) / float(want.abs().max())
    one_pass = G.conv2d(_lib.BF16, x1, w, b, x2=x2, stride=spec.get("stride", 1), upsample=spec.get("up", False))
    err1 = float((one_pass.double() - want).abs().max()) / float(want.abs().max())
    print(f"{shape}: bf16x3 rel err {err:.2e}, single-pass bf16 {err1:.2e}")
    assert err <= 3e-5
    assert err1 > 10 * err  # the mode buys what it costs


def test_bf16x3_training_step_gradients_vs_fp32():
    """the mode also trains (data gradients through the same three-pass kernels, weight gradients on the fp32 MFMA path):
    loss and flat gradient against precision="fp32" on the same batch, injected t / noise / masks"""
    import dmme_amd

    cfg = O.UNetConfig()
    outs = {}
    for prec in ("fp32", "bf16x3"):
        net = _build(cfg, 9, prec)
        net.train()
        masks = O.make_drop_masks(cfg, 4, 7)
        net.inject_dropout_masks(torch.cat([masks[k].reshape(-1) for k in O.res_block_names(cfg)]).cuda())
        ddpm = dmme_amd.DDPM(net, 1000).cuda()
        loss = ddpm.training_step(synth.uniform(1, (4, 3, 32, 32)).cuda(), t=torch.tensor([5, 300, 650, 999]).cuda(), noise=synth.normal(2, (4, 3, 32, 32)).cuda())
        loss.backward()
        outs[prec] = (float(loss), net.flat_grad().clone())
    assert abs(outs["fp32"][0] - outs["bf16x3"][0]) <= 1e-4 * abs(outs["fp32"][0])
    g32, g3 = outs["fp32"][1], outs["bf16x3"][1]
    assert float((g32 - g3).norm() / g32.norm()) <= 1e-3


@pytest.mark.parametrize("N,S,C,heads", [(3, 256, 256, 1), (2, 256, 128, 1), (3, 256, 256, 4), (5, 64, 256, 4), (2, 128, 64, 1)])
def test_attention_bf16x3_vs_fp64(N, S, C, heads):
    """the three-pass MFMA attention kernel (fp32 in / out) against an fp64 softmax(q k^T C^-0.5) v of the same fp32 operands, single- and
    multi-head (the IDDPM head view with the reference's batch-mixing merge): within 2e-5 of the output's max"""
    from dmme_amd import _lib

    qkv = synth.normal(S + C + heads, (N, S, 3 * C)) * 1.5
    d = C // heads
    x = qkv.double().reshape(N, S, heads, 3 * d).permute(0, 2, 1, 3).reshape(N * heads, S, 3 * d)
    q, k, v = x[..., :d], x[..., d : 2 * d], x[..., 2 * d :]
    w = torch.softmax(torch.bmm(q, k.transpose(1, 2) * C**-0.5), dim=2)
    want = torch.bmm(w, v).reshape(heads, N, S, d).permute(1, 2, 0, 3).reshape(N, S, C)  # rows b*heads + h re-read as (head', b')
    qd = qkv.cuda().contiguous()
    out = torch.empty((N, S, C), dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().dmme_attention_heads(_lib.BF16X3, _lib.ptr(qd), N, S, C, heads, _lib.ptr(out), 0, _lib.stream_ptr()))
    err = float((out.double().cpu() - want).abs().max() / want.abs().max())
    print(f"attention bf16x3 N={N} S={S} C={C} heads={heads}: rel err {err:.2e}")
    assert err <= 2e-5
