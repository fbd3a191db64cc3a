"""-m gpu: Improved-DDPM path -- the HIP UNet with scale-shift ResBlocks and the reference's 4-head
attention, the learned-variance sampler and the hybrid loss -- against the golden vectors the reference produced
(tests/golden/iddpm_*.npz) and the CPU oracle (oracle/iddpm.py) on the same inputs."""

import numpy as np
import pytest
import torch

from oracle import iddpm as OI
from oracle import synth

pytestmark = pytest.mark.gpu

FP32_ATOL = 1e-5
BF16_REL_RMS = 1.5e-2  # same bounds as the DDPM network (tests/test_gpu_unet.py)
BF16_MAX_ABS = 6e-2


def _build(cfg, seed, precision, train=False):
    import dmme_amd
    from dmme_amd.models import iddpm

    net = iddpm.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                     cfg.attention_depths, precision=precision, num_heads=cfg.num_heads)
    sd = OI.make_state_dict(cfg, seed)
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    net.cuda()
    net.train(train)
    return net, sd


@pytest.mark.parametrize("tag,cfg", [("tiny", OI.TINY), ("attn", OI.TINY_ATTN)])
def test_unet_fp32_vs_reference_golden(golden, tag, cfg):
    g = golden("iddpm_unet")
    net, sd = _build(cfg, int(g[f"{tag}_seed"]), "fp32")
    with torch.no_grad():
        for c in range(int(g[f"{tag}_ncases"])):
            B = int(g[f"{tag}_case{c}_B"])
            x = synth.normal(int(g[f"{tag}_case{c}_xseed"]), (B, 3, 32, 32)).cuda()
            y = net(x, torch.from_numpy(g[f"{tag}_case{c}_t"]).cuda())
            assert y.shape == (B, 6, 32, 32)
            np.testing.assert_allclose(y.cpu().numpy(), g[f"{tag}_case{c}_y"], atol=FP32_ATOL, rtol=0, err_msg=f"case {c}")
        # B = 2: per-module activations (the head merge mixes the two samples)
        x = synth.normal(int(g[f"{tag}_acts_xseed"]), (2, 3, 32, 32)).cuda()
        y = net(x, torch.from_numpy(g[f"{tag}_acts_t"]).cuda())
        worst = {}
        for k in [k for k in g.files if k.startswith(f"{tag}_act::")]:
            name = k.split("::")[1]
            want = g[k]
            got = net.debug_activation(name).cpu().numpy()[: want.size].reshape(want.shape)
            worst[name] = float(np.abs(got - want).max())
        bad = {k: v for k, v in worst.items() if v > FP32_ATOL}
        assert not bad, f"activations off: {bad}"
        np.testing.assert_allclose(y.cpu().numpy(), g[f"{tag}_acts_y"], atol=FP32_ATOL, rtol=0)
    # train mode, injected Dropout2d masks
    B, mseed, xseed = (int(v) for v in g[f"{tag}_train_meta"])
    masks = OI.make_drop_masks(cfg, B, mseed)
    flat = torch.cat([masks[k].reshape(-1) for k in OI.res_block_names(cfg)])
    net.train(True)
    net.inject_dropout_masks(flat.cuda())
    try:
        with torch.no_grad():
            y = net(synth.normal(xseed, (B, 3, 32, 32)).cuda(), torch.from_numpy(g[f"{tag}_train_t"]).cuda())
    finally:
        net.inject_dropout_masks(None)
        net.train(False)
    np.testing.assert_allclose(y.cpu().numpy(), g[f"{tag}_train_y"], atol=FP32_ATOL, rtol=0)


def test_unet_full_fp32_vs_reference_golden(golden):
    g = golden("iddpm_unet")
    cfg = OI.IUNetConfig()
    net, sd = _build(cfg, int(g["full_seed"]), "fp32")
    assert sum(p.numel() for p in net.parameters()) == 36_168_070
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32)).cuda()
    with torch.no_grad():
        y = net(x, torch.from_numpy(g["full_t_one"]).cuda())
    for k in [k for k in g.files if k.startswith("full_actdigest::")]:
        name = k.split("::")[1]
        got = net.debug_activation(name).cpu()
        if name == "condition":
            got = got[:512]
        ok, err = synth.digest_close(got, g[k], atol=5e-5, rtol=1e-5)
        assert ok, (k, err)
    np.testing.assert_allclose(y.cpu().numpy(), g["full_y_one"], atol=5e-5, rtol=0)


def test_unet_full_bf16_vs_oracle():
    cfg = OI.IUNetConfig()
    net, sd = _build(cfg, 41, "bf16")
    B = 4
    x = synth.normal(9, (B, 3, 32, 32))
    t = torch.tensor([900])
    want = OI.unet_forward(sd, cfg, x, t)
    with torch.no_grad():
        got = net(x.cuda(), t.cuda()).cpu()
    err = (got - want).abs()
    rel_rms = float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    assert rel_rms < BF16_REL_RMS and float(err.max()) < BF16_MAX_ABS * max(1.0, float(want.abs().max())), (rel_rms, float(err.max()))


def test_unet_64x64_config4_fp32_vs_oracle():
    """BASELINE configs[3] geometry: attention_depths=(3, 4) at 64x64 (S = 256 and 64, d = 64), small batch."""
    cfg = OI.IUNetConfig(attention_depths=(3, 4))
    net, sd = _build(cfg, 43, "fp32")
    B = 2
    x = synth.normal(10, (B, 3, 64, 64))
    t = torch.tensor([17, 3011])
    want = OI.unet_forward(sd, cfg, x, t)
    with torch.no_grad():
        got = net(x.cuda(), t.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=5e-5, rtol=0)
