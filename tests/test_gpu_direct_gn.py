"""-m gpu: GroupNorms finished by the producing convolution's epilogue (8x8 / 4x4 maps at the benchmark batch: tiles of whole
images - conv_epilogue_store_direct).  The path only exists at batches that fill the chip without split-K, so the small-batch golden
tests do not reach it: here the default UNet runs at B = 128 against the reference's golden rows, and against the same network with
the path switched off (DMME_NO_GN_DIRECT, read when a plan is built)."""

import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import unet as O

pytestmark = pytest.mark.gpu


def _build(seed, precision, direct):
    import dmme_amd

    cfg = O.UNetConfig()
    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                        cfg.attention_depths, precision=precision)
    net.load_state_dict(O.make_state_dict(cfg, seed), strict=True)
    net = net.cuda().eval()
    return net


def _run(net, x, t, direct, switch="DMME_NO_GN_DIRECT"):
    # these epilogue / parameter-fill paths serve the 8x8 and 4x4 levels only where the level engine (round 3, csrc/lvl_engine.hip;
    # tests/test_gpu_level.py) does not take those levels: it is switched off here, so the older route stays covered
    os.environ["DMME_NO_LVL"] = "1"
    if direct:
        os.environ.pop(switch, None)
    else:
        os.environ[switch] = "1"
    try:
        with torch.no_grad():
            y = net(x, t).float().cpu()
        n = net._last_plan.lib.dmme_unet_plan_num_launches(net._last_plan.h)
    finally:
        os.environ.pop(switch, None)
        os.environ.pop("DMME_NO_LVL", None)
    return y, n


@pytest.mark.parametrize("precision,atol_ref,rel_ab,fewer", [("fp32", 1e-5, 2e-6, 8), ("bf16", 1.7e-2, 1.0e-2, 20)])
def test_batch128_direct_groupnorm_vs_reference_and_vs_launched_norms(golden, precision, atol_ref, rel_ab, fewer):
    g = golden("unet_full")
    seed = int(g["full_seed"])
    base = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    x = base.repeat(64, 1, 1, 1).cuda()
    t = torch.from_numpy(g["full_t_one"]).cuda()
    ya, na = _run(_build(seed, precision, True), x, t, True)
    yb, nb = _run(_build(seed, precision, False), x, t, False)
    # the path is on: bf16 - the norms of the 8x8 and 4x4 levels whose sources all come from whole-image tiles are no launches any
    # more; fp32 - the 8x8 level only (its 4x4 convs take split-K, whose finish kernel sees no whole image)
    assert nb - na >= fewer, (na, nb)
    rows = ya.reshape(64, 2, 3, 32, 32)
    assert torch.equal(rows, rows[:1].expand_as(rows))  # every image pair took the same arithmetic
    ref = torch.from_numpy(g["full_y_one"])
    e_ref = float((rows[0] - ref).abs().max())
    e_ab = float((ya - yb).pow(2).mean().sqrt() / yb.pow(2).mean().sqrt())
    print(f"{precision}: launches {nb} -> {na}; max|err| vs reference {e_ref:.3e}; relative rms between the two paths {e_ab:.3e}")
    assert e_ref <= atol_ref
    assert e_ab <= rel_ab


def test_batch128_groupnorm_finished_by_consumer_vs_reference_and_vs_finalize_launches(golden):
    """norms whose statistics are the producers' partials, merged by the CONSUMING conv's parameter fill (gn_in_scale_shift: the
    wave-specialised 3x3 kernel, the activation-stationary 1x1 kernel) instead of a finalize launch: B = 128 against the reference's
    golden rows and against the same network with the launches (DMME_NO_GN_IN, read when a plan is built)"""
    g = golden("unet_full")
    seed = int(g["full_seed"])
    base = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    x = base.repeat(64, 1, 1, 1).cuda()
    t = torch.from_numpy(g["full_t_one"]).cuda()
    ya, na = _run(_build(seed, "bf16", True), x, t, True, "DMME_NO_GN_IN")
    yb, nb = _run(_build(seed, "bf16", False), x, t, False, "DMME_NO_GN_IN")
    assert nb - na >= 10, (na, nb)  # 14 on the default UNet: every norm in front of those two kernels at the 32x32 / 16x16 levels
    rows = ya.reshape(64, 2, 3, 32, 32)
    assert torch.equal(rows, rows[:1].expand_as(rows))  # every image pair took the same arithmetic
    ref = torch.from_numpy(g["full_y_one"])
    e_ref = float((rows[0] - ref).abs().max())
    e_ab = float((ya - yb).pow(2).mean().sqrt() / yb.pow(2).mean().sqrt())
    print(f"launches {nb} -> {na}; max|err| vs reference {e_ref:.3e}; relative rms between the two paths {e_ab:.3e}")
    assert e_ref <= 1.7e-2  # the bf16 network's max-abs bound (tests/test_gpu_unet.py: BF16_MAX_ABS)
    assert e_ab <= 1.0e-2    # (the two merges differ in rounding only: equal-count batch form vs sequential Chan updates)


def test_batch128_training_forward_direct_groupnorm_saves_mean_rstd():
    """training mode (per-image time rows inside a 4-image tile, Dropout2d masks in the pre-activated input): the loss of one step
    with and without the path, same seed"""
    import dmme_amd

    def loss(direct):
        os.environ["DMME_NO_LVL"] = "1"
        if direct:
            os.environ.pop("DMME_NO_GN_DIRECT", None)
        else:
            os.environ["DMME_NO_GN_DIRECT"] = "1"
        try:
            torch.manual_seed(0)
            net = dmme_amd.UNet(precision="bf16").cuda().train()
            x = torch.randn(128, 3, 32, 32, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
            t = torch.arange(128, device="cuda") * 7 % 1000
            y = net(x, t)
            l = (y.float() ** 2).mean()
            l.backward()
            gsq = float(net.flat_grad().float().pow(2).sum())
            return float(l.detach()), gsq
        finally:
            os.environ.pop("DMME_NO_GN_DIRECT", None)
            os.environ.pop("DMME_NO_LVL", None)

    la, ga = loss(True)
    lb, gb = loss(False)
    print(f"training forward loss {la:.6f} (direct) vs {lb:.6f}; |grad|^2 {ga:.4e} vs {gb:.4e}")
    assert abs(la - lb) <= 2e-3 * abs(lb)
    assert abs(ga - gb) <= 3e-2 * gb
