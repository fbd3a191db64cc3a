"""-m gpu: every A/B route switch the library still reads (a dozen product switches as variables of their own, the rest as
keys of DMME_DEBUG_ROUTE - tests/gpu_util.py: route_env translates the historical names) selects a kernel route, never a CPU path - and no route
may rot: each switch is set, a fresh plan is built (the switches are read when a plan is built or a kernel is launched) and the default
UNet is held against the default route on the same inputs - the forward switches at the benchmark batch against the reference's golden
rows as well, the backward switches through the flat gradient of one training step."""

import os

import pytest
import torch

from oracle import synth
from oracle import unet as O

pytestmark = pytest.mark.gpu

# (switches set together, note).  Routes of the 8x8 / 4x4 levels only exist with the level engine off.
FORWARD = [
    ({"DMME_NO_LVL": "1"}, "per-layer launches on the small maps"),
    ({"DMME_LVL_MASK": "4"}, "level engine on the 4x4 maps only"),
    ({"DMME_LVL_MASK": "8"}, "level engine on the 8x8 maps only"),
    ({"DMME_LVL_GB1": "1"}, "level engine: one pixel group per iteration"),
    ({"DMME_LVL_NJ1": "1"}, "level engine: 32-cout slices (8 per group) on every level"),
    ({"DMME_LVL_NO_XRUN": "1"}, "level engine: skip tensors re-normalised by the run that reads them"),
    ({"DMME_LVL_NO_RES_MERGE": "1"}, "level engine: the blocks' 1x1 residual convs as ops of their own"),
    ({"DMME_LVL_MAX_ITER": "1"}, "level engine only where a workgroup owns one iteration per op"),
    ({"DMME_NO_LVL": "1", "DMME_NO_KW": "1"}, "small maps on the four-wave pipelined kernel"),
    ({"DMME_KW_NO_BM32": "1"}, "K-split kernel without its 32-pixel tiles (they apply at batch 1-2; same route at this batch)"),
    ({"DMME_NO_LVL": "1", "DMME_KW_BM64": "1"}, "8x8 level on 64-pixel K-split tiles"),
    ({"DMME_NO_LVL": "1", "DMME_NO_GN_DIRECT": "1"}, "norms of whole-image tiles as launches"),
    ({"DMME_NO_LVL": "1", "DMME_NO_GN_DIRECT_WS": "1"}, "16x16 norms not finished by the persistent kernel's two-pass epilogue"),
    ({"DMME_NO_LVL": "1", "DMME_NO_PREACT": "1"}, "convs apply GroupNorm / SiLU themselves everywhere"),
    ({"DMME_NO_LVL": "1", "DMME_NO_GN_SMALL": "1"}, "no one-workgroup-per-image norm kernel"),
    ({"DMME_NO_LVL": "1", "DMME_NO_ATTN_S16": "1"}, "4x4 attention on the generic kernel"),
    ({"DMME_NO_LVL": "1", "DMME_NO_GN_IN_KW": "1"}, "K-split kernel: finalize launches in front"),
    ({"DMME_NO_ATTN_FULL": "1"}, "16x16 attention on the online-softmax kernel"),
    ({"DMME_ATTN_SLEEP": "0"}, "whole-row attention kernel without pacing"),
    ({"DMME_NO_CONV1X1_AS": "1"}, "1x1 convs on the tiled kernel"),
    ({"DMME_NO_GN_IN": "1"}, "finalize launches instead of consumer-side merges"),
    ({"DMME_NO_GN_IN_PIPE": "1"}, "four-wave kernels: finalize launches in front"),
    ({"DMME_NO_GN_IN_PIPE1": "1"}, "tiled 1x1 kernel: finalize launches in front"),
    ({"DMME_NO_PIPE_DMA": "1"}, "filter tiles through registers in the 64-cout pipelined kernel"),
    ({"DMME_NO_CONV_THIN": "1"}, "output conv on the tiled kernels"),
    ({"DMME_NO_WS": "1"}, "no wave-specialised persistent kernel"),
    ({"DMME_NO_RSEG": "1"}, "the blocks' 1x1 residual convs of the 32x32 / 16x16 levels as launches of their own (+7), not as a segment of conv2"),
    ({"DMME_NO_WS128": "1"}, "persistent kernel without its 128-pixel tiles (the 128-cout layers of the 16x16 level back on the four-wave kernel)"),
    ({"DMME_NO_WS_E16": "1"}, "persistent kernel: the fp32-staged two-pass epilogue everywhere (round 5: 16-bit staging on transposed accumulators where no residual tensor is read)"),
    ({"DMME_LVL_NO_XCD": "1"}, "level engine: a pixel group's slices in index order (round 5: on one XCD)"),
    ({"DMME_LVL_KEEP_RAW": "1"}, "level engine under no_grad: every conv stores its raw output (round 5: not where only the pre-activated copy is read)"),
    ({"DMME_NO_ATTN_PROJ": "1"}, "attention blocks of full launches: proj conv + residual as their own launch (round 5: inside the attention launch)"),
    ({"DMME_NO_FUSED_GN": "1"}, "every GroupNorm reads its tensor"),
    ({"DMME_NO_XCD_ORDER": "1"}, "plain workgroup order in attention / 1x1 convs"),
    ({"DMME_NO_SPLITK": "1"}, "no split-K in the four-wave kernel"),
    ({"DMME_NO_CONV_IN_MFMA": "1"}, "input conv on the VALU kernel"),
]
BACKWARD = ["DMME_NO_WG_ACT", "DMME_NO_WG_DMA", "DMME_NO_WG_S2", "DMME_NO_WGRAD_GROUP", "DMME_NO_GN_BWD_IMAGE", "DMME_NO_GN_BWD_FUSED_FIN",
            "DMME_NO_GN_BWD_ROWS", "DMME_NO_GN_BWD_REGS", "DMME_NO_RES_EXTRA", "DMME_NO_GN_BWD_SLICES", "DMME_NO_DGRAD_DIRECT", "DMME_NO_RES_ALIAS", "DMME_NO_COLSUM_GROUP",
            "DMME_NO_BIAS_GROUP", "DMME_NO_WGRAD_THIN", "DMME_NO_TIME_PRE", "DMME_NO_SMALL_GEMM_MFMA", "DMME_NO_LVL",
            "DMME_NO_RSEG", "DMME_NO_WS_E16", "DMME_NO_ATTN_PROJ"]  # (forward routes: at this batch three blocks' residual convs run inside conv2 - the training step with them as launches)


from tests.gpu_util import route_env as _env  # (sets product switches as variables, everything else as DMME_DEBUG_ROUTE keys)


@pytest.fixture(scope="module")
def fwd_case(golden):
    import dmme_amd

    g = golden("unet_full")
    seed = int(g["full_seed"])
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32)).repeat(64, 1, 1, 1).cuda()
    t = torch.from_numpy(g["full_t_one"]).cuda()
    sd = O.make_state_dict(O.UNetConfig(), seed)

    def run(env):
        with _env(env):
            net = dmme_amd.UNet(precision="bf16")
            net.load_state_dict(sd, strict=True)
            net = net.cuda().eval()
            with torch.no_grad():
                y = net(x, t).float().cpu()
            n = net._last_plan.lib.dmme_unet_plan_num_launches(net._last_plan.h)
        return y, n

    base, n0 = run({})
    return run, base, n0, torch.from_numpy(g["full_y_one"])


@pytest.mark.parametrize("env,note", FORWARD, ids=["+".join(f"{k}={v}" if v != "1" else k for k, v in e.items()) for e, _ in FORWARD])
def test_forward_route_switch_vs_default_route_and_reference(fwd_case, env, note):
    run, base, n0, ref = fwd_case
    y, n = run(env)
    rows = y.reshape(64, 2, 3, 32, 32)
    assert torch.equal(rows, rows[:1].expand_as(rows)), note  # every image pair took the same arithmetic
    e_ref = float((rows[0] - ref).abs().max())
    r_ref = float((rows[0] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    e_ab = float((y - base).pow(2).mean().sqrt() / base.pow(2).mean().sqrt())
    print(f"{env} ({note}): launches {n0} -> {n}; vs reference: rel-RMS {r_ref:.3e}, max|err| {e_ref:.3e}; rel-rms vs the default route {e_ab:.3e}")
    # the bf16 network's error budget (tests/test_gpu_unet.py: BF16_REL_RMS, BF16_MAX_ABS - every route is another rounding sequence)
    assert r_ref <= 1.0e-2 and e_ref <= 1.7e-2
    assert e_ab <= 1.0e-2    # two bf16 evaluations of one network (other tile shapes / summation orders), or identical bits


def test_residual_segment_saves_seven_launches(fwd_case):
    """at the benchmark batch the seven channel-changing ResBlocks of the 32x32 / 16x16 levels run their 1x1 residual conv inside conv2
    (conv_pipe.hip RSEG; five on the 256-pixel tiles, two on the 128-pixel ones): seven launches fewer than with the switch off, and
    not the same bits (the residual tensor's rounding is gone)"""
    run, base, n0, ref = fwd_case
    y, n = run({"DMME_NO_RSEG": "1"})
    assert n == n0 + 7, (n0, n)
    assert not torch.equal(y, base)


def test_ws128_on_32x32_maps_at_batch_32(golden):
    """the 128-pixel tiles of the persistent kernel on the 32x32 maps (4 x 32 pixels, 204 halo rows in 7 units - another geometry than
    the 8 x 16 tiles the batch-128 case above reaches): batch 32 with and without them, against each other and the golden rows"""
    import dmme_amd

    g = golden("unet_full")
    sd = O.make_state_dict(O.UNetConfig(), int(g["full_seed"]))
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32)).repeat(16, 1, 1, 1).cuda()
    t = torch.from_numpy(g["full_t_one"]).cuda()
    ref = torch.from_numpy(g["full_y_one"])
    ys = []
    for env in ({}, {"DMME_NO_WS128": "1"}):
        with _env(env):
            net = dmme_amd.UNet(precision="bf16")
            net.load_state_dict(sd, strict=True)
            net = net.cuda().eval()
            with torch.no_grad():
                y = net(x, t).float().cpu()
        rows = y.reshape(16, 2, 3, 32, 32)
        assert torch.equal(rows, rows[:1].expand_as(rows))
        assert float((rows[0] - ref).abs().max()) <= 1.7e-2 and float((rows[0] - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) <= 1.0e-2
        ys.append(y)
    e_ab = float((ys[0] - ys[1]).pow(2).mean().sqrt() / ys[1].pow(2).mean().sqrt())
    assert 0.0 < e_ab <= 1.0e-2, e_ab  # (> 0: the switch did select another kernel for the 32x32 layers)


@pytest.fixture(scope="module")
def bwd_case():
    import dmme_amd

    def step(env):
        with _env(env):
            torch.manual_seed(0)
            net = dmme_amd.UNet(precision="bf16").cuda().train()
            x = torch.randn(32, 3, 32, 32, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
            t = torch.arange(32, device="cuda") * 31 % 1000
            y = net(x, t)
            l = (y.float() ** 2).mean()
            l.backward()
            return float(l.detach()), net.flat_grad().float().clone()

    return step, step({})


@pytest.mark.parametrize("switch", BACKWARD)
def test_backward_route_switch_vs_default_route(bwd_case, switch):
    step, (l0, g0) = bwd_case
    l, g = step({switch: "1"})
    rel = float((g - g0).norm() / g0.norm())
    print(f"{switch}: loss {l:.6f} vs {l0:.6f}; flat gradient relative difference {rel:.3e}")
    assert abs(l - l0) <= 2e-3 * abs(l0)
    assert rel <= 3e-2  # two bf16 backward passes (per-tensor budget: DESIGN.md section 2), or identical bits
