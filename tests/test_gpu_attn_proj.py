"""The attention block's tail in ONE launch: dst = res + proj(attention(qkv)) (models/ddpm.py:54-75 behind the norm + qkv conv),
dmme_attention_proj, against (i) the separate launches it replaces (dmme_attention, then the 1x1 conv + residual in fp32 from the
SAME 16-bit context: differences are summation order only) and (ii) the fp64 restatement of the block from the 16-bit operands."""
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu

BF16_RTOL = 5e-3      # of max |want| (tests/test_gpu_ops.py: attention) + the proj conv's 16-bit input rounding
FP16_RTOL = 5e-3 / 8


def _run(dt, qkv, w, bias, res, want_ctx, gn_cg):
    from dmme_amd import _lib
    from tests.gpu_util import TD

    N, S, C3 = qkv.shape
    C = C3 // 3
    td = TD[dt]
    q = qkv.to(td).contiguous()
    wd, rd = w.to(td).contiguous(), res.to(td).contiguous()
    b = bias.float().contiguous()
    dst = torch.zeros((N, S, C), dtype=td, device=q.device)
    ctx = torch.zeros((N, S, C), dtype=td, device=q.device) if want_ctx else None
    part = torch.zeros((N, S // 32, C // gn_cg, 2), dtype=torch.float32, device=q.device) if gn_cg else None
    _lib.check(
        _lib.lib().dmme_attention_proj(dt, _lib.ptr(q), N, S, C, _lib.ptr(wd), _lib.ptr(b), _lib.ptr(rd), _lib.ptr(dst), _lib.ptr(ctx), _lib.ptr(part),
                                       gn_cg, _lib.stream_ptr()),
        "dmme_attention_proj",
    )
    torch.cuda.synchronize()
    return dst, ctx, part


@pytest.mark.parametrize("C,dtname,cg", [(256, "bf16", 8), (256, "fp16", 8), (128, "bf16", 4), (128, "fp16", 4),
                                         (256, "bf16", 4), (128, "bf16", 8)])  # (the last two: 64 / 16 groups - the other group width per head width)
def test_attention_proj_one_launch(C, dtname, cg):
    from dmme_amd import _lib
    from tests import gpu_util as G

    N, S = 128, 256
    dt = _lib.dtype_code(dtname)
    td = G.TD[dt]
    dev = torch.device("cuda:0")
    qkv = synth.normal(11, (N, S, 3 * C))
    qkv[:, :, :C] *= 2.0  # sharper softmax
    w = synth.normal(12, (C, C)) * C**-0.5
    bias = synth.normal(13, (C,)) * 0.1
    res = synth.normal(14, (N, S, C))
    qkv, w, bias, res = qkv.to(dev), w.to(dev), bias.to(dev), res.to(dev)
    dst, ctx, part = _run(dt, qkv, w, bias, res, True, cg)

    # (i) the launches it replaces
    ctx_sep = G.attention(dt, qkv, False).to(td)
    assert torch.equal(ctx, ctx_sep), "context tensor differs from dmme_attention's"
    w16, r16 = w.to(td).double(), res.to(td).double()
    want_sep = (r16 + ctx_sep.double() @ w16.t() + bias.double()).to(td)
    ulp = 2.0**-7 if dtname == "bf16" else 2.0**-10
    diff = (dst.double() - want_sep.double()).abs()
    bound = ulp * want_sep.double().abs().clamp_min(1.0)  # one rounding step: the fp32 sums differ in order only
    assert bool((diff <= bound).all()), f"max excess {(diff - bound).max().item():.3e}"
    assert (diff > 0).double().mean().item() < 0.02, "more than 2 % of the outputs differ from the separate launches"

    # (ii) the block in fp64 from the 16-bit operands
    src = qkv.to(td).double()
    q, k, v = src[:, :, :C], src[:, :, C : 2 * C], src[:, :, 2 * C :]
    att = torch.softmax(q @ (k.transpose(1, 2) * C**-0.5), dim=2) @ v
    want = r16 + att @ w16.t() + bias.double()
    err = (dst.double() - want).abs().max().item()
    tol = (BF16_RTOL if dtname == "bf16" else FP16_RTOL) * want.abs().max().item() * 2
    print(f"attention_proj C={C} {dtname}: err {err:.3e} (tol {tol:.3e})")
    assert err <= tol

    # the next norm's partials: (mean, M2) of the ROUNDED outputs per (image, 32 tokens, group)
    x = dst.double().view(N, S // 32, 32, C // cg, cg)
    mean = x.mean(dim=(2, 4))
    m2 = ((x - mean[:, :, None, :, None]) ** 2).sum(dim=(2, 4))
    assert (part[..., 0].double() - mean).abs().max().item() <= 1e-5 * max(1.0, mean.abs().max().item())
    assert ((part[..., 1].double() - m2).abs() / m2.clamp_min(1e-6)).max().item() <= 1e-4

    # without the context tensor and without statistics: the same outputs
    dst2, _, _ = _run(dt, qkv, w, bias, res, False, 0)
    assert torch.equal(dst2, dst)


def test_attention_proj_refuses_what_the_whole_row_kernel_does_not_serve():
    from dmme_amd import _lib

    dev = torch.device("cuda:0")
    dt = _lib.dtype_code("bf16")
    z = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=dev)  # noqa: E731
    b = torch.zeros(256, device=dev)
    for N, S, C in ((4, 256, 256), (128, 64, 256), (128, 256, 64)):
        rc = _lib.lib().dmme_attention_proj(dt, _lib.ptr(z(N, S, 3 * C)), N, S, C, _lib.ptr(z(C, C)), _lib.ptr(b), _lib.ptr(z(N, S, C)), _lib.ptr(z(N, S, C)), None,
                                            None, 0, _lib.stream_ptr())
        assert rc != 0, (N, S, C)


def _net(seed, precision):
    import dmme_amd
    from oracle import unet as O

    cfg = O.UNetConfig()
    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                        cfg.attention_depths, precision=precision)
    net.load_state_dict(O.make_state_dict(cfg, seed), strict=True)
    return net.cuda().eval()


@pytest.mark.parametrize("precision,atol_ref", [("bf16", 1.7e-2), ("fp16", 2.0e-3)])  # (the networks' bounds: test_gpu_unet.py BF16_MAX_ABS, test_gpu_fp16.py FP16_MAX_ABS)
def test_batch128_network_with_the_blocks_fused_vs_reference_rows_and_vs_separate_launches(golden, precision, atol_ref):
    """the default UNet at the benchmark batch: its five 16x16 attention blocks run attention + proj + residual in one launch each
    (plan.hip: assign_attn_proj); against the reference's golden rows and against the same network with DMME_DEBUG_ROUTE=no_attn_proj"""
    from tests.gpu_util import route_env

    g = golden("unet_full")
    seed = int(g["full_seed"])
    base = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    x = base.repeat(64, 1, 1, 1).cuda()
    t = torch.from_numpy(g["full_t_one"]).cuda()

    def run(env):
        with route_env(env):
            net = _net(seed, precision)
            with torch.no_grad():
                y = net(x, t).float().cpu()
            return y, net._last_plan.lib.dmme_unet_plan_num_launches(net._last_plan.h)

    ya, na = run({})
    yb, nb = run({"DMME_NO_ATTN_PROJ": "1"})
    assert nb - na == 5, (na, nb)
    rows = ya.reshape(64, 2, 3, 32, 32)
    assert torch.equal(rows, rows[:1].expand_as(rows))  # every image pair took the same arithmetic
    ref = torch.from_numpy(g["full_y_one"])
    e_ref = float((rows[0] - ref).abs().max())
    e_ab = float((ya - yb).pow(2).mean().sqrt() / yb.pow(2).mean().sqrt())
    print(f"{precision}: launches {nb} -> {na}; max|err| vs reference {e_ref:.3e}; relative rms between the two routes {e_ab:.3e}")
    assert e_ref <= atol_ref
    assert e_ab <= (1.0e-2 if precision == "bf16" else 2.5e-3)  # (fp32 summation order inside the proj conv, then the network's roundings)


def test_batch128_training_step_with_the_blocks_fused():
    """a forward that a backward follows also writes the context tensor (the proj conv's weight gradient reads it): loss and gradient of
    one training step with and without the fusion, same seed and dropout masks"""
    import dmme_amd
    from tests.gpu_util import route_env

    def step(env):
        with route_env(env):
            torch.manual_seed(0)
            net = dmme_amd.UNet(precision="bf16").cuda().train()
            x = torch.randn(128, 3, 32, 32, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
            t = torch.arange(128, device="cuda") * 7 % 1000
            y = net(x, t)
            l = (y.float() ** 2).mean()
            l.backward()
            g = net.flat_grad().float().clone()
            proj = {n: p.grad.float().clone() for n, p in net.named_parameters() if ".attention.proj." in n}
            return float(l.detach()), g, proj

    la, ga, pa = step({})
    lb, gb, pb = step({"DMME_NO_ATTN_PROJ": "1"})
    assert len(pa) == 12  # six blocks: five of them on the 16x16 maps
    rel = float((ga - gb).norm() / gb.norm())
    worst = max(float((pa[n] - pb[n]).norm() / pb[n].norm().clamp_min(1e-12)) for n in pa)
    print(f"training step loss {la:.6f} (fused) vs {lb:.6f}; relative gradient difference {rel:.3e}; worst proj-parameter difference {worst:.3e}")
    assert abs(la - lb) <= 2e-3 * abs(lb)
    assert rel <= 5e-2 and worst <= 5e-2  # (bf16 networks of 100 rounded tensors: two runs of ONE route differ by ~1e-2 through the atomics' order)


def test_backward_refuses_the_workspace_of_a_no_grad_forward():
    """dmme_unet_forward_nograd leaves out what only the backward pass reads (the fused blocks' context tensors, raw conv outputs of
    the level engine): dmme_unet_backward on that workspace fails loudly (the C entry point itself - the host module has its own
    generation check in front of it); after a dmme_unet_forward it runs"""
    import dmme_amd
    from dmme_amd import _lib

    torch.manual_seed(0)
    net = dmme_amd.UNet(precision="bf16").cuda().train()
    x = torch.randn(128, 3, 32, 32, device="cuda")
    t = (torch.arange(128, device="cuda") * 7 % 1000).to(torch.int64)
    net(x, t).float().pow(2).mean().backward()  # creates the backward workspace and the packed backward weights
    plan = net._last_plan
    lib = plan.lib
    with torch.no_grad():
        net(x, t)                               # same plan, same workspace: dmme_unet_forward_nograd
    packed = net._packed_for(plan)
    d = torch.zeros(128, 3, 32, 32, device="cuda")
    g = net.flat_grad()
    args = (plan.h, _lib.ptr(packed), _lib.ptr(plan.packed_bwd), _lib.ptr(x), _lib.ptr(t), 128, _lib.ptr(d), _lib.ptr(plan.workspace), _lib.ptr(plan.bws),
            _lib.ptr(plan.masks), _lib.ptr(g), None, _lib.stream_ptr())
    rc = lib.dmme_unet_backward(*args)
    assert rc != 0 and b"dmme_unet_forward_nograd" in lib.dmme_last_error()
    net.zero_grad()
    net(x, t).float().pow(2).mean().backward()  # a forward with autograd fills the workspace again
    torch.cuda.synchronize()
    assert float(net.flat_grad().float().abs().sum()) > 0
