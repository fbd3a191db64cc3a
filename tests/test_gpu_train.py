"""-m gpu: the training step (HIP forward + HIP backward + fused optimiser) against the golden
gradients the reference produced and against torch autograd of the CPU oracle."""

import numpy as np
import pytest
import torch

from oracle import diffusion as D
from oracle import synth
from oracle import unet as O

pytestmark = pytest.mark.gpu


def _build(cfg, seed, precision="fp32"):
    import dmme_amd

    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth,
                        cfg.num_blocks, cfg.attention_depths, precision=precision)
    net.load_state_dict(O.make_state_dict(cfg, seed), strict=True)
    return net.cuda()


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_training_step_loss_and_grads_vs_reference_golden(golden, mode):
    import dmme_amd

    g = golden("train_tiny")
    seed, T, B, sx, st, sz, sm = [int(v) for v in g["train_meta"]]
    cfg = O.TINY
    net = _build(cfg, seed)
    net.train(mode == "train")
    if mode == "train":
        masks = O.make_drop_masks(cfg, B, sm)
        net.inject_dropout_masks(torch.cat([masks[k].reshape(-1) for k in O.res_block_names(cfg)]).cuda())
    ddpm = dmme_amd.DDPM(net, T).cuda()
    x0 = synth.uniform(sx, (B, 3, 32, 32)).cuda()
    t = synth.randint(st, 1, T, B).cuda()
    z = synth.normal(sz, (B, 3, 32, 32)).cuda()
    loss = ddpm.training_step(x0, t=t, noise=z)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"train_{mode}_loss"], rtol=1e-5)
    worst = {}
    n = 0
    for name, p in net.named_parameters():
        want = g[f"train_{mode}_grad::{name}"]
        got = p.grad.cpu().numpy()
        err = np.abs(got - want).max()
        tol = 2e-6 + 1e-4 * np.abs(want).max()
        if err > tol:
            worst[name] = (float(err), float(np.abs(want).max()))
        n += 1
    assert n == 410 or n > 100
    assert not worst, f"{len(worst)} gradients off, e.g. {list(worst.items())[:6]}"
    # gradients accumulate across backward calls, like torch autograd
    loss2 = ddpm.training_step(x0, t=t, noise=z)
    loss2.backward()
    p = dict(net.named_parameters())["output_conv.2.weight"]
    np.testing.assert_allclose(p.grad.cpu().numpy(), 2 * g[f"train_{mode}_grad::output_conv.2.weight"], rtol=1e-4, atol=2e-6)


def test_full_size_grads_vs_oracle_autograd():
    """default UNet, B = 2, fp32: exercises the MFMA data-gradient convolutions (transposed, tap-flipped
    weights, zero-insertion for the stride-2 convs, 2x2 sum-pool for the upsample convs)."""
    import dmme_amd

    cfg = O.UNetConfig()
    seed, T, B = 31, 1000, 2
    net = _build(cfg, seed).eval()
    sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in O.make_state_dict(cfg, seed).items()}
    x0 = synth.uniform(1, (B, 3, 32, 32))
    t = torch.tensor([17, 803])
    z = synth.normal(2, (B, 3, 32, 32))
    _, abar = D.alpha_tables(D.linear_beta(T))
    want = D.training_loss(lambda xt, tt: O.unet_forward(sd, cfg, xt, tt), x0, t, z, abar)
    want.backward()
    ddpm = dmme_amd.DDPM(net, T).cuda()
    loss = ddpm.training_step(x0.cuda(), t=t.cuda(), noise=z.cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), want.item(), rtol=1e-5)
    bad = {}
    for name, p in net.named_parameters():
        w = sd[name].grad.numpy()
        err = np.abs(p.grad.cpu().numpy() - w).max()
        if err > 1e-6 + 2e-4 * np.abs(w).max():
            bad[name] = (float(err), float(np.abs(w).max()))
    assert not bad, f"{len(bad)} gradients off, e.g. {list(bad.items())[:6]}"


def test_fused_adam_matches_torch_adam():
    import dmme_amd
    from dmme_amd.optim import FusedAdam

    cfg = O.TINY
    net = _build(cfg, 5)
    ref = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items() if k != "condition.0.embeddings"}
    ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ref.items()}
    opt_ref = torch.optim.Adam(list(ref.values()), lr=1e-3)
    opt = FusedAdam(net.parameters(), lr=1e-3, max_grad_norm=1.0, ema_decay=0.9)
    ema = {k: v.detach().clone() for k, v in ref.items()}
    for step in range(3):
        flat_g = net.flat_grad()
        gen = torch.Generator().manual_seed(step)
        for k, p in net.named_parameters():
            gk = torch.randn(p.shape, generator=gen) * 3.0
            p.grad.copy_(gk.cuda())
            ref[k].grad = gk.clone()
        torch.nn.utils.clip_grad_norm_(list(ref.values()), 1.0)
        opt_ref.step()
        for k in ema:
            ema[k] = 0.9 * ema[k] + 0.1 * ref[k].detach()
        opt.step()
        assert flat_g.data_ptr() == net.flat_grad().data_ptr()
    for k, p in net.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), ref[k].detach().numpy(), atol=2e-6, rtol=1e-5, err_msg=k)
    ema_flat = opt.ema_parameters(net)
    table = {name: (off, int(np.prod(shape))) for name, shape, off, isb in net._table}
    off, n = table["input_conv.weight"]
    np.testing.assert_allclose(ema_flat[off : off + n].cpu().numpy().reshape(-1), ema["input_conv.weight"].numpy().reshape(-1), atol=2e-6, rtol=1e-5)


def test_train_loop_tiny_matches_oracle_two_steps():
    """two optimisation steps (injected t / noise, eval-mode dropout) through LitDDPM + FusedAdam +
    WarmupLR equal the same two steps of the oracle under torch.optim.Adam."""
    import dmme_amd

    cfg = O.TINY
    net = _build(cfg, 9).eval()
    lit = dmme_amd.LitDDPM(lr=1e-3, warmup=4, decay=0.0, diffusion_model=dmme_amd.DDPM(net, 100)).cuda()
    opts, scheds = lit.configure_optimizers()
    opt, sched = opts[0], scheds[0]["scheduler"]
    for gk in opt.param_groups:
        gk["max_grad_norm"] = 1.0
    sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in O.make_state_dict(cfg, 9).items()}
    params = [v for k, v in sd.items() if v.requires_grad]
    ropt = torch.optim.Adam(params, lr=1e-3)
    _, abar = D.alpha_tables(D.linear_beta(100))
    for step in range(2):
        x0 = synth.uniform(40 + step, (4, 3, 32, 32))
        t = synth.randint(50 + step, 1, 100, 4)
        z = synth.normal(60 + step, (4, 3, 32, 32))
        lr = 1e-3 * min(1.0, (step + 1) / 4)
        for gk in ropt.param_groups:
            gk["lr"] = lr
        ropt.zero_grad()
        want = D.training_loss(lambda xt, tt: O.unet_forward(sd, cfg, xt, tt), x0, t, z, abar)
        want.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        ropt.step()
        assert abs(opt.param_groups[0]["lr"] - lr) < 1e-12
        loss = lit.diffusion_model.training_step(x0.cuda(), t=t.cuda(), noise=z.cuda())
        loss.backward()
        opt.step()
        sched.step()
        opt.zero_grad()
        np.testing.assert_allclose(loss.item(), want.item(), rtol=2e-5)
    for k, p in net.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), sd[k].detach().numpy(), atol=3e-5, rtol=1e-4, err_msg=k)


def test_bf16_train_step_close_to_fp32():
    """bf16 compute path (MFMA wgrad with transposed LDS reads): gradients track the fp32 path."""
    import dmme_amd

    cfg = O.UNetConfig(channels_per_depth=(64, 128), num_blocks=1, attention_depths=(2,), emb_dim=128, pos_dim=32)
    grads = {}
    for prec in ("fp32", "bf16"):
        net = _build(cfg, 3, prec).eval()
        ddpm = dmme_amd.DDPM(net, 1000).cuda()
        loss = ddpm.training_step(synth.uniform(1, (4, 3, 32, 32)).cuda(), t=torch.tensor([5, 300, 700, 999]).cuda(), noise=synth.normal(2, (4, 3, 32, 32)).cuda())
        loss.backward()
        grads[prec] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
    from tests.test_gpu_grad_b128 import CLASS_BOUNDS, _class_errors, _classes

    worst = _class_errors(grads["bf16"], grads["fp32"], _classes(net))
    print("worst relative gradient error bf16 vs fp32 per tensor class:", {c: f"{v[0]:.3e} ({v[1]})" for c, v in worst.items()})
    for c, (rel, name) in worst.items():  # the per-class budget of the batch-128 parity test (derived there)
        assert rel <= CLASS_BOUNDS[c], (c, name, rel)


def test_swap_ema_weights_for_sampling():
    """EMA copy follows ema = d*ema + (1-d)*p after every step; swap_ema puts it in the model for evaluation and
    restores the live weights afterwards (forward outputs follow the swap: the packed weights are rebuilt)."""
    import dmme_amd
    from dmme_amd.optim import FusedAdam

    torch.manual_seed(3)
    from oracle import unet as O

    cfg = O.TINY
    model = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, 0.0, cfg.channels_per_depth, cfg.num_blocks,
                          cfg.attention_depths, precision="fp32").cuda()
    model.eval()
    opt = FusedAdam(model.parameters(), lr=1e-2, ema_decay=0.5)
    x = torch.randn(2, cfg.in_channels, 8, 8, device="cuda")
    t = torch.tensor([5], device="cuda")
    p0 = model.flat_parameters().clone()
    for _ in range(2):
        model.train()
        loss = (model(x, t) ** 2).mean()
        loss.backward()
        opt.step()
        opt.zero_grad()
    model.eval()
    p2 = model.flat_parameters().clone()
    ema = opt.ema_parameters(model).clone()
    assert not torch.equal(p2, p0) and not torch.equal(ema, p2)
    with torch.no_grad():
        y_live = model(x, t).clone()
        with opt.swap_ema(model):
            assert torch.equal(model.flat_parameters(), ema)
            y_ema = model(x, t).clone()
        assert torch.equal(model.flat_parameters(), p2)
        y_back = model(x, t).clone()
    assert torch.equal(y_live, y_back)
    assert not torch.equal(y_live, y_ema)


def test_checkpoint_layout_round_trip_and_torch_adam_interop(tmp_path):
    """The optimiser state is written in the reference's EMAOptimizer layout ({"opt", "ema", "current_step", ...},
    callbacks/ema.py:339-359): a stock torch.optim.Adam loads "opt" and takes the same next step; a fresh model +
    FusedAdam restored from the file continues bit-identically."""
    import dmme_amd
    from dmme_amd.checkpoint import load_checkpoint, save_checkpoint
    from dmme_amd.optim import FusedAdam

    cfg = O.TINY

    def make():
        lit = dmme_amd.LitDDPM(model=_build(cfg, 5), timesteps=100, warmup=3).cuda()
        opts, scheds = lit.configure_optimizers()
        return lit, opts[0], scheds[0]["scheduler"]

    def set_grads(net, step):
        net.flat_grad()
        gen = torch.Generator().manual_seed(100 + step)
        for _, p in net.named_parameters():
            p.grad.copy_((torch.randn(p.shape, generator=gen) * 0.1).cuda())

    lit, opt, sched = make()
    net = lit.diffusion_model.model
    for step in range(3):
        set_grads(net, step)
        opt.step()
        sched.step()
    path = str(tmp_path / "last.ckpt")
    save_checkpoint(path, lit, opt, sched, global_step=3)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) >= {"state_dict", "optimizer_states", "lr_schedulers", "global_step"}
    assert all(k.startswith("diffusion_model.model.") for k in ck["state_dict"])
    osd = ck["optimizer_states"][0]
    assert set(osd) >= {"opt", "ema", "current_step", "decay", "every_n_steps", "device"}
    assert osd["current_step"] == 3 and len(osd["ema"]) == len(list(net.parameters())) == len(osd["opt"]["state"])
    # a stock Adam takes the same 4th step from the saved moments
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in net.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=1.0)
    ref_opt.load_state_dict(osd["opt"])
    set_grads(net, 3)
    for rp, p in zip(ref_params, net.parameters()):
        rp.grad = p.grad.detach().cpu().clone()
    lr_now = opt.param_groups[0]["lr"]
    assert ref_opt.param_groups[0]["lr"] == lr_now
    ref_opt.step()
    # restored copy continues identically
    lit2, opt2, sched2 = make()
    with torch.no_grad():
        lit2.diffusion_model.model.flat_parameters().add_(1.0)  # make sure the weights really come from the file
    load_checkpoint(path, lit2, opt2, sched2)
    net2 = lit2.diffusion_model.model
    set_grads(net2, 3)
    opt.step()
    opt2.step()
    for rp, p, p2 in zip(ref_params, net.parameters(), net2.parameters()):
        assert torch.equal(p, p2)
        np.testing.assert_allclose(p.detach().cpu().numpy(), rp.detach().numpy(), rtol=0, atol=2e-7)
    assert torch.equal(opt.ema_parameters(net), opt2.ema_parameters(net2))
    assert sched2.state_dict()["last_epoch"] == sched.state_dict()["last_epoch"]


@pytest.mark.parametrize("which", ["tiny_fp32", "full_bf16", "iddpm_bf16"])
def test_bucketed_backward_matches_and_reports_its_buckets(which):
    """dmme_unet_backward_buckets (gradient exchange overlapped with backward): same gradients as the one-piece backward; the
    hand-overs tile the flat buffer exactly once, arrive in the order backward finishes them (output conv / last up blocks first, the
    time MLP + input conv + first down blocks last) and - for the default UNet - there are at least four buckets, none above a
    quarter of the bytes, the LAST (the only exchange no compute can hide) at most 15 %."""
    import ctypes as C

    import dmme_amd
    from dmme_amd import _lib
    from dmme_amd.distributed import OverlappedGradReducer

    if which == "tiny_fp32":
        net = _build(O.TINY, 5, "fp32")
        B, side = 3, 32
    elif which == "full_bf16":
        net = _build(O.UNetConfig(), 5, "bf16")
        B, side = 4, 32
    else:
        from dmme_amd.models import iddpm

        net = iddpm.UNet(precision="bf16").cuda()
        B, side = 4, 32
    net.train()
    x = synth.normal(1, (B, 3, side, side)).cuda()
    t = torch.tensor([5, 60, 99, 7][:B]).cuda()
    w = synth.normal(2, (B, net.out_channels, side, side)).cuda()

    def run():
        net.zero_grad(set_to_none=True)
        net._mask_calls = 0
        torch.manual_seed(0)
        (net(x, t) * w).sum().backward()
        return net.flat_grad().clone()

    want = run()
    red = OverlappedGradReducer(net)
    got = run()
    rep = list(red.reported)
    assert red.finish() is False  # single process: nothing to reduce, the caller's path applies
    net._bucket_hook = None
    n = want.numel()
    # what the plan announces = what backward reported, in order
    plan = net._last_plan
    offs, nums, bks = (C.c_int64 * 64)(), (C.c_int64 * 64)(), (C.c_int * 64)()
    cnt = plan.lib.dmme_unet_plan_grad_buckets(plan.h, offs, nums, bks, 64)
    table = [(int(offs[i]), int(nums[i]), int(bks[i])) for i in range(cnt)]
    assert [(o, m) for o, m, _ in table] == rep, (table, rep)
    # exact tiling of [0, n)
    cover = sorted(rep)
    names = dict((name, (o, isb)) for name, shape, o, isb in net._table)
    pos = 0
    for o, m in cover:
        assert o == pos, (o, pos, cover)
        pos = o + m
    assert pos == n
    nb = max(b for _, _, b in table) + 1
    per_bucket = [sum(m for _, m, b in table if b == k) for k in range(nb)]
    print(f"{which}: {nb} gradient buckets, shares {[round(v / n, 3) for v in per_bucket]}")
    # order: the bucket holding output_conv first, the one holding the time MLP last
    out_off = names["output_conv.2.weight"][0]
    cond_off = names["condition.1.weight"][0]
    assert any(o <= out_off < o + m for o, m, b in table if b == 0) and any(o <= cond_off < o + m for o, m, b in table if b == nb - 1)
    if which != "tiny_fp32":
        assert nb >= 4 and max(per_bucket) <= 0.26 * n and per_bucket[-1] <= 0.15 * n, per_bucket
    scale = float(want.abs().max())
    # fp32: only the order of the float atomics differs; bf16: the grouped weight-gradient launch is cut into the buckets' launches
    assert float((got - want).abs().max()) <= (1e-5 if which == "tiny_fp32" else 2e-3) * scale


@pytest.mark.parametrize("which", ["ddpm", "iddpm"])
def test_training_learns_bf16(which):
    """End to end in the benchmark precision: 120 optimisation steps (HIP backward, fused clip + Adam + EMA, warm-up) on a small set of
    smooth images bring the loss from ~1.1 to well under a fifth of that, for L_simple and for the hybrid loss."""
    import dmme_amd
    from dmme_amd.train_loop import train_step

    torch.manual_seed(0)
    if which == "ddpm":
        lit = dmme_amd.LitDDPM(model=dmme_amd.UNet(precision="bf16"), warmup=50)
    else:
        from dmme_amd.models import iddpm

        lit = dmme_amd.LitIDDPM(model=iddpm.UNet(precision="bf16"), warmup=50)
    lit = lit.cuda()
    lit.train()
    opts, scheds = lit.configure_optimizers()
    opt, sched = opts[0], scheds[0]["scheduler"]
    for g in opt.param_groups:
        g["max_grad_norm"] = 1.0
    base = torch.nn.functional.interpolate(torch.rand(256, 3, 4, 4, device="cuda") * 2 - 1, size=32, mode="bilinear")
    first = last = None
    for step in range(120):
        loss = float(train_step(lit, opt, sched, base[torch.randint(0, 256, (64,), device="cuda")]).detach())
        assert loss == loss, f"NaN loss at step {step}"
        if step == 0:
            first = loss
        last = loss
    assert first > 0.8 and last < 0.2 * first, (first, last)
    ema = opt.ema_parameters(lit.diffusion_model.model)
    assert ema is not None and bool(torch.isfinite(ema).all())
