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
# single-pass bf16 against the fp32 oracle: (rel-RMS, max-abs / |want|max) per geometry at 1.25 x the measured values (the error
# budget is the DDPM network's, tests/test_gpu_unet.py / DESIGN.md section 2; the accurate mode is precision="bf16x3")
BF16_BOUNDS = {"iddpm_default_32": (9.7e-3, 1.08e-2),  # measured 7.72e-3, 8.64e-3
               "iddpm_imagenet64": (8.4e-3, 8.9e-3)}   # measured 6.65e-3, 7.06e-3


def _assert_bf16_close(got, want, tag):
    err = (got - want).abs()
    rel_rms = float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    max_rel = float(err.max() / want.abs().max())
    print(f"bf16 [{tag}]: rel-RMS {rel_rms:.3e}, max-abs / |want|max {max_rel:.3e} (|want|max {float(want.abs().max()):.3f})")
    lim = BF16_BOUNDS[tag]
    assert rel_rms <= lim[0] and max_rel <= lim[1], (tag, rel_rms, max_rel, lim)


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
    _assert_bf16_close(got, want, "iddpm_default_32")


def test_unet_64x64_config4_bf16_vs_oracle():
    """BASELINE configs[3] geometry in the benchmark precision (MFMA head-view attention at S = 256 and S = 64, batch-mixing merge)."""
    cfg = OI.IUNetConfig(attention_depths=(3, 4))
    net, sd = _build(cfg, 43, "bf16")
    B = 4
    x = synth.normal(10, (B, 3, 64, 64))
    t = torch.tensor([2500])
    want = OI.unet_forward(sd, cfg, x, t)
    with torch.no_grad():
        got = net(x.cuda(), t.cuda()).cpu()
    _assert_bf16_close(got, want, "iddpm_imagenet64")


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


@pytest.mark.parametrize("name", ["default32", "imagenet64"])
def test_unet_fp16r32_vs_oracle_within_1e_3(name):
    """round 5: the reduced-precision mode INSIDE north_star's 1e-3 serves the Improved-DDPM UNet too - fp32
    tensors and three-pass split-fp16 products on the full-resolution level (the scale-shift norms are per-(image, channel) rows either
    way; the six-cout output conv runs the thin kernel's split form), plain half below it.  Bound: 1e-3 itself.  Plain fp16 on the same
    inputs, for scale: 1.4e-3 / 1.8e-3."""
    cfg, side, t = (OI.IUNetConfig(), 32, [900]) if name == "default32" else (OI.IUNetConfig(attention_depths=(3, 4)), 64, [17, 3011])
    net, sd = _build(cfg, 43, "fp16r32")
    x = synth.normal(10, (2, 3, side, side))
    tt = torch.tensor(t)
    want = OI.unet_forward(sd, cfg, x, tt)
    with torch.no_grad():
        got = net(x.cuda(), tt.cuda()).cpu()
    e = (got - want).abs()
    rms = float(e.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
    print(f"IDDPM {name} fp16r32 vs oracle: max|err| {float(e.max()):.3e} rel-rms {rms:.3e}")
    assert float(e.max()) <= 1.0e-3 and rms <= 6.0e-4
    # the sampler on top of it: one IDDPM step stays finite and deterministic
    import dmme_amd

    proc = dmme_amd.IDDPM(net, 100).cuda() if hasattr(dmme_amd, "IDDPM") else None
    if proc is not None:
        torch.manual_seed(3)
        a = proc.sampling_step(x.cuda(), torch.tensor([50]).cuda())
        torch.manual_seed(3)
        b = proc.sampling_step(x.cuda(), torch.tensor([50]).cuda())
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


# ------------------------------------------------------------------------------------------ process: loss, gradients, sampler


def test_vlb_kernel_value_and_gradient_vs_reference_golden(golden):
    """dmme_iddpm_loss alone on synthetic tensors: L_vlb and d L_vlb / d(model output) (a t == 1 row included)."""
    import dmme_amd
    from dmme_amd import _lib

    g = golden("iddpm_process")
    seed, T, B, x0s, zs, ms = (int(v) for v in g["train_meta"])
    t = torch.from_numpy(g["train_t"]).cuda()
    idd = dmme_amd.IDDPM(torch.nn.Identity(), timesteps=T).cuda()
    x0 = synth.uniform(x0s, (B, 3, 32, 32)).cuda()
    mo = (0.5 * synth.normal(int(g["vlb_meta"][0]), (B, 6, 32, 32))).cuda()
    x_t = synth.normal(int(g["vlb_meta"][1]), (B, 3, 32, 32)).cuda()
    loss = torch.empty(3, device="cuda")
    d_out = torch.empty_like(mo)
    scratch = torch.empty(1024, device="cuda")
    _lib.check(_lib.lib().dmme_iddpm_loss(_lib.ptr(mo), _lib.ptr(x_t), _lib.ptr(x0), _lib.ptr(x_t), _lib.ptr(t), _lib.ptr(idd._coef), B, 3 * 32 * 32,
                                          0.0, 1.0, _lib.ptr(loss), _lib.ptr(d_out), 1.0, _lib.ptr(scratch), _lib.stream_ptr()))
    np.testing.assert_allclose(loss[0].item(), g["vlb_value"], rtol=2e-5)
    want = g["vlb_dout"]
    got = d_out.cpu().numpy()
    assert np.all(got[:, :3] == 0)  # stop-gradient on the predicted noise
    # the t == 1 rows difference two fp32 CDFs: allow the cancellation noise of erf there
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=1e-4 * float(np.abs(want).max()))
    np.testing.assert_allclose(got[1:], want[1:], rtol=1e-4, atol=1e-6 * float(np.abs(want).max()))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_training_step_loss_and_grads_vs_reference_golden(golden, mode):
    import dmme_amd

    g = golden("iddpm_process")
    cfg = OI.TINY
    seed, T, B, x0s, zs, ms = (int(v) for v in g["train_meta"])
    t = torch.from_numpy(g["train_t"]).cuda()
    x0, z = synth.uniform(x0s, (B, 3, 32, 32)).cuda(), synth.normal(zs, (B, 3, 32, 32)).cuda()
    for sched in ("cosine", "linear"):
        net, _ = _build(cfg, seed, "fp32", train=(mode == "train"))
        if mode == "train":
            masks = OI.make_drop_masks(cfg, B, ms)
            net.inject_dropout_masks(torch.cat([masks[k].reshape(-1) for k in OI.res_block_names(cfg)]).cuda())
        idd = dmme_amd.IDDPM(net, timesteps=T, schedule=sched).cuda()
        loss = idd.training_step(x0, t=t, noise=z)
        np.testing.assert_allclose(loss.item(), g[f"train_{sched}_{mode}_loss"], rtol=2e-5)
        if sched != "cosine":
            continue
        loss.backward()
        worst, n = {}, 0
        for name, p in net.named_parameters():
            want = g[f"train_{mode}_grad::{name}"]
            err = np.abs(p.grad.cpu().numpy() - want).max()
            # 1e-3 of the tensor's largest entry: the t == 1 row feeds d L_vlb / dv through the difference of two fp32 normal
            # CDFs at +-1/255 around a mean divided by sigma_1 ~ 1e-2, which amplifies the 1e-6 forward differences
            if err > 2e-6 + 1e-3 * np.abs(want).max():
                worst[name] = (float(err), float(np.abs(want).max()))
            n += 1
        assert n == 450
        assert not worst, f"{len(worst)} gradients off, e.g. {list(worst.items())[:6]}"
    net, _ = _build(cfg, seed, "fp32")
    with torch.no_grad():
        vlb = dmme_amd.IDDPM(net, timesteps=T, loss_type="vlb").cuda().training_step(x0, t=t, noise=z)
        assert dmme_amd.IDDPM(net, timesteps=T, loss_type="simple").cuda().training_step(x0, t=t, noise=z) is None
    np.testing.assert_allclose(vlb.item(), g["train_vlb_only_loss"], rtol=5e-4)  # dominated by the ill-conditioned t == 1 NLL row


def test_hybrid_grads_without_t1_row_vs_oracle_autograd():
    """Same step with every t > 1 (KL rows only, well conditioned): the tight gradient bound the DDPM path meets."""
    import dmme_amd

    cfg, seed, T, B = OI.TINY, 31, 100, 4
    net, _ = _build(cfg, seed, "fp32")
    sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in OI.make_state_dict(cfg, seed).items()}
    x0, z = synth.uniform(810, (B, 3, 32, 32)), synth.normal(812, (B, 3, 32, 32))
    t = torch.tensor([3, 57, 99, 2])
    want = OI.training_loss(lambda xt, tt: OI.unet_forward(sd, cfg, xt, tt), x0, t, z, OI.schedule_tables(T), gamma=0.05)
    want.backward()
    idd = dmme_amd.IDDPM(net, timesteps=T, gamma=0.05).cuda()
    loss = idd.training_step(x0.cuda(), t=t.cuda(), noise=z.cuda())
    loss.backward()
    np.testing.assert_allclose(loss.item(), want.item(), rtol=2e-5)
    bad = {}
    for name, p in net.named_parameters():
        wg = sd[name].grad.numpy()
        err = np.abs(p.grad.cpu().numpy() - wg).max()
        if err > 2e-6 + 2e-4 * np.abs(wg).max():
            bad[name] = (float(err), float(np.abs(wg).max()))
    assert not bad, f"{len(bad)} gradients off, e.g. {list(bad.items())[:6]}"


def test_attention_config_grads_vs_oracle_autograd():
    """TINY_ATTN (4-head blocks at S = 256, 64 and the middle block) with a shared timestep row (t.shape == (1,)):
    exercises the multi-head backward incl. the sample-mixing merge and the nt == 1 accumulation of the projection rows."""
    import dmme_amd

    cfg, seed, T, B = OI.TINY_ATTN, 32, 100, 3
    net, _ = _build(cfg, seed, "fp32")
    sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in OI.make_state_dict(cfg, seed).items()}
    x = synth.normal(3, (B, 3, 32, 32))
    for t in (torch.tensor([40]), torch.tensor([5, 60, 99])):
        for v in sd.values():
            v.grad = None
        net.zero_grad(set_to_none=True)
        w = synth.normal(4, (B, 6, 32, 32))
        want = (OI.unet_forward(sd, cfg, x, t) * w).sum()
        want.backward()
        y = net(x.cuda(), t.cuda())
        (y * w.cuda()).sum().backward()
        bad = {}
        for name, p in net.named_parameters():
            wg = sd[name].grad.numpy()
            err = np.abs(p.grad.cpu().numpy() - wg).max()
            if err > 1e-5 + 2e-4 * np.abs(wg).max():
                bad[name] = (float(err), float(np.abs(wg).max()))
        assert not bad, f"t={t.tolist()}: {len(bad)} gradients off, e.g. {list(bad.items())[:6]}"


def test_full_size_bf16_training_step_runs_and_matches_oracle_loss():
    import dmme_amd

    cfg, seed, T, B = OI.IUNetConfig(), 41, 1000, 4
    net, sd = _build(cfg, seed, "bf16")
    x0, z = synth.uniform(1, (B, 3, 32, 32)), synth.normal(2, (B, 3, 32, 32))
    t = torch.tensor([1, 17, 803, 999])
    with torch.no_grad():
        want = OI.training_loss(lambda xt, tt: OI.unet_forward(sd, cfg, xt, tt), x0, t, z, OI.schedule_tables(T))
    idd = dmme_amd.IDDPM(net, timesteps=T).cuda()
    loss = idd.training_step(x0.cuda(), t=t.cuda(), noise=z.cuda())
    loss.backward()
    assert abs(loss.item() - want.item()) < 3e-2 * abs(want.item())
    gn = float(net.flat_grad().norm())
    assert np.isfinite(gn) and gn > 0


def test_sampler_trajectories_vs_reference_golden(golden):
    import dmme_amd

    g = golden("iddpm_process")
    cfg = OI.TINY
    seed, T, B, xs, zs0 = (int(v) for v in g["traj_meta"])
    net, _ = _build(cfg, seed, "fp32")
    shape = (B, 3, 32, 32)
    for sched in ("cosine", "linear"):
        idd = dmme_amd.IDDPM(net, timesteps=T, schedule=sched).cuda()
        x = synth.normal(xs, shape).cuda()
        with torch.no_grad():
            for k in range(T):
                x = idd.sampling_step(x, torch.tensor([T - k], device="cuda"), noise=synth.normal(zs0 + k, shape).cuda())
                if f"traj_{sched}_step{k}" in g.files:
                    want = g[f"traj_{sched}_step{k}"]
                    np.testing.assert_allclose(x.cpu().numpy(), want, atol=2e-5 * max(1.0, float(np.abs(want).max())), rtol=0, err_msg=f"{sched} step {k}")
    with torch.no_grad():
        img = dmme_amd.LitIDDPM(model=net, timesteps=20).cuda().generate((2, 3, 32, 32))
    assert img.shape == (2, 3, 32, 32) and bool(torch.isfinite(img).all())


# ------------------------------------------------------------------------------------------ multi-head attention kernels


def _mha_core_reference(qkv, heads, round_scaled_k=False):
    """qkv (N, S, 3C) fp32 -> (N, S, C): oracle.iddpm.multi_head_attention without norm / projections.
    round_scaled_k: K * C^-0.5 rounded to bf16 before the product, as a 16-bit autocast of the reference (and the generic kernel) does"""
    N, S, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    x = qkv.reshape(N, S, heads, 3 * d).permute(0, 2, 1, 3).reshape(N * heads, S, 3 * d)
    q, k, v = x[..., :d], x[..., d : 2 * d], x[..., 2 * d :]
    ks = k * C**-0.5
    if round_scaled_k:
        ks = ks.to(torch.bfloat16).to(torch.float32)
    w = torch.softmax(torch.bmm(q, ks.transpose(1, 2)), dim=2)
    o = torch.bmm(w, v).reshape(heads, N, S, d)  # rows b*heads + h re-read as (head', b')
    return o.permute(1, 2, 0, 3).reshape(N, S, C)


@pytest.mark.parametrize("S,C,heads,N", [(256, 256, 4, 3), (64, 256, 4, 5), (16, 256, 4, 2), (256, 128, 1, 2), (64, 128, 2, 3)])
def test_attention_heads_kernels(S, C, heads, N):
    from dmme_amd import _lib

    qkv = synth.normal(S + C + heads, (N, S, 3 * C)) * 1.5
    want = _mha_core_reference(qkv.to(torch.bfloat16).to(torch.float32), heads)
    for dt, tdt, tol in ((_lib.F32, torch.float32, 2e-5), (_lib.BF16, torch.bfloat16, 5e-3 * max(1.0, float(want.abs().max())))):  # bf16: K scale, probability and output roundings (test_gpu_ops.BF16_ATTN_RTOL)
        for force_generic in (1, 0):
            mfma = dt == _lib.BF16 and not force_generic and (C // heads) in (64, 128, 256) and S >= 64
            ref = _mha_core_reference(qkv, heads) if dt == _lib.F32 else (want if mfma else _mha_core_reference(qkv.to(torch.bfloat16).to(torch.float32), heads, True))
            q = qkv.to(tdt).cuda().contiguous()
            out = torch.empty((N, S, C), dtype=tdt, device="cuda")
            _lib.check(_lib.lib().dmme_attention_heads(dt, _lib.ptr(q), N, S, C, heads, _lib.ptr(out), force_generic, _lib.stream_ptr()))
            err = float((out.float().cpu() - ref).abs().max())
            assert err < tol, (dt, force_generic, err)


def test_full_size_bf16_grads_track_fp32_grads():
    """default IDDPM UNet, B = 3: the bf16 step (MFMA multi-head attention forward + backward, grouped weight gradients) against the
    fp32 step of the same library (generic kernels, pinned against autograd on the small configurations above)."""
    import dmme_amd

    cfg, seed, T, B = OI.IUNetConfig(), 41, 1000, 3
    x0, z = synth.uniform(1, (B, 3, 32, 32)).cuda(), synth.normal(2, (B, 3, 32, 32)).cuda()
    t = torch.tensor([17, 803, 400]).cuda()
    grads = {}
    for prec in ("fp32", "bf16"):
        net, _ = _build(cfg, seed, prec)
        idd = dmme_amd.IDDPM(net, timesteps=T, gamma=0.05).cuda()
        idd.training_step(x0, t=t, noise=z).backward()
        grads[prec] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    from tests.test_gpu_grad_b128 import CLASS_BOUNDS, _class_errors, _classes

    worst = _class_errors({k: v.cpu() for k, v in grads["bf16"].items()}, {k: v.cpu() for k, v in grads["fp32"].items()}, _classes(net))
    print("IDDPM default UNet B=3 bf16 vs fp32, worst relative error per tensor class:", {c: f"{v[0]:.3e} ({v[1]})" for c, v in worst.items()})
    for c, (rel, name) in worst.items():  # the per-class budget of the batch-128 parity test (derived there)
        assert rel <= CLASS_BOUNDS[c], (c, name, rel)


def test_config4_bf16_grads_vs_oracle_autograd_b2():
    """BASELINE configs[3] geometry (64x64, 4-head attention at 16x16 / 8x8, scale-shift ResBlocks) at B = 2, benchmark precision: the
    hybrid-loss training step's gradients against AUTOGRAD THROUGH THE ORACLE (fp32 torch restatement of models/iddpm.py:125-265 and
    equations/iddpm/losses.py) - not against this library's own fp32 path as the self-comparison above.  Per tensor class, inside the
    budget the batch-128 DDPM parity test derives for bf16 (tests/test_gpu_grad_b128.py: CLASS_BOUNDS)."""
    import dmme_amd
    from tests.test_gpu_grad_b128 import CLASS_BOUNDS, _class_errors, _classes

    cfg, seed, T, B = OI.IUNetConfig(attention_depths=(3, 4)), 47, 4000, 2
    net, _ = _build(cfg, seed, "bf16")
    sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in OI.make_state_dict(cfg, seed).items()}
    x0, z = synth.uniform(21, (B, 3, 64, 64)), synth.normal(22, (B, 3, 64, 64))
    t = torch.tensor([37, 2811])  # (t > 1: KL rows - the t == 1 NLL row is ill-conditioned, see above)
    want = OI.training_loss(lambda xt, tt: OI.unet_forward(sd, cfg, xt, tt), x0, t, z, OI.schedule_tables(T), gamma=0.05)
    want.backward()
    idd = dmme_amd.IDDPM(net, timesteps=T, gamma=0.05).cuda()
    loss = idd.training_step(x0.cuda(), t=t.cuda(), noise=z.cuda())
    loss.backward()
    assert abs(loss.item() - want.item()) < 3e-2 * abs(want.item()), (loss.item(), want.item())
    got = {k: p.grad.detach().float().cpu() for k, p in net.named_parameters()}
    ref = {k: sd[k].grad for k in got}
    worst = _class_errors(got, ref, _classes(net))
    print("IDDPM config-4 geometry B=2 bf16 vs oracle autograd, worst relative error per tensor class:", {c: f"{v[0]:.3e} ({v[1]})" for c, v in worst.items()})
    for c, (rel, name) in worst.items():
        assert rel <= CLASS_BOUNDS[c], (c, name, rel)
