"""-m gpu: the replayable denoising step (dmme_chain_*) and the host-side contracts around a forward:
stale packed weights after in-place parameter writes, a second forward before backward, the gradient w.r.t. the input."""

import ctypes as C

import numpy as np
import pytest
import torch

from oracle import diffusion as D
from oracle import synth
from oracle import unet as O

pytestmark = pytest.mark.gpu


def _tiny(precision="fp32", seed=11):
    import dmme_amd

    cfg = O.TINY
    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                        cfg.attention_depths, precision=precision)
    sd = O.make_state_dict(cfg, seed)
    net.load_state_dict(sd, strict=True)
    return net.cuda().eval(), sd, cfg


def _state(runner):
    torch.cuda.synchronize()
    return [int(v) for v in runner.state[:4].cpu()]


# ------------------------------------------------------------------------------------------ the update kernel alone
@pytest.mark.parametrize("kind", ["ddpm", "ddim", "iddpm"])
def test_chain_update_bit_exact_vs_eager_kernels_and_state_advance(kind):
    """dmme_chain_update (coefficients, timestep and Philox offset read from device memory, noise drawn in the kernel) against
    the eager pair dmme_randn + dmme_{ddpm,ddim,iddpm}_step fed host scalars - bit for bit, including t == 1 (no noise added, the
    offset advances all the same) - and the loop state it leaves behind."""
    import dmme_amd
    from dmme_amd import _lib
    from dmme_amd.models import iddpm as iddpm_models

    lib = _lib.lib()
    B, shape = 3, (3, 3, 16, 16)
    if kind == "ddim":
        proc = dmme_amd.DDIM(torch.nn.Identity(), 100, 5).cuda()
        first = 5
    elif kind == "iddpm":
        proc = dmme_amd.IDDPM(torch.nn.Identity(), 50).cuda()
        first = 3
    else:
        proc = dmme_amd.DDPM(torch.nn.Identity(), 50).cuda()
        first = 3
    n, rows, ttab = proc._chain_tables()
    coef = torch.tensor(rows, dtype=torch.float32).reshape(-1).cuda()
    tt = torch.tensor(ttab, dtype=torch.int64).cuda()
    state = torch.zeros(8, dtype=torch.int64, device="cuda")
    out_ch = 6 if kind == "iddpm" else 3
    x = synth.normal(1, shape).cuda()
    want = x.clone()
    seed, off = 0xC0FFEE, 1000
    quads = x.numel() // 4
    _lib.check(lib.dmme_chain_set(_lib.ptr(state), first, _lib.ptr(tt), seed, off, _lib.stream_ptr()))
    i = first
    while i >= 1:
        out = synth.normal(100 + i, (B, out_ch, 16, 16)).cuda()
        _lib.check(lib.dmme_chain_update(proc._chain_kind, _lib.ptr(x), _lib.ptr(out), _lib.ptr(coef), _lib.ptr(tt), _lib.ptr(state), B, 3 * 16 * 16, _lib.stream_ptr()))
        z = torch.empty_like(want)
        _lib.check(lib.dmme_randn(_lib.ptr(z), z.numel(), seed, off + (first - i) * quads, _lib.stream_ptr()))
        if kind == "ddim":
            proc._ddim_update(want, out, i)
        else:
            proc._reverse_update(want, out, ttab[i], z)
        torch.cuda.synchronize()
        assert torch.equal(x, want), f"{kind}: update differs at loop index {i}"
        i -= 1
        st = [int(v) for v in state[:4].cpu()]
        assert st == [i, ttab[i], off + (first - i) * quads, seed], f"{kind}: state {st} after stepping to {i}"
        assert int(state[4].cpu()) & 0xFFFFFFFF == 0  # ticket back to zero
        if i == 1 and kind != "ddim":
            assert ttab[i] == 1  # the next update runs the t == 1 branch


# ------------------------------------------------------------------------------------------ whole chains
def test_generate_through_the_captured_step_equals_the_eager_loop():
    """DDPM / DDIM / IDDPM `generate` (one hipGraph of UNet + noise + update + state advance, replayed) against the eager host loop
    (per-step launches, host scalars, dmme_randn) under the same torch seed: bit-identical samples."""
    import dmme_amd
    from dmme_amd.models import iddpm as iddpm_models

    net, _, _ = _tiny()
    shape = (2, 3, 32, 32)
    for proc in (dmme_amd.DDPM(net, 30).cuda(), dmme_amd.DDIM(net, 100, 7).cuda()):
        torch.manual_seed(77)
        got = proc.generate(shape).clone()
        assert proc._runner is not None and (proc._runner.graph is not None or getattr(net, "_graph_disabled", False))
        torch.manual_seed(77)
        x = dmme_amd.gaussian(shape, device="cuda")
        with torch.no_grad():
            if isinstance(proc, dmme_amd.DDIM):
                for i in range(proc.sub_timesteps, 0, -1):
                    proc._ddim_update(x, net(x, proc.tau_tensor(i, x.device)), i)
            else:
                for t in range(proc.timesteps, 0, -1):
                    proc._reverse_update(x, net(x, proc.timestep_tensor(t, x.device)), t, None)
        assert torch.equal(got, x), type(proc).__name__
        torch.manual_seed(78)  # a captured graph serves another seed (the seed is device state, not a captured constant)
        other = proc.generate(shape)
        assert isinstance(proc, dmme_amd.DDIM) or not torch.equal(other, got)
    inet = iddpm_models.UNet(3, 4, 8, 2, 0.0, (4, 8), 1, (2,)).cuda().eval()
    idd = dmme_amd.IDDPM(inet, 20).cuda()
    torch.manual_seed(5)
    got = idd.generate(shape).clone()
    torch.manual_seed(5)
    x = dmme_amd.gaussian(shape, device="cuda")
    with torch.no_grad():
        for t in range(20, 0, -1):
            idd._reverse_update(x, inet(x, idd.timestep_tensor(t, x.device)), t, None)
    assert torch.equal(got, x) and bool(torch.isfinite(got).all())


def test_ddim_chain_through_the_runner_vs_reference_golden(golden):
    """the DDIM chain has no noise, so the replayed step can be held against the reference's own trajectory directly"""
    import dmme_amd

    g = golden("traj_tiny")
    seed, T, B, sx, sz = [int(v) for v in g["traj_meta"]]
    net, _, _ = _tiny()
    for T_, S_, sch in ((100, 5, "quadratic"), (1000, 50, "quadratic")):
        ddim = dmme_amd.DDIM(net, T_, S_, sch).cuda()
        x = synth.normal(sx, (B, 3, 32, 32)).cuda()
        runner = ddim.chain_runner(x)
        runner.set(S_, 0, 0)
        for i in range(S_, 0, -1):
            runner.step()
            if i in (S_, S_ - 1, 2, 1):
                np.testing.assert_allclose(x.cpu().numpy(), g[f"traj_ddim_{sch}_{T_}_{S_}_i{i}"], atol=1e-4, rtol=0, err_msg=f"ddim {sch} i={i}")
        assert _state(runner)[:2] == [0, 0]


def test_lit_forward_with_host_integer_t_matches_sampling_step():
    """LitDDPM.forward(x_t, t: int) (reference lit_modules/ddpm.py:65-79; the per-step caller is callbacks/generate.py:82): same
    draw as sampling_step with a (1,) tensor under the same seed, a NEW tensor each call, input untouched."""
    import dmme_amd

    net, _, _ = _tiny()
    lit = dmme_amd.LitDDPM(model=net, timesteps=40).cuda().eval()
    x = synth.normal(3, (2, 3, 32, 32)).cuda()
    keep = x.clone()
    with torch.no_grad():
        for t in (40, 17, 1):
            torch.manual_seed(9)
            a = lit(x, t)
            torch.manual_seed(9)
            b = lit.diffusion_model.sampling_step(x, torch.tensor([t], device="cuda"))
            assert torch.equal(a, b) and a.data_ptr() != x.data_ptr() and torch.equal(x, keep), t
    lit2 = dmme_amd.LitDDIM(model=net, timesteps=100, sample_steps=5).cuda().eval()
    with torch.no_grad():
        for i in (5, 2, 1):
            a = lit2(x, i)
            b = lit2.diffusion_model.sampling_step(x, torch.tensor([i], device="cuda"))
            assert torch.equal(a, b), i


# ------------------------------------------------------------------------------------------ stale packed weights
def test_parameter_writes_after_cuda_reach_the_kernels():
    """After .cuda() every Parameter is rebound with `p.data = view`; load_state_dict, a torch.optim step and a manual copy_ then
    write through counters flat._version never sees.  Forward, backward weights and the captured sampling graph must follow."""
    import dmme_amd

    cfg = O.TINY
    net, sd1, _ = _tiny(seed=11)
    x = synth.normal(1, (2, 3, 32, 32))
    t = torch.tensor([37])
    xc, tc = x.cuda(), t.cuda()
    with torch.no_grad():
        y1 = net(xc, tc).cpu()
    assert float((y1 - O.unet_forward(sd1, cfg, x, t)).abs().max()) < 1e-5
    sd2 = O.make_state_dict(cfg, 12)
    net.load_state_dict(sd2)  # on the GPU module: p.copy_ through the rebound Parameters
    with torch.no_grad():
        y2 = net(xc, tc).cpu()
    assert float((y2 - O.unet_forward(sd2, cfg, x, t)).abs().max()) < 1e-5, "forward ran on stale packed weights after load_state_dict"
    # stock torch.optim.Adam on the GPU module (torch optimisers work on it)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    sd3 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    assert float((sd3["input_conv.weight"] - sd2["input_conv.weight"]).abs().max()) > 1e-3
    with torch.no_grad():
        y3 = net(xc, tc).cpu()
    assert float((y3 - O.unet_forward(sd3, cfg, x, t)).abs().max()) < 1e-5, "forward ran on stale packed weights after an optimiser step"
    # the captured chain step re-captures on new weights
    ddim = dmme_amd.DDIM(net, 100, 5).cuda()
    xs = synth.normal(4, (2, 3, 32, 32)).cuda()
    r = ddim.chain_runner(xs)
    r.set(5, 0, 0)
    r.step()
    first = xs.clone()
    with torch.no_grad():
        net.input_conv.weight.mul_(1.5)
    xs.copy_(synth.normal(4, (2, 3, 32, 32)).cuda())
    r.set(5, 0, 0)
    r.step()
    sd4 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tau = D.tau_table(100, 5)
    _, abar = D.alpha_tables(D.linear_beta(100))
    x0 = synth.normal(4, (2, 3, 32, 32))
    want = D.ddim_step(x0, int(tau[5]), int(tau[4]), O.unet_forward(sd4, cfg, x0, torch.tensor([int(tau[5])])), abar)
    assert not torch.equal(xs, first) and float((xs.cpu() - want).abs().max()) < 1e-5, "replayed step ran on stale weights"


def test_second_forward_of_the_same_shape_before_backward_is_refused():
    """the saved activations live in the per-shape plan workspace: a backward whose forward was overwritten must raise"""
    import dmme_amd

    net, _, _ = _tiny()
    net.train()
    x1 = synth.normal(1, (2, 3, 32, 32)).cuda()
    x2 = synth.normal(2, (2, 3, 32, 32)).cuda()
    t = torch.tensor([5, 9]).cuda()
    y1 = net(x1, t)
    y2 = net(x2, t)  # same (B, H, W, dtype): overwrites y1's activations
    with pytest.raises(RuntimeError, match="overwrote"):
        (y1.sum() + y2.sum()).backward()
    net.zero_grad()
    y3 = net(x1, t)
    with torch.no_grad():
        net(x2[:1], t[:1])  # another batch size: its own plan and workspace, y3's graph stays valid
    y3.sum().backward()
    assert float(net.flat_grad().abs().sum()) > 0


@pytest.mark.parametrize("full,prec,tol", [(False, "fp32", 2e-5), (True, "fp32", 2e-4), (True, "bf16", 3e-2)])
def test_input_gradient_vs_oracle_autograd(full, prec, tol):
    """x.requires_grad: autograd receives dL/dx (the data gradient of input_conv) instead of None - against torch autograd of
    the oracle (relative to the gradient's max)."""
    import dmme_amd

    cfg = O.UNetConfig(dropout=0.0) if full else O.TINY
    sd = O.make_state_dict(cfg, 3)
    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks,
                        cfg.attention_depths, precision=prec)
    net.load_state_dict(sd)
    net.cuda().eval()
    x = synth.normal(1, (2, 3, 32, 32))
    t = torch.tensor([11, 700 if full else 70])
    w = synth.normal(2, (2, 3, 32, 32))
    xr = x.clone().requires_grad_(True)
    sdr = {k: v.clone() for k, v in sd.items()}
    (O.unet_forward(sdr, cfg, xr, t) * w).sum().backward()
    xg = x.clone().cuda().requires_grad_(True)
    y = net(xg, t.cuda())
    (y * w.cuda()).sum().backward()
    assert xg.grad is not None and xg.grad.shape == x.shape
    scale = float(xr.grad.abs().max())
    assert float((xg.grad.cpu() - xr.grad).abs().max()) <= tol * scale
