"""-m gpu: gradient parity AT THE BENCHMARK TRAINING CONFIGURATION (BASELINE configs[1]: default UNet, batch 128, bf16, train
mode) - the persistent 3x3 kernel as data-gradient kernel, the grouped weight-gradient launches over their long pixel runs, the
grouped column sums.  The mean-loss gradient is linear in the batch mean, so a batch of 2 distinct images tiled 64 times (same
timesteps, noise and Dropout2d masks per copy) has exactly the gradient of the 2-image batch, which torch autograd of the fp32
CPU oracle provides in seconds.  Reference: DDPM.training_step, src/dmme/diffusion_models/ddpm.py:53-81."""

import ctypes as C

import numpy as np
import pytest
import torch

from oracle import diffusion as D
from oracle import synth
from oracle import unet as O

pytestmark = pytest.mark.gpu

# bf16 gradients against the fp32 oracle, per tensor class: worst ||g - g_ref|| / ||g_ref|| over the class's tensors (tensors of
# fewer than 1024 values are pooled by class).  Every product dY * act(x) carries two bf16 roundings (u = 2^-9 each, independent),
# and a gradient entry is a sum of N*H*W such products with heavy cancellation, so its relative error is u * sqrt(2) times the
# cancellation ratio ||terms||_2 / |sum| - a few per cent for the deep 3x3 filters, less for the sums that cancel less.
# The worst tensors are the ones fed by the 4x4 maps (middle_layers.0: 16 pixels per image, so few terms to average over): measured
# conv3x3 4.3e-2, linear 4.1e-2, conv1x1 3.8e-2, conv / linear biases 5.6e-3, GroupNorm gamma 7.4e-3 / beta 6.1e-3 (the 1-D tensors
# sum over every pixel of a channel and cancel far less); the whole flat gradient is off by 7.9e-3.  IDDPM-64 at batch 32 measures
# the same (4.3e-2 / 4.1e-2 / 3.8e-2 / 3.3e-3 / 3.5e-3 / 3.7e-3).  Bounds = 1.25 x measured.
CLASS_BOUNDS = {"conv3x3": 5.4e-2, "conv1x1": 4.8e-2, "linear": 5.1e-2, "bias": 7.0e-3, "gn_gamma": 9.3e-3, "gn_beta": 7.7e-3}
FLAT_BOUND = 1.0e-2  # whole flat gradient, measured 7.9e-3
# The batch-128 run against the batch-2 run of the same library (other kernels: persistent vs 4-wave tiles, other summation orders,
# hence other roundings): two estimates with the SAME error statistics - measured B = 2 vs oracle 4.6e-2 / B = 128 vs oracle
# 4.3e-2 / B = 128 vs B = 2 3.8e-2 - so the batch-128 kernels are as accurate as the small-batch ones, bounded by the same table.


def _classes(net):
    out = {}
    named = dict(net.named_parameters())
    for name, p in named.items():
        if p.ndim == 4:
            out[name] = "conv3x3" if p.shape[-1] == 3 else "conv1x1"
        elif p.ndim == 2:
            out[name] = "linear"
        else:
            sib = named.get(name.rsplit(".", 1)[0] + ".weight")
            is_norm = sib is not None and sib.ndim == 1
            out[name] = ("gn_gamma" if name.endswith(".weight") else "gn_beta") if is_norm else "bias"
    return out


def _class_errors(got, want, classes):
    """worst per-tensor relative error by class (small tensors pooled per class)"""
    worst, pooled = {}, {}
    for k, c in classes.items():
        a, b = got[k].double().reshape(-1), want[k].double().reshape(-1)
        if a.numel() >= 1024:
            rel = float((a - b).norm() / (b.norm() + 1e-30))
            if rel > worst.get(c, (0.0, ""))[0]:
                worst[c] = (rel, k)
        else:
            e, n = pooled.get(c, (0.0, 0.0))
            pooled[c] = (e + float((a - b).pow(2).sum()), n + float(b.pow(2).sum()))
    for c, (e, n) in pooled.items():
        rel = (e / (n + 1e-30)) ** 0.5
        if rel > worst.get(c, (0.0, ""))[0]:
            worst[c] = (rel, "<pooled small tensors>")
    return worst


def _bwd_summary(net, B, H):
    from dmme_amd import _lib

    plan = net._plan_for(B, H, H, torch.device("cuda", 0))
    buf = C.create_string_buffer(4096)
    _lib.check(_lib.lib().dmme_unet_plan_bwd_summary(plan.h, buf, 4096))
    return dict(kv.split("=") for kv in buf.value.decode().split())


def _fwd_labels(net, B, H):
    from dmme_amd import _lib

    plan = net._plan_for(B, H, H, torch.device("cuda", 0))
    lib, buf, f, b = _lib.lib(), C.create_string_buffer(128), C.c_double(), C.c_double()
    out = []
    for i in range(lib.dmme_unet_plan_num_ops(plan.h)):
        _lib.check(lib.dmme_unet_plan_op_info(plan.h, i, buf, 128, C.byref(f), C.byref(b)))
        out.append(buf.value.decode())
    return out


def test_ddpm_batch128_bf16_train_gradients_vs_fp32_oracle():
    import dmme_amd

    cfg = O.UNetConfig()
    sd = O.make_state_dict(cfg, 23)
    T, reps = 1000, 64
    x0 = synth.uniform(1, (2, 3, 32, 32))
    t = torch.tensor([137, 862])
    z = synth.normal(2, (2, 3, 32, 32))
    masks = O.make_drop_masks(cfg, 2, 5)
    names = O.res_block_names(cfg)
    # ---- oracle: fp32 autograd on the 2-image batch
    sdr = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in sd.items()}
    _, abar = D.alpha_tables(D.linear_beta(T))
    loss_ref = D.training_loss(lambda xt, tt: O.unet_forward(sdr, cfg, xt, tt, drop_masks=masks), x0, t, z, abar)
    loss_ref.backward()
    want = {k: v.grad for k, v in sdr.items() if v.requires_grad}

    def run(B):
        r = B // 2
        net = dmme_amd.UNet(precision="bf16")
        net.load_state_dict(sd)
        net.cuda().train()
        # mask layout: per ResBlock [B][Cout]; the tiled batch repeats the two rows
        flat = torch.cat([masks[k].repeat(r, 1).reshape(-1) for k in names])
        net.inject_dropout_masks(flat.cuda())
        ddpm = dmme_amd.DDPM(net, T).cuda()
        loss = ddpm.training_step(x0.repeat(r, 1, 1, 1).cuda(), t=t.repeat(r).cuda(), noise=z.repeat(r, 1, 1, 1).cuda())
        loss.backward()
        torch.cuda.synchronize()
        return net, float(loss.detach()), {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}

    net2, loss2, g2 = run(2)
    net, loss128, g128 = run(2 * reps)
    # ---- the configuration really runs the kernels it is here for
    labels = _fwd_labels(net, 128, 32)
    assert sum(1 for l in labels if l.startswith("conv3x3_ws2_kernel")) >= 18, labels
    bw = _bwd_summary(net, 128, 32)
    assert int(bw["wgrad_group3x3_jobs"]) > 1000 and int(bw["wgrad_group3x3_layers"]) >= 40 and int(bw["wgrad_group1x1_layers"]) >= 20, bw
    assert int(bw["colsum_group_jobs"]) > 0 and int(bw["bias_group_jobs"]) > 0, bw
    assert int(bw.get("dgrad[conv3x3_ws2_kernel<11>]", 0)) >= 18, bw
    # ---- loss and gradients
    assert abs(loss128 - float(loss_ref)) <= 5e-3 * abs(float(loss_ref)), (loss128, float(loss_ref))
    assert abs(loss128 - loss2) <= 1e-3 * abs(loss2)
    classes = _classes(net)
    vs_ref = _class_errors(g128, want, classes)
    vs_b2 = _class_errors(g128, g2, classes)
    b2_vs_ref = _class_errors(g2, want, classes)
    print("B=128 bf16 vs fp32 oracle, worst relative error per tensor class:", {c: f"{v[0]:.3e} ({v[1]})" for c, v in vs_ref.items()})
    print("B=2   bf16 vs fp32 oracle:", {c: f"{v[0]:.3e}" for c, v in b2_vs_ref.items()})
    print("B=128 bf16 vs B=2 bf16 (same roundings, other kernels):", {c: f"{v[0]:.3e} ({v[1]})" for c, v in vs_b2.items()})
    total = float(torch.cat([(g128[k] - want[k]).reshape(-1) for k in want]).norm() / torch.cat([want[k].reshape(-1) for k in want]).norm())
    print(f"whole flat gradient: relative error {total:.3e}")
    for c, (rel, name) in vs_ref.items():
        assert rel <= CLASS_BOUNDS[c], f"{c}: {rel:.3e} > {CLASS_BOUNDS[c]} at {name}"
    assert total <= FLAT_BOUND, total
    for c, (rel, name) in vs_b2.items():
        assert rel <= CLASS_BOUNDS[c], f"{c}: B=128 vs B=2 {rel:.3e} at {name}"
    for c, (rel, name) in b2_vs_ref.items():  # and the large batch is no worse than the small one by more than the table's margin
        assert vs_ref[c][0] <= 1.25 * max(rel, 1e-3), (c, vs_ref[c], rel)


def test_iddpm64_batch32_bf16_train_gradients_vs_fp32_path():
    """BASELINE configs[3] shard: IDDPM ImageNet-64 UNet (attention at 16x16 / 8x8, 4 heads), batch 32, bf16, hybrid loss.  The
    reference's head merge mixes samples across the batch, so a tiled small batch is NOT equivalent here; the bf16 step is held
    against the library's own fp32 step on the same batch (that path is pinned against autograd of the oracle in test_gpu_iddpm)."""
    import dmme_amd
    from dmme_amd.models import iddpm
    from oracle import iddpm as OI

    B, T = 32, 4000
    x0, z = synth.uniform(1, (B, 3, 64, 64)).cuda(), synth.normal(2, (B, 3, 64, 64)).cuda()
    t = synth.randint(3, 1, T, B).cuda()
    cfg = OI.IUNetConfig(attention_depths=(3, 4))
    sd = OI.make_state_dict(cfg, 41)
    masks = OI.make_drop_masks(cfg, B, 7)
    flat = torch.cat([masks[k].reshape(-1) for k in OI.res_block_names(cfg)]).cuda()
    grads, losses, nets = {}, {}, {}
    for prec in ("fp32", "bf16"):
        net = iddpm.UNet(attention_depths=(3, 4), precision=prec)
        net.load_state_dict(sd)
        net.cuda().train()
        net.inject_dropout_masks(flat)
        idd = dmme_amd.IDDPM(net, timesteps=T).cuda()
        loss = idd.training_step(x0, t=t, noise=z)
        loss.backward()
        losses[prec] = float(loss)
        grads[prec] = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
        nets[prec] = net
    bw = _bwd_summary(nets["bf16"], B, 64)
    assert int(bw["wgrad_group3x3_jobs"]) > 1000 and int(bw.get("dgrad[conv3x3_ws2_kernel<11>]", 0)) >= 10, bw
    assert abs(losses["bf16"] - losses["fp32"]) <= 5e-3 * abs(losses["fp32"]), losses
    classes = _classes(nets["bf16"])
    worst = _class_errors(grads["bf16"], grads["fp32"], classes)
    print("IDDPM-64 B=32 bf16 vs fp32 path, worst relative error per tensor class:", {c: f"{v[0]:.3e} ({v[1]})" for c, v in worst.items()})
    for c, (rel, name) in worst.items():
        assert rel <= CLASS_BOUNDS[c], f"{c}: {rel:.3e} > {CLASS_BOUNDS[c]} at {name}"
