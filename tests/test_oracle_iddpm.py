"""Pin the Improved-DDPM CPU oracle (oracle/iddpm.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py: iddpm_unet, iddpm_process).  CPU only."""

import numpy as np
import pytest
import torch

from oracle import iddpm as OI
from oracle import synth

FP32_ATOL = 1e-5


def test_param_table_default_counts(golden):
    g = golden("iddpm_unet")
    tab = OI.param_table(OI.IUNetConfig())
    n = sum(int(np.prod(s)) for k, s, r in tab if r != "buffer")
    assert n == int(g["full_nparams"]) == 36_168_070


@pytest.mark.parametrize("tag,cfg", [("tiny", OI.TINY), ("attn", OI.TINY_ATTN)])
def test_unet_vs_reference(golden, tag, cfg):
    g = golden("iddpm_unet")
    sd = OI.make_state_dict(cfg, int(g[f"{tag}_seed"]))
    for c in range(int(g[f"{tag}_ncases"])):
        B = int(g[f"{tag}_case{c}_B"])
        x = synth.normal(int(g[f"{tag}_case{c}_xseed"]), (B, 3, 32, 32))
        y = OI.unet_forward(sd, cfg, x, torch.from_numpy(g[f"{tag}_case{c}_t"]))
        assert y.shape == (B, 6, 32, 32)
        np.testing.assert_allclose(y.numpy(), g[f"{tag}_case{c}_y"], atol=FP32_ATOL, rtol=0)
    # B = 2 with per-module activations: the head merge mixes the two samples
    x = synth.normal(int(g[f"{tag}_acts_xseed"]), (2, 3, 32, 32))
    cap = {}
    y = OI.unet_forward(sd, cfg, x, torch.from_numpy(g[f"{tag}_acts_t"]), capture=cap)
    np.testing.assert_allclose(y.numpy(), g[f"{tag}_acts_y"], atol=FP32_ATOL, rtol=0)
    keys = [k for k in g.files if k.startswith(f"{tag}_act::")]
    assert len(keys) == len(cap)
    for k in keys:
        np.testing.assert_allclose(cap[k.split("::")[1]].numpy(), g[k], atol=FP32_ATOL, rtol=0, err_msg=k)
    # train mode with injected Dropout2d masks
    B, mseed, xseed = (int(v) for v in g[f"{tag}_train_meta"])
    y = OI.unet_forward(sd, cfg, synth.normal(xseed, (B, 3, 32, 32)), torch.from_numpy(g[f"{tag}_train_t"]), drop_masks=OI.make_drop_masks(cfg, B, mseed))
    np.testing.assert_allclose(y.numpy(), g[f"{tag}_train_y"], atol=FP32_ATOL, rtol=0)


def test_head_merge_mixes_samples():
    """batched output differs from per-sample output (the quirk is reproduced, not fixed)."""
    cfg = OI.TINY_ATTN
    sd = OI.make_state_dict(cfg, 32)
    x = synth.normal(1, (2, 3, 32, 32))
    t = torch.tensor([5])
    both = OI.unet_forward(sd, cfg, x, t)
    solo = OI.unet_forward(sd, cfg, x[:1], t)
    assert float((both[:1] - solo).abs().max()) > 1e-3


def test_unet_full_vs_reference(golden):
    g = golden("iddpm_unet")
    cfg = OI.IUNetConfig()
    sd = OI.make_state_dict(cfg, int(g["full_seed"]))
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    cap = {}
    y = OI.unet_forward(sd, cfg, x, torch.from_numpy(g["full_t_one"]), capture=cap)
    np.testing.assert_allclose(y.numpy(), g["full_y_one"], atol=FP32_ATOL, rtol=0)
    for k in [k for k in g.files if k.startswith("full_actdigest::")]:
        ok, err = synth.digest_close(cap[k.split("::")[1]], g[k], atol=FP32_ATOL, rtol=1e-5)
        assert ok, (k, err)
    pre = "down_layers.3.attention"
    y = OI.multi_head_attention(sd, pre, synth.normal(771, (3, 256, 8, 8)), 32, 4)
    np.testing.assert_allclose(y.numpy(), g["layer_mha256_y"], atol=FP32_ATOL, rtol=0)


def test_schedules_bit_exact(golden):
    g = golden("iddpm_process")
    for T in (100, 1000, 4000):
        beta, alpha, abar = OI.schedule_tables(T, "cosine")
        assert np.array_equal(beta.numpy(), g[f"cos_beta_{T}"])
        assert np.array_equal(alpha.numpy(), g[f"cos_alpha_{T}"])
        assert np.array_equal(abar.numpy(), g[f"cos_abar_{T}"])
    beta, _, abar = OI.schedule_tables(4000, "linear", start=2.5e-5, end=0.005)
    assert np.array_equal(beta.numpy(), g["lin_beta_4000"])
    assert np.array_equal(abar.numpy(), g["lin_abar_4000"])
    assert int(g["bad_schedule_raises"]) == 1
    with pytest.raises(NotImplementedError):
        OI.schedule_tables(100, "sqrt")


def test_interpolate_variance(golden):
    g = golden("iddpm_process")
    v = synth.uniform(800, (4, 3, 8, 8), -0.5, 1.5)
    bt = torch.tensor([0.02, 1e-4, 0.3, 0.999]).reshape(4, 1, 1, 1)
    btt = torch.tensor([0.01, 0.0, 0.2, 0.5]).reshape(4, 1, 1, 1)
    np.testing.assert_allclose(OI.interpolate_variance(v, bt, btt).numpy(), g["interp_var"], rtol=1e-6, atol=0)


def test_vlb_value_and_gradient(golden):
    g = golden("iddpm_process")
    seed, T, B, x0s, zs, ms = (int(v) for v in g["train_meta"])
    t = torch.from_numpy(g["train_t"])
    tabs = OI.schedule_tables(T, "cosine")
    x0 = synth.uniform(x0s, (B, 3, 32, 32))
    mo = (0.5 * synth.normal(int(g["vlb_meta"][0]), (B, 6, 32, 32))).requires_grad_(True)
    x_t = synth.normal(int(g["vlb_meta"][1]), (B, 3, 32, 32))
    b, a, ab, abp = (OI._col(tabs[0], t), OI._col(tabs[1], t), OI._col(tabs[2], t), OI._col(tabs[2], t - 1))
    eps, var = OI.forward_model(mo, b, ab, abp)
    vlb = OI.loss_vlb(eps, var, x_t, t, x0, b, a, ab, abp)
    vlb.backward()
    np.testing.assert_allclose(vlb.detach().numpy(), g["vlb_value"], rtol=1e-5)
    np.testing.assert_allclose(mo.grad.numpy(), g["vlb_dout"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_training_loss_and_grads_vs_reference(golden, mode):
    g = golden("iddpm_process")
    cfg = OI.TINY
    seed, T, B, x0s, zs, ms = (int(v) for v in g["train_meta"])
    t = torch.from_numpy(g["train_t"])
    x0, z = synth.uniform(x0s, (B, 3, 32, 32)), synth.normal(zs, (B, 3, 32, 32))
    masks = OI.make_drop_masks(cfg, B, ms) if mode == "train" else None
    for sched in ("cosine", "linear"):
        sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in OI.make_state_dict(cfg, seed).items()}
        loss = OI.training_loss(lambda x, tt: OI.unet_forward(sd, cfg, x, tt, drop_masks=masks), x0, t, z, OI.schedule_tables(T, sched))
        np.testing.assert_allclose(loss.detach().numpy(), g[f"train_{sched}_{mode}_loss"], rtol=2e-5)
        if sched == "cosine":
            loss.backward()
            for k in [k for k in g.files if k.startswith(f"train_{mode}_grad::")]:
                want = g[k]
                got = sd[k.split("::")[1]].grad.numpy()
                np.testing.assert_allclose(got, want, atol=1e-5 + 1e-4 * float(np.abs(want).max()), rtol=0, err_msg=k)
    sd = OI.make_state_dict(cfg, seed)
    tabs = OI.schedule_tables(T, "cosine")
    with torch.no_grad():
        vlb = OI.training_loss(lambda x, tt: OI.unet_forward(sd, cfg, x, tt), x0, t, z, tabs, loss_type="vlb")
        assert OI.training_loss(lambda x, tt: OI.unet_forward(sd, cfg, x, tt), x0, t, z, tabs, loss_type="simple") is None
    assert int(g["train_simple_returns_none"]) == 1
    np.testing.assert_allclose(vlb.numpy(), g["train_vlb_only_loss"], rtol=2e-5)


def test_sampler_trajectories_vs_reference(golden):
    g = golden("iddpm_process")
    cfg = OI.TINY
    seed, T, B, xs, zs0 = (int(v) for v in g["traj_meta"])
    sd = OI.make_state_dict(cfg, seed)
    shape = (B, 3, 32, 32)
    for sched in ("cosine", "linear"):
        tabs = OI.schedule_tables(T, sched)
        x = synth.normal(xs, shape)
        with torch.no_grad():
            for k in range(T):
                t = T - k
                out = OI.unet_forward(sd, cfg, x, torch.tensor([t]))
                x = OI.sampling_step(out, x, t, synth.normal(zs0 + k, shape), tabs)
                if f"traj_{sched}_step{k}" in g.files:
                    want = g[f"traj_{sched}_step{k}"]
                    np.testing.assert_allclose(x.numpy(), want, atol=1e-5 * max(1.0, float(np.abs(want).max())), rtol=0, err_msg=f"{sched} step {k}")
