"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; no reference import at test time."""

import numpy as np
import pytest
import torch

from oracle import unet as O
from oracle import diffusion as D
from oracle import synth

FP32_ATOL = 1e-5  # north_star: 1e-5 fp32


def test_param_table_default_counts():
    tab = O.param_table(O.UNetConfig())
    assert len(tab) == 305
    n = sum(int(np.prod(s)) for k, s, r in tab if r != "buffer")
    assert n == 32_416_643


def test_unet_tiny_vs_reference(golden):
    g = golden("unet_tiny")
    sd = O.make_state_dict(O.TINY, int(g["tiny_seed"]))
    for c in range(int(g["tiny_ncases"])):
        B = int(g[f"tiny_case{c}_B"])
        x = synth.normal(int(g[f"tiny_case{c}_xseed"]), (B, 3, 32, 32))
        t = torch.from_numpy(g[f"tiny_case{c}_t"])
        y = O.unet_forward(sd, O.TINY, x, t)
        np.testing.assert_allclose(y.numpy(), g[f"tiny_case{c}_y"], atol=FP32_ATOL, rtol=0)


def test_unet_tiny_activations_vs_reference(golden):
    g = golden("unet_tiny")
    sd = O.make_state_dict(O.TINY, int(g["tiny_seed"]))
    x = synth.normal(int(g["tiny_acts_xseed"]), (2, 3, 32, 32))
    cap = {}
    y = O.unet_forward(sd, O.TINY, x, torch.from_numpy(g["tiny_acts_t"]), capture=cap)
    np.testing.assert_allclose(y.numpy(), g["tiny_acts_y"], atol=FP32_ATOL, rtol=0)
    keys = [k for k in g.files if k.startswith("tiny_act::")]
    assert len(keys) == len(cap)
    for k in keys:
        np.testing.assert_allclose(cap[k.split("::")[1]].numpy(), g[k], atol=FP32_ATOL, rtol=0, err_msg=k)


def test_unet_full_vs_reference(golden):
    g = golden("unet_full")
    cfg = O.UNetConfig()
    sd = O.make_state_dict(cfg, int(g["full_seed"]))
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    cap = {}
    y1 = O.unet_forward(sd, cfg, x, torch.from_numpy(g["full_t_one"]), capture=cap)
    np.testing.assert_allclose(y1.numpy(), g["full_y_one"], atol=FP32_ATOL, rtol=0)
    y2 = O.unet_forward(sd, cfg, x, torch.from_numpy(g["full_t_per"]))
    np.testing.assert_allclose(y2.numpy(), g["full_y_per"], atol=FP32_ATOL, rtol=0)
    for k in [k for k in g.files if k.startswith("full_actdigest::")]:
        ok, err = synth.digest_close(cap[k.split("::")[1]], g[k], atol=FP32_ATOL, rtol=1e-5)
        assert ok, (k, err)


def test_layers_vs_reference(golden):
    g = golden("layers")
    cfg = O.UNetConfig()
    sd = O.make_state_dict(cfg, 21)
    temb = 0.5 * synth.normal(int(g["layer_temb_seed"]), (2, 512))
    y = O.res_block(sd, cfg, O.Node("res", "down_layers.0", 128, 128, False), synth.normal(400, (2, 128, 32, 32)), temb)
    assert synth.digest_close(y, g["layer_rb128_y"], FP32_ATOL, 1e-5)[0]
    y = O.res_block(sd, cfg, O.Node("res", "up_layers.8", 512, 256, True), synth.normal(401, (2, 512, 16, 16)), temb)
    assert synth.digest_close(y, g["layer_rb512a_y"], FP32_ATOL, 1e-5)[0]
    y = O.attention_block(sd, "up_layers.10.attention", synth.normal(403, (2, 128, 16, 16)), 32)
    assert synth.digest_close(y, g["layer_attn128_y"], FP32_ATOL, 1e-5)[0]


def test_schedules_bit_exact(golden):
    g = golden("schedules")
    for T in (100, 1000):
        beta = D.linear_beta(T)
        alpha, abar = D.alpha_tables(beta)
        assert np.array_equal(beta.numpy(), g[f"sched_beta_{T}"])
        assert np.array_equal(alpha.numpy(), g[f"sched_alpha_{T}"])
        assert np.array_equal(abar.numpy(), g[f"sched_abar_{T}"])
    _, abar = D.alpha_tables(D.linear_beta(50, 2.5e-5, 0.005))
    assert np.array_equal(abar.numpy(), g["sched_abar_50_custom"])
    for T, S in ((1000, 50), (100, 5), (1000, 7)):
        for sch in ("linear", "quadratic"):
            assert np.array_equal(D.tau_table(T, S, sch).numpy(), g[f"tau_{sch}_{T}_{S}"])
    assert int(g["tau_bad_raises"]) == 1
    with pytest.raises(NotImplementedError):
        D.tau_table(100, 5, "cosine")
    assert int(g["uniform_int_max_100"]) == 99  # t = T is never drawn in training


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_training_loss_and_grads_vs_reference(golden, mode):
    g = golden("train_tiny")
    seed, T, B, sx, st, sz, sm = [int(v) for v in g["train_meta"]]
    cfg = O.TINY
    sd = {k: v.clone().requires_grad_(k != "condition.0.embeddings") for k, v in O.make_state_dict(cfg, seed).items()}
    x0 = synth.uniform(sx, (B, 3, 32, 32)).requires_grad_(True)
    t = synth.randint(st, 1, T, B)
    assert np.array_equal(t.numpy(), g["train_t"])
    z = synth.normal(sz, (B, 3, 32, 32))
    masks = O.make_drop_masks(cfg, B, sm) if mode == "train" else None
    _, abar = D.alpha_tables(D.linear_beta(T))
    loss = D.training_loss(lambda xt, tt: O.unet_forward(sd, cfg, xt, tt, drop_masks=masks), x0, t, z, abar)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"train_{mode}_loss"], rtol=1e-5)
    np.testing.assert_allclose(x0.grad.numpy(), g[f"train_{mode}_dx0"], atol=1e-6, rtol=1e-4)
    n = 0
    for k in g.files:
        if k.startswith(f"train_{mode}_grad::"):
            name = k.split("::")[1]
            np.testing.assert_allclose(sd[name].grad.numpy(), g[k], atol=2e-6, rtol=1e-4, err_msg=name)
            n += 1
    assert n == len(sd) - 1


def test_sampler_trajectories_vs_reference(golden):
    g = golden("traj_tiny")
    seed, T, B, sx, sz = [int(v) for v in g["traj_meta"]]
    cfg = O.TINY
    sd = O.make_state_dict(cfg, seed)
    shape = (B, 3, 32, 32)
    x_T = synth.normal(sx, shape)
    zs = [synth.normal(sz + k, shape) for k in range(T)]
    with torch.no_grad():
        model = lambda x, t: O.unet_forward(sd, cfg, x, t)
        traj = D.ddpm_generate(model, x_T, zs, T)
        for k in (0, 1, 9, 49, 97, 98, 99):
            np.testing.assert_allclose(traj[k].numpy(), g[f"traj_ddpm_step{k}"], atol=5e-5, rtol=0, err_msg=str(k))
        for T_, S_, sch in ((100, 5, "quadratic"), (100, 5, "linear"), (1000, 50, "quadratic")):
            tr = D.ddim_generate(model, x_T, T_, S_, sch)
            for i in (S_, S_ - 1, 2, 1):
                np.testing.assert_allclose(
                    tr[S_ - i].numpy(), g[f"traj_ddim_{sch}_{T_}_{S_}_i{i}"], atol=5e-5, rtol=0, err_msg=f"{sch} {i}"
                )
