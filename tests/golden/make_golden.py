#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (dmme v0.5.2 at /root/reference).

Runs only in the build container (the reference does not exist on the GPU box).
The reference files are imported unmodified through a namespace stub (its package
__init__ pulls torchvision / pytorch_lightning, which are not installed).
All randomness is injected from frozen numpy streams (oracle/synth.py), weights come
from oracle.make_state_dict(cfg, seed) and are loaded into the reference modules with
load_state_dict(strict=True) -- so a fixture is {seeds, reference outputs}.

Harness-side patches (reference files untouched):
  * torch.normal       -> mean + std * z_injected   (Normal.sample(), DDPM noise)
  * F.dropout2d        -> x * injected (B, C) mask  (train-mode parity)
  * dmme.uniform_int / dmme.gaussian on the stub -> injected t / x_T
  * Distribution.set_default_validate_args(False) so DDIM's Normal(mean, 0) at
    tau_{i-1} = 0 does not raise

usage:  PYTHONDONTWRITEBYTECODE=1 TQDM_DISABLE=1 python tests/golden/make_golden.py
"""

import importlib
import os
import sys
import types

os.environ.setdefault("TQDM_DISABLE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import unet as O  # noqa: E402
from oracle import iddpm as OI  # noqa: E402
from oracle import synth  # noqa: E402

REF = "/root/reference/src/dmme"


def import_reference():
    def stub(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    d = stub("dmme", REF)
    stub("dmme.common", REF + "/common")
    stub("dmme.models", REF + "/models")
    noise = importlib.import_module("dmme.common.noise")
    norm = importlib.import_module("dmme.common.norm")
    for k in ("gaussian", "gaussian_like", "uniform_int", "pad"):
        setattr(d, k, getattr(noise, k))
    for k in ("norm", "denorm"):
        setattr(d, k, getattr(norm, k))
    eq = importlib.import_module("dmme.equations")
    dm = importlib.import_module("dmme.diffusion_models")
    mm = importlib.import_module("dmme.models.ddpm")
    mi = importlib.import_module("dmme.models.iddpm")
    return d, eq, dm, mm, mi


dmme, eq, dm, mm, mi = import_reference()
torch.distributions.Distribution.set_default_validate_args(False)
torch.set_num_threads(8)


def ref_unet(cfg: O.UNetConfig, seed: int):
    net = mm.UNet(
        in_channels=cfg.in_channels,
        pos_dim=cfg.pos_dim,
        emb_dim=cfg.emb_dim,
        num_groups=cfg.num_groups,
        dropout=cfg.dropout,
        channels_per_depth=cfg.channels_per_depth,
        num_blocks=cfg.num_blocks,
        attention_depths=cfg.attention_depths,
    )
    sd = O.make_state_dict(cfg, seed)
    # key order and shapes must agree with the oracle's param_table
    ref_keys = list(net.state_dict().keys())
    assert ref_keys == [k for k, _, _ in O.param_table(cfg)], "param_table order mismatch"
    net.load_state_dict(sd, strict=True)
    return net, sd


class InjectNormal:
    """Replace torch.normal(mean, std) by mean + std * next(z) for the duration."""

    def __init__(self, zs):
        self.zs = list(zs)
        self.k = 0

    def __enter__(self):
        self.orig = torch.normal

        def fake(mean, std, *a, **kw):
            z = self.zs[self.k]
            self.k += 1
            return mean + std * z

        torch.normal = fake
        return self

    def __exit__(self, *a):
        torch.normal = self.orig


class InjectDropout:
    """Replace F.dropout2d by multiplication with injected (B, C) masks, in call order."""

    def __init__(self, masks):
        self.masks = list(masks)
        self.k = 0

    def __enter__(self):
        self.orig = F.dropout2d

        def fake(x, p=0.5, training=True, inplace=False):
            if not training:
                return x
            m = self.masks[self.k]
            self.k += 1
            return x * m[:, :, None, None]

        F.dropout2d = fake
        torch.nn.functional.dropout2d = fake
        return self

    def __exit__(self, *a):
        F.dropout2d = self.orig
        torch.nn.functional.dropout2d = self.orig


def hook_outputs(net):
    acts = {}

    def mk(name):
        def fn(mod, inp, out):
            acts[name] = out.detach().clone()

        return fn

    hs = [net.condition.register_forward_hook(mk("condition")), net.input_conv.register_forward_hook(mk("input_conv"))]
    for grp in ("down_layers", "middle_layers", "up_layers"):
        for i, m in enumerate(getattr(net, grp)):
            hs.append(m.register_forward_hook(mk(f"{grp}.{i}")))
    return acts, hs


def gen_unet_tiny(out):
    cfg, seed = O.TINY, 11
    net, _ = ref_unet(cfg, seed)
    net.eval()
    out["tiny_seed"] = np.int64(seed)
    case = 0
    for B in (1, 3):
        for tshape in ("one", "per"):
            x = synth.normal(100 + case, (B, 3, 32, 32))
            t = synth.randint(200 + case, 1, 100, 1 if tshape == "one" else B)
            with torch.no_grad():
                y = net(x, t)
            out[f"tiny_case{case}_B"] = np.int64(B)
            out[f"tiny_case{case}_xseed"] = np.int64(100 + case)
            out[f"tiny_case{case}_t"] = t.numpy()
            out[f"tiny_case{case}_y"] = y.numpy()
            case += 1
    out["tiny_ncases"] = np.int64(case)
    # per-module activations for B=2, per-sample t
    x = synth.normal(150, (2, 3, 32, 32))
    t = torch.tensor([7, 93])
    acts, hs = hook_outputs(net)
    with torch.no_grad():
        y = net(x, t)
    for h in hs:
        h.remove()
    out["tiny_acts_xseed"] = np.int64(150)
    out["tiny_acts_t"] = t.numpy()
    out["tiny_acts_y"] = y.numpy()
    for k, v in acts.items():
        out[f"tiny_act::{k}"] = v.numpy()


def gen_unet_full(out):
    cfg, seed = O.UNetConfig(), 21
    net, _ = ref_unet(cfg, seed)
    net.eval()
    out["full_seed"] = np.int64(seed)
    x = synth.normal(300, (2, 3, 32, 32))
    out["full_xseed"] = np.int64(300)
    acts, hs = hook_outputs(net)
    with torch.no_grad():
        y1 = net(x, torch.tensor([500]))
    for h in hs:
        h.remove()
    with torch.no_grad():
        y2 = net(x, torch.tensor([3, 999]))
    out["full_t_one"] = np.array([500], dtype=np.int64)
    out["full_y_one"] = y1.numpy()
    out["full_t_per"] = np.array([3, 999], dtype=np.int64)
    out["full_y_per"] = y2.numpy()
    for k, v in acts.items():
        out[f"full_actdigest::{k}"] = synth.digest(v)


def gen_layers(out):
    """Stand-alone reference modules at real channel counts (digests)."""
    cfg = O.UNetConfig()
    full_sd = O.make_state_dict(cfg, 21)

    def sub(prefix):
        return {k[len(prefix) + 1 :]: v for k, v in full_sd.items() if k.startswith(prefix + ".")}

    temb = 0.5 * synth.normal(402, (2, 512))
    out["layer_temb_seed"] = np.int64(402)
    # ResBlock 128->128 @32x32 (down_layers.0), no attention
    rb = mm.ResBlock(128, 128, False)
    rb.load_state_dict(sub("down_layers.0"))
    rb.eval()
    x = synth.normal(400, (2, 128, 32, 32))
    with torch.no_grad():
        out["layer_rb128_y"] = synth.digest(rb(x, temb))
    # ResBlock 512->256 + attention @16x16 (up_layers.8): 1x1 residual, S=256 C=256
    rb = mm.ResBlock(512, 256, True)
    rb.load_state_dict(sub("up_layers.8"))
    rb.eval()
    x = synth.normal(401, (2, 512, 16, 16))
    with torch.no_grad():
        out["layer_rb512a_y"] = synth.digest(rb(x, temb))
    # Attention alone C=128 S=256 (up_layers.10.attention)
    at = mm.Attention(128, 32)
    at.load_state_dict(sub("up_layers.10.attention"))
    x = synth.normal(403, (2, 128, 16, 16))
    with torch.no_grad():
        out["layer_attn128_y"] = synth.digest(at(x))
    # UpSample 256 @8->16 (up_layers.7), DownSample 128 @32->16 (down_layers.2)
    up = mm.UpSample(256, 256)
    up.load_state_dict(sub("up_layers.7"))
    x = synth.normal(404, (2, 256, 8, 8))
    with torch.no_grad():
        out["layer_up256_y"] = synth.digest(up(x))
    dn = mm.DownSample(128, 128)
    dn.load_state_dict(sub("down_layers.2"))
    x = synth.normal(405, (2, 128, 32, 32))
    with torch.no_grad():
        out["layer_down128_y"] = synth.digest(dn(x))


def gen_schedules(out):
    dummy = torch.nn.Identity()
    for T in (100, 1000):
        d = dm.DDPM(dummy, timesteps=T)
        out[f"sched_beta_{T}"] = d.beta.reshape(-1).numpy()
        out[f"sched_alpha_{T}"] = d.alpha.reshape(-1).numpy()
        out[f"sched_abar_{T}"] = d.alpha_bar.reshape(-1).numpy()
    d = dm.DDPM(dummy, timesteps=50, start=2.5e-5, end=0.005)
    out["sched_abar_50_custom"] = d.alpha_bar.reshape(-1).numpy()
    for T, S in ((1000, 50), (100, 5), (1000, 7)):
        for sch in ("linear", "quadratic"):
            out[f"tau_{sch}_{T}_{S}"] = dm.DDIM(dummy, T, S, sch).tau.numpy()
    try:
        dm.DDIM(dummy, 100, 5, "cosine")
        out["tau_bad_raises"] = np.int64(0)
    except NotImplementedError:
        out["tau_bad_raises"] = np.int64(1)
    # uniform_int(1, T) never returns T (randint high-exclusive)
    out["uniform_int_max_100"] = np.int64(int(dmme.uniform_int(1, 100, 100000).max()))


def gen_train(out):
    """DDPM.training_step on the tiny UNet: loss + every gradient, eval and train mode."""
    cfg, seed, T, B = O.TINY, 11, 100, 3
    x0 = synth.uniform(500, (B, 3, 32, 32))
    t = synth.randint(501, 1, T, B)
    z = synth.normal(502, (B, 3, 32, 32))
    out["train_meta"] = np.array([seed, T, B, 500, 501, 502, 503], dtype=np.int64)
    out["train_t"] = t.numpy()
    for mode in ("eval", "train"):
        net, _ = ref_unet(cfg, seed)
        net.train(mode == "train")
        ddpm = dm.DDPM(net, timesteps=T)
        x0g = x0.clone().requires_grad_(True)
        masks = O.make_drop_masks(cfg, B, 503)
        order = O.res_block_names(cfg)
        orig_ui = dmme.uniform_int
        dmme.uniform_int = lambda lo, hi, count=1, device=None: t
        try:
            with InjectNormal([z]), InjectDropout([masks[k] for k in order]) as dr:
                loss = ddpm.training_step(x0g)
                if mode == "train":
                    assert dr.k == len(order)
        finally:
            dmme.uniform_int = orig_ui
        loss.backward()
        out[f"train_{mode}_loss"] = loss.detach().numpy()
        out[f"train_{mode}_dx0"] = x0g.grad.numpy()
        for k, p in net.named_parameters():
            out[f"train_{mode}_grad::{k}"] = p.grad.numpy()


def gen_traj(out):
    cfg, seed = O.TINY, 11
    net, _ = ref_unet(cfg, seed)
    net.eval()
    B, T = 2, 100
    shape = (B, 3, 32, 32)
    x_T = synth.normal(600, shape)
    zs = [synth.normal(1000 + k, shape) for k in range(T)]
    out["traj_meta"] = np.array([seed, T, B, 600, 1000], dtype=np.int64)
    ddpm = dm.DDPM(net, timesteps=T)
    all_t = torch.arange(0, T + 1).unsqueeze(1)
    keep = {0, 1, 9, 49, 97, 98, 99}
    x = x_T
    with torch.no_grad(), InjectNormal(zs):
        for k in range(T):
            x = ddpm.sampling_step(x, all_t[T - k])
            if k in keep:
                out[f"traj_ddpm_step{k}"] = x.numpy()
    # the reference's own generate() must land on the same final image
    orig_g = dmme.gaussian
    dmme.gaussian = lambda shape, dtype=None, device=None: x_T
    try:
        with torch.no_grad(), InjectNormal(zs):
            xg = ddpm.generate(shape)
    finally:
        dmme.gaussian = orig_g
    assert torch.equal(xg, x), "generate() and the sampling_step loop disagree"
    # DDIM: (T=100, S=5) and (T=1000, S=50), quadratic; (100, 5) linear
    for T_, S_, sch in ((100, 5, "quadratic"), (100, 5, "linear"), (1000, 50, "quadratic")):
        ddim = dm.DDIM(net, T_, S_, sch)
        all_i = torch.arange(0, S_ + 1).unsqueeze(1)
        x = x_T
        with torch.no_grad():
            for i in range(S_, 0, -1):
                x = ddim.sampling_step(x, all_i[i])
                if i in (S_, S_ - 1, 2, 1):
                    out[f"traj_ddim_{sch}_{T_}_{S_}_i{i}"] = x.numpy()


# ---------------------------------------------------------------------------- Improved DDPM


def ref_iunet(cfg: OI.IUNetConfig, seed: int):
    net = mi.UNet(
        in_channels=cfg.in_channels,
        pos_dim=cfg.pos_dim,
        emb_dim=cfg.emb_dim,
        num_groups=cfg.num_groups,
        dropout=cfg.dropout,
        channels_per_depth=cfg.channels_per_depth,
        num_blocks=cfg.num_blocks,
        attention_depths=cfg.attention_depths,
    )
    sd = OI.make_state_dict(cfg, seed)
    assert list(net.state_dict().keys()) == [k for k, _, _ in OI.param_table(cfg)], "iddpm param_table order mismatch"
    net.load_state_dict(sd, strict=True)
    return net, sd


def gen_iddpm_unet(out):
    for tag, cfg, seed in (("tiny", OI.TINY, 31), ("attn", OI.TINY_ATTN, 32)):
        net, _ = ref_iunet(cfg, seed)
        net.eval()
        out[f"{tag}_seed"] = np.int64(seed)
        case = 0
        for B in (1, 3):
            for tshape in ("one", "per"):
                x = synth.normal(700 + case, (B, 3, 32, 32))
                t = synth.randint(720 + case, 1, 100, 1 if tshape == "one" else B)
                with torch.no_grad():
                    y = net(x, t)
                out[f"{tag}_case{case}_B"] = np.int64(B)
                out[f"{tag}_case{case}_xseed"] = np.int64(700 + case)
                out[f"{tag}_case{case}_t"] = t.numpy()
                out[f"{tag}_case{case}_y"] = y.numpy()
                case += 1
        out[f"{tag}_ncases"] = np.int64(case)
        # per-module activations at B = 2 (the head merge mixes the two samples)
        x = synth.normal(750, (2, 3, 32, 32))
        t = torch.tensor([7, 93])
        acts, hs = hook_outputs(net)
        with torch.no_grad():
            y = net(x, t)
        for h in hs:
            h.remove()
        out[f"{tag}_acts_xseed"] = np.int64(750)
        out[f"{tag}_acts_t"] = t.numpy()
        out[f"{tag}_acts_y"] = y.numpy()
        for k, v in acts.items():
            out[f"{tag}_act::{k}"] = v.numpy()
        # train mode, injected Dropout2d masks
        B = 3
        masks = OI.make_drop_masks(cfg, B, 760)
        x = synth.normal(761, (B, 3, 32, 32))
        t = torch.tensor([3, 50, 99])
        net.train(True)
        with torch.no_grad(), InjectDropout([masks[k] for k in OI.res_block_names(cfg)]) as dr:
            y = net(x, t)
            assert dr.k == len(OI.res_block_names(cfg))
        out[f"{tag}_train_meta"] = np.array([B, 760, 761], dtype=np.int64)
        out[f"{tag}_train_t"] = t.numpy()
        out[f"{tag}_train_y"] = y.numpy()
    # default-size network (36.2 M parameters): digests only
    cfg, seed = OI.IUNetConfig(), 41
    net, _ = ref_iunet(cfg, seed)
    net.eval()
    out["full_seed"] = np.int64(seed)
    out["full_nparams"] = np.int64(sum(p.numel() for p in net.parameters()))
    x = synth.normal(770, (2, 3, 32, 32))
    out["full_xseed"] = np.int64(770)
    acts, hs = hook_outputs(net)
    with torch.no_grad():
        y1 = net(x, torch.tensor([500]))
    for h in hs:
        h.remove()
    out["full_t_one"] = np.array([500], dtype=np.int64)
    out["full_y_one"] = y1.numpy()
    for k, v in acts.items():
        out[f"full_actdigest::{k}"] = synth.digest(v)
    # one multi-head attention block alone at real width: C = 256, S = 64, B = 3
    at = mi.MultiHeadAttention(256, 32, 4)
    full_sd = OI.make_state_dict(cfg, seed)
    pre = "down_layers.3.attention."
    at.load_state_dict({k[len(pre):]: v for k, v in full_sd.items() if k.startswith(pre)})
    x = synth.normal(771, (3, 256, 8, 8))
    with torch.no_grad():
        out["layer_mha256_y"] = at(x).numpy()


def gen_iddpm_process(out):
    dummy = torch.nn.Identity()
    for T in (100, 1000, 4000):
        d = dm.IDDPM(dummy, timesteps=T)
        out[f"cos_beta_{T}"] = d.beta.reshape(-1).numpy()
        out[f"cos_alpha_{T}"] = d.alpha.reshape(-1).numpy()
        out[f"cos_abar_{T}"] = d.alpha_bar.reshape(-1).numpy()
    d = dm.IDDPM(dummy, timesteps=4000, schedule="linear", start=2.5e-5, end=0.005)  # configs/iddpm/cifar10.yaml:78-81
    out["lin_beta_4000"] = d.beta.reshape(-1).numpy()
    out["lin_abar_4000"] = d.alpha_bar.reshape(-1).numpy()
    try:
        dm.IDDPM(dummy, schedule="sqrt")
        out["bad_schedule_raises"] = np.int64(0)
    except NotImplementedError:
        out["bad_schedule_raises"] = np.int64(1)
    # equations on synthetic tensors
    v = synth.uniform(800, (4, 3, 8, 8), -0.5, 1.5)
    bt = torch.tensor([0.02, 1e-4, 0.3, 0.999]).reshape(4, 1, 1, 1)
    btt = torch.tensor([0.01, 0.0, 0.2, 0.5]).reshape(4, 1, 1, 1)
    out["interp_var"] = eq.iddpm.interpolate_variance(v, bt, btt).numpy()

    # training_step on the tiny network: hybrid loss + every gradient (a t == 1 row included)
    cfg, seed, T, B = OI.TINY, 31, 100, 4
    x0 = synth.uniform(810, (B, 3, 32, 32))
    z = synth.normal(812, (B, 3, 32, 32))
    t = torch.tensor([1, 57, 99, 2])
    out["train_meta"] = np.array([seed, T, B, 810, 812, 813], dtype=np.int64)
    out["train_t"] = t.numpy()
    for sched in ("cosine", "linear"):
        for mode in ("eval", "train"):
            net, _ = ref_iunet(cfg, seed)
            net.train(mode == "train")
            idd = dm.IDDPM(net, timesteps=T, schedule=sched)
            masks = OI.make_drop_masks(cfg, B, 813)
            order = OI.res_block_names(cfg)
            orig_ui = dmme.uniform_int
            dmme.uniform_int = lambda lo, hi, count=1, device=None: t
            try:
                with InjectNormal([z]), InjectDropout([masks[k] for k in order]):
                    loss = idd.training_step(x0)
            finally:
                dmme.uniform_int = orig_ui
            loss.backward()
            out[f"train_{sched}_{mode}_loss"] = loss.detach().numpy()
            if sched == "cosine":
                for k, p in net.named_parameters():
                    out[f"train_{mode}_grad::{k}"] = p.grad.numpy()
    # the two loss terms by themselves, and the gradient w.r.t. the raw network output, on synthetic tensors
    idd = dm.IDDPM(dummy, timesteps=T)
    mo = (0.5 * synth.normal(820, (B, 6, 32, 32))).requires_grad_(True)
    x_t = synth.normal(821, (B, 3, 32, 32))
    bt, at_, abt, abp = idd.beta[t], idd.alpha[t], idd.alpha_bar[t], idd.alpha_bar[t - 1]
    eps_, v_ = mo.chunk(2, dim=1)
    var = eq.iddpm.interpolate_variance(v_, bt, (1 - abp) / (1 - abt) * bt)
    vlb = eq.iddpm.loss_vlb(eps_, var, x_t, t, x0, bt, at_, abt, abp)
    vlb.backward()
    out["vlb_meta"] = np.array([820, 821], dtype=np.int64)
    out["vlb_value"] = vlb.detach().numpy()
    out["vlb_dout"] = mo.grad.numpy()
    net, _ = ref_iunet(cfg, seed)
    net.eval()
    idd_v = dm.IDDPM(net, timesteps=T, loss_type="vlb")
    idd_s = dm.IDDPM(net, timesteps=T, loss_type="simple")
    orig_ui = dmme.uniform_int
    dmme.uniform_int = lambda lo, hi, count=1, device=None: t
    try:
        with InjectNormal([z]), torch.no_grad():
            out["train_vlb_only_loss"] = idd_v.training_step(x0).numpy()
        with InjectNormal([z]), torch.no_grad():
            out["train_simple_returns_none"] = np.int64(idd_s.training_step(x0) is None)
    finally:
        dmme.uniform_int = orig_ui

    # sampling: 100-step cosine chain on the tiny network, injected noise
    Bs = 2
    shape = (Bs, 3, 32, 32)
    x_T = synth.normal(830, shape)
    zs = [synth.normal(2000 + k, shape) for k in range(T)]
    out["traj_meta"] = np.array([seed, T, Bs, 830, 2000], dtype=np.int64)
    for sched in ("cosine", "linear"):
        idd = dm.IDDPM(net, timesteps=T, schedule=sched)
        all_t = torch.arange(0, T + 1).unsqueeze(1)
        x = x_T
        with torch.no_grad(), InjectNormal(zs):
            for k in range(T):
                x = idd.sampling_step(x, all_t[T - k])
                if k in (0, 1, 9, 49, 97, 98, 99):
                    out[f"traj_{sched}_step{k}"] = x.numpy()


def gen_data(out):
    """the reference's norm / denorm (common/norm.py) on every byte value as torchvision's ToTensor would hand it over"""
    x = torch.arange(256, dtype=torch.float32).div(255)
    out["norm_u8_table"] = dmme.norm(x).numpy()
    out["denorm_of_norm"] = dmme.denorm(dmme.norm(x)).numpy()
    out["denorm_clip"] = dmme.denorm(torch.tensor([-3.0, -1.0, 0.0, 0.25, 1.0, 7.0])).numpy()


def main():
    torch.manual_seed(0)
    groups = {
        "unet_tiny": gen_unet_tiny,
        "unet_full": gen_unet_full,
        "layers": gen_layers,
        "schedules": gen_schedules,
        "train_tiny": gen_train,
        "traj_tiny": gen_traj,
        "iddpm_unet": gen_iddpm_unet,
        "iddpm_process": gen_iddpm_process,
        "data": gen_data,
    }
    only = sys.argv[1:]
    for name, fn in groups.items():
        if only and name not in only:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
