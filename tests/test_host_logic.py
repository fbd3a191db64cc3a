"""CPU: host-side mirror of the reference interface (no GPU compute): state_dict contract,
schedules / tau tables against the reference's golden values, error behaviour, YAML runner,
LR warm-up, helpers."""

import os

import numpy as np
import pytest
import torch

import dmme_amd
from oracle import unet as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", [O.UNetConfig(), O.TINY, O.UNetConfig(dropout=0.0), O.UNetConfig(channels_per_depth=(32, 64), num_blocks=1, attention_depths=(1, 2))])
def test_state_dict_contract_matches_reference_table(cfg):
    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks, cfg.attention_depths)
    table = O.param_table(cfg)  # pinned against the reference by make_golden.py (load_state_dict strict + key order)
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in table]
    assert [tuple(v.shape) for v in sd.values()] == [s for _, s, _ in table]
    assert sum(p.numel() for p in net.parameters()) == sum(int(np.prod(s)) for _, s, r in table if r != "buffer")
    assert [n for n, _ in net.named_buffers()] == ["condition.0.embeddings"]


def test_load_state_dict_roundtrip_keeps_flat_views():
    cfg = O.TINY
    net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks)
    sd = O.make_state_dict(cfg, 3)
    net.load_state_dict(sd, strict=True)
    flat = net.flat_parameters()
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k])
        assert flat.data_ptr() <= v.data_ptr() < flat.data_ptr() + 4 * flat.numel()
    net.double().float()  # _apply moves tensors one by one; the module re-flattens afterwards
    flat2 = net.flat_parameters()
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]) and flat2.data_ptr() <= v.data_ptr() < flat2.data_ptr() + 4 * flat2.numel()
    with pytest.raises(RuntimeError):
        net.load_state_dict({k: v for k, v in list(sd.items())[:-1]}, strict=True)


def test_default_init_statistics():
    torch.manual_seed(0)
    net = dmme_amd.UNet()
    sd = net.state_dict()
    w = sd["down_layers.0.conv1.2.weight"]
    bound = 1 / np.sqrt(128 * 9)
    assert w.abs().max() <= bound and w.abs().max() > 0.9 * bound and abs(w.mean()) < 1e-3
    assert torch.all(sd["down_layers.0.conv1.0.weight"] == 1) and torch.all(sd["down_layers.0.conv1.0.bias"] == 0)
    assert torch.all(sd["up_layers.8.attention.norm.weight"] == 1)
    f = sd["condition.0.embeddings"]
    assert f.shape == (1, 64) and f[0, 0] == 1 and abs(f[0, -1].item() - 1e-4) < 1e-9


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = dmme_amd.UNet(pos_dim=4, emb_dim=8, num_groups=2, channels_per_depth=(4, 8), num_blocks=1)
    with pytest.raises(dmme_amd._lib.DmmeError):
        net(torch.zeros(1, 3, 8, 8), torch.tensor([1]))


def test_ddpm_buffers_match_reference_golden(golden):
    g = golden("schedules")
    for T in (100, 1000):
        d = dmme_amd.DDPM(torch.nn.Identity(), T)
        assert d.beta.shape == (T + 1, 1, 1, 1)
        assert np.array_equal(d.beta.reshape(-1).numpy(), g[f"sched_beta_{T}"])
        assert np.array_equal(d.alpha.reshape(-1).numpy(), g[f"sched_alpha_{T}"])
        assert np.array_equal(d.alpha_bar.reshape(-1).numpy(), g[f"sched_abar_{T}"])
        assert d.state_dict() == {}  # schedules are non-persistent, like the reference
    d = dmme_amd.DDPM(torch.nn.Identity(), 50, start=2.5e-5, end=0.005)
    assert np.array_equal(d.alpha_bar.reshape(-1).numpy(), g["sched_abar_50_custom"])
    # per-step scalars of the reverse update, same fp32 ops as reverse_process
    b, a, ab = (torch.from_numpy(g[f"sched_{k}_1000"]) for k in ("beta", "alpha", "abar"))
    d = dmme_amd.DDPM(torch.nn.Identity(), 1000)
    # index 0 (t = 0) is never used: beta_0 / sqrt(1 - abar_0) is 0/0
    assert d._c1 == (1 / torch.sqrt(a)).tolist() and d._c2[1:] == (b / torch.sqrt(1 - ab)).tolist()[1:] and d._sigma == torch.sqrt(b).tolist()


def test_ddim_tau_matches_reference_golden(golden):
    g = golden("schedules")
    for T, S in ((1000, 50), (100, 5), (1000, 7)):
        for sch in ("linear", "quadratic"):
            d = dmme_amd.DDIM(torch.nn.Identity(), T, S, sch.upper() if S == 7 else sch)
            assert np.array_equal(d.tau.numpy(), g[f"tau_{sch}_{T}_{S}"])
    with pytest.raises(NotImplementedError):
        dmme_amd.DDIM(torch.nn.Identity(), 100, 5, "cosine")
    assert dmme_amd.DDIM(torch.nn.Identity()).tau[:4].tolist() == [0, 0, 2, 4]  # tau_0 = tau_1 = 0


def test_helpers_match_reference_contracts():
    torch.manual_seed(0)
    t = dmme_amd.uniform_int(1, 100, 100000)
    assert t.min() == 1 and t.max() == 99  # randint is high-exclusive: t = T never drawn
    assert torch.equal(dmme_amd.pad(torch.tensor([1.0, 2.0])), torch.tensor([0.0, 1.0, 2.0]))
    assert torch.equal(dmme_amd.pad(torch.tensor([1.0]), value=1), torch.tensor([1.0, 1.0]))
    x = torch.rand(4, 3)
    assert torch.equal(dmme_amd.norm(x), (x - 0.5) * 2) and torch.equal(dmme_amd.denorm(dmme_amd.norm(x) * 3), torch.clip((dmme_amd.norm(x) * 3 + 1) / 2, 0, 1))
    assert dmme_amd.gaussian((2, 3)).shape == (2, 3) and dmme_amd.gaussian_like(x).shape == x.shape


def test_warmup_lr_schedule():
    p = torch.nn.Parameter(torch.zeros(2))
    opt = torch.optim.Adam([p], lr=2e-4)
    sched = dmme_amd.lr_scheduler.WarmupLR(opt, 5)
    lrs = []
    for _ in range(7):
        lrs.append(opt.param_groups[0]["lr"])
        p.grad = torch.ones(2)
        opt.step()
        sched.step()
    # lr = base * min(1, (step_count + 1) / warmup), evaluated when the scheduler steps (reference warmup.py:10-19)
    np.testing.assert_allclose(lrs, [2e-4 * v for v in (0.2, 0.4, 0.6, 0.8, 1, 1, 1)], rtol=1e-12)


def test_lit_modules_and_yaml_runner():
    from dmme_amd import trainer

    for name, cls, dm in (("ddpm", dmme_amd.LitDDPM, dmme_amd.DDPM), ("ddim", dmme_amd.LitDDIM, dmme_amd.DDIM)):
        conf = trainer.parse_config(os.path.join(ROOT, "configs", name, "cifar10.yaml"))
        assert conf["batch_size"] == 128 and conf["max_steps"] == 800000 and conf["gradient_clip_val"] == 1.0 and conf["precision"] == "fp16"  # `precision: 16`: IEEE half + loss scaling, as the reference
        module = trainer.build_module(conf)
        assert isinstance(module, cls) and type(module.diffusion_model) is dm
        assert module.lr == 2e-4 and module.warmup == 5000 and module.decay == 0.9999
        assert module.diffusion_model.model.precision == "fp16"
        keys = list(module.state_dict().keys())
        assert keys[0] == "diffusion_model.model.condition.0.embeddings" and len(keys) == 305
    opts, scheds = module.configure_optimizers()
    assert opts[0].defaults["lr"] == 2e-4 and scheds[0]["interval"] == "step"
    # configs/iddpm/cifar10.yaml: LitIDDPM, linear schedule over 4000 steps (reference configs/iddpm/cifar10.yaml:72-81)
    conf = trainer.parse_config(os.path.join(ROOT, "configs", "iddpm", "cifar10.yaml"))
    module = trainer.build_module(conf)
    assert isinstance(module, dmme_amd.LitIDDPM) and type(module.diffusion_model) is dmme_amd.IDDPM
    idd = module.diffusion_model
    assert module.lr == 1e-4 and idd.timesteps == 4000 and idd.loss_type == "hybrid" and idd.gamma == 0.001
    assert abs(float(idd.beta[1]) - 2.5e-5) < 1e-12 and abs(float(idd.beta[4000]) - 0.005) < 1e-9
    assert idd.model.out_channels == 6 and len(module.state_dict()) == 335
    m2 = dmme_amd.LitDDIM(sample_steps=10, tau_schedule="linear", timesteps=100)
    assert m2.diffusion_model.sub_timesteps == 10 and m2.diffusion_model.tau[-1] == 100


# ------------------------------------------------------------------------------------------ input pipeline (host side)


def test_norm_table_vs_reference_golden(golden):
    import dmme_amd
    from oracle import data as OD

    g = golden("data")
    x = torch.arange(256, dtype=torch.float32).div(255)
    assert np.array_equal(OD.norm(x).numpy(), g["norm_u8_table"])
    assert np.array_equal(dmme_amd.norm(x).numpy(), g["norm_u8_table"])
    assert np.array_equal(dmme_amd.denorm(dmme_amd.norm(x)).numpy(), g["denorm_of_norm"])
    assert np.array_equal(dmme_amd.denorm(torch.tensor([-3.0, -1.0, 0.0, 0.25, 1.0, 7.0])).numpy(), g["denorm_clip"])


def test_cifar10_pickle_reader_and_module_arguments(tmp_path):
    import pickle

    from dmme_amd.data_modules import CIFAR10, RandomHorizontalFlip, read_cifar10_batches

    d = tmp_path / "cifar-10-batches-py"
    d.mkdir()
    rs = np.random.RandomState(0)
    want_x, want_y = [], []
    for i in range(1, 6):
        data = rs.randint(0, 256, size=(7, 3072)).astype(np.uint8)
        labels = [int(v) for v in rs.randint(0, 10, size=7)]
        with open(d / f"data_batch_{i}", "wb") as f:
            pickle.dump({"data": data, "labels": labels, "batch_label": f"b{i}"}, f)
        want_x.append(data.reshape(7, 3, 32, 32))
        want_y += labels
    x, y = read_cifar10_batches(str(tmp_path))
    assert x.shape == (35, 3, 32, 32) and x.dtype == np.uint8 and np.array_equal(x, np.concatenate(want_x))
    assert np.array_equal(y, np.array(want_y))
    dm = CIFAR10(data_dir=str(tmp_path), batch_size=16, device="cpu")  # reference signature: data_dir, batch_size, augs
    dm.prepare_data()
    assert isinstance(dm.augs[0], RandomHorizontalFlip) and dm.augs[0].p == 0.5
    with pytest.raises(FileNotFoundError):
        CIFAR10(data_dir=str(tmp_path / "nowhere")).prepare_data()
    with pytest.raises(NotImplementedError):
        CIFAR10(augs=[object()])
    dm.setup("fit")
    loader = dm.train_dataloader()
    assert len(loader) == 3  # 35 images in batches of 16: the partial batch is kept, like the reference's DataLoader
    # DistributedSampler-style sharding: rank-strided slices of the padded permutation partition the epoch
    from dmme_amd.data_modules import GpuBatchLoader

    xs = torch.from_numpy(x)
    parts = [GpuBatchLoader(xs, None, 4, True, 0.5, rank=r, world=4, seed=9)._indices() for r in range(4)]
    assert all(p.numel() == 9 for p in parts)
    assert set(torch.cat(parts).tolist()) == set(range(35))


def test_trainer_maps_the_yaml_data_section():
    from dmme_amd import trainer
    from dmme_amd.data_modules import CIFAR10, RandomHorizontalFlip

    spec = {"class_path": "dmme.CIFAR10", "init_args": {"data_dir": ".", "batch_size": 128, "augs": [{"class_path": "torchvision.transforms.RandomHorizontalFlip"}]}}
    dm = trainer._instantiate(spec)
    assert isinstance(dm, CIFAR10) and dm.batch_size == 128 and isinstance(dm.augs[0], RandomHorizontalFlip)


# ------------------------------------------------------------------------------------------ the reference's YAML forms (jsonargparse typing)

_REFERENCE_FORM_YAML = """
seed_everything: true
trainer:
  callbacks:
    - class_path: pytorch_lightning.callbacks.ModelCheckpoint
      init_args:
        every_n_train_steps: 100_000
  gradient_clip_val: 1.0
  devices: 1
  max_steps: 800_000
  log_every_n_steps: 50
  precision: 16
  strategy: null
ckpt_path: null
model:
  class_path: dmme.LitDDPM
  init_args:
    lr: 2e-5
    warmup: 5000
    decay: 0.9999
    model:
      class_path: dmme.models.ddpm.UNet
      init_args:
        dropout: 0.0
        channels_per_depth:
          - 32
          - 64
        attention_depths:
          - 2
data:
  class_path: dmme.CIFAR10
  init_args:
    data_dir: "."
    batch_size: 128
    augs:
      - class_path: torchvision.transforms.RandomHorizontalFlip
"""


def test_yaml_in_the_reference_literal_forms_drives_module_and_optimizer(tmp_path):
    """`lr: 2e-5` is a str and `800_000` an int to YAML 1.1; jsonargparse converts by the constructor's annotations
    (reference configs/ddpm/cifar10.yaml:41,72-77, lsun_church.yaml:75-90) and so must the bundled runner."""
    from dmme_amd import trainer

    import yaml

    assert isinstance(yaml.safe_load("lr: 2e-4")["lr"], str)  # the trap itself
    path = tmp_path / "ref_form.yaml"
    path.write_text(_REFERENCE_FORM_YAML)
    conf = trainer.parse_config(str(path))
    assert conf["max_steps"] == 800000 and conf["gradient_clip_val"] == 1.0 and conf["precision"] == "fp16" and conf["ckpt_path"] is None
    module = trainer.build_module(conf)
    assert isinstance(module.lr, float) and module.lr == 2e-5 and isinstance(module.warmup, int)
    unet = module.diffusion_model.model
    assert type(unet) is dmme_amd.UNet and unet.dropout == 0.0 and tuple(unet._cfg.channels_per_depth[:2]) == (32, 64)
    assert any(k.endswith("conv2.2.weight") for k in module.state_dict())  # dropout 0: conv index 2 (models/ddpm.py:25-35)
    opts, scheds = module.configure_optimizers()
    assert opts[0].param_groups[0]["lr"] == pytest.approx(2e-5 / 5000)  # WarmupLR already applied its first factor
    # scalar rules
    assert trainer._coerce("1e-4", float) == 1e-4 and trainer._coerce("800_000", int) == 800000 and trainer._coerce(3, float) == 3.0
    assert trainer._coerce(None, typing_optional_float()) is None and trainer._coerce("2.5e-5", typing_optional_float()) == 2.5e-5
    with pytest.raises((TypeError, ValueError)):
        trainer._coerce("fast", float)
    with pytest.raises(TypeError):
        trainer._instantiate({"class_path": "dmme.LitDDPM", "init_args": {"lr": "quick"}})


def typing_optional_float():
    import typing

    return typing.Optional[float]


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference checkout absent (GPU box)")
@pytest.mark.parametrize("rel,cls,numel", [("ddpm/cifar10", "LitDDPM", 32416643), ("ddim/cifar10", "LitDDIM", 32416643),
                                           ("iddpm/cifar10", "LitIDDPM", 36168070), ("ddpm/lsun_church", "LitDDPM", 97689219)])
def test_reference_yaml_files_parse_and_build_unchanged(rel, cls, numel):
    """build container only: the reference's own files, read in place, build module + optimizer + scheduler"""
    from dmme_amd import trainer

    conf = trainer.parse_config(f"/root/reference/configs/{rel}.yaml")
    module = trainer.build_module(conf)
    assert type(module).__name__ == cls and isinstance(module.lr, float)
    assert sum(p.numel() for p in module.parameters()) == numel
    opts, scheds = module.configure_optimizers()
    assert isinstance(opts[0].param_groups[0]["lr"], float) and scheds[0]["interval"] == "step"
    assert conf["precision"] == "fp16" and conf["gradient_clip_val"] == 1.0


# ------------------------------------------------------------------------------------------ stale packed weights


def test_weights_key_follows_writes_through_rebound_parameters():
    """.cuda()/.to() rebind every Parameter with `p.data = view`; such a Parameter keeps its own version counter, so
    flat._version alone misses load_state_dict / torch.optim / p.copy_ writes.  Reproduced on CPU by a device-less _apply."""
    net = dmme_amd.UNet(pos_dim=4, emb_dim=8, num_groups=2, channels_per_depth=(4, 8), num_blocks=1)
    net._apply(lambda t: t.clone())  # what .cuda() does: new tensors, then _ensure_flat re-flattens and rebinds .data
    flat = net.flat_parameters()
    p = net.input_conv.weight
    off = (p.data_ptr() - flat.data_ptr()) // 4
    assert 0 < off < flat.numel()  # the Parameter is a view of the flat buffer again
    k0 = net._weights_key(net._ensure_flat())
    assert net._weights_key(net._ensure_flat()) == k0  # stable without writes
    with torch.no_grad():
        p.add_(1.0)
    k1 = net._weights_key(net._ensure_flat())
    assert k1 != k0 and float(flat[off]) == float(p.detach().reshape(-1)[0])  # the write landed in the flat buffer and moved the key
    sd = {k: v.clone() + 0.5 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    k2 = net._weights_key(net._ensure_flat())
    assert k2 != k1
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for q in net.parameters():
        q.grad = torch.ones_like(q)
    opt.step()
    assert net._weights_key(net._ensure_flat()) != k2
    net.mark_params_updated()
    assert net._weights_key(net._ensure_flat())[-1] == 1


# ------------------------------------------------------------------------------------------ device-resident loop tables (dmme_chain_*)


def test_chain_tables_hold_the_reference_update_scalars(golden):
    """the per-index scalars a replayed step reads from device memory are the ones the eager loop passes as kernel arguments:
    DDPM 1/sqrt(alpha_t), beta_t/sqrt(1-abar_t), sqrt(beta_t) (equations/ddpm/ddpm.py:65-71); DDIM sqrt(1-abar_tau_i),
    sqrt(abar_tau_{i-1}) with t_table = tau (equations/ddim/ddim.py:52-57); IDDPM the log-variance pair (equations/iddpm/losses.py:34-37)"""
    from oracle import diffusion as OD

    T = 1000
    d = dmme_amd.DDPM(torch.nn.Identity(), T)
    n, rows, ttab = d._chain_tables()
    beta = OD.linear_beta(T)
    alpha, abar = OD.alpha_tables(beta)
    assert n == T and ttab == list(range(T + 1)) and len(rows) == T + 1
    for t in (1, 2, 500, 1000):
        want = (float(1 / torch.sqrt(alpha[t])), float(beta[t] / torch.sqrt(1 - abar[t])), float(torch.sqrt(beta[t])))
        assert rows[t][:3] == want, (t, rows[t], want)
    assert all(v == v for r in rows for v in r)  # row 0 (never stepped from) holds no NaN either
    dd = dmme_amd.DDIM(torch.nn.Identity(), T, 50)
    n, rows, ttab = dd._chain_tables()
    tau = OD.tau_table(T, 50)
    assert n == 50 and ttab == [int(v) for v in tau]
    for i in (1, 2, 25, 50):
        want = (float(torch.sqrt(1 - abar[int(tau[i])])), float(torch.sqrt(abar[int(tau[i - 1])])))
        assert rows[i][:2] == want, (i, rows[i], want)
    idd = dmme_amd.IDDPM(torch.nn.Identity(), 100)
    n, rows, ttab = idd._chain_tables()
    assert n == 100 and ttab == list(range(101))
    for t in (1, 50, 100):
        assert list(rows[t]) == list(idd._coef_host[t][:4])


def test_bench_defaults_and_rank_launcher_command(monkeypatch):
    """`python bench.py` defaults finish within minutes; `--gpus N` without a torch.distributed environment builds the
    torch.distributed.run command line the contract names (127.0.0.1, one rank per GPU) instead of exiting"""
    import importlib
    import sys as _sys

    monkeypatch.setattr(_sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    a = bench.parse()
    assert a.gpus == 1 and a.steps <= 100 and a.warmup <= 20 and a.batch == 128 and a.precision == "bf16" and a.mode == "sample"
    seen = {}

    class _P:
        stdout = iter(['{"metric": "m", "value": 1}\n', "noise\n"])

        def wait(self):
            return 0

    def fake_popen(cmd, **kw):
        seen["cmd"] = cmd
        return _P()

    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    assert bench.spawn_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_bench_cpu_plumbing_mode_runs_the_reference_yaml_on_the_oracle():
    """BASELINE configs[0]: `python bench.py --mode cpu-plumbing` - the shipped YAML through the product's parser and constructors, ten
    sampling steps and two training steps at batch 1 in the CPU oracle, ONE JSON line with finite numbers (no GPU involved)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "cpu-plumbing"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 0 and rec["steps"] == 10 and rec["finite"] is True and rec["value"] > 0 and len(rec["train_losses"]) == 2
    assert "configs/ddpm/cifar10.yaml" in rec["config"]["workload"] and "LitDDPM" in rec["config"]["workload"]


def test_trainer_sample_geometry_and_precision_follow_the_yaml(tmp_path):
    """`trainer sample` takes the image size from the YAML's data module (dmme.LSUN: init_args.imgsize, configs/ddpm/lsun_church.yaml:94)
    and maps `precision: 16` to IEEE half (training under dynamic loss scaling, and sampling)"""
    import os

    from dmme_amd import trainer

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = trainer.parse_config(os.path.join(root, "configs", "ddpm", "cifar10.yaml"))
    assert conf["image_size"] == 32 and conf["precision"] == "fp16" and conf["sample_precision"] == "fp16"
    ref = "/root/reference/configs/ddpm/lsun_church.yaml"
    if os.path.exists(ref):  # (not on the GPU box)
        assert trainer.parse_config(ref)["image_size"] == 256
    y = tmp_path / "c.yaml"
    y.write_text("trainer:\n  precision: bf16\nmodel:\n  class_path: dmme.LitDDPM\ndata:\n  class_path: dmme.LSUN\n  init_args:\n    imgsize: 64\n")
    c2 = trainer.parse_config(str(y))
    assert c2["image_size"] == 64 and c2["sample_precision"] == "bf16"
