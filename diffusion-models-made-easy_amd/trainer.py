"""Minimal runner for the reference's LightningCLI YAML files
(`dmme.trainer fit --config configs/ddpm/cifar10.yaml`, reference src/dmme/trainer.py:4-9).

pytorch_lightning / jsonargparse are not part of this image, so the runner parses the
same YAML with PyYAML, honours the keys that affect the hot path and ignores the
Lightning-only ones (loggers, checkpoint callbacks, ...):

  model.class_path / init_args[.model.init_args]   -> dmme_amd.LitDDPM / LitDDIM (+ UNet overrides)
  data.init_args.batch_size                         -> synthetic batches of that size (default), or with `--data config`
                                                       the YAML's data module (dmme.CIFAR10: HBM-resident uint8 set, GPU flip + norm)
  trainer.max_steps, gradient_clip_val, precision, log_every_n_steps, devices
  seed_everything

  python -m dmme_amd.trainer fit    --config configs/ddpm/cifar10.yaml [--max-steps N] [--batch-size B]
  python -m dmme_amd.trainer sample --config configs/ddim/cifar10.yaml [--num-images N] [--steps K]
"""

from __future__ import annotations

import argparse
import importlib
import json
import sys
import time
from typing import Any, Dict

import torch
import yaml


def _resolve(class_path: str):
    """`dmme.X` in a reference YAML means this package's drop-in X."""
    mod, _, name = class_path.rpartition(".")
    if class_path == "torchvision.transforms.RandomHorizontalFlip":  # the flip runs on the GPU inside dmme_image_batch
        mod = "dmme_amd.data_modules"
    if mod == "dmme" or mod.startswith("dmme."):
        mod = "dmme_amd" + mod[4:]
    return getattr(importlib.import_module(mod), name)


def _instantiate(spec: Any):
    if isinstance(spec, dict) and "class_path" in spec:
        kwargs = {k: _instantiate(v) for k, v in (spec.get("init_args") or {}).items()}
        return _resolve(spec["class_path"])(**kwargs)
    if isinstance(spec, list):
        return [_instantiate(v) for v in spec]
    return spec


def parse_config(path: str) -> Dict[str, Any]:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    trainer = cfg.get("trainer") or {}
    data_args = ((cfg.get("data") or {}).get("init_args")) or {}
    precision = trainer.get("precision", 32)
    return {
        "model_spec": cfg["model"],
        "data_spec": cfg.get("data"),
        "batch_size": int(data_args.get("batch_size", 128)),
        "max_steps": int(trainer.get("max_steps") or -1),
        "gradient_clip_val": trainer.get("gradient_clip_val"),
        "precision": "bf16" if str(precision) in ("16", "bf16", "16-mixed", "bf16-mixed") else "fp32",
        "log_every_n_steps": int(trainer.get("log_every_n_steps") or 50),
        "devices": trainer.get("devices", 1),
        "seed": cfg.get("seed_everything", 1337),
        "ckpt_path": cfg.get("ckpt_path"),
    }


def build_module(conf: Dict[str, Any]):
    module = _instantiate(conf["model_spec"])
    unet = module.diffusion_model.model
    if hasattr(unet, "set_precision"):
        unet.set_precision(conf["precision"])
    return module


def main(argv=None):
    ap = argparse.ArgumentParser(prog="dmme_amd.trainer")
    ap.add_argument("command", choices=["fit", "sample"])
    ap.add_argument("--config", required=True)
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "config"],
                    help="fit: 'config' instantiates the YAML's data module (falls back to a random byte set when the CIFAR10 files are absent)")
    ap.add_argument("--ckpt-path", default=None, help="resume / sample from a Lightning-layout checkpoint (the YAML's ckpt_path key)")
    ap.add_argument("--save-checkpoint", default=None, help="fit: write a Lightning-layout checkpoint (weights, Adam moments, EMA copy) at the end")
    ap.add_argument("--num-images", type=int, default=16)
    ap.add_argument("--steps", type=int, default=None, help="sample: stop after this many denoising steps")
    args = ap.parse_args(argv)

    conf = parse_config(args.config)
    seed = conf["seed"] if isinstance(conf["seed"], int) and not isinstance(conf["seed"], bool) else 1337
    torch.manual_seed(seed)
    module = build_module(conf).cuda()
    B = args.batch_size or conf["batch_size"]

    ckpt_path = args.ckpt_path or conf.get("ckpt_path")
    if args.command == "sample":
        if ckpt_path:
            from .checkpoint import load_checkpoint

            load_checkpoint(ckpt_path, module, strict=False)
        module.eval()
        dm = module.diffusion_model
        t0 = time.perf_counter()
        if args.steps is None:
            imgs = module.generate((args.num_images, 3, 32, 32))
        else:
            import dmme_amd

            imgs = dmme_amd.gaussian((args.num_images, 3, 32, 32), device="cuda")
            for k in range(args.steps):
                imgs = module(imgs, dm.timesteps - k)
        torch.cuda.synchronize()
        print(json.dumps({"images": list(imgs.shape), "seconds": round(time.perf_counter() - t0, 3), "finite": bool(torch.isfinite(imgs).all())}))
        return 0

    from .train_loop import fit

    steps = args.max_steps if args.max_steps is not None else conf["max_steps"]
    loader = None
    if args.data == "config" and conf["data_spec"]:
        dm = _instantiate(conf["data_spec"])
        dm.batch_size = B
        try:
            dm.prepare_data()
        except FileNotFoundError as e:
            print(json.dumps({"data": "random bytes (dataset files absent)", "reason": str(e)[:160]}), flush=True)
            dm.synthetic = True
        dm.setup("fit")
        loader = dm.train_dataloader()
    fit(module, batch_size=B, max_steps=steps, clip=conf["gradient_clip_val"], log_every=conf["log_every_n_steps"], loader=loader,
        ckpt_path=ckpt_path, save_path=args.save_checkpoint)
    return 0


if __name__ == "__main__":
    sys.exit(main())
