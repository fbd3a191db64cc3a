"""Minimal runner for the reference's LightningCLI YAML files
(`dmme.trainer fit --config configs/ddpm/cifar10.yaml`, reference src/dmme/trainer.py:4-9).

pytorch_lightning / jsonargparse are not part of this image, so the runner parses the
same YAML with PyYAML and applies jsonargparse's rule itself: every `init_args` value is
converted to the type its constructor parameter is annotated with (`lr: 2e-4` is a string
to YAML 1.1 and a float to `LitDDPM.__init__(lr: float)`; `800_000` an int; an
`Optional[nn.Module]` parameter takes a nested `class_path` / `init_args` block).  It
honours the keys that affect the hot path and ignores the Lightning-only ones (loggers,
checkpoint callbacks, ...):

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
import inspect
import typing
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


def _number(text: str, kind):
    """YAML 1.1 scalars jsonargparse reads as numbers: `2e-4` (no dot: a string to PyYAML), `800_000`"""
    cleaned = text.strip().replace("_", "")
    if kind is int:
        try:
            return int(cleaned, 0)
        except ValueError:
            as_float = float(cleaned)
            if as_float != int(as_float):
                raise
            return int(as_float)
    return float(cleaned)


def _coerce(value: Any, annotation: Any) -> Any:
    """Convert a parsed YAML value to the annotated parameter type (the subset of jsonargparse's typing rules the
    reference's configs exercise: float / int / bool / str, Optional and Union, Sequence / List / Tuple of those,
    sub-class specs as `class_path` + `init_args`)."""
    if isinstance(value, dict) and "class_path" in value:
        return _instantiate(value)
    if annotation is inspect.Parameter.empty or annotation is Any or annotation is None:
        return _instantiate(value) if isinstance(value, list) else value
    origin = typing.get_origin(annotation)
    args = typing.get_args(annotation)
    if origin is typing.Union:
        if value is None and type(None) in args:
            return None
        last = None
        for cand in args:
            if cand is type(None):
                continue
            try:
                return _coerce(value, cand)
            except (TypeError, ValueError) as exc:
                last = exc
        raise last if last else TypeError(f"{value!r} matches no member of {annotation}")
    if origin in (list, tuple, typing.Sequence) or (isinstance(origin, type) and issubclass(origin, (list, tuple, typing.Sequence))) \
            or origin is __import__("collections").abc.Sequence:
        if not isinstance(value, (list, tuple)):
            raise TypeError(f"expected a sequence for {annotation}, got {value!r}")
        if origin is tuple and args and not (len(args) == 2 and args[1] is Ellipsis):
            items = [_coerce(v, a) for v, a in zip(value, args)]
        else:
            inner = args[0] if args else Any
            items = [_coerce(v, inner) for v in value]
        return tuple(items) if origin is tuple else items
    if annotation is float:
        if isinstance(value, bool):
            raise TypeError(f"expected a float, got {value!r}")
        if isinstance(value, (int, float)):
            return float(value)
        if isinstance(value, str):
            return _number(value, float)
        raise TypeError(f"expected a float, got {value!r}")
    if annotation is int:
        if isinstance(value, bool):
            raise TypeError(f"expected an int, got {value!r}")
        if isinstance(value, int):
            return value
        if isinstance(value, float) and value == int(value):
            return int(value)
        if isinstance(value, str):
            return _number(value, int)
        raise TypeError(f"expected an int, got {value!r}")
    if annotation is bool:
        if isinstance(value, bool):
            return value
        if isinstance(value, str) and value.lower() in ("true", "false"):
            return value.lower() == "true"
        raise TypeError(f"expected a bool, got {value!r}")
    if annotation is str:
        if isinstance(value, str):
            return value
        raise TypeError(f"expected a str, got {value!r}")
    if isinstance(value, list):
        return [_instantiate(v) for v in value]
    return value


def _init_annotations(cls) -> Dict[str, Any]:
    """parameter name -> annotation over the class and its bases (a subclass that forwards **kwargs inherits its parent's)"""
    out: Dict[str, Any] = {}
    for klass in reversed(inspect.getmro(cls)):
        init = klass.__dict__.get("__init__")
        if init is None:
            continue
        try:
            hints = typing.get_type_hints(init)
        except Exception:  # noqa: BLE001 - unresolved forward references: fall back to the raw annotations
            hints = getattr(init, "__annotations__", {})
        for name, par in inspect.signature(init).parameters.items():
            if name == "self":
                continue
            ann = hints.get(name, par.annotation)
            if isinstance(ann, str):
                ann = inspect.Parameter.empty
            out[name] = ann
    return out


def _instantiate(spec: Any):
    if isinstance(spec, dict) and "class_path" in spec:
        cls = _resolve(spec["class_path"])
        hints = _init_annotations(cls)
        kwargs = {}
        for k, v in (spec.get("init_args") or {}).items():
            try:
                kwargs[k] = _coerce(v, hints.get(k, inspect.Parameter.empty))
            except (TypeError, ValueError) as exc:
                raise TypeError(f"{spec['class_path']}: init_args.{k} = {v!r} does not fit the parameter's type {hints.get(k)}: {exc}") from exc
        return cls(**kwargs)
    if isinstance(spec, list):
        return [_instantiate(v) for v in spec]
    return spec


def parse_config(path: str) -> Dict[str, Any]:
    with open(path) as f:
        cfg = yaml.safe_load(f)
    trainer = cfg.get("trainer") or {}
    data_args = ((cfg.get("data") or {}).get("init_args")) or {}
    precision = trainer.get("precision", 32)
    p16 = str(precision) in ("16", "16-mixed")
    pbf = str(precision) in ("bf16", "bf16-mixed")
    return {
        "model_spec": cfg["model"],
        "data_spec": cfg.get("data"),
        "batch_size": _coerce(data_args.get("batch_size", 128), int),
        "max_steps": _coerce(trainer.get("max_steps") if trainer.get("max_steps") is not None else -1, int),
        "gradient_clip_val": _coerce(trainer.get("gradient_clip_val"), typing.Optional[float]),
        # `precision: 16` (configs/ddpm/cifar10.yaml:53) is fp16 autocast under a GradScaler in the reference: here IEEE-half tensors and
        # MFMA operands with fp32 accumulation and master weights, under the device-resident dynamic loss scaling of the fused optimiser
        # pass (optim.FusedAdam(amp=True), include/dmme_hip.h: dmme_amp_*) - for `fit` and for sampling alike
        "precision": "fp16" if p16 else "bf16" if pbf else "fp32",
        "sample_precision": "fp16" if p16 else "bf16" if pbf else "fp32",
        # images the YAML's data module yields (dmme.CIFAR10: 32 x 32; dmme.LSUN: init_args.imgsize, configs/ddpm/lsun_church.yaml:94)
        "image_size": _coerce(data_args.get("imgsize", 32), int),
        "log_every_n_steps": _coerce(trainer.get("log_every_n_steps") or 50, int),
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
    ap.add_argument("--image-size", type=int, default=None, help="sample: image height = width (default: what the YAML's data module yields)")
    ap.add_argument("--precision", default=None, help="override the YAML's trainer.precision (fp32 | bf16 | fp16 | bf16x3)")
    ap.add_argument("--steps", type=int, default=None, help="sample: stop after this many denoising steps")
    args = ap.parse_args(argv)

    from . import _lib
    from . import distributed as D

    conf = parse_config(args.config)
    if args.precision:
        conf["precision"] = conf["sample_precision"] = args.precision
    if args.command == "sample":
        conf["precision"] = conf["sample_precision"]
    _lib.require_gpu()  # the product path is the HIP denoiser: no CPU fallback (the CPU plumbing run of BASELINE configs[0] is bench.py --mode cpu-plumbing)
    # one process per GPU under `python -m torch.distributed.run` (the reference reaches DDP through trainer.devices / strategy)
    rank, local, world = D.init_from_env()
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    seed = conf["seed"] if isinstance(conf["seed"], int) and not isinstance(conf["seed"], bool) else 1337
    torch.manual_seed(seed)  # identical initial weights on every rank ...
    module = build_module(conf).cuda()
    if world > 1:
        torch.manual_seed(D.rank_seed(seed, rank))  # ... then each rank's own stream for timesteps, noise and dropout masks
    B = args.batch_size or conf["batch_size"]

    ckpt_path = args.ckpt_path or conf.get("ckpt_path")
    if args.command == "sample":
        if ckpt_path:
            from .checkpoint import load_checkpoint

            load_checkpoint(ckpt_path, module, strict=False)
        module.eval()
        dm = module.diffusion_model
        t0 = time.perf_counter()
        hw = args.image_size or conf["image_size"]
        shape = (args.num_images, dm.model.in_channels, hw, hw)
        if args.steps is None:
            imgs = module.generate(shape)
        else:
            import dmme_amd

            imgs = dmme_amd.gaussian(shape, device="cuda")
            for k in range(args.steps):
                imgs = module(imgs, dm.timesteps - k)
        torch.cuda.synchronize()
        print(json.dumps({"images": list(imgs.shape), "precision": conf["precision"], "seconds": round(time.perf_counter() - t0, 3),
                          "finite": bool(torch.isfinite(imgs).all())}))
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
