#!/usr/bin/env python3
"""Headline benchmark: DDPM denoising steps/s on the default CIFAR10 UNet (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch: eps = UNet(x_t, t) at batch 128 per GPU in
bf16 followed by the DDPM reverse update (Philox noise included), t running down from T = 1000.
Inputs and weights are resident in HBM before the timed region.  With --gpus N every rank runs its
own independent batch-128 chain (sampling shards with no data-path collective: weak scaling).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sample|ddim] [--precision bf16|fp32]

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel, measured live with HIP events
on the launch stream) and `cpu_baseline` (the CPU oracle timed on this host's cores, rank 0, N=1).
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")

import torch  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA TFLOP/s (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0
METRIC = "denoising steps/sec + training images/sec, DDPM UNet CIFAR10 32×32 @1/2/4/8 GPU"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--mode", default="sample", choices=["sample", "ddim", "train"])
    ap.add_argument("--model", default="ddpm", choices=["ddpm", "iddpm64"],
                    help="ddpm: BASELINE configs[1]/[2] (default UNet, 32x32); iddpm64: configs[3] (IDDPM ImageNet-64 UNet, attention at "
                         "16x16/8x8, cosine schedule, T=4000; use --batch 32 for the 256-over-8-GPUs shard)")
    ap.add_argument("--train-steps", type=int, default=8, help="training steps timed for train_images_per_s (0: skip)")
    ap.add_argument("--graph", action="store_true", help="replay the UNet forward from a hipGraph (small-batch sampling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def roofline_leg(model, x, t_dev, precision):
    """Per-op HIP-event timing of the same forward (dmme_unet_forward_profiled), grouped by kernel
    symbol; the dominant kernel's algorithmic FLOPs / its summed launch time is `achieved`."""
    from dmme_amd import _lib

    plan = model._last_plan
    lib = plan.lib
    n = lib.dmme_unet_plan_num_ops(plan.h)
    labels, flops, nbytes = [], [], []
    buf = C.create_string_buffer(128)
    f, b = C.c_double(), C.c_double()
    for i in range(n):
        _lib.check(lib.dmme_unet_plan_op_info(plan.h, i, buf, 128, C.byref(f), C.byref(b)))
        labels.append(buf.value.decode())
        flops.append(f.value)
        nbytes.append(b.value)
    packed = model._packed_for(plan)
    y = torch.empty_like(x)
    ms = (C.c_float * n)()
    acc = [0.0] * n
    reps = 5
    for r in range(reps + 1):
        _lib.check(lib.dmme_unet_forward_profiled(plan.h, _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t_dev), 1, _lib.ptr(y),
                                                  _lib.ptr(plan.workspace), _lib.ptr(None), _lib.stream_ptr(), ms))
        if r:  # first repetition is a warm-up
            for i in range(n):
                acc[i] += ms[i] / reps
    groups = {}
    for i in range(n):
        g = groups.setdefault(labels[i], {"ms": 0.0, "count": 0, "flops": 0.0, "bytes": 0.0})
        g["ms"] += acc[i]
        g["count"] += 1
        g["flops"] += flops[i]
        g["bytes"] += nbytes[i]
    order = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
    total_ms = sum(g["ms"] for _, g in order)
    table = []
    for name, g in order[:8]:
        table.append({
            "kernel": name, "launches": g["count"], "ms_per_step": round(g["ms"], 4),
            "avg_launch_us": round(1e3 * g["ms"] / g["count"], 2), "share": round(g["ms"] / total_ms, 4),
            "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 2) if g["ms"] > 0 else 0.0,
            "algo_gbs": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1) if g["ms"] > 0 else 0.0,
        })
    name, g = order[0]
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and x.shape[0] == 128 and x.shape[-1] == 32 and precision == "bf16":  # the PMC passes (tools/prof.sh) run this configuration only
        try:
            traffic = json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None
    if g["flops"] > 0:
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK.get(precision, 2500.0), "unit": "TFLOP/s",
                "frac": round(ach / PEAK.get(precision, 2500.0), 4)}
    else:
        ach = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    roof.update({"traffic": traffic, "launches_per_step": g["count"], "avg_launch_us": round(1e3 * g["ms"] / g["count"], 2),
                 "algo_flops_per_launch": g["flops"] / g["count"], "algo_bytes_per_launch": g["bytes"] / g["count"],
                 "algo_gbs": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1), "step_gpu_ms_sum": round(total_ms, 3), "top_kernels": table})
    return roof


def cpu_baseline_leg(batch_ref):
    """The CPU oracle (oracle/, parity-pinned against the reference) timed on this host: a bounded
    sample of the same workload (same UNet, same update, fp32), scaled to batch-`batch_ref` steps."""
    from oracle import diffusion as D
    from oracle import synth
    from oracle import unet as O

    # the GPU box exposes every host core but grants a 16-core share per GPU: more threads only thrash
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg = O.UNetConfig()
    sd = O.make_state_dict(cfg, 1337)
    Bc = 8
    x = synth.normal(1, (Bc, 3, 32, 32))
    beta = D.linear_beta(1000)
    alpha, abar = D.alpha_tables(beta)
    z = synth.normal(2, x.shape)
    done, t0 = 0, None
    with torch.no_grad():
        for k in range(200):
            t = 1000 - k
            eps = O.unet_forward(sd, cfg, x, torch.tensor([t]))
            x = D.ddpm_step(x, t, eps, z, beta, alpha, abar)
            if k == 0:
                t0 = time.perf_counter()  # first step is the warm-up
                continue
            done += 1
            if time.perf_counter() - t0 > 12.0:
                break
    dt = time.perf_counter() - t0
    img_steps = done * Bc / dt
    return {"value": round(img_steps / batch_ref, 4), "unit": f"denoising steps/s at batch {batch_ref} (scaled from the sample)",
            "image_steps_per_s": round(img_steps, 2), "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} DDPM steps of batch {Bc}, fp32, oracle UNet + update, {dt:.1f} s"}


def workload(dmme_amd, name, precision):
    """(UNet, image side, T, process class, Lit class, label) of a --model choice"""
    if name == "iddpm64":
        from dmme_amd.models import iddpm

        return (iddpm.UNet(attention_depths=(3, 4), precision=precision), 64, 4000, dmme_amd.IDDPM, dmme_amd.LitIDDPM,
                "IDDPM ImageNet-64 UNet (36,562,822 params, 4-head attention at 16x16/8x8, dropout 0.3), cosine schedule, T=4000")
    return dmme_amd.UNet(precision=precision), 32, 1000, dmme_amd.DDPM, dmme_amd.LitDDPM, "default UNet (32,416,643 params, random init)"


def train_leg(dmme_amd, dev, B, precision, steps, warmup, dist, model_name="ddpm"):
    """training images/s: q_sample -> UNet fwd (train mode, Dropout2d on) -> MSE -> HIP backward -> (RCCL mean
    all-reduce of the flat gradient) -> fused clip(1.0)+Adam+EMA -> warm-up LR step; per-GPU batch B."""
    from dmme_amd.train_loop import synthetic_batch, train_step

    net, side, T, _, lit_cls, _ = workload(dmme_amd, model_name, precision)
    lit = lit_cls(model=net, timesteps=T).to(dev)
    lit.train()
    opts, scheds = lit.configure_optimizers()
    opt, sched = opts[0], scheds[0]["scheduler"]
    for g in opt.param_groups:
        g["max_grad_norm"] = 1.0
    x0 = synthetic_batch(B, dev, (3, side, side))

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(warmup):
        loss = train_step(lit, opt, sched, x0)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(lit, opt, sched, x0)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(loss).all(), "non-finite training loss"
    return dt, float(loss.detach())


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist

        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
        # rehearsal on a box with fewer GPUs than ranks: DMME_DIST_BACKEND=gloo lets the ranks share devices (RCCL refuses that)
        backend = os.environ.get("DMME_DIST_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)

    import dmme_amd

    torch.manual_seed(1337 + rank)
    B = args.batch
    model, side, T, proc_cls, _, label = workload(dmme_amd, args.model, args.precision)
    if args.mode == "train":
        del model
        dt, loss = train_leg(dmme_amd, dev, B, args.precision, args.steps, args.warmup, dist, args.model)
        if rank == 0:
            print(json.dumps({
                "metric": METRIC, "value": round(world * args.steps * B / dt, 2),
                "unit": "training images/s (q_sample + UNet fwd/bwd + grad all-reduce + clip + Adam + EMA), summed over GPUs",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": (f"DDPM CIFAR10 32x32 training step, default UNet, batch {B} per GPU, dropout 0.1" if args.model == "ddpm" else
                                        f"{label}: hybrid-loss training step at 64x64, batch {B} per GPU"), "global_batch": B * world,
                           "parallelism": f"dp{world} (RCCL mean all-reduce of the flat fp32 gradient)"}, "final_loss": round(loss, 5)}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    model = model.to(dev).eval()
    if args.mode == "ddim":
        proc = dmme_amd.DDIM(model, T, 50).to(dev)
    else:
        proc = proc_cls(model, T).to(dev)
    x = dmme_amd.gaussian((B, 3, side, side), device=dev)
    all_t = torch.arange(0, T + 1, device=dev).unsqueeze(1)
    tau = proc._tau_host if args.mode == "ddim" else None

    t_buf = all_t[T].clone()

    def fwd(tt):
        if args.graph:
            t_buf.copy_(tt)
            return model.graphed_forward(x, t_buf)
        return model(x, tt)

    def one_step(k):
        with torch.no_grad():
            if args.mode == "ddim":
                i = 50 - (k % 50)
                eps = fwd(all_t[tau[i]])
                proc._ddim_update(x, eps, i)
            else:
                t = T - (k % T)
                eps = fwd(all_t[t])
                proc._reverse_update(x, eps, t, None)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(x).all(), "non-finite samples"

    out = {
        "metric": METRIC,
        "value": round(world * args.steps / elapsed, 3),
        "unit": f"denoising steps/s (one step = UNet forward + {'DDIM' if args.mode == 'ddim' else 'IDDPM learned-variance' if args.model == 'iddpm64' else 'DDPM'} update on a batch of {B}), summed over GPUs",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": {
            "workload": (f"{'DDIM 50-step (quadratic tau)' if args.mode == 'ddim' else 'DDPM T=1000'} sampling, CIFAR10 32x32, {label}, "
                         f"batch {B} per GPU, t.shape=(1,)") if args.model == "ddpm" else f"{label}: learned-variance sampling at 64x64, batch {B} per GPU, t.shape=(1,)",
            "global_batch": B * world,
            "parallelism": f"dp{world} (independent chains per GPU, no data-path collective)",
        },
        "image_steps_per_s": round(world * args.steps * B / elapsed, 1),
        "train_images_per_s": None,
        "launches_per_step": int(model._last_plan.lib.dmme_unet_plan_num_launches(model._last_plan.h)) + 2,
        "hip_graph": bool(args.graph and not getattr(model, "_graph_disabled", False)),
    }
    def rank0_legs():  # the per-kernel roofline of the measured forward and the CPU baseline
        if rank == 0:
            if not args.no_roofline:
                xin = dmme_amd.gaussian((B, 3, side, side), device=dev)
                out["roofline"] = roofline_leg(model, xin, all_t[500], args.precision)
                del xin
            if world == 1 and not args.no_cpu_baseline and args.model == "ddpm":
                out["cpu_baseline"] = cpu_baseline_leg(B)

    # several ranks: rank 0's legs first, so that a stuck collective in the training leg cannot cost them; one rank: after it (no
    # collective to get stuck in, and the event-bracketed kernel times sit closer to rocprofv3's with the device in its training-leg state)
    if world > 1 or args.train_steps <= 0:
        rank0_legs()
    if args.train_steps > 0:
        del x
        # Secondary figure.  Neither an exception nor a stuck collective in it may cost the headline line: past the deadline every
        # rank leaves through the watchdog, rank 0 printing the line it already has.
        import threading

        def bail():
            if rank == 0:
                out["train_error"] = "training leg exceeded its deadline"
                print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(240.0, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            dt_tr, _ = train_leg(dmme_amd, dev, B, args.precision, args.train_steps, 2, dist, args.model)
            out["train_images_per_s"] = round(world * args.train_steps * B / dt_tr, 1)
            out["train_ms_per_step"] = round(1e3 * dt_tr / args.train_steps, 2)
        except Exception as exc:  # noqa: BLE001
            out["train_error"] = f"{type(exc).__name__}: {exc}"[:300]
        watchdog.cancel()
        if world == 1:
            rank0_legs()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
