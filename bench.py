#!/usr/bin/env python3
"""Headline benchmark: DDPM denoising steps/s on the default CIFAR10 UNet (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch: eps = UNet(x_t, t) at batch 128 per GPU in
bf16 followed by the DDPM reverse update (Philox noise included), t running down from T = 1000 -
the same replayable step `DDPM.generate` runs (one captured hipGraph: time MLP + UNet + noise +
update + t -> t-1, all loop state resident on the device).  Inputs and weights are resident in HBM
before the timed region.  With --gpus N every rank runs its own independent batch-128 chain
(sampling shards with no data-path collective: weak scaling).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sample|ddim|train|cpu-plumbing] [--precision bf16|fp16|bf16x3|fp32]

`python bench.py --gpus N` with N > 1 and no torch.distributed environment starts its N ranks itself
(`python -m torch.distributed.run`, before this process touches the GPU) and relays rank 0's line.
Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel, measured live with HIP events
on the launch stream) and `cpu_baseline` (the CPU oracle timed on this host's cores, rank 0, N=1).
"""

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")

import torch  # noqa: E402

# dense MFMA TFLOP/s (MI355X_MICROARCH.md, chip-level parameters); bf16x3 runs three bf16 matrix products per algorithmic one
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp16r32": 2500.0, "bf16x3": 2500.0 / 3.0, "fp32": 157.3}
HBM_PEAK_GBS = 8000.0
METRIC = "denoising steps/sec + training images/sec, DDPM UNet CIFAR10 32×32 @1/2/4/8 GPU"
FWD_GFLOP_PER_IMAGE = {"ddpm": 9.809, "iddpm64": 37.50}  # conv + linear + QK^T + AV, 2 x MAC; training = 3x


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3, help="timed blocks of --steps steps each (value = the median block; min and all blocks are reported)")
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--mode", default="sample", choices=["sample", "ddim", "train", "cpu-plumbing"])
    ap.add_argument("--model", default="ddpm", choices=["ddpm", "iddpm64"],
                    help="ddpm: BASELINE configs[1]/[2] (default UNet, 32x32); iddpm64: configs[3] (IDDPM ImageNet-64 UNet, attention at "
                         "16x16/8x8, cosine schedule, T=4000; use --batch 32 for the 256-over-8-GPUs shard)")
    ap.add_argument("--train-steps", type=int, default=30, help="training steps timed for train_images_per_s (0: skip)")
    ap.add_argument("--no-graph", action="store_true", help="issue the step's launches eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--graph", action="store_true", help=argparse.SUPPRESS)  # accepted for older command lines: the graph is the default now
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-accurate-leg", action="store_true", help="skip the short bf16x3 (accurate mode) and fp16 throughput legs")
    ap.add_argument("--no-small-batch-leg", action="store_true", help="skip the batch-1 / batch-32 legs of the N = 1 line")
    ap.add_argument("--no-ddim-leg", action="store_true", help="skip the BASELINE configs[2] leg (DDIM 50 steps, batch 512) of the N = 1 line")
    ap.add_argument("--config", default=None, help="cpu-plumbing: the LightningCLI YAML to drive (default configs/ddpm/cifar10.yaml)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- launcher (python bench.py --gpus N)
def spawn_ranks(n: int) -> int:
    """Start the N ranks as children under torch.distributed.run BEFORE this process has touched the GPU, relay their output,
    and fail when any rank fails or rank 0's JSON line is missing.  (Never re-exec a process that initialised the GPU.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL peer access)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0:
        sys.stderr.write(f"bench.py: the {n}-rank run exited with status {rc}\n")
        return rc
    return 0 if line is not None else 4


# ---------------------------------------------------------------------------------------------- per-kernel roofline
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
    y = torch.empty((x.shape[0], model.out_channels, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device)
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
        if labels[i].startswith("("):  # an op that launches nothing (a GroupNorm finished inside a neighbouring conv): its "time" is
            continue                   # the event pair's own cost
        g = groups.setdefault(labels[i], {"ms": 0.0, "count": 0, "flops": 0.0, "bytes": 0.0})
        g["ms"] += acc[i]
        g["count"] += 1
        g["flops"] += flops[i]
        g["bytes"] += nbytes[i]
    order = sorted(groups.items(), key=lambda kv: -kv[1]["ms"])
    total_ms = sum(g["ms"] for _, g in order)
    # Counters cannot be read from inside this process: MFMA-busy and HBM-traffic figures are those of the committed rocprofv3 --pmc
    # passes of this workload (tools/prof.sh -> profiles/*_latest.json) and are only handed on when that file was measured on THIS
    # build - its recorded source hash equals the hash of the sources the loaded library was built from - else null, with the reason
    here = _lib.csrc_sha16()

    def profile_file(fname):
        path = os.path.join(ROOT, "profiles", fname)
        if not (os.path.exists(path) and x.shape[0] == 128 and x.shape[-1] == 32 and precision == "bf16"):
            return None, "no committed counter pass for this workload"
        try:
            rec = json.load(open(path))
        except Exception:  # noqa: BLE001
            return None, f"profiles/{fname}: unreadable"
        meta = rec.get("_meta") or {}
        if meta.get("csrc_sha16") != here:
            return None, f"profiles/{fname} was measured on another build (sources {meta.get('csrc_sha16')}, loaded {here}): dropped"
        return rec, f"file: profiles/{fname} ({meta.get('label')}; rocprofv3 --pmc passes of this workload on a builder box with this build, not this run)"

    busy_rec, busy_source = profile_file("mfma_busy_latest.json")
    pmc_busy = {k: v.get("mfma_busy_pct") for k, v in (busy_rec or {}).items() if isinstance(v, dict) and k != "_meta"}
    table = []
    for name, g in order[:8]:
        table.append({
            "kernel": name, "launches": g["count"], "ms_per_step": round(g["ms"], 4),
            "avg_launch_us": round(1e3 * g["ms"] / g["count"], 2), "share": round(g["ms"] / total_ms, 4),
            "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 2) if g["ms"] > 0 else 0.0,
            "algo_gbs": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1) if g["ms"] > 0 else 0.0,
            # both roofs per kernel, so that an HBM-shaped kernel (1x1 convs, attention over a materialised qkv) is read against the right one
            "mfma_frac": round(g["flops"] / (g["ms"] * 1e-3) / 1e12 / PEAK.get(precision, 2500.0), 4) if g["ms"] > 0 else 0.0,
            "hbm_frac": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if g["ms"] > 0 else 0.0,
            "mfma_busy_pct_from_profile": pmc_busy.get(name),
        })
    name, g = order[0]
    traffic_rec, traffic_source = profile_file("traffic_latest.json")
    traffic = (traffic_rec or {}).get(name, {}).get("hbm_bytes_per_launch")
    peak = PEAK.get(precision, 2500.0)
    if g["flops"] > 0:
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
    else:
        ach = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    roof.update({"traffic": traffic, "traffic_source": traffic_source, "mfma_busy_source": busy_source, "csrc_sha16": here, "launches_per_step": g["count"], "avg_launch_us": round(1e3 * g["ms"] / g["count"], 2),
                 "algo_flops_per_launch": g["flops"] / g["count"], "algo_bytes_per_launch": g["bytes"] / g["count"],
                 "algo_gbs": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1), "step_gpu_ms_sum": round(total_ms, 3), "top_kernels": table})
    return roof


# ---------------------------------------------------------------------------------------------- CPU baseline (the oracle)
def _cpu_threads():
    """threads for the CPU legs: the GPU box shows every host core but grants a 16-core share per GPU - more threads only thrash"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return min(16, avail), avail


def cpu_baseline_leg(batch):
    """The CPU oracle (oracle/, parity-pinned against the reference) timed on this host at the benchmark's own batch size:
    same UNet, same DDPM update, fp32; one warm-up step, then steps for about 20 s (at least 3)."""
    from oracle import diffusion as D
    from oracle import synth
    from oracle import unet as O

    threads, avail = _cpu_threads()
    torch.set_num_threads(threads)
    cfg = O.UNetConfig()
    sd = O.make_state_dict(cfg, 1337)
    x = synth.normal(1, (batch, 3, 32, 32))
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
            if done >= 3 and time.perf_counter() - t0 > 20.0:
                break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": f"denoising steps/s at batch {batch}", "image_steps_per_s": round(done * batch / dt, 2),
            "cores": threads, "host_cpus_visible": avail, "os_cpu_count": os.cpu_count(), "kind": "port",
            "kind_detail": f"oracle/ (CPU restatement of the reference's UNet + DDPM update, pinned by the reference's own outputs), fp32, {threads} threads",
            "sample": f"{done} DDPM steps of batch {batch} (the benchmark's own batch, not scaled) after 1 warm-up, fp32, oracle UNet + update, "
                      f"{dt:.1f} s, torch intra-op threads = {threads} (the per-GPU core share of the box)"}


def cpu_plumbing(args):
    """BASELINE configs[0]: `configs/ddpm/cifar10.yaml` driving the CPU reference path at batch 1, T = 1000 - ten sampling
    steps and two training steps, no GPU.  The YAML goes through the product's own parser / constructors (host code); the math
    runs in the CPU oracle (this leg is the `cpu_baseline` side of the bench: the product path itself has no CPU fallback)."""
    from dmme_amd import trainer
    from oracle import diffusion as D
    from oracle import synth
    from oracle import unet as O

    threads, avail = _cpu_threads()
    torch.set_num_threads(threads)
    path = args.config or os.path.join(ROOT, "configs", "ddpm", "cifar10.yaml")
    conf = trainer.parse_config(path)
    torch.manual_seed(1337)
    module = trainer.build_module(conf)  # CPU-resident parameters (torch default init), never moved to a GPU here
    proc = module.diffusion_model
    unet = proc.model
    c = unet._cfg
    cfg = O.UNetConfig(c.in_channels, c.pos_dim, c.emb_dim, c.num_groups, float(c.dropout), tuple(c.channels_per_depth[: c.num_depths]), c.num_blocks,
                       tuple(c.attention_depths[: c.num_attention_depths]))
    sd = {k: v.detach().clone() for k, v in unet.state_dict().items()}
    T = proc.timesteps
    beta = D.linear_beta(T)
    alpha, abar = D.alpha_tables(beta)
    x = synth.normal(1, (1, 3, 32, 32))
    t0 = time.perf_counter()
    with torch.no_grad():
        for k in range(10):
            t = T - k
            x = D.ddpm_step(x, t, O.unet_forward(sd, cfg, x, torch.tensor([t])), synth.normal(10 + k, x.shape), beta, alpha, abar)
    dt_sample = time.perf_counter() - t0
    params = {k: v.requires_grad_(True) for k, v in sd.items() if k != "condition.0.embeddings"}
    sdp = dict(sd)
    sdp.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=module.lr)
    losses = []
    t0 = time.perf_counter()
    for k in range(2):
        x0 = synth.uniform(20 + k, (1, 3, 32, 32))
        tt = synth.randint(30 + k, 1, T, 1)
        masks = O.make_drop_masks(cfg, 1, 40 + k) if cfg.dropout > 0 else None
        loss = D.training_loss(lambda xx, t_: O.unet_forward(sdp, cfg, xx, t_, drop_masks=masks), x0, tt, synth.normal(50 + k, x0.shape), abar)
        opt.zero_grad()
        loss.backward()
        if conf["gradient_clip_val"]:
            torch.nn.utils.clip_grad_norm_(list(params.values()), conf["gradient_clip_val"])
        opt.step()
        losses.append(float(loss.detach()))
    dt_train = time.perf_counter() - t0
    ok = bool(torch.isfinite(x).all()) and all(v == v for v in losses)
    print(json.dumps({
        "metric": METRIC, "value": round(10 / dt_sample, 3), "unit": "denoising steps/s at batch 1 on the CPU reference path (oracle)", "n_gpus": 0,
        "steps": 10, "warmup": 0, "ms_per_step": round(1e3 * dt_sample / 10, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[0]: {os.path.relpath(path, ROOT) if path.startswith(ROOT) else path} -> {type(module).__name__} / "
                               f"UNet {sum(v.numel() for v in sd.values())} values, T={T}, batch 1, CPU oracle: 10 sampling steps + 2 training steps"},
        "train_ms_per_step": round(1e3 * dt_train / 2, 1), "train_losses": [round(v, 5) for v in losses], "finite": ok, "cores": threads,
        "host_cpus_visible": avail}), flush=True)
    return 0 if ok else 5


# ---------------------------------------------------------------------------------------------- workloads
def workload(dmme_amd, name, precision):
    """(UNet, image side, T, process class, Lit class, label) of a --model choice"""
    if name == "iddpm64":
        from dmme_amd.models import iddpm

        return (iddpm.UNet(attention_depths=(3, 4), precision=precision), 64, 4000, dmme_amd.IDDPM, dmme_amd.LitIDDPM,
                "IDDPM ImageNet-64 UNet (36,562,822 params, 4-head attention at 16x16/8x8, dropout 0.3), cosine schedule, T=4000")
    return dmme_amd.UNet(precision=precision), 32, 1000, dmme_amd.DDPM, dmme_amd.LitDDPM, "default UNet (32,416,643 params, random init)"


def _fence(dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(dt, dist, dev):
    if dist is None:
        return dt
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def train_leg(dmme_amd, dev, B, precision, steps, warmup, dist, model_name="ddpm", reduce=True, seed=1337, exchange=None):
    """training images/s: q_sample -> UNet fwd (train mode, Dropout2d on) -> loss -> HIP backward -> (RCCL mean all-reduce of
    the flat gradient, first bucket overlapped with the rest of backward) -> fused clip(1.0)+Adam+EMA -> warm-up LR step.
    Identical initial weights on every rank (same seed), per-rank noise / timesteps / masks afterwards."""
    from dmme_amd import distributed as DD
    from dmme_amd.train_loop import synthetic_batch, train_step

    rank = dist.get_rank() if dist is not None else 0
    torch.manual_seed(seed)
    net, side, T, _, lit_cls, _ = workload(dmme_amd, model_name, precision)
    lit = lit_cls(model=net, timesteps=T).to(dev)
    lit.train()
    opts, scheds = lit.configure_optimizers()
    opt, sched = opts[0], scheds[0]["scheduler"]
    for g in opt.param_groups:
        g["max_grad_norm"] = 1.0
    torch.manual_seed(DD.rank_seed(seed, rank))
    x0 = synthetic_batch(B, dev, (3, side, side))
    loss = None
    for _ in range(warmup):
        loss = train_step(lit, opt, sched, x0, reduce=reduce, exchange=exchange)
    _fence(dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(lit, opt, sched, x0, reduce=reduce, exchange=exchange)
    _fence(dist)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, dev)
    assert torch.isfinite(loss).all(), "non-finite training loss"
    numel = net.flat_parameters().numel()
    del lit, opt, sched, net
    torch.cuda.empty_cache()
    return dt, float(loss.detach()), numel


def allreduce_alone_ms(dist, dev, numel, reps=10):
    """the step's one collective by itself: mean all-reduce of a flat fp32 gradient buffer, sliced as the training step slices it"""
    from dmme_amd import distributed as DD

    buf = torch.zeros(numel, dtype=torch.float32, device=dev)
    for _ in range(2):
        DD.allreduce_mean_flat(buf)
    _fence(dist)
    t0 = time.perf_counter()
    for _ in range(reps):
        DD.allreduce_mean_flat(buf)
    _fence(dist)
    return 1e3 * _max_over_ranks(time.perf_counter() - t0, dist, dev) / reps


def chain_leg(proc, runner, n_steps, warmup, steps, dist, dev, reps=1):
    """W untimed + `reps` x (K timed replays of the denoising step, each block bracketed by barrier + synchronize), loop index wrapping
    to the top of the chain when it runs out.  Returns the list of per-block times (max over ranks each)."""
    from dmme_amd.common.noise import philox_reserve

    left = 0

    def one():
        nonlocal left
        if left == 0:
            seed, off = philox_reserve(dev, runner.x.numel() * n_steps)
            runner.set(n_steps, seed, off)
            left = n_steps
        runner.step()
        left -= 1

    times = []
    with torch.no_grad():
        for _ in range(warmup):
            one()
        for _ in range(reps):
            _fence(dist)
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            _fence(dist)
            times.append(_max_over_ranks(time.perf_counter() - t0, dist, dev))
    runner.plan.check()  # a level-engine hand-off that gave up anywhere in the warm-up or the timed region voids the run (DmmeError)
    return times


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def mfma_calibration(dev):
    """dense bf16 MFMA rate of THIS box under a sustained all-CU loop (dmme_debug_mfma_valu, MFMA waves only: 256 workgroups x 4
    waves x 8 independent 32x32x16 MFMAs per iteration, ~1.5 ms): the pool's boards are power-managed and differ by a few percent,
    this figure says which kind the line was measured on"""
    from dmme_amd import _lib

    lib = _lib.lib()
    sink = torch.zeros(4096, dtype=torch.float32, device=dev)
    iters, blocks = 10000, 256
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for r in range(4):
        e0.record()
        _lib.check(lib.dmme_debug_mfma_valu(1, iters, blocks, _lib.ptr(sink), _lib.stream_ptr()), "mfma_valu")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if r and (best is None or ms < best):
            best = ms
    return round(blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16 / (best * 1e-3) / 1e12, 1)


def main():
    args = parse()
    if args.mode == "cpu-plumbing":
        return cpu_plumbing(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist

        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or plain `python bench.py --gpus N`)")
        # rehearsal on a box with fewer GPUs than ranks: DMME_DIST_BACKEND=gloo lets the ranks share devices (RCCL refuses that)
        backend = os.environ.get("DMME_DIST_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        ones = torch.ones(1, device=torch.device("cuda", local))
        dist.all_reduce(ones)  # the first collective: how many ranks the backend really connected
        ranks_seen = int(round(float(ones.item())))
        assert ranks_seen == dist.get_world_size() == world, f"backend connected {ranks_seen} ranks, expected {world}"
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)

    import dmme_amd
    from dmme_amd import distributed as DD

    B = args.batch
    gflop = FWD_GFLOP_PER_IMAGE[args.model]
    peak = PEAK.get(args.precision, 2500.0)
    if args.mode == "train":
        dt, loss, _ = train_leg(dmme_amd, dev, B, args.precision, args.steps, args.warmup, dist, args.model)
        if rank == 0:
            tf = world * args.steps * B * 3 * gflop / dt / 1e3
            print(json.dumps({
                "metric": METRIC, "value": round(world * args.steps * B / dt, 2),
                "unit": "training images/s (q_sample + UNet fwd/bwd + grad all-reduce + clip + Adam + EMA), summed over GPUs",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": (f"DDPM CIFAR10 32x32 training step, default UNet, batch {B} per GPU, dropout 0.1" if args.model == "ddpm" else
                                        f"IDDPM ImageNet-64 hybrid-loss training step at 64x64, batch {B} per GPU"), "global_batch": B * world,
                           "parallelism": f"dp{world} (RCCL mean all-reduce of the flat fp32 gradient)"}, "final_loss": round(loss, 5),
                "ranks_seen": ranks_seen, "step_tflops": round(tf, 1), "step_frac_of_peak": round(tf / (world * peak), 4)}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    torch.manual_seed(1337)  # same weights on every rank ...
    model, side, T, proc_cls, _, label = workload(dmme_amd, args.model, args.precision)
    model = model.to(dev).eval()
    torch.manual_seed(DD.rank_seed(1337, rank))  # ... each rank's own chains
    if args.mode == "ddim":
        proc = dmme_amd.DDIM(model, T, 50).to(dev)
        n_steps = 50
    else:
        proc = proc_cls(model, T).to(dev)
        n_steps = T
    x = dmme_amd.gaussian((B, 3, side, side), device=dev)
    runner = proc.chain_runner(x, use_graph=not args.no_graph)
    assert runner is not None, "the replayable denoising step does not apply to this configuration"
    times = chain_leg(proc, runner, n_steps, args.warmup, args.steps, dist, dev, reps=args.reps)
    assert torch.isfinite(x).all(), "non-finite samples"
    elapsed = _median(times)  # K steps; the line also carries the fastest block and every block's time

    step_tf = world * args.steps * B * gflop / elapsed / 1e3
    update = "DDIM" if args.mode == "ddim" else "IDDPM learned-variance" if args.model == "iddpm64" else "DDPM"
    out = {
        "metric": METRIC,
        "value": round(world * args.steps / elapsed, 3),
        "unit": f"denoising steps/s (one step = UNet forward + {update} update on a batch of {B}), summed over GPUs",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "timed_blocks": {"reps": len(times), "steps_each": args.steps, "ms_per_step_each": [round(1e3 * v / args.steps, 4) for v in times],
                         "ms_per_step_min": round(1e3 * min(times) / args.steps, 4), "value_is": "median block"},
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
        # whole-step arithmetic rate: B x algorithmic forward FLOPs per step over the measured step time - the
        # figure the dominant kernel's roofline fraction must not be mistaken for
        "step_tflops": round(step_tf, 1),
        "step_frac_of_peak": round(step_tf / (world * peak), 4),
        "ranks_seen": ranks_seen,
        "train_images_per_s": None,
        "launches_per_step": int(runner.plan.lib.dmme_unet_plan_num_launches(runner.plan.h)) + 1,
        "hip_graph": bool(runner.graph is not None),
        "box_mfma_tfps": mfma_calibration(dev),
    }

    plan = runner.plan

    def rank0_legs():  # the per-kernel roofline of the measured forward and the CPU baseline
        if rank == 0:
            if not args.no_roofline:
                xin = dmme_amd.gaussian((B, 3, side, side), device=dev)
                model._last_plan = plan
                out["roofline"] = roofline_leg(model, xin, proc.timestep_tensor(min(500, T), dev), args.precision)
                del xin
            if world == 1 and not args.no_cpu_baseline and args.model == "ddpm":
                out["cpu_baseline"] = cpu_baseline_leg(B)

    def accurate_leg():
        """the same sampling step in the modes that are closer to the reference than bf16: precision="fp16r32" (`accurate_mode`:
        fp16 with the full-resolution level in fp32 tensors and three-pass split-fp16 products - max|err| 6-7e-4, INSIDE north_star's 1e-3),
        precision="bf16x3" (fp32 tensors everywhere, three bf16 passes - 1.7e-5) and precision="fp16" (IEEE half, the reference's own AMP
        dtype, same kernels and rate as bf16 - 1.5e-3)"""
        if args.no_accurate_leg or args.precision != "bf16" or args.model != "ddpm" or args.mode != "sample":
            return
        for key, prec in (("accurate_mode", "fp16r32"), ("bf16x3_mode", "bf16x3"), ("fp16_mode", "fp16")):
            if prec not in dmme_amd._lib.DTYPES:
                continue
            try:
                torch.manual_seed(1337)
                m3 = dmme_amd.UNet(precision=prec).to(dev).eval()
                p3 = dmme_amd.DDPM(m3, T).to(dev)
                x3 = dmme_amd.gaussian((B, 3, side, side), device=dev)
                r3 = p3.chain_runner(x3, use_graph=not args.no_graph)
                k = max(5, min(20, args.steps))
                dt3 = _median(chain_leg(p3, r3, T, 3, k, dist, dev, reps=2))
                out[key] = {"precision": prec, "steps_per_s": round(world * k / dt3, 3), "ms_per_step": round(1e3 * dt3 / k, 3),
                            "steps": k, "step_tflops_algorithmic": round(world * k * B * gflop / dt3 / 1e3, 1),
                            "max_abs_err_vs_reference": {"fp16r32": "6.0e-4 at batch 2, 7.2e-4 at batch 128 (rel-RMS 3.6e-4): inside north_star's 1e-3", "bf16x3": "1.7e-5",
                                                         "fp16": "1.5e-3 (rel-RMS 9.8e-4)"}[prec] + " (tests/test_gpu_fp16.py, test_gpu_x3.py)"}
                del m3, p3, x3, r3
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001 - a secondary figure must not cost the headline line
                out[key] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
        # north_star asks "within 1e-3" of the reduced-precision path: `value` above is timed in bf16 (max |err| 1.2e-2 against the
        # reference); this top-level key is the same step in the fastest mode INSIDE that tolerance, so that the
        # in-tolerance figure cannot be overlooked
        am = out.get("accurate_mode") or {}
        if "steps_per_s" in am:
            out["value_within_north_star_tolerance"] = {"value": am["steps_per_s"], "unit": out["unit"], "precision": am["precision"], "ms_per_step": am["ms_per_step"],
                                                        "max_abs_err_vs_reference": "6.0e-4 at batch 2, 7.2e-4 at batch 128 (bound 1e-3, tests/test_gpu_fp16.py)",
                                                        "value_precision_max_abs_err": "bf16: 1.2e-2 (bound 1.7e-2, tests/test_gpu_unet.py)"}

    def ddim_leg():
        """BASELINE configs[2] as a secondary key of the N = 1 line: DDIM, 50-step quadratic tau, batch 512, the full chain once"""
        if args.no_ddim_leg or world != 1 or args.precision != "bf16" or args.model != "ddpm" or args.mode != "sample":
            return
        try:
            torch.manual_seed(1337)
            md = dmme_amd.UNet(precision="bf16").to(dev).eval()
            pd = dmme_amd.DDIM(md, T, 50).to(dev)
            xd = dmme_amd.gaussian((512, 3, side, side), device=dev)
            rd = pd.chain_runner(xd, use_graph=not args.no_graph)
            dtd = chain_leg(pd, rd, 50, 5, 50, dist, dev)[0]
            out["ddim_b512"] = {"config": "DDIM 50-step (quadratic tau), batch 512, bf16", "steps_per_s": round(50 / dtd, 3), "ms_per_step": round(1e3 * dtd / 50, 3),
                                "image_steps_per_s": round(50 * 512 / dtd, 1), "chain_seconds": round(dtd, 4),
                                "step_tflops": round(50 * 512 * gflop / dtd / 1e3, 1)}
            del md, pd, xd, rd
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            out["ddim_b512"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    def small_batch_leg():
        """the same captured step at batch 1 and batch 32 (the launch-bound end of `DDPM.generate`), secondary keys of the N = 1 line"""
        if args.no_small_batch_leg or world != 1 or args.precision != "bf16" or args.model != "ddpm" or args.mode != "sample" or B != 128:
            return
        out["small_batch"] = {}
        for bs in (1, 32):
            try:
                torch.manual_seed(1337)
                ms_ = dmme_amd.UNet(precision="bf16").to(dev).eval()
                ps = dmme_amd.DDPM(ms_, T).to(dev)
                xs = dmme_amd.gaussian((bs, 3, side, side), device=dev)
                rs = ps.chain_runner(xs, use_graph=not args.no_graph)
                ts = chain_leg(ps, rs, T, 30, 200, dist, dev, reps=3)
                out["small_batch"][f"b{bs}"] = {"steps_per_s": round(200 / _median(ts), 1), "ms_per_step": round(1e3 * _median(ts) / 200, 4),
                                                "ms_per_step_min": round(1e3 * min(ts) / 200, 4),
                                                "launches_per_step": int(rs.plan.lib.dmme_unet_plan_num_launches(rs.plan.h)) + 1}
                del ms_, ps, xs, rs
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001
                out["small_batch"][f"b{bs}"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}

    # several ranks: rank 0's legs first, so that a stuck collective in the training leg cannot cost them; one rank: after it (no
    # collective to get stuck in, and the event-bracketed kernel times sit closer to rocprofv3's with the device in its training-leg state)
    if world > 1 or args.train_steps <= 0:
        rank0_legs()
    accurate_leg()
    ddim_leg()
    small_batch_leg()
    rc = 0
    if args.train_steps > 0:
        del x, runner
        proc._runner = None
        # Secondary figures.  Neither an exception nor a stuck collective in them may cost the headline line: past the deadline
        # every rank leaves through the watchdog - rank 0 printing the line it already has - with a NON-zero status, so the launcher
        # and CI see the hang.
        import threading

        def bail():
            if rank == 0:
                out["train_error"] = "training leg exceeded its deadline (stuck collective?)"
                print(json.dumps(out), flush=True)
            os._exit(3)

        watchdog = threading.Timer(300.0, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            k = args.train_steps
            dt_tr, _, numel = train_leg(dmme_amd, dev, B, args.precision, k, 3, dist, args.model)
            out["train_images_per_s"] = round(world * k * B / dt_tr, 1)
            out["train_ms_per_step"] = round(1e3 * dt_tr / k, 2)
            out["train_steps"] = k
            ttf = world * k * B * 3 * gflop / dt_tr / 1e3
            out["train_step_tflops"] = round(ttf, 1)
            out["train_step_frac_of_peak"] = round(ttf / (world * peak), 4)
            if world > 1:
                n_buckets, bucket_bytes = 0, []
                try:  # the gradient buckets the plan's backward hands to the exchange, in hand-over order
                    import ctypes as C_

                    mb = dmme_amd.UNet(precision=args.precision).to(dev) if args.model == "ddpm" else None
                    if mb is not None:
                        pl = mb._plan_for(B, side, side, dev)
                        bks, offs, nums = (C_.c_int * 64)(), (C_.c_int64 * 64)(), (C_.c_int64 * 64)()
                        cnt = pl.lib.dmme_unet_plan_grad_buckets(pl.h, offs, nums, bks, 64)
                        n_buckets = max(bks[i] for i in range(min(cnt, 64))) + 1
                        bucket_bytes = [4 * sum(int(nums[i]) for i in range(min(cnt, 64)) if bks[i] == b) for b in range(n_buckets)]
                        del mb, pl
                except Exception:  # noqa: BLE001
                    n_buckets, bucket_bytes = 0, []
                # the same step without its collective, and the collective alone: what the overlap hides
                dt_nc, _, _ = train_leg(dmme_amd, dev, B, args.precision, k, 3, dist, args.model, reduce=False)
                ar_ms = allreduce_alone_ms(dist, dev, numel)
                step_ms, nocomm_ms = 1e3 * dt_tr / k, 1e3 * dt_nc / k
                exposed = max(0.0, step_ms - nocomm_ms)
                out["train_dp"] = {
                    "semantics": f"per-rank batch {B} (the reference under Lightning DDP: YAML batch_size is per rank), global batch {B * world}",
                    "ms_per_step": round(step_ms, 3), "ms_per_step_without_allreduce": round(nocomm_ms, 3),
                    "allreduce_alone_ms": round(ar_ms, 3), "allreduce_exposed_ms": round(exposed, 3),
                    "allreduce_hidden_ms": round(max(0.0, ar_ms - exposed), 3), "gradient_bytes": numel * 4,
                    "exchange": DD.default_exchange(B), "gradient_buckets": n_buckets, "bucket_bytes": bucket_bytes,
                    "ranks_seen": out.get("ranks_seen")}
                # the other wire format of the gradient mean (distributed.Bf16ShardExchange: bf16 all-to-all + all-gather, fp32 accumulation)
                try:
                    if os.environ.get("DMME_BENCH_FAIL_BF16_LEG"):  # (test hook: the status of a run whose exchange leg fails)
                        raise RuntimeError("DMME_BENCH_FAIL_BF16_LEG")
                    dt_bf, _, _ = train_leg(dmme_amd, dev, B, args.precision, k, 3, dist, args.model, exchange="bf16-rs-ag")
                    out["train_dp"]["ms_per_step_bf16_rs_ag"] = round(1e3 * dt_bf / k, 3)
                except Exception as exc:  # noqa: BLE001  (the line still goes out, the status says the exchange failed: VERDICT round 4)
                    out["train_dp"]["bf16_rs_ag_error"] = f"{type(exc).__name__}: {exc}"[:200]
                    rc = 4
                if B % world == 0 and B // world >= 1:
                    bs = B // world  # north_star wording: the batch of 128 sharded over the ranks
                    dt_g, _, _ = train_leg(dmme_amd, dev, bs, args.precision, k, 3, dist, args.model)
                    out["train_global_batch"] = {"semantics": f"global batch {B} sharded: {bs} per rank (strong scaling of one batch-{B} step)",
                                                 "images_per_s": round(k * B / dt_g, 1), "ms_per_step": round(1e3 * dt_g / k, 3)}
        except Exception as exc:  # noqa: BLE001
            out["train_error"] = f"{type(exc).__name__}: {exc}"[:300]
            if world > 1:  # a failing data-parallel leg is a failed run: the line is printed, the status is not 0
                rc = 4
        watchdog.cancel()
        if world == 1:
            rank0_legs()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
