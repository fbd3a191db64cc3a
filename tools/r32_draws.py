#!/usr/bin/env python3
"""precision="fp16r32": max |err| against the CPU oracle over many (input, timestep, weight seed) draws, per set of two-pass convs.
usage: python tools/r32_draws.py <ndraws> <mask> [<mask> ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth, unet as O
n = int(sys.argv[1]); masks = [int(m) for m in sys.argv[2:]]
torch.set_num_threads(16)
cfg = O.UNetConfig()
draws = []
for seed in (11, 23):
    sd = O.make_state_dict(cfg, seed)
    for k in range(n // 2):
        t = 1 + (k * 997 + seed * 31) % 999
        x = synth.normal(1000 * seed + k, (2, 3, 32, 32)); tt = torch.tensor([t, max(1, 1000 - t)])
        draws.append((seed, x, tt, O.unet_forward(sd, cfg, x, tt)))
import dmme_amd
for m in masks:
    os.environ["DMME_DEBUG_ROUTE"] = f"r32_2pass={m}"
    errs = []
    for seed in (11, 23):
        net = dmme_amd.UNet(precision="fp16r32"); net.load_state_dict(O.make_state_dict(cfg, seed), strict=True); net = net.cuda().eval()
        for s, x, tt, want in draws:
            if s != seed: continue
            with torch.no_grad(): got = net(x.cuda(), tt.cuda()).cpu()
            errs.append(float((got - want).abs().max()))
    errs.sort()
    print(f"mask {m}: {len(errs)} draws: max {errs[-1]:.3e}, second {errs[-2]:.3e}, median {errs[len(errs)//2]:.3e}", flush=True)
