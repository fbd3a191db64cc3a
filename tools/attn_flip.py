#!/usr/bin/env python3
"""One process, one box: forwards with the whole-row attention kernel and with the online-softmax kernel, alternating (the switch is read
per launch).  usage: python tools/attn_flip.py [extra idle us per forward]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dmme_amd
m = dmme_amd.UNet(precision="bf16").cuda().eval()
x = dmme_amd.gaussian((128, 3, 32, 32), device="cuda"); t = torch.tensor([500], device="cuda")
def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m(x, t)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    for _ in range(20): m(x, t)
    for rnd in range(4):
        os.environ.pop("DMME_NO_ATTN_FULL", None)
        a = run(100)
        os.environ["DMME_NO_ATTN_FULL"] = "1"
        b = run(100)
        os.environ.pop("DMME_NO_ATTN_FULL", None)
        c = []
        for lvl in ("1", "2", "3", "4"):
            os.environ["DMME_DEBUG_ROUTE"] = "attn_sleep=" + lvl
            c.append(run(100))
        os.environ.pop("DMME_DEBUG_ROUTE", None)
        print(f"round {rnd}: whole-row kernel {a:.4f} ms / forward, online kernel {b:.4f}, whole-row with s_sleep 2 / 4 / 8 / 16 per tile: " + " ".join(f"{v:.4f}" for v in c))
