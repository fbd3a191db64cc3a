"""Level engine (csrc/lvl_engine.hip) against the per-op launch path on the same inputs: outputs, the per-module activations of the
small-map levels, the engine's error word, and step times of both paths.  `python tools/lvl_check.py [B ...]`"""

import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import dmme_amd
from dmme_amd import _lib
from oracle import synth
from oracle import unet as O

MODS = ["down_layers.7", "down_layers.8", "down_layers.9", "down_layers.10", "middle_layers.0", "middle_layers.1", "up_layers.0", "up_layers.1",
        "up_layers.2", "up_layers.3", "up_layers.4", "up_layers.5", "up_layers.6", "up_layers.7"]


def level_info(net):
    buf = C.create_string_buffer(2048)
    _lib.check(net._last_plan.lib.dmme_unet_plan_level_info(net._last_plan.h, buf, 2048), "level_info")
    return buf.value.decode()


def build(precision, no_lvl):
    if no_lvl:
        os.environ["DMME_NO_LVL"] = "1"
    else:
        os.environ.pop("DMME_NO_LVL", None)
    net = dmme_amd.UNet(precision=precision)
    net.load_state_dict(O.make_state_dict(O.UNetConfig(), 5))
    net.cuda().eval()
    return net


def timed(net, x, t, n=30):
    with torch.no_grad():
        for _ in range(5):
            net(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(x, t)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    batches = [int(a) for a in sys.argv[1:]] or [2, 1, 8, 128]
    precision = os.environ.get("LVL_PRECISION", "bf16")
    for B in batches:
        x = synth.normal(1, (B, 3, 32, 32)).cuda()
        t = torch.tensor([321]).cuda()
        outs, acts, ms = {}, {}, {}
        for no_lvl in (1, 0):
            net = build(precision, no_lvl)
            with torch.no_grad():
                y = net(x, t)
                torch.cuda.synchronize()
                outs[no_lvl] = y.cpu()
                acts[no_lvl] = {m: net.debug_activation(m).cpu() for m in MODS}
            info = level_info(net)
            ms[no_lvl] = timed(net, x, t)
            print(f"B={B} no_lvl={no_lvl}: launches={net._last_plan.lib.dmme_unet_plan_num_launches(net._last_plan.h)} {ms[no_lvl]:.3f} ms/forward  {info}", flush=True)
            if not no_lvl:
                print("   after timing:", level_info(net), flush=True)
            del net
        d = (outs[0] - outs[1]).abs()
        print(f"B={B}: engine vs per-op output: max|diff| {float(d.max()):.3e}  (|y|max {float(outs[1].abs().max()):.3f}); speedup {ms[1] / ms[0]:.3f}x", flush=True)
        for m in MODS:
            a, b = acts[0][m], acts[1][m]
            dd = (a - b).abs()
            print(f"   {m:18s} max|diff| {float(dd.max()):.3e}  rel-rms {float(dd.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()):.3e}  nan={bool(torch.isnan(a).any())}")
        if B <= 2:
            want = O.unet_forward(O.make_state_dict(O.UNetConfig(), 5), O.UNetConfig(), x.cpu(), t.cpu())
            for k in (1, 0):
                print(f"   vs oracle no_lvl={k}: max|err| {float((outs[k] - want).abs().max()):.3e}")


if __name__ == "__main__":
    main()
