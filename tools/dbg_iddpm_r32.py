import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth, iddpm as OI
from tests.test_gpu_iddpm import _build
torch.set_num_threads(16)
for name, cfg, side, B, t in (("default32", OI.IUNetConfig(), 32, 2, [900]), ("imagenet64", OI.IUNetConfig(attention_depths=(3, 4)), 64, 2, [17, 3011])):
    try:
        net, sd = _build(cfg, 43, "fp16r32")
        x = synth.normal(10, (B, 3, side, side)); tt = torch.tensor(t)
        want = OI.unet_forward(sd, cfg, x, tt)
        with torch.no_grad(): got = net(x.cuda(), tt.cuda()).cpu()
        e = (got - want).abs()
        print(name, "fp16r32: max|err|", float(e.max()), "rel-rms", float(e.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()), "max|want|", float(want.abs().max()))
        net2, _ = _build(cfg, 43, "fp16")
        with torch.no_grad(): g2 = net2(x.cuda(), tt.cuda()).cpu()
        e2 = (g2 - want).abs()
        print(name, "fp16   : max|err|", float(e2.max()), "rel-rms", float(e2.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()))
    except Exception as exc:
        print(name, "FAILED:", type(exc).__name__, str(exc)[:600])
