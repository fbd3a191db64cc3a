"""Soak check (run on the GPU box): two full 1000-step DDPM chains at batch 128 through the captured step - finite, bit-identical under
the same seed, and the level engine's status word clean afterwards.  usage: python tools/soak_chain.py"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import dmme_amd
torch.manual_seed(0)
net = dmme_amd.UNet(precision="bf16").cuda().eval()
ddpm = dmme_amd.DDPM(net, timesteps=1000).cuda()
outs = []
for rep in range(2):
    torch.manual_seed(123)
    t0 = time.time()
    x = ddpm.generate((128, 3, 32, 32)) if hasattr(ddpm, "generate") else None
    torch.cuda.synchronize()
    print("chain", rep, "seconds", round(time.time() - t0, 2), "finite", bool(torch.isfinite(x).all()), "absmax", float(x.abs().max()))
    outs.append(x.clone())
print("bit-identical repeat:", torch.equal(outs[0], outs[1]))
net.check_engine()
print("engine ok")
