"""which backward kernel breaks in fp16?  per-class gradient error of the library's fp16 step against its own fp32 step, under A/B switches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth, unet as O
import dmme_amd
from dmme_amd.optim import FusedAdam
from tests.test_gpu_grad_b128 import _classes, _class_errors
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else "fp16"
cfg = O.UNetConfig(); sd = O.make_state_dict(cfg, 23)
x0 = synth.uniform(1, (2, 3, 32, 32)).repeat(B // 2, 1, 1, 1).cuda(); t = torch.tensor([137, 862]).repeat(B // 2).cuda(); z = synth.normal(2, (2, 3, 32, 32)).repeat(B // 2, 1, 1, 1).cuda()
def run(p, amp=True):
    net = dmme_amd.UNet(precision=p, dropout=0.0); 
    sd2 = {k.replace("conv2.3", "conv2.2"): v for k, v in sd.items()}
    net.load_state_dict(sd2); net.cuda().train()
    opt = FusedAdam(net.parameters(), amp=amp)
    ddpm = dmme_amd.DDPM(net, 1000).cuda()
    loss = ddpm.training_step(x0, t=t, noise=z); loss.backward(); torch.cuda.synchronize()
    S = opt.loss_scale()
    return net, float(loss), {k: (p_.grad.detach() / S).cpu().clone() for k, p_ in net.named_parameters()}
net32, l32, g32 = run("fp32")
net, l16, g16 = run(prec)
cl = _classes(net)
print("loss", l32, l16, "env", {k: v for k, v in os.environ.items() if k.startswith("DMME_")})
print({c: f"{v[0]:.2e} ({v[1]})" for c, v in _class_errors(g16, g32, cl).items()})
rows = sorted(((float((g16[k].double() - g32[k].double()).norm() / (g32[k].double().norm() + 1e-30)), k) for k in g32), reverse=True)
print("worst:", rows[:12]); print("best:", rows[-6:])
order = [k for k in g32 if k.endswith("conv2.2.bias") or k.endswith(".proj.bias") or k.endswith("qkv_proj.bias") or k.endswith("conv1.2.bias") or k in ("output_conv.2.bias", "input_conv.bias") or (k.count(".") == 2 and k.endswith(".bias"))]
print("in module order (bias gradients = column sums of dY at that conv's output):")
for k in order:
    a, b = g16[k].double(), g32[k].double()
    print(f"  {k:42s} rel {float((a-b).norm()/(b.norm()+1e-30)):.2e}  |g16| {float(a.norm()):.3e} |g32| {float(b.norm()):.3e}")
