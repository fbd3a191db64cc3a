import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dmme_amd
from oracle import unet as O, synth
g = np.load("tests/golden/train_tiny.npz")
seed, T, B, sx, st, sz, sm = [int(v) for v in g["train_meta"]]
cfg = O.TINY
net = dmme_amd.UNet(cfg.in_channels, cfg.pos_dim, cfg.emb_dim, cfg.num_groups, cfg.dropout, cfg.channels_per_depth, cfg.num_blocks, cfg.attention_depths)
net.load_state_dict(O.make_state_dict(cfg, seed)); net.cuda().eval()
ddpm = dmme_amd.DDPM(net, T).cuda()
loss = ddpm.training_step(synth.uniform(sx, (B,3,32,32)).cuda(), t=synth.randint(st,1,T,B).cuda(), noise=synth.normal(sz,(B,3,32,32)).cuda())
loss.backward()
names = [n for n,_ in net.named_parameters()]
order = ["output_conv", "up_layers.14", "up_layers.13", "up_layers.12", "up_layers.11", "up_layers.10", "up_layers.9", "middle_layers.1", "middle_layers.0", "down_layers.14", "down_layers.0", "input_conv", "condition"]
for pre in order:
    for n in names:
        if n.startswith(pre + "."):
            want = g[f"train_eval_grad::{n}"]; got = dict(net.named_parameters())[n].grad.cpu().numpy()
            print(f"{n:45s} err {np.abs(got-want).max():.3e} ref {np.abs(want).max():.3e} ratio {np.abs(got).max()/max(np.abs(want).max(),1e-30):.3f}")
