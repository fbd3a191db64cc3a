"""Error-budget study for the half-precision inference mode (test infrastructure; not collected by pytest).

Emulates on the CPU, with the oracle's functional UNet walk (oracle/unet.py), WHICH roundings of a 16-bit forward pass cost
how much of north_star's 1e-3: every matrix product runs in fp32 on operands rounded (or not) to the 16-bit type, which is what an
MFMA with fp32 accumulation computes up to summation order.  Switches:

  w   conv / linear weights rounded            a   matrix operands (GN+SiLU outputs, raw 1x1 inputs, q/k/v/p) rounded
  s   residual stream rounded when stored      h   the tensor between conv1 and conv2 of a block rounded when stored
      (block outputs, skips, up/down outputs)  q   qkv / attention context tensors rounded when stored
  W2  weights as hi + lo (two passes)          A2  operands as hi + lo (two passes)

Run:  python tests/study_precision_budget.py            (prints one line per variant, against the reference's golden output)
"""

import itertools
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth  # noqa: E402
from oracle import unet as O  # noqa: E402

DT = torch.float16


def rnd(x, on, dt=None):
    return x.to(dt or DT).float() if on else x


class Emu:
    def __init__(self, sd, cfg, flags, dt=DT):
        self.sd, self.cfg, self.f, self.dt = sd, cfg, set(flags), dt

    def R(self, x, flag):
        return rnd(x, flag in self.f, self.dt)

    def W(self, w):
        if "W2" in self.f:
            hi = w.to(self.dt).float()
            return hi + (w - hi).to(self.dt).float()
        return self.R(w, "w")

    def A(self, a):
        if "A2" in self.f:
            hi = a.to(self.dt).float()
            return hi + (a - hi).to(self.dt).float()
        return self.R(a, "a")

    def conv(self, a, key, **kw):
        return F.conv2d(self.A(a), self.W(self.sd[key + ".weight"]), self.sd[key + ".bias"], **kw)

    def gn(self, p, x):
        return F.group_norm(x, self.cfg.num_groups, self.sd[p + ".weight"], self.sd[p + ".bias"], eps=1e-5)

    def attn(self, p, x):
        b, c, hh, ww = x.shape
        qkv = self.R(self.conv(self.gn(p + ".norm", x), p + ".qkv_proj"), "q")
        qkv = qkv.reshape(b, 3 * c, hh * ww).transpose(1, 2)
        q, k, v = qkv[:, :, :c], qkv[:, :, c : 2 * c], qkv[:, :, 2 * c :]
        s = torch.bmm(self.A(q), self.A(k).transpose(1, 2)) * (c**-0.5)  # the MFMA kernels scale the fp32 scores
        w = torch.softmax(s, dim=2)
        o = torch.bmm(self.A(w), self.A(v)).transpose(1, 2).reshape(b, c, hh, ww)
        o = self.R(o, "q")
        return self.conv(o, p + ".proj") + x

    def res(self, n, x, temb):
        p = n.prefix
        h = self.conv(F.silu(self.gn(p + ".conv1.0", x)), p + ".conv1.2", padding=1)
        h = h + F.linear(temb, self.sd[p + ".condition.0.weight"], self.sd[p + ".condition.0.bias"])[:, :, None, None]
        h = self.R(h, "h")
        ck = f"{p}.conv2.{O._conv2_index(self.cfg)}"
        h2 = self.conv(F.silu(self.gn(p + ".conv2.0", h)), ck, padding=1)
        if n.c_in != n.c_out:
            h2 = h2 + self.conv(x, p + ".residual")
        else:
            h2 = h2 + x
        h2 = self.R(h2, "s")
        if n.attn:
            h2 = self.R(self.attn(p + ".attention", h2), "s")
        return h2

    def forward(self, x, t):
        g = O.build_graph(self.cfg)
        temb = O.time_embedding(self.sd, t)
        h = self.R(F.conv2d(x, self.sd["input_conv.weight"], self.sd["input_conv.bias"], padding=1), "s")
        skips = [h]
        for n in g.down:
            if n.kind == "res":
                h = self.res(n, h, temb)
            else:
                h = self.R(self.conv(h, n.prefix, stride=2, padding=1), "s")
            skips.append(h)
        for n in g.mid:
            h = self.res(n, h, temb)
        for n in g.up:
            if n.kind == "res":
                h = self.res(n, torch.cat([h, skips.pop()], dim=1), temb)
            else:
                h = self.R(self.conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), n.prefix + ".conv", padding=1), "s")
        return self.conv(F.silu(self.gn("output_conv.0", h)), "output_conv.2", padding=1)


def main():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_full.npz"))
    cfg = O.UNetConfig()
    sd = O.make_state_dict(cfg, int(g["full_seed"]))
    x = synth.normal(int(g["full_xseed"]), (2, 3, 32, 32))
    ref_one, ref_per = torch.from_numpy(g["full_y_one"]), torch.from_numpy(g["full_y_per"])
    t_one, t_per = torch.from_numpy(g["full_t_one"]), torch.from_numpy(g["full_t_per"])
    variants = [
        ("fp32 walk (sanity)", ""),
        ("all rounded (today's fp16)", "w a s h q"),
        ("weights only", "w"),
        ("operands only", "a"),
        ("stream only", "s"),
        ("h only", "h"),
        ("qkv/ctx only", "q"),
        ("fp32 stream (w a h q)", "w a h q"),
        ("fp32 stream+h (w a q)", "w a q"),
        ("fp32 stream+h+q (w a)", "w a"),
        ("W2 + a (fp32 tensors)", "W2 a"),
        ("A2 + w (fp32 tensors)", "A2 w"),
        ("W2 + a s h q", "W2 a s h q"),
        ("W2 + a h q", "W2 a h q"),
    ]
    which = sys.argv[1:] or None
    dts = {"fp16": torch.float16, "bf16": torch.bfloat16}
    for dname in (os.environ.get("STUDY_DT", "fp16").split(",")):
        for name, flags in variants:
            if which and not any(w in name for w in which):
                continue
            e = Emu(sd, cfg, flags.split(), dts[dname])
            with torch.no_grad():
                y1, y2 = e.forward(x, t_one), e.forward(x, t_per)
            e1, e2 = (y1 - ref_one).abs(), (y2 - ref_per).abs()
            print(f"{dname} {name:34s} max|err| {float(max(e1.max(), e2.max())):.3e}  rel-rms {float(e1.pow(2).mean().sqrt() / ref_one.pow(2).mean().sqrt()):.3e}", flush=True)


if __name__ == "__main__":
    main()
