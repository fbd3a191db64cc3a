#!/usr/bin/env python3
"""fused attention block vs the separate launches: where do the context tensors differ?  usage: python tools/dbg_attn_proj.py [fp16|bf16] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmme_amd import _lib
from tests import gpu_util as G
from tests.test_gpu_attn_proj import _run
from oracle import synth
dtname = sys.argv[1] if len(sys.argv) > 1 else "fp16"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N, S = 128, 256
dt = _lib.dtype_code(dtname); td = G.TD[dt]; dev = torch.device("cuda:0")
qkv = synth.normal(11, (N, S, 3 * C)); qkv[:, :, :C] *= 2.0
w = synth.normal(12, (C, C)) * C**-0.5; bias = synth.normal(13, (C,)) * 0.1; res = synth.normal(14, (N, S, C))
qkv, w, bias, res = qkv.to(dev), w.to(dev), bias.to(dev), res.to(dev)
dst, ctx, part = _run(dt, qkv, w, bias, res, True, C // 32)
sep = G.attention(dt, qkv, False).to(td)
d = (ctx.float() - sep.float())
nz = d != 0
print("mismatches", int(nz.sum()), "of", d.numel(), "max", float(d.abs().max()))
idx = nz.nonzero()[:20]
for i in idx:
    n, s, c = (int(v) for v in i)
    print(n, s, c, float(ctx[n, s, c]), float(sep[n, s, c]))
print("channels with mismatches:", sorted(set(int(v) for v in nz.nonzero()[:, 2].tolist()))[:64])
print("tokens with mismatches:", sorted(set(int(v) for v in nz.nonzero()[:, 1].tolist()))[:64])
