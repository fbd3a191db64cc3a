#!/usr/bin/env python3
"""Cycle stamps of wave 0 / workgroup 0 of the attention block's launch (attn_full_kernel<.., PROJ>); needs a library whose attn_mfma.hip was
built with -DAT_STAMPS (make FLAGS_attn_mfma=-DAT_STAMPS; DMME_LIB_PATH selects it).  usage: python tools/stamp_attn.py [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmme_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, S = 128, 256
dev = torch.device("cuda:0"); lib = _lib.lib(); dt = _lib.BF16
qkv = torch.randn(N, S, 3 * C, device=dev).bfloat16()
w = (torch.randn(C, C, device=dev) * C**-0.5).bfloat16(); b = torch.randn(C, device=dev)
res = torch.randn(N, S, C, device=dev).bfloat16(); dst = torch.empty_like(res)
part = torch.zeros(N, S // 32, 32, 2, device=dev)
st = _lib.stream_ptr()
run = lambda: _lib.check(lib.dmme_attention_proj(dt, _lib.ptr(qkv), N, S, C, _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), _lib.ptr(dst), None, _lib.ptr(part), C // 32, st))
for _ in range(3): run()
stamps = torch.zeros(128, dtype=torch.int64, device=dev)
_lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
run(); torch.cuda.synchronize()
_lib.check(lib.dmme_debug_set_stamps(None))
v = [int(t) for t in stamps.cpu() if int(t) != 0]
d = [v[i + 1] - v[i] for i in range(len(v) - 1)]
print(f"C={C}: {len(v)} stamps, total {v[-1] - v[0]} cycles")
print("deltas:", d)
