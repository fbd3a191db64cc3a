#!/usr/bin/env python3
"""event-bracketed launch times of the attention block's tail: separate launches vs one.  usage: python tools/time_attn_proj.py [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmme_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, S = 128, 256
dev = torch.device("cuda:0"); lib = _lib.lib(); dt = _lib.BF16
qkv = torch.randn(N, S, 3 * C, device=dev).bfloat16()
w = (torch.randn(C, C, device=dev) * C**-0.5).bfloat16(); b = torch.randn(C, device=dev)
res = torch.randn(N, S, C, device=dev).bfloat16(); dst = torch.empty_like(res); ctx = torch.empty_like(res)
part = torch.zeros(N, S // 32, 32, 2, device=dev)
st = _lib.stream_ptr()
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
att = lambda: _lib.check(lib.dmme_attention(dt, _lib.ptr(qkv), N, S, C, _lib.ptr(ctx), 0, st))
fused = lambda: _lib.check(lib.dmme_attention_proj(dt, _lib.ptr(qkv), N, S, C, _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), _lib.ptr(dst), None, _lib.ptr(part), C // 32, st))
fused_ctx = lambda: _lib.check(lib.dmme_attention_proj(dt, _lib.ptr(qkv), N, S, C, _lib.ptr(w), _lib.ptr(b), _lib.ptr(res), _lib.ptr(dst), _lib.ptr(ctx), _lib.ptr(part), C // 32, st))
print(f"C={C} route={os.environ.get('DMME_DEBUG_ROUTE', '')}: attention {t(att):.1f} us; attention+proj {t(fused):.1f} us; with the context tensor {t(fused_ctx):.1f} us")
