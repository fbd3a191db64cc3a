#!/usr/bin/env python3
"""Workgroup timeline of one pipelined-conv launch (start/end cycle, CU, XCC of every workgroup) from the in-kernel
stamps.  usage: python tools/timeline_conv.py 32x128x128 [gn]"""
import ctypes as C, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
shp = sys.argv[1] if len(sys.argv) > 1 else "32x128x128"
gn = len(sys.argv) > 2 and sys.argv[2] == "gn"
hw, cin, cout = (int(v) for v in shp.split("x"))
B = 128
dev = torch.device("cuda:0")
lib = _lib.lib()
x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, 9, cin, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(cout, device=dev)
scale = torch.rand(B, cin, device=dev) + 0.5
shift = torch.randn(B, cin, device=dev) * 0.1
out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.bfloat16)
d = _lib.ConvDesc()
d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = _lib.BF16, B, hw, hw, cin, 0
d.upsample, d.stride, d.taps, d.Cout = 0, 1, 9, cout
d.pro_silu = int(gn)
d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
st = _lib.stream_ptr()
sc, sh = (scale, shift) if gn else (None, None)
def run():
    _lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(sh), None, None, None, None, cout,
                               _lib.ptr(out), st), "conv")
for _ in range(3): run()
stamps = torch.zeros(8 * 64 + 4096 * 4, dtype=torch.int64, device=dev)
_lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
run()
torch.cuda.synchronize()
_lib.check(lib.dmme_debug_set_stamps(None))
rec = stamps.cpu()[8 * 64:].view(4096, 4).tolist()
rec = [(i, r) for i, r in enumerate(rec) if r[1] != 0]
print("workgroups", len(rec))
bycu = collections.defaultdict(list)
for i, (t0, t1, hw_id, xcc) in rec:
    bycu[(xcc & 0xf, (hw_id >> 13) & 0x7, (hw_id >> 8) & 0xf)].append((t0, t1, i))
spans = []
for n, key in enumerate(sorted(bycu)):
    L = sorted(bycu[key])
    base = L[0][0]
    spans.append(max(t1 for _, t1, _ in L) - base)
    if n < 12:
        print(key, [(t0 - base, t1 - base, i) for t0, t1, i in L])
lives = [((r[3] >> 8) & 0xffffff, r[1] - r[0]) for _, r in rec]
print("core clock estimate (MHz): ", sum(c for _, c in lives) / sum(w for w, _ in lives) * 100.0)
w0 = [(r[3] >> 32) & 0x7fffffff for _, r in rec]
w1 = [((r[3] >> 32) & 0x7fffffff) + ((r[3] >> 8) & 0xffffff) for _, r in rec]
print("wall-clock: first start -> last end (us):", (max(w1) - min(w0)) / 100.0, " start spread of first 512 (us):", (sorted(w0)[511] - min(w0)) / 100.0)
print("per-CU span: min", min(spans), "mean", sum(spans) // len(spans), "max", max(spans), " CUs", len(spans))
