#!/usr/bin/env python3
"""Phase breakdown of one pipelined-conv launch from in-kernel cycle stamps (dmme_debug_set_stamps).
usage: python tools/stamp_conv.py 32x128x128 [gn]"""
import ctypes as C, os, sys
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
stamps = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
_lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
run()
torch.cuda.synchronize()
_lib.check(lib.dmme_debug_set_stamps(None))
s = stamps.cpu().view(8, 64)
for slot in range(8):
    v = [int(t) for t in s[slot] if int(t) != 0]
    if not v: continue
    d0 = [v[i + 1] - v[i] for i in range(len(v) - 1)]
    print(f"slot {slot}: total {v[-1] - v[0]} cyc; start offset vs slot0 {v[0] - int(s[0][0])}")
    if slot > 0 and len(sys.argv) > 3: continue
    print("   prologue", d0[0])
    body = d0[1:-1]
    for i in range(0, len(body), 4):
        print("   interval", i // 4, "mfma", body[i], "bar1", body[i + 1] if i + 1 < len(body) else None, "store", body[i + 2] if i + 2 < len(body) else None,
              "bar2", body[i + 3] if i + 3 < len(body) else None)
    print("   epilogue", d0[-1])
