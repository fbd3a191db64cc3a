#!/usr/bin/env python3
"""debug: 1x1 conv through the C ABI vs torch, error per 64-cout unit and per 32-pixel block"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
B, hw, cin, cout = 2, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 768
mode = sys.argv[3] if len(sys.argv) > 3 else "gn"
dev = torch.device("cuda:0")
lib = _lib.lib()
torch.manual_seed(0)
x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, 1, cin, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(cout, device=dev)
scale = torch.rand(B, cin, device=dev) + 0.5
shift = torch.randn(B, cin, device=dev) * 0.1
out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.bfloat16)
d = _lib.ConvDesc()
d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = _lib.BF16, B, hw, hw, cin, 0
d.upsample, d.stride, d.taps, d.Cout = 0, 1, 1, cout
d.pro_silu = 1 if mode == "silu" else 0
d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = 0
d.force_generic = int(os.environ.get("FG", "0"))
gn = mode != "plain"
_lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(scale if gn else None), _lib.ptr(shift if gn else None), None, None,
                           None, None, cout, _lib.ptr(out), _lib.stream_ptr()), "conv")
torch.cuda.synchronize()
xf = x.float()
if gn:
    xf = xf * scale[:, None, None, :] + shift[:, None, None, :]
    if mode == "silu": xf = torch.nn.functional.silu(xf)
    xf = xf.to(torch.bfloat16).float()
ref = xf.reshape(-1, cin) @ w.float().reshape(cout, cin).t() + b
err = (out.float().reshape(-1, cout) - ref).abs()
print("max err", float(err.max()))
e = err.reshape(-1, 32, cout // 64, 64).amax(dim=(1, 3))
torch.set_printoptions(linewidth=250, precision=2, sci_mode=False)
print(e.cpu())
o = out.float().reshape(-1, cout)
print("out[0,:8]", o[0, :8].cpu().numpy().round(2), "\nref[0,:8]", ref[0, :8].cpu().numpy().round(2))
# does output row 0 (first 64 couts) match some reference row / cout permutation?
for p in (0, 1, 33, 70):
    d2 = ((ref[:128, :64] - o[p, :64][None, :]) ** 2).sum(1)
    print("out px", p, "closest ref px", int(d2.argmin()), float(d2.min()))
for c in (0, 1, 9, 40):
    d2 = ((ref[:128, :64] - o[:128, c][:, None]) ** 2).sum(0)
    print("out cout", c, "closest ref cout", int(d2.argmin()), float(d2.min()))
nb = (xf.reshape(-1, cin) @ w.float().reshape(cout, cin).t())
print("err without bias in ref:", float((o - nb).abs().max()))
