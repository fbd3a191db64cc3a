#!/usr/bin/env python3
"""Shader clock under a workload: a stand-alone conv3x3_ws2 launch (cycle stamps by clock64 = shader clock, wall time by HIP events)
right behind five forwards of the UNet, repeated.  usage: [DMME_NO_ATTN_FULL=1] python tools/clock_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dmme_amd
from dmme_amd import _lib
dev = torch.device("cuda:0")
m = dmme_amd.UNet(precision="bf16").cuda().eval()
x = dmme_amd.gaussian((128, 3, 32, 32), device="cuda"); t = torch.tensor([500], device="cuda")
lib = _lib.lib()
B, hw, cin, cout = 128, 32, 128, 128
xi = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, 9, cin, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(cout, device=dev)
out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.bfloat16)
d = _lib.ConvDesc()
d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = _lib.BF16, B, hw, hw, cin, 0
d.upsample, d.stride, d.taps, d.Cout = 0, 1, 9, cout
d.pro_silu = d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
st = _lib.stream_ptr()
def conv():
    _lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(xi), None, _lib.ptr(w), _lib.ptr(b), None, None, None, None, None, None, cout, _lib.ptr(out), st), "conv")
stamps = torch.zeros(8 * 64 + 4096 * 4, dtype=torch.int64, device=dev)
cyc, wall = [], []
with torch.no_grad():
    for _ in range(10): m(x, t)
    for it in range(12):
        for _ in range(5): m(x, t)
        stamps.zero_()
        _lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(); e1.record()
        torch.cuda.synchronize()
        _lib.check(lib.dmme_debug_set_stamps(None))
        v = [int(a) for a in stamps[:64].cpu() if int(a) != 0]
        cyc.append(v[-1] - v[0]); wall.append(e0.elapsed_time(e1) * 1e3)
cyc.sort(); wall.sort()
print(f"stand-alone 128->128@32x32 conv behind 5 forwards: {cyc[len(cyc)//2]} cycles (first..last stamp of one wave), {wall[len(wall)//2]:.1f} us wall")
