#!/usr/bin/env python3
"""Phase stamps (100 MHz wall clock -> us) of the software-pipelined 3x3 kernel's workgroups 0 and 1, next to the dispatch's
HIP-event duration.  usage: python tools/stamp_pipe.py 4x256x256 [batch]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
shp = sys.argv[1] if len(sys.argv) > 1 else "4x256x256"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
plain = len(sys.argv) > 3 and sys.argv[3] == "plain"
hw, cin, cout = (int(v) for v in shp.split("x"))
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
d.pro_silu = 0 if plain else 1
d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
st = _lib.stream_ptr()
def run():
    _lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(None if plain else scale), _lib.ptr(None if plain else shift), None, None, None, None, cout,
                               _lib.ptr(out), st), "conv")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"{shp} B={B}: {e0.elapsed_time(e1) * 50:.1f} us per launch (back to back)")
stamps = torch.zeros(4096, dtype=torch.int64, device=dev)
_lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
run(); torch.cuda.synchronize()
_lib.check(lib.dmme_debug_set_stamps(None))
v = stamps.cpu()[:16].tolist()
for wg in range(2):
    t = v[wg * 8: wg * 8 + 5]
    e = v[wg * 8 + 5: wg * 8 + 8]
    if all(e): print(f"   kw: setup +{(e[0]-t[0])/100:.2f} us, filter units primed +{(e[1]-e[0])/100:.2f}, halo DMA issued +{(t[1]-e[1])/100:.2f}, landed +{(e[2]-t[1])/100:.2f}, transformed +{(t[2]-e[2])/100:.2f}")
    print(f"wg{wg}: loads issued +{(t[1]-t[0])/100:.2f} us, tiles staged +{(t[2]-t[1])/100:.2f}, main loop +{(t[3]-t[2])/100:.2f}, epilogue +{(t[4]-t[3])/100:.2f}, total {(t[4]-t[0])/100:.2f} us")
