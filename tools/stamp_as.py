#!/usr/bin/env python3
"""Phase stamps (100 MHz wall clock -> us) of the activation-stationary 1x1 kernel's workgroups 0 and 1.
usage: python tools/stamp_as.py 16x256x768 [batch] [plain|res]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
shp = sys.argv[1] if len(sys.argv) > 1 else "16x256x768"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mode = sys.argv[3] if len(sys.argv) > 3 else "gn"
hw, cin, cout = (int(v) for v in shp.split("x"))
dev = torch.device("cuda:0")
lib = _lib.lib()
x = torch.randn(B, hw, hw, cin, device=dev).to(torch.bfloat16)
w = (torch.randn(cout, 1, cin, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(cout, device=dev)
scale = torch.rand(B, cin, device=dev) + 0.5
shift = torch.randn(B, cin, device=dev) * 0.1
res = torch.randn(B, hw, hw, cout, device=dev).to(torch.bfloat16)
out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.bfloat16)
d = _lib.ConvDesc()
d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = _lib.BF16, B, hw, hw, cin, 0
d.upsample, d.stride, d.taps, d.Cout = 0, 1, 1, cout
d.pro_silu = d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
st = _lib.stream_ptr()
gn = mode == "gn"
def run():
    _lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(scale if gn else None), _lib.ptr(shift if gn else None), None, None,
                               _lib.ptr(res if mode == "res" else None), None, cout, _lib.ptr(out), st), "conv")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"{shp} B={B} {mode}: {e0.elapsed_time(e1) * 50:.1f} us per launch (back to back)")
stamps = torch.zeros(4096, dtype=torch.int64, device=dev)
_lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
run(); torch.cuda.synchronize()
_lib.check(lib.dmme_debug_set_stamps(None))
v = stamps.cpu()[:64].tolist()
mn = {1: "DMA issued", 2: "tile landed", 3: "fragments", 4: "u0 mfma", 5: "u0 staged+barrier", 7: "u1 mfma", 8: "u1 staged+barrier", 10: "u2 mfma", 11: "u2 staged+barrier", 13: "end"}
sn = {4: "u0 barrier", 5: "u0 stored", 7: "u1 barrier", 8: "u1 stored", 10: "u2 barrier", 11: "u2 stored", 13: "end"}
for wg in range(2):
    for team, names in ((0, mn), (1, sn)):
        t = v[wg * 32 + team * 16: wg * 32 + team * 16 + 14]
        if not t[0]: continue
        prev, out = t[0], []
        for k in sorted(names):
            if t[k]:
                out.append(f"{names[k]} +{(t[k] - prev) / 100:.2f}")
                prev = t[k]
        print(f"wg{wg} {'mfma ' if team == 0 else 'store'}: " + ", ".join(out) + f"; total {(t[13] - t[0]) / 100:.2f} us")
