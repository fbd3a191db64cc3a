#!/usr/bin/env python3
"""Per-stage cycle stamps of the wave-specialised conv kernel (consumer wave 0 and producer wave 4 of two workgroups).
usage: python tools/stamp_ws.py 32x128x128 [gn] [r32] [res=256+128]
  r32: the split-pass form of precision="fp16r32" (fp32 tensors, hi / lo halves); res=A+B: with the residual segment over raw inputs of
  A and B channels (dmme_conv2d_res)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
shp = sys.argv[1] if len(sys.argv) > 1 else "32x128x128"
gn = "gn" in sys.argv[2:]
r32 = "r32" in sys.argv[2:]
res = next((v[4:] for v in sys.argv[2:] if v.startswith("res=")), None)
hw, cin, cout = (int(v) for v in shp.split("x"))
B = 128
dev = torch.device("cuda:0")
lib = _lib.lib()
x = torch.randn(B, hw, hw, cin, device=dev).to(torch.float32 if r32 else torch.bfloat16)
w = (torch.randn(cout, 9, (2 if r32 else 1) * cin, device=dev) * 0.05).to(torch.float16 if r32 else torch.bfloat16)
b = torch.randn(cout, device=dev)
scale = torch.rand(B, cin, device=dev) + 0.5
shift = torch.randn(B, cin, device=dev) * 0.1
out = torch.empty(B, hw, hw, cout, device=dev, dtype=torch.float32 if r32 else torch.bfloat16)
d = _lib.ConvDesc()
d.dtype, d.N, d.Hin, d.Win, d.C1, d.C2 = (_lib.F16R32 if r32 else _lib.BF16), B, hw, hw, cin, 0
d.upsample, d.stride, d.taps, d.Cout = 0, 1, 9, cout
d.pro_silu = int(gn)
d.out_silu = d.nt = d.tproj_ld = d.in_nchw = d.out_nchw = d.force_generic = 0
st = _lib.stream_ptr()
sc, sh = (scale, shift) if gn else (None, None)
if res:
    rc1, rc2 = (int(v) for v in res.split("+"))
    xr1 = torch.randn(B, hw, hw, rc1, device=dev).to(torch.bfloat16)
    xr2 = torch.randn(B, hw, hw, rc2, device=dev).to(torch.bfloat16) if rc2 else None
    wr = (torch.randn(cout, rc1 + rc2, device=dev) * 0.05).to(torch.bfloat16)
    br = torch.randn(cout, device=dev)
def run():
    if res:
        _lib.check(lib.dmme_conv2d_res(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(sh), None, _lib.ptr(xr1), _lib.ptr(xr2),
                                       rc1, rc2, _lib.ptr(wr), _lib.ptr(br), _lib.ptr(out), st), "conv_res")
        return
    _lib.check(lib.dmme_conv2d(C.byref(d), _lib.ptr(x), None, _lib.ptr(w), _lib.ptr(b), _lib.ptr(sc), _lib.ptr(sh), None, None, None, None, cout,
                               _lib.ptr(out), st), "conv")
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"{shp} gn={gn} r32={r32}: {e0.elapsed_time(e1) * 100:.1f} us per launch (event-bracketed, 10 launches)")
stamps = torch.zeros(8 * 64 + 4096 * 4, dtype=torch.int64, device=dev)
_lib.check(lib.dmme_debug_set_stamps(_lib.ptr(stamps)))
run()
torch.cuda.synchronize()
_lib.check(lib.dmme_debug_set_stamps(None))
v = [int(t) for t in stamps.cpu()[:64] if int(t) != 0]
rel = [t - v[0] for t in v]
d = [rel[i + 1] - rel[i] for i in range(len(rel) - 1)]
print("stamps", len(v), "total", rel[-1])
print("wait preamble", d[0])
print("work per stage:", d[1::2][:31])
print("wait per stage:", d[2::2][:31])

e = [int(t) for t in stamps.cpu()[64:88] if int(t) != 0]
print("epilogue stamps (pass start, after conv_epilogue, after barrier) deltas:", [e[i + 1] - e[i] for i in range(len(e) - 1)])

pv = [int(t) for t in stamps.cpu()[128:628] if int(t) != 0]
if pv and os.environ.get("PSTAMPS_TAGGED"):
    # tagged stamps (tag in the top byte): 0 = the five stage stamps, 1 / 2 / 3 = after a unit's store / coordinates / load
    tg = [(t >> 56) & 0xff for t in pv]; ck = [t & ((1 << 56) - 1) for t in pv]
    line = []; nstage = 0; zeros = 0
    for i in range(1, len(pv)):
        if tg[i] == 0:
            zeros += 1
        line.append(f"{'sulS'[tg[i]] if tg[i] else 'S'}{ck[i] - ck[i - 1]}")
        if tg[i] == 0 and zeros % 5 == 0:
            print(f"  st{nstage:2d} (tap {nstage % 9}): " + " ".join(line)); line = []; nstage += 1
            if nstage >= 22: break
    pv = []
if pv:
    # five stamps per producer stage: entry, halo data arrived, DMA issued, units done, DMA retired (then the barrier)
    NP = int(os.environ.get("PSTAMPS_PER_STAGE", "5"))
    rows = [pv[i:i + NP] for i in range(0, len(pv) - NP + 1, NP)]
    print("producer wave 4, per stage: [halo-arrive wait, dma issue, units, vm wait, barrier+next] (cycles)")
    for i in range(min(len(rows) - 1, 40)):
        r = rows[i]
        print(f"  st{i:2d} (tap {i % 9}): " + " ".join(f"{r[k+1]-r[k]:5d}" for k in range(NP - 1)) + f" {rows[i+1][0]-r[NP-1]:5d}")
