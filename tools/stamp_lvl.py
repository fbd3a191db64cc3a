#!/usr/bin/env python3
"""Per-op phase times inside the level engine (csrc/lvl_engine.hip) from its in-kernel stamps.  usage: python tools/stamp_lvl.py [B] [run] [wg]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmme_amd
from dmme_amd import _lib
from oracle import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
run = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
net = dmme_amd.UNet(precision=os.environ.get("LVL_PRECISION", "bf16")).cuda().eval()
x = synth.normal(1, (B, 3, 32, 32)).cuda()
t = torch.tensor([321]).cuda()
lib = _lib.lib()
with torch.no_grad():
    for _ in range(3):
        net(x, t)
    torch.cuda.synchronize()
    stamps = torch.zeros(120 * 8, dtype=torch.int64, device="cuda")
    _lib.check(lib.dmme_debug_level_stamps(_lib.ptr(stamps), run, wg))
    net(x, t)
    torch.cuda.synchronize()
    _lib.check(lib.dmme_debug_level_stamps(None, -1, 0))
v = stamps.cpu().view(120, 8)
t0 = int(v[0, 0])
print(f"B={B} run={run} wg={wg}: per op iteration, us: start(abs) | wait | gather+unit0 | main | epilogue | drain+barrier | flag+prime | total")
tot = 0
for i in range(120):
    r = [int(a) for a in v[i]]
    if r[0] == 0:
        break
    def d(a, b):
        return (r[b] - r[a]) / 100 if r[a] and r[b] else float("nan")
    conv = r[1] != 0 or r[2] != 0
    if conv:
        s1 = r[1] if r[1] else r[0]
        print(f"  it{i:3d} conv  {(r[0]-t0)/100:8.2f} | {(s1-r[0])/100:6.2f} | {(r[2]-s1)/100:6.2f} | {d(2,3):6.2f} | {d(3,4):6.2f} | {d(4,5):6.2f} | {d(5,6):6.2f} | {d(0,6):6.2f}")
    else:
        print(f"  it{i:3d} other {(r[0]-t0)/100:8.2f} | {'':6s} | {'':6s} | {'':6s} | {d(0,4):6.2f} | {d(4,5):6.2f} | {d(5,6):6.2f} | {d(0,6):6.2f}")
    last = r[6]
print(f"  whole run: {(last - t0) / 100:.2f} us over {i} op iterations")
