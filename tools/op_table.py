"""Per-op timing table of one UNet forward (event-bracketed; run on the GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dmme_amd
from dmme_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02  # ms
m = dmme_amd.UNet(precision=prec).cuda().eval()
x = dmme_amd.gaussian((B, 3, 32, 32), device="cuda"); t = torch.tensor([500], device="cuda")
with torch.no_grad(): m(x, t)
plan = m._last_plan; lib = plan.lib
n = lib.dmme_unet_plan_num_ops(plan.h); buf = C.create_string_buffer(128); f, b = C.c_double(), C.c_double()
info = []
for i in range(n):
    lib.dmme_unet_plan_op_info(plan.h, i, buf, 128, C.byref(f), C.byref(b)); info.append((buf.value.decode(), f.value, b.value))
ms = (C.c_float * n)(); acc = [0.0] * n; y = torch.empty_like(x); packed = m._packed_for(plan)
for r in range(6):
    _lib.check(lib.dmme_unet_forward_profiled(plan.h, _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t), 1, _lib.ptr(y), _lib.ptr(plan.workspace), _lib.ptr(None), _lib.stream_ptr(), ms))
    if r:
        for i in range(n): acc[i] += ms[i] / 5
for i, (lab, fl, by) in enumerate(info):
    if acc[i] >= thr and not lab.startswith("("): print(f"{i:3d} {lab:38s} {acc[i]*1e3:8.1f} us  {fl/1e9:7.2f} GFLOP {fl/acc[i]/1e9 if acc[i] else 0:8.1f} TF  {by/1e6:7.1f} MB {by/acc[i]/1e6:8.1f} GB/s")
print("total ms", sum(a for a, (lab, _, _) in zip(acc, info) if not lab.startswith("(")))
