#!/usr/bin/env python3
"""Do MFMAs and VALU work of two waves on one SIMD overlap (dmme_debug_mfma_valu)?  Prints the time of the MFMA waves alone, the VALU
waves alone and both together, with the accumulators in arch VGPRs (what hipcc picks at two waves per SIMD) and in AccVGPRs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
sink = torch.zeros(4096, device=dev)
iters = 20000
def run(mode):
    st = _lib.stream_ptr()
    _lib.check(lib.dmme_debug_mfma_valu(mode, iters, 256, _lib.ptr(sink), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.dmme_debug_mfma_valu(mode, iters, 256, _lib.ptr(sink), st)); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for name, mode in (("MFMA waves alone (arch VGPR accumulators)", 1), ("MFMA waves alone (AccVGPR accumulators)", 5), ("VALU waves alone", 2),
                   ("both, arch VGPR accumulators", 3), ("both, AccVGPR accumulators", 7)):
    ms = run(mode)
    print(f"{name:46s} {ms:8.3f} ms  ({ms * 1e6 / iters / 8:6.1f} ns per MFMA slot)")
