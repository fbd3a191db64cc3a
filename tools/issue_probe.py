#!/usr/bin/env python3
"""What does one KIND of vector instruction cost beside MFMAs of the partner wave on the same SIMD (dmme_debug_issue_probe)?
Per kind and count n (instructions per 32x32x16 MFMA slot): cycles per MFMA slot of the MFMA wave alone, of the vector wave alone,
and of both when they run together (in-kernel s_memtime cycles, median over workgroups: the clock does not enter)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmme_amd import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
BLOCKS = 256
sink = torch.zeros(2 * BLOCKS, dtype=torch.int64, device=dev)
src = torch.randn(2 << 20, device=dev)  # 8 MB; the probe's request kinds read its first 1.2 MB
ITERS = int(os.environ.get("ITERS", "4000"))
KINDS = {0: "v_fma_f32", 1: "v_pk_fma_f32", 2: "v_exp_f32", 3: "v_cvt_pk_bf16_f32", 4: "v_pk_mul_f32", 5: "v_pk_add_f32", 6: "v_add_f32",
         7: "v_rcp_f32", 8: "prologue dword, plain (13 instr)", 9: "prologue dword, packed (9 instr)", 10: "ds_write_b128", 11: "ds_read_b128",
         12: "bf16 unpack (shift/and)", 13: "LDS-DMA 1 KB (tap shape)", 14: "global_load 1 KB (tap shape)"}
def run(kind, n, flags):
    st = _lib.stream_ptr()
    sink.zero_()
    for _ in range(2):
        _lib.check(lib.dmme_debug_issue_probe(kind, n, ITERS, flags, BLOCKS, _lib.ptr(sink), _lib.ptr(src), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.dmme_debug_issue_probe(kind, n, ITERS, flags, BLOCKS, _lib.ptr(sink), _lib.ptr(src), st)); e1.record(); torch.cuda.synchronize()
    s = sink.view(BLOCKS, 2).float()
    slots = ITERS * 8
    return s[:, 0].median().item() / slots, s[:, 1].median().item() / slots, e0.elapsed_time(e1)
def main():
    extra = int(os.environ.get("FLAGS", "0"))  # 4: prio on MFMA waves, 8: prio on vector waves, 16: 16x16x32
    m_alone, _, ms = run(0, 0, 1 | extra)
    print(f"MFMA waves alone: {m_alone:6.1f} cycles per 32x32x16 slot ({ms:.3f} ms)")
    kinds = [int(k) for k in os.environ.get("KINDS", "0,1,2,3,4,5,6,7,8,9,10,11,12").split(",")]
    print(f"{'kind':34s} {'n/slot':>6s} {'vec alone':>10s} {'both: mfma':>11s} {'both: vec':>10s} {'cost/instr':>10s}")
    for k in kinds:
        per = 13 if k == 8 else 9 if k == 9 else 8
        for n in ((1, 2) if k in (8, 9) else (1, 2, 4, 8) if k >= 13 else (1, 2, 4, 6)):
            _, v_alone, _ = run(k, n, 2 | extra)
            m_both, v_both, _ = run(k, n, 3 | extra)
            ninstr = n * per / 8.0  # vector instructions per MFMA slot
            if k >= 13: ninstr = n / 8.0
            print(f"{KINDS[k]:34s} {ninstr:6.2f} {v_alone:10.1f} {m_both:11.1f} {v_both:10.1f} {(max(m_both, v_both) - m_alone) / ninstr:10.2f}")
if __name__ == "__main__":
    main()
