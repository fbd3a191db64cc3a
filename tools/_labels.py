"""rocprofv3 kernel names -> the kernel labels bench.py prints (csrc/plan.hip: op_account) for the batch-128 bf16 sampling workload."""
import re


def bench_label(name):
    m = re.match(r"_ZN4dmme(\d+)", name)
    if m:
        k = int(m.group(1)); base = name[m.end():m.end() + k]; rest = name[m.end() + k:]
    else:
        mm = re.match(r"(?:void )?dmme::(\w+)", name)
        if not mm: return None
        base, rest = mm.group(1), name[mm.end():]
    ints = re.findall(r"Li(\d+)E", rest) or re.findall(r"\b(\d+)\b", rest.split("(")[0])
    if base == "lvl_engine_kernel": return "lvl_engine_kernel<8x8>" if ints[:1] == ["2"] else "lvl_engine_kernel<4x4>"  # (batch 128: GB = 2 on the 8x8 maps)
    if base == "conv3x3_ws2_kernel":  # (the RSEG instances - the residual 1x1 conv as a second K segment - end in Lb1E; demangled: "true")
        bools = re.findall(r"Lb([01])E", rest) or [("1" if b == "true" else "0") for b in re.findall(r"\b(true|false)\b", rest.split("(")[0])]
        res = ",res" if bools[:1] == ["1"] else ""  # template <PIPE_UA, T, BM, SPLIT, RSEG, E16>: the first bool is RSEG
        return f"conv3x3_ws2_kernel<7,128{res}>" if ints[:1] == ["7"] else f"conv3x3_ws2_kernel<11{res}>"
    if base in ("conv1x1_as_kernel",): return f"{base}<{ints[0]}>"
    if base == "attn_mfma_kernel": return "attn_mfma_kernel<bf16>"
    if base == "attn_full_kernel":  # template <C, T, PROJ>: the proj conv + residual inside the launch
        bools = re.findall(r"Lb([01])E", rest) or [("1" if b == "true" else "0") for b in re.findall(r"\b(true|false)\b", rest.split("(")[0])]
        # (rocprofv3 prints the PROJ = true instances through a demangler that loses the value - "<256, bool _Accum, bool, E>" - and leaves
        #  the PROJ = false ones mangled)
        return "attn_full_kernel<bf16,proj>" if bools[:1] == ["1"] or "bool _Accum, bool" in rest else "attn_full_kernel<bf16>"
    if base in ("conv3x3_pipe_kernel", "conv1x1_pipe_kernel"): return f"{base}<bf16,{','.join(ints[:4 if base.startswith('conv3') else 2])}>"
    if base == "conv3x3_kw_kernel": return f"{base}<{ints[0]},{ints[1]},{ints[3] if len(ints) > 3 else ints[-1]}>"
    return base
