#!/usr/bin/env python3
"""The "state of the roofline" page at the top of DESIGN.md, GENERATED from the committed measurements so that it cannot drift:
   profiles/<tag>_bench_n1.json            the N = 1 bench line (bench.py's live per-kernel HIP-event timing: roofline.top_kernels)
   profiles/<tag>_sample_b128_bf16_hbm_traffic.json   FETCH / WRITE counter passes (tools/traffic.py)
   profiles/<tag>_sample_b128_bf16_mfma_busy.json     SQ_VALU_MFMA_BUSY_CYCLES pass (tools/mfma_busy.py)
   profiles/limiters.json                  per kernel label: the named limiter and the lever (the only hand-written input)
usage: python tools/roofline_state.py <tag>      (rewrites the block between the ROOFLINE_STATE markers of DESIGN.md)"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = lambda n: os.path.join(root, "profiles", n)
bench = json.load(open(P(f"{tag}_bench_n1.json")))
traffic = json.load(open(P(f"{tag}_sample_b128_bf16_hbm_traffic.json"))) if os.path.exists(P(f"{tag}_sample_b128_bf16_hbm_traffic.json")) else {}
busy = json.load(open(P(f"{tag}_sample_b128_bf16_mfma_busy.json"))) if os.path.exists(P(f"{tag}_sample_b128_bf16_mfma_busy.json")) else {}
lim = json.load(open(P("limiters.json")))
r = bench["roofline"]
rows = []
for k in r["top_kernels"][:8]:
    name = k["kernel"]
    algo_bytes = k["algo_gbs"] * 1e9 * k["avg_launch_us"] * 1e-6
    t = traffic.get(name, {}).get("hbm_bytes_per_launch")
    ratio = f"{t / algo_bytes:.2f}x" if t and algo_bytes > 0 else "n/a"
    b = busy.get(name, {}).get("mfma_busy_pct", k.get("mfma_busy_pct_from_profile"))
    L = lim.get(name, {})
    rows.append(f"| `{name}` | {k['launches']} | {k['ms_per_step']:.3f} | {100 * k['share']:.1f} % | {k['tflops']:.0f} | {100 * k['mfma_frac']:.1f} % | "
                f"{'n/a' if b is None else f'{b:.1f} %'} | {ratio} | {L.get('limiter', '-')} | {L.get('lever', '-')} |")
am = bench.get("value_within_north_star_tolerance") or {}
head = [
    f"## 0. State of the roofline (generated: `python tools/roofline_state.py {tag}` from `profiles/{tag}_*`; do not edit by hand)",
    "",
    f"DDPM sampling step, batch 128, bf16, 1 x MI355X: **{bench['value']:.1f} steps/s = {bench['ms_per_step']:.3f} ms/step** "
    f"(whole step {bench.get('step_tflops', 0):.0f} TFLOP/s = {100 * bench.get('step_frac_of_peak', 0):.1f} % of the 2.5 PF/s dense bf16 peak; the box's own "
    f"MFMA-only loop sustains {bench.get('box_mfma_tfps', 0):.0f} TF/s); within north_star's 1e-3: **{am.get('value', 'n/a')} steps/s** ({am.get('precision', '-')}). "
    f"Kernel times: HIP events on the launch stream inside `bench.py` (sum {r.get('step_gpu_ms_sum', 0):.3f} ms); fractions against the NOMINAL 2.5 PF/s; "
    "counter ÷ algorithmic bytes from separate FETCH_SIZE / WRITE_SIZE passes of the same build (source hash "
    f"`{r.get('csrc_sha16', '?')}`).",
    "",
    "| kernel | launches / step | ms / step | share | TF/s | of MFMA peak | MFMA busy (counter) | HBM bytes ÷ algorithmic | limiter (measured) | lever |",
    "|---|---|---|---|---|---|---|---|---|---|",
] + rows + [""]
design = open(os.path.join(root, "DESIGN.md")).read()
B, E = "<!-- ROOFLINE_STATE:BEGIN -->", "<!-- ROOFLINE_STATE:END -->"
block = B + "\n" + "\n".join(head) + E
if B in design:
    design = design[:design.index(B)] + block + design[design.index(E) + len(E):]
else:  # first time: right behind the title paragraph
    cut = design.index("## 1. The path and its boundary")
    design = design[:cut] + block + "\n\n" + design[cut:]
open(os.path.join(root, "DESIGN.md"), "w").write(design)
print("\n".join(head))
