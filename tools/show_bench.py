#!/usr/bin/env python3
"""Print the headline and the per-kernel table of a bench.py JSON line: python tools/show_bench.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(d["value"], d["unit"], d["ms_per_step"], "ms/step; launches", d.get("launches_per_step"), "; roofline frac", r.get("frac"))
for k in r.get("top_kernels", []):
    print(f"{k['kernel']:45s} n={k['launches']:3d} {k['ms_per_step']*1000:7.1f} us  avg {k['avg_launch_us']:6.1f}  {k.get('tflops', 0):6.1f} TF  share {k['share']:.3f}")
for key in ("small_batch", "fp16_mode", "accurate_mode", "train", "ddim_b512"):
    if key in d:
        print(key, json.dumps(d[key])[:400])
