#!/usr/bin/env python3
"""MFMA-busy percentage per kernel from the pmc1 pass of tools/prof.sh -> profiles/mfma_busy_latest.json (+ <label>_mfma_busy.json).
busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz), summed over the dispatches of a symbol in the pass
(counter and duration of the same dispatch, from the same run).  Keys are bench.py's kernel labels.
usage: python tools/mfma_busy.py <tag> <label>"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
tag, label = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
newest = lambda pat: sorted(glob.glob(os.path.join(root, pat)), key=os.path.getmtime)[-1]
cc = newest(f"gpurun_out/prof_{tag}_pmc1/*/*counter_collection.csv")
kt = newest(f"gpurun_out/prof_{tag}_pmc1/*/*kernel_trace.csv")
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
busy, ns, n = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES" or r["Dispatch_Id"] not in dur:
        continue
    busy[r["Kernel_Name"]] += float(r["Counter_Value"])
    ns[r["Kernel_Name"]] += dur[r["Dispatch_Id"]]
    n[r["Kernel_Name"]] += 1
from _labels import bench_label
out = {}
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k in busy:
    lab = bench_label(k)
    if lab is None: continue
    agg[lab][0] += busy[k]; agg[lab][1] += ns[k]; agg[lab][2] += n[k]
for lab, (b, t, c) in agg.items():
    out[lab] = {"mfma_busy_pct": round(100.0 * b / (1024 * t * 2.4), 2), "dispatches": c, "avg_us_under_pmc": round(t / c / 1e3, 1)}
json.dump(out, open(os.path.join(root, "profiles", "mfma_busy_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", f"{label}_mfma_busy.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_pct"])[:12]: print(k, v)
