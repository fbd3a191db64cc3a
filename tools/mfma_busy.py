#!/usr/bin/env python3
"""MFMA-busy percentage per kernel from the pmc1 pass of tools/prof.sh -> profiles/mfma_busy_latest.json (+ <label>_mfma_busy.json).
busy % = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x the dispatch's own cycles): cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over
the 8 XCDs) of the SAME pass - not a nominal clock, which the boards' power management moves; `mfma_busy_pct_nominal_2p4ghz` keeps
the older duration x 2.4 GHz figure beside it.  Keys are bench.py's kernel labels; `_meta` records the source hash of the build.
usage: python tools/mfma_busy.py <tag> <label>"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, label = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
newest = lambda pat: sorted(glob.glob(os.path.join(root, pat)), key=os.path.getmtime)[-1]
cc = newest(f"gpurun_out/prof_{tag}_pmc1/*/*counter_collection.csv")
kt = newest(f"gpurun_out/prof_{tag}_pmc1/*/*kernel_trace.csv")
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
busy, ns, n, gui = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int), collections.defaultdict(float)
for r in csv.DictReader(open(cc)):
    if r["Dispatch_Id"] not in dur:
        continue
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui[r["Kernel_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
        continue
    busy[r["Kernel_Name"]] += float(r["Counter_Value"])
    ns[r["Kernel_Name"]] += dur[r["Dispatch_Id"]]
    n[r["Kernel_Name"]] += 1
from _labels import bench_label
out = {}
agg = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])
for k in busy:
    lab = bench_label(k)
    if lab is None: continue
    agg[lab][0] += busy[k]; agg[lab][1] += ns[k]; agg[lab][2] += n[k]; agg[lab][3] += gui[k]
for lab, (b, t, c, gcy) in agg.items():
    nominal = round(100.0 * b / (1024 * t * 2.4), 2)
    out[lab] = {"mfma_busy_pct": round(100.0 * b / (1024 * gcy / 8.0), 2) if gcy > 0 else nominal, "mfma_busy_pct_nominal_2p4ghz": nominal,
                "clock_ghz_under_pmc": round(gcy / 8.0 / t, 3) if gcy > 0 else None, "dispatches": c, "avg_us_under_pmc": round(t / c / 1e3, 1)}
from dmme_amd._lib import csrc_sha16
out["_meta"] = {"csrc_sha16": csrc_sha16(), "label": label, "workload": "bench.py sampling leg, batch 128, bf16 (tools/prof.sh pmc1 pass)"}
json.dump(out, open(os.path.join(root, "profiles", "mfma_busy_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", f"{label}_mfma_busy.json"), "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["mfma_busy_pct"])[:12]: print(k, v)
