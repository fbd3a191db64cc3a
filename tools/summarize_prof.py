#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_<tag>_*) into small tracked files under profiles/.
usage: python tools/summarize_prof.py <tag> <round-label>"""
import collections
import csv
import glob
import os
import shutil
import sys

tag, label = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
stats = sorted(glob.glob(os.path.join(root, f"gpurun_out/prof_{tag}_stats/*/*kernel_stats.csv")), key=os.path.getmtime)[::-1]
if stats:
    shutil.copy(stats[0], os.path.join(out, f"{label}_kernel_stats.csv"))
rows_out = []
for pmc in ("pmc1", "pmc2", "pmc3"):
    files = sorted(glob.glob(os.path.join(root, f"gpurun_out/prof_{tag}_{pmc}/*/*counter_collection.csv")), key=os.path.getmtime)[::-1]
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    meta = {}
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
    for k in agg:
        for c, v in sorted(agg[k].items()):
            rows_out.append([k, len(disp[k]), c, f"{v / len(disp[k]):.6g}", *meta[k]])
with open(os.path.join(out, f"{label}_pmc_per_dispatch.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "counter", "avg_per_dispatch", "vgpr", "agpr", "sgpr", "lds_bytes", "wg_size"])
    w.writerows(sorted(rows_out))
print("wrote", [p for p in os.listdir(out) if p.startswith(label)])
