#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats run. usage: python tools/top_kernels.py <dir under gpurun_out> <launch divisor> [n]"""
import csv, glob, os, sys
d, div = sys.argv[1], float(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(root, "gpurun_out", d, "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms/step", round(tot / div / 1e6, 3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{r['Name'][:70]:70s} {float(r['Calls'])/div:6.1f} {float(r['TotalDurationNs'])/div/1e6:7.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
