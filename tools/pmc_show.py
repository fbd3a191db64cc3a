#!/usr/bin/env python3
"""Per-launch averages of every counter tools/pmc_train.sh collected, for kernels whose name matches a regex.
usage: python tools/pmc_show.py <tag> <regex>"""
import collections, csv, glob, os, re, sys
tag, pat = sys.argv[1], re.compile(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for leg in "abfw":
    fs = sorted(glob.glob(os.path.join(root, f"gpurun_out/pmct_{tag}_{leg}/*/*counter_collection.csv")), key=os.path.getmtime)
    if not fs: continue
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(fs[-1])):
        if pat.search(r["Kernel_Name"]):
            k = (r["Kernel_Name"][:48], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in sorted(tot): print(f"{k[0]:50s} {k[1]:28s} {tot[k]/len(disp[k]):16.1f}  ({len(disp[k])} launches)")
