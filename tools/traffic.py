#!/usr/bin/env python3
"""HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh -> profiles/traffic_latest.json.
Counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so the read side is doubled.  usage: python tools/traffic.py <tag> <label>"""
import collections, csv, glob, json, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _labels import bench_label
from dmme_amd._lib import csrc_sha16

tag, label = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def per_kernel(pmc, counter):
    f = sorted(glob.glob(os.path.join(root, f"gpurun_out/prof_{tag}_{pmc}/*/*counter_collection.csv")), key=os.path.getmtime)[-1]
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: tot[k] / len(disp[k]) for k in tot}
def norm(name):
    m = re.match(r"_ZN4dmme(\d+)", name)
    if m:  # hand-demangle our own kernels (binutils' c++filt does not know DF16b = __bf16)
        n = int(m.group(1)); rest = name[m.end():]; kname, rest = rest[:n], rest[n:]
        args = []
        if rest.startswith("I"):
            rest = rest[1:]
            while rest and not rest.startswith("E"):
                if rest.startswith("DF16b"): args.append("bf16"); rest = rest[5:]
                elif rest.startswith("DF16_"): args.append("f16"); rest = rest[5:]
                elif rest.startswith("f"): args.append("float"); rest = rest[1:]
                else:
                    mm = re.match(r"L([ib])(\d+)E", rest)  # integer and BOOL template arguments: Lb1E tells <.., res> from the plain instance
                    if not mm: break
                    args.append(mm.group(2) if mm.group(1) == "i" else ("true" if mm.group(2) == "1" else "false")); rest = rest[mm.end():]
        return kname + ("<" + ",".join(args) + ">" if args else "")
    m = re.match(r"(?:void )?dmme::(\w+)(<[^(]*>)?\(", name)
    if not m: return name
    # (rocprofv3's demangler does not know DF16b either: "IDF16bLi1E" comes out as "<bool _Accum, int, E")
    targs = (m.group(2) or "").replace("__bf16", "bf16").replace("bool _Accum, int, E", "bf16,1").replace(" ", "")
    return m.group(1) + targs
rd, wr = per_kernel("pmc3", "FETCH_SIZE"), per_kernel("pmc4", "WRITE_SIZE")
out = {}
for k in rd:
    out[norm(k)] = {"hbm_bytes_per_launch": round((2.0 * rd[k] + wr.get(k, 0.0)) * 1024), "fetch_kib_raw": round(rd[k], 1), "write_kib": round(wr.get(k, 0.0), 1),
                    "note": "2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE half-count correction), averaged over the launches of this symbol"}
# the same figures under bench.py's kernel labels (launch-weighted where several symbols share a label)
lab_tot, lab_n = collections.defaultdict(float), collections.defaultdict(int)
f3 = sorted(glob.glob(os.path.join(root, f"gpurun_out/prof_{tag}_pmc3/*/*counter_collection.csv")), key=os.path.getmtime)[-1]
ndisp = collections.defaultdict(set)
for r in csv.DictReader(open(f3)):
    ndisp[r["Kernel_Name"]].add(r["Dispatch_Id"])
for k in rd:
    lab = bench_label(k)
    if lab:
        lab_tot[lab] += (2.0 * rd[k] + wr.get(k, 0.0)) * 1024 * len(ndisp[k]); lab_n[lab] += len(ndisp[k])
for lab in lab_tot:
    if lab not in out:
        out[lab] = {"hbm_bytes_per_launch": round(lab_tot[lab] / lab_n[lab]), "note": "bench.py label: launch-weighted mean of the symbols above"}
out["_meta"] = {"csrc_sha16": csrc_sha16(), "label": label, "workload": "bench.py sampling leg, batch 128, bf16 (tools/prof.sh pmc3 / pmc4 passes)"}
json.dump(out, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", f"{label}_hbm_traffic.json"), "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]: print(k, v["hbm_bytes_per_launch"])
