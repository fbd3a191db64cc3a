#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels in a remarks file written by tools/isa.sh: tools/resources.py conv_pipe [filter]"""
import re, subprocess, sys
name = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
txt = open(f"/tmp/isa/{name}.remarks").read()
blocks = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", txt, re.S)
seen = set()
for fn, v, a, s, o, l in blocks:
    if fn in seen:
        continue
    seen.add(fn)
    d = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(.*", "", d).replace("dmme::", "").replace("void ", "")
    if flt in d:
        print(f"{d[:100]:100s} VGPR {v:>3s} AGPR {a:>3s} scratch {s:>4s} occ {o} lds {l}")
