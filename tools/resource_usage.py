#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: one line per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for ln in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, ln)
        if m and cur is not None:
            cur[key] = int(m.group(1))
try:
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
except Exception:
    names = [r["name"] for r in rows]
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void dxh::", "")
    if filt and filt not in n:
        continue
    print(f"{n:60s} vgpr={r.get('vgpr')} sgpr={r.get('sgpr')} scratch={r.get('scratch')} occ={r.get('occ')} lds={r.get('lds')}")
