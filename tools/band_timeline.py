#!/usr/bin/env python3
"""Merge the kernel and memory-copy traces of a rocprofv3 run and print the timeline of the LAST banded call:
copies individually, kernels grouped into runs of the same name."""
import csv, sys, glob
path = sys.argv[1]
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
ev = []
for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:48], r.get("Queue_Id", "?")))
for f in glob.glob(path + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?")), "-"))
ev.sort()
end = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= end - window_ms * 1e6]
t0 = ev[0][0]
i = 0
while i < len(ev):
    s, e, kind, name, q = ev[i]
    j = i
    if kind == "K":
        while j + 1 < len(ev) and ev[j + 1][2] == "K" and ev[j + 1][3] == name and ev[j + 1][4] == q:
            j += 1
    last = ev[j][1]
    print(f"{(s - t0) / 1e6:8.3f} - {(last - t0) / 1e6:8.3f} ms  {kind} x{j - i + 1:<3} q{q:>2}  {name}")
    i = j + 1
