#!/usr/bin/env python3
"""Print the kernel timeline (start, duration, gap, queue, name) of the last N kernels of a rocprofv3 kernel-trace CSV."""
import csv, sys, glob
path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:60]
    print(f"{(st - t0) / 1e3:10.1f} us  dur {(en - st) / 1e3:8.1f}  gap {(st - prev_end) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {name}")
    prev_end = max(prev_end, en)
