#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/gpu_profile.sh) into profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  <tag>_pmc_summary.csv    per-kernel mean FETCH_SIZE / WRITE_SIZE
  traffic.json             HBM bytes per launch of the dominant fused kernel, corrected as
                           MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950; KiB units),
                           with the k_scale launch (exactly one plane read and written) as calibration.
"""
import csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]  # noqa: E731
stats = newest(f"{src}/trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats[0])))
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    print("dominant kernel:", top["Name"][:90], "avg ns", top["AverageNs"], "calls", top["Calls"])
else:
    top = None


def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


f = newest(f"{src}/pmc_fetch/*/*_counter_collection.csv")
w = newest(f"{src}/pmc_write/*/*_counter_collection.csv")
if f and w:
    fe, wr = per_kernel(f[0], "FETCH_SIZE"), per_kernel(w[0], "WRITE_SIZE")
    with open(f"profiles/{tag}_pmc_summary.csv", "w") as o:
        o.write("kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean\n")
        for k in fe:
            o.write(f"\"{k}\",{fe[k][1]},{fe[k][0]:.3f},{wr.get(k, (0, 0))[0]:.3f}\n")
    fused = [k for k in fe if "k_fused" in k]
    scale = [k for k in fe if "k_scale" in k]
    if fused:
        # the kernel the trace found dominant (the 10-iteration PMC runs also launch a K = 2 remainder)
        k = top["Name"] if top and top["Name"] in fe else max(fused, key=lambda n: (fe[n][1], fe[n][0]))
        fetch_b = fe[k][0] * 1024 * 2.0      # gfx950: FETCH_SIZE reports half the bytes of a coalesced stream
        write_b = wr[k][0] * 1024
        cal = None
        if scale:
            ks = scale[0]
            cal = {"kernel": "k_scale (reads one plane, writes one plane)",
                   "fetch_x2_bytes": fe[ks][0] * 2048, "write_bytes": wr[ks][0] * 1024}
        t = {"kernel": k, "hbm_bytes_per_launch": round(fetch_b + write_b),
             "read_bytes_per_launch": round(fetch_b), "write_bytes_per_launch": round(write_b),
             "launches_sampled": fe[k][1],
             "source": f"profiles/{tag}_pmc_summary.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                       "passes; FETCH_SIZE x 2 (gfx950 reports half the bytes of coalesced reads), KiB -> bytes",
             "calibration": cal}
        json.dump(t, open("profiles/traffic.json", "w"), indent=1)
        print(json.dumps(t, indent=1))
