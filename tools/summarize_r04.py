#!/usr/bin/env python3
"""Condense the round-4 evidence run (tools/profile_r04.sh, raw output in gpurun_out/) into profiles/:
  r04_kernel_stats.csv      rocprofv3 --kernel-trace --stats of bench.py (both dtype flows run in it)
  r04_pmc_summary.csv       per-kernel mean FETCH_SIZE / WRITE_SIZE
  traffic.json              HBM bytes per launch of the K-stage kernel of EACH flow ({"native": ..., "numpy2": ...})
  r04_sq_counters.csv       SQ / traffic counters per launch (native float pipeline, the numpy2 pipelines)
  r04_marker_trace.txt      roctx ranges of a banded whole call
"""
import csv, glob, json, os, shutil, collections
os.makedirs("profiles", exist_ok=True)
src = "gpurun_out/prof_r04"
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1:]  # noqa: E731
stats = newest(f"{src}/trace/**/*_kernel_stats.csv")
if stats:
    shutil.copy(stats[0], "profiles/r04_kernel_stats.csv")
    for r in sorted(csv.DictReader(open(stats[0])), key=lambda r: -float(r["TotalDurationNs"]))[:6]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]} %')


def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


f, w = newest(f"{src}/pmc_fetch/**/*_counter_collection.csv"), newest(f"{src}/pmc_write/**/*_counter_collection.csv")
if f and w:
    fe, wr = per_kernel(f[0], "FETCH_SIZE"), per_kernel(w[0], "WRITE_SIZE")
    with open("profiles/r04_pmc_summary.csv", "w") as o:
        o.write("kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean\n")
        for k in fe:
            o.write(f"\"{k}\",{fe[k][1]},{fe[k][0]:.3f},{wr.get(k, (0, 0))[0]:.3f}\n")
    scale = [k for k in fe if "k_scale<" in k]
    cal = None
    if scale:
        ks = scale[0]
        cal = {"kernel": "k_scale<float> (reads one float32 plane, writes one)", "fetch_x2_bytes": fe[ks][0] * 2048, "write_bytes": wr[ks][0] * 1024}
    traffic = {}
    for flow, pat in (("native", "k_fused_multi<float, 3, true, false, 4"), ("numpy2", "k_fused_multi_np2<3, true, 4, false, false")):
        ks = [k for k in fe if pat in k]
        if not ks:
            continue
        k = ks[0]
        fetch_b, write_b = fe[k][0] * 2048, wr[k][0] * 1024
        traffic[flow] = {"kernel": k, "hbm_bytes_per_launch": round(fetch_b + write_b), "read_bytes_per_launch": round(fetch_b),
                         "write_bytes_per_launch": round(write_b), "launches_sampled": fe[k][1],
                         "source": "profiles/r04_pmc_summary.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
                                   "FETCH_SIZE x 2 (gfx950 reports half the bytes of coalesced 16-byte-per-lane reads; the numpy2 "
                                   "kernel's 8-byte-per-lane reads of the float32 plane are 4 of its 28 read bytes and uncalibrated), "
                                   "KiB -> bytes", "calibration": cal}
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}))

rows = []
for v in ("aniso", "np2", "np2arr", "np2gray"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"gpurun_out/sq_r04_{v}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_fused" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        n = max(len(x) for x in d.values())
        if n < 2:
            continue
        m = {c: sum(x) / len(x) for c, x in d.items()}
        valu_cycles, launch_cycles = m.get("SQ_INSTS_VALU", 0) * 4 / 1024, m.get("GRBM_GUI_ACTIVE", 0) / 8
        rows.append({"variant": v, "kernel": k, "launches": n,
                     **{c: f"{m.get(c, float('nan')):.6g}" for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS",
                                                                   "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                                                                   "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")},
                     "valu_issue_cycles_per_simd": f"{valu_cycles:.6g}", "launch_cycles": f"{launch_cycles:.6g}",
                     "valu_utilisation": f"{valu_cycles / launch_cycles:.3f}" if launch_cycles else "",
                     "hbm_read_bytes": f"{m.get('FETCH_SIZE', 0) * 2048:.6g}", "hbm_write_bytes": f"{m.get('WRITE_SIZE', 0) * 1024:.6g}"})
if rows:
    with open("profiles/r04_sq_counters.csv", "w", newline="") as fo:
        wtr = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        wtr.writeheader()
        wtr.writerows(rows)
    for r in rows:
        print(r["variant"], r["kernel"][:64], "VALU insts", r["SQ_INSTS_VALU"], "util", r["valu_utilisation"])
mk = newest("gpurun_out/marker_r04/**/*marker_api_trace.csv")
if mk:
    rws = list(csv.DictReader(open(mk[0])))
    t0 = min(int(r["Start_Timestamp"]) for r in rws)
    with open("profiles/r04_marker_trace.txt", "w") as fo:
        fo.write("roctx ranges of da.split_bregman_tvd(4000 x 8000 x 3 float32 host image, 40 iterations): DXH_ROCTX=1 rocprofv3 --marker-trace\n")
        fo.write("start_ms  duration_ms  range\n")
        for r in sorted(rws, key=lambda r: int(r["Start_Timestamp"])):
            fo.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:8.2f}  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6:8.2f}  {r.get("Function", r.get("Message", ""))}\n')
    print(open("profiles/r04_marker_trace.txt").read()[:1500])
