#!/usr/bin/env python3
"""Condense the round-3 evidence run (tools/profile_r03.sh, raw output in gpurun_out/) into profiles/:
  r03_kernel_stats.csv, r03_pmc_summary.csv, traffic.json     (tools/summarize_prof.py r03: bench.py)
  r03_sq_counters.csv        SQ / traffic counters per launch of every shipped pipeline
  r03_other_kernel_stats.csv rocprofv3 --kernel-trace --stats of tools/bench_configs.py
  r03_other_configs.txt      its timings
"""
import csv, glob, os, re, shutil, collections
os.makedirs("profiles", exist_ok=True)
rows = []
for v in ("aniso", "iso", "gray", "np2", "np2arr", "jacobi3d", "warp"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/sq_r03_{v}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not any(s in k for s in ("k_fused", "k_jacobi_vol", "k_warp")):
                continue
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        n = max(len(x) for x in d.values())
        if n < 2 and v not in ("warp",):
            continue
        m = {c: sum(x) / len(x) for c, x in d.items()}
        # derived: VALU issue cycles per SIMD (4 cycles per wave instruction, 1024 SIMDs), cycles the launch took
        # (GRBM_GUI_ACTIVE is summed over the 8 XCDs), hence the VALU-issue utilisation; HBM bytes (FETCH_SIZE x 2 on
        # gfx950, KiB units)
        valu_cycles = m.get("SQ_INSTS_VALU", 0) * 4 / 1024
        launch_cycles = m.get("GRBM_GUI_ACTIVE", 0) / 8
        rows.append({
            "variant": v, "kernel": k, "launches": n,
            **{c: f"{m.get(c, float('nan')):.6g}" for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU",
                                                          "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                                          "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT",
                                                          "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE")},
            "valu_issue_cycles_per_simd": f"{valu_cycles:.6g}", "launch_cycles": f"{launch_cycles:.6g}",
            "valu_utilisation": f"{valu_cycles / launch_cycles:.3f}" if launch_cycles else "",
            "hbm_read_bytes": f"{m.get('FETCH_SIZE', 0) * 2048:.6g}", "hbm_write_bytes": f"{m.get('WRITE_SIZE', 0) * 1024:.6g}",
        })
with open("profiles/r03_sq_counters.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
for r in rows:
    print(r["variant"], r["kernel"][:70], "VALU util", r["valu_utilisation"], "read", r["hbm_read_bytes"], "write", r["hbm_write_bytes"])
