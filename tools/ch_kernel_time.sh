#!/bin/bash
# kernel-only durations of the Chambolle march for two library builds (rocprofv3 --kernel-trace --stats)
export TMPDIR=/tmp
for lib in "" gpurun_variants/libdarsia_div_ieee.so; do
  rm -rf gpurun_out/chk; mkdir -p gpurun_out/chk
  if [ -n "$lib" ]; then export DARSIA_HIP_LIB=$PWD/$lib; else unset DARSIA_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chk -- python3 tools/run_variant.py chambolle60 > /dev/null 2>&1
  python3 - "$lib" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/chk/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ch_march" in r["Name"]:
            print(sys.argv[1] or "default", r["Name"][:40], r["Calls"], f"{float(r['AverageNs'])/1e3:.1f} us")
PY
done
