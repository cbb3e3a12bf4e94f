#!/bin/bash
# Developer: SQ / traffic counters of the kernels one python script launches.
# usage: tools/sq_any.sh TAG KERNEL_SUBSTRING script.py [args...]   -> gpurun_out/sq_TAG/ + summary on stdout
set -uo pipefail
TAG="$1"; FILT="$2"; shift 2
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="${REPO}/gpurun_out/sq_${TAG}"
rm -rf "${OUT}"; mkdir -p "${OUT}"
export TMPDIR=/tmp
cd "${REPO}"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc ${set} --output-format csv -d "${OUT}/p${i}" -- python3 "$@" > "${OUT}/p${i}.log" 2>&1 || { echo "pass ${i} failed"; tail -3 "${OUT}/p${i}.log"; }
done
python3 - "$OUT" "$FILT" <<'PY'
import csv, glob, sys, collections
out, filt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt not in k: continue
        acc[k.split("(")[0][:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
