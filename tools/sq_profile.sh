#!/bin/bash
# Developer: SQ counters of one configuration of the TV sweep (env knobs exported by the caller).
# usage: tools/sq_profile.sh TAG   -> gpurun_out/sq_TAG/*.csv + a summary on stdout
set -uo pipefail
TAG="${1:-x}"
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="${REPO}/gpurun_out/sq_${TAG}"
rm -rf "${OUT}"; mkdir -p "${OUT}"
export TMPDIR=/tmp
cd "${REPO}"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc ${set} --output-format csv -d "${OUT}/p${i}" -- python3 tools/run_k.py 8 > "${OUT}/p${i}.log" 2>&1 || { echo "pass ${i} failed"; tail -3 "${OUT}/p${i}.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_fused" not in k: continue
        acc[k.split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
