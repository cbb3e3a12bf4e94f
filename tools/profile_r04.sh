#!/bin/bash
# Round-4 evidence run (on the GPU box): kernel trace + traffic counters of bench.py (BOTH dtype flows run in it), SQ counters of
# the pipelines that changed this round, a marker trace (roctx ranges) of a banded whole call.  Raw output under gpurun_out/;
# tools/summarize_r04.py condenses into profiles/.
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "${REPO}"
export TMPDIR=/tmp
bash tools/gpu_profile.sh r04 > gpurun_out/prof_r04.log 2>&1; tail -3 gpurun_out/prof_r04.log
for v in aniso np2 np2arr np2gray; do
  bash tools/sq_any.sh r04_$v k_fused tools/run_variant.py $v > gpurun_out/sq_r04_$v.txt 2>&1
  echo "== $v"; tail -20 gpurun_out/sq_r04_$v.txt | head -24
done
# roctx ranges: the default-flow whole call on a host image (row bands)
rm -rf gpurun_out/marker_r04; mkdir -p gpurun_out/marker_r04
DXH_ROCTX=1 timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d gpurun_out/marker_r04 -- python3 tools/run_variant.py banded > gpurun_out/marker_r04.log 2>&1 || tail -5 gpurun_out/marker_r04.log
find gpurun_out/marker_r04 -name "*marker*csv" | head -3
