#!/bin/bash
# Round-3 evidence run (on the GPU box): kernel trace + traffic counters of bench.py, SQ counters of the pipelines that
# changed this round.  Raw output under gpurun_out/; tools/summarize_r03.py condenses into profiles/.
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "${REPO}"
export TMPDIR=/tmp
bash tools/gpu_profile.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -3 gpurun_out/prof_r03.log
for v in aniso iso gray np2 np2arr jacobi3d warp; do
  case $v in
    aniso|iso|gray|np2|np2arr) filt=k_fused;; jacobi3d) filt=k_jacobi_vol;; warp) filt=k_warp;;
  esac
  bash tools/sq_any.sh r03_$v $filt tools/run_variant.py $v > gpurun_out/sq_r03_$v.txt 2>&1
  echo "== $v"; tail -22 gpurun_out/sq_r03_$v.txt | head -40
done
