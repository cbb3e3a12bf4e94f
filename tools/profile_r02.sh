#!/bin/bash
# Round-2 evidence run (on the GPU box): kernel trace + traffic counters of bench.py, SQ counters of the shipped
# pipelines, kernel trace of the other configurations.  Raw output under gpurun_out/; tools/summarize_r02.py condenses.
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "${REPO}"
export TMPDIR=/tmp
bash tools/gpu_profile.sh r02 > gpurun_out/prof_r02.log 2>&1; tail -3 gpurun_out/prof_r02.log
for v in aniso iso het chambolle warp vol3d gray f64; do
  case $v in
    aniso|iso|het|gray|f64) filt=k_fused;; chambolle) filt=k_ch_;; warp) filt=k_warp;; vol3d) filt=k_march3d;;
  esac
  bash tools/sq_any.sh r02_$v $filt tools/run_variant.py $v > gpurun_out/sq_r02_$v.txt 2>&1
  echo "== $v"; tail -22 gpurun_out/sq_r02_$v.txt | head -40
done
bash tools/profile_configs.sh r02 > gpurun_out/prof_r02_configs.log 2>&1; tail -60 gpurun_out/prof_r02_configs.log
