#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in np2 np2arr; do
  bash tools/sq_any.sh r03_$v k_fused tools/run_variant.py $v > gpurun_out/sq_r03_$v.txt 2>&1
  echo "== $v"; tail -24 gpurun_out/sq_r03_$v.txt | head -30
done
