#!/bin/bash
# Runs on the GPU box: kernel trace of tools/bench_configs.py (all the non-headline configurations).
set -uo pipefail
TAG="${1:-r01}"
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="${REPO}/gpurun_out/prof_${TAG}_configs"
rm -rf "${OUT}"; mkdir -p "${OUT}"
export TMPDIR=/tmp
cd "${REPO}"
rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}/trace" -- python3 tools/bench_configs.py > "${OUT}/bench_configs.log" 2>&1 || { echo "trace failed"; tail -5 "${OUT}/bench_configs.log"; exit 1; }
find "${OUT}" -name "*kernel_stats.csv" | head -3
tail -40 "${OUT}/bench_configs.log"
