#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + two PMC passes of bench.py; raw output under gpurun_out/.
set -uo pipefail
TAG="${1:-r01}"
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="${REPO}/gpurun_out/prof_${TAG}"
rm -rf "${OUT}"; mkdir -p "${OUT}"
export TMPDIR=/tmp
cd "${REPO}"
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d "${OUT}/trace" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-self-check > "${OUT}/trace_bench.log" 2>&1 || { echo "trace failed"; tail -5 "${OUT}/trace_bench.log"; exit 1; }
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "${OUT}/pmc_fetch" -- python3 bench.py --steps 1 --warmup 0 --iters 10 --no-cpu-baseline --no-self-check > "${OUT}/pmc_fetch.log" 2>&1 || { echo "pmc fetch failed"; tail -5 "${OUT}/pmc_fetch.log"; exit 1; }
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "${OUT}/pmc_write" -- python3 bench.py --steps 1 --warmup 0 --iters 10 --no-cpu-baseline --no-self-check > "${OUT}/pmc_write.log" 2>&1 || { echo "pmc write failed"; tail -5 "${OUT}/pmc_write.log"; exit 1; }
find "${OUT}" -name "*.csv" | head -20
du -sh "${OUT}"
