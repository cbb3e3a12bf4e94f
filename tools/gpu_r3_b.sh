#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_config5_gpu.py tests/test_parallel_gpu.py -x -q -m gpu > gpurun_out/r3b_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r3b_tests.log
[ $rc -eq 0 ] || exit $rc
echo "== trimmed halos (default)" > gpurun_out/r3b_slab.log
python tools/bench_slab_overlap.py >> gpurun_out/r3b_slab.log 2>&1 || exit 1
echo "== whole slab every launch (DXH_FUSED_TUNE=131072)" >> gpurun_out/r3b_slab.log
DXH_FUSED_TUNE=131072 python tools/bench_slab_overlap.py >> gpurun_out/r3b_slab.log 2>&1 || exit 1
cat gpurun_out/r3b_slab.log
