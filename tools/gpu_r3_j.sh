#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_numpy2_flow_gpu.py tests/test_concentration_gpu.py -x -q -m gpu > gpurun_out/r3j_tests.log 2>&1 || { tail -40 gpurun_out/r3j_tests.log; exit 1; }
tail -2 gpurun_out/r3j_tests.log
python tests/fuzz/fuzz_r3.py 3 600 > gpurun_out/r3j_fuzz.log 2>&1 || { tail -20 gpurun_out/r3j_fuzz.log; exit 1; }
tail -1 gpurun_out/r3j_fuzz.log
python -m pytest tests/test_parallel_gpu.py tests/test_tvd_gpu.py -x -q -m gpu > gpurun_out/r3j_tests2.log 2>&1 || { tail -40 gpurun_out/r3j_tests2.log; exit 1; }
tail -2 gpurun_out/r3j_tests2.log
python tools/np2_timing.py > gpurun_out/r3j_np2.log 2>&1; cat gpurun_out/r3j_np2.log
echo "== K = 1 only (DXH_FUSE_K=1)"; DXH_FUSE_K=1 python tools/np2_timing.py 2>&1 | grep numpy2
