#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_tvd_gpu.py -x -q -m gpu -k "volume or jacobi or solver or dim3 or vol" > gpurun_out/r3e_tests.log 2>&1 || { tail -20 gpurun_out/r3e_tests.log; exit 1; }
tail -2 gpurun_out/r3e_tests.log
python -m pytest tests/test_solvers_gpu.py -x -q -m gpu > gpurun_out/r3e_tests2.log 2>&1 || { tail -20 gpurun_out/r3e_tests2.log; exit 1; }
tail -2 gpurun_out/r3e_tests2.log
echo "== new sweep" > gpurun_out/r3e_dim3.log
python tools/bench_dim3.py >> gpurun_out/r3e_dim3.log 2>&1
echo "== old sweep (DXH_FUSED_TUNE=262144)" >> gpurun_out/r3e_dim3.log
DXH_FUSED_TUNE=262144 python tools/bench_dim3.py >> gpurun_out/r3e_dim3.log 2>&1
cat gpurun_out/r3e_dim3.log
python tools/np2_timing.py > gpurun_out/r3e_np2.log 2>&1; cat gpurun_out/r3e_np2.log
