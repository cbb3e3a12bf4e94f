#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_tvd_gpu.py -x -q -m gpu -k "volume or march or vol3d or dim3" > gpurun_out/r3i_tests.log 2>&1 || { tail -30 gpurun_out/r3i_tests.log; exit 1; }
tail -2 gpurun_out/r3i_tests.log
{
echo "== four adjacent columns per thread (k_march3d_v, default)"
python tools/bench_dim3.py
echo "== one column at a time (k_march3d, DXH_FUSED_TUNE=1048576)"
DXH_FUSED_TUNE=1048576 python tools/bench_dim3.py
} > gpurun_out/r3i_march_ab.log 2>&1
cat gpurun_out/r3i_march_ab.log
