#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_tvd_gpu.py tests/test_solvers_gpu.py tests/test_concentration_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r3g_tests.log 2>&1 || { tail -30 gpurun_out/r3g_tests.log; exit 1; }
tail -2 gpurun_out/r3g_tests.log
python tools/ab_iso_libs.py default tools/_ab/libdarsia_hip_isoieee.so > gpurun_out/r3g_iso_ab.log 2>&1; cat gpurun_out/r3g_iso_ab.log
echo "== XCD-contiguous generic stencils (default)" > gpurun_out/r3g_dim3.log
python tools/bench_dim3.py >> gpurun_out/r3g_dim3.log 2>&1
echo "== round-robin block order (DXH_FUSED_TUNE=524288)" >> gpurun_out/r3g_dim3.log
DXH_FUSED_TUNE=524288 python tools/bench_dim3.py >> gpurun_out/r3g_dim3.log 2>&1
cat gpurun_out/r3g_dim3.log
