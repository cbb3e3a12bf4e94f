#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_tvd_gpu.py tests/test_solvers_gpu.py tests/test_parallel_gpu.py -x -q -m gpu > gpurun_out/r3m_tests.log 2>&1 || { tail -30 gpurun_out/r3m_tests.log; exit 1; }
tail -2 gpurun_out/r3m_tests.log
python tests/fuzz/fuzz_r3.py 21 300 2>&1 | tail -2
python tests/fuzz/fuzz_tvd.py 22 300 2>&1 | tail -2
{ echo "== scalar planes skipped (default)"; python tools/bench_dim3.py 2>&1 | grep "array-valued"; echo "== all four planes (DXH_FUSED_TUNE=2097152)"; DXH_FUSED_TUNE=2097152 python tools/bench_dim3.py 2>&1 | grep "array-valued"; } > gpurun_out/r3m_het3d.log; cat gpurun_out/r3m_het3d.log
