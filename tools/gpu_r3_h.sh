#!/bin/bash
# stage-skipping chunk start-up: tests, then A/B against a -DDXH_NO_STAGE_SKIP build
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_tvd_gpu.py tests/test_parallel_gpu.py tests/test_config5_gpu.py tests/test_numpy2_flow_gpu.py tests/test_curvature_gpu.py -x -q -m gpu > gpurun_out/r3h_tests.log 2>&1 || { tail -30 gpurun_out/r3h_tests.log; exit 1; }
tail -2 gpurun_out/r3h_tests.log
{
echo "== headline (4000 x 8000 x 3 anisotropic, ms per iteration): default = stage-skipping start-up"
python tools/ab_libs.py default tools/_ab/libdarsia_hip_nostageskip.so
echo "== isotropic RGB and gray"
python tools/ab_iso_libs.py default tools/_ab/libdarsia_hip_nostageskip.so
echo "== N = 8 / 4 / 2 slabs of configs[4], default library"
python tools/bench_slab_overlap.py 2>&1 | grep "^{"
echo "== the same, -DDXH_NO_STAGE_SKIP"
DARSIA_HIP_LIB=tools/_ab/libdarsia_hip_nostageskip.so python tools/bench_slab_overlap.py 2>&1 | grep "^{"
} > gpurun_out/r3h_stage_skip_ab.log 2>&1
cat gpurun_out/r3h_stage_skip_ab.log
