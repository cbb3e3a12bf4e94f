#!/bin/bash
# round 3, first GPU call: new configs[4] tests, slab tests, N=2 rehearsal of the bench line, N=1 bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_config5_gpu.py tests/test_parallel_gpu.py -x -q -m gpu > gpurun_out/r3a_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r3a_tests.log
[ $rc -eq 0 ] || exit $rc
DXH_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3a_bench_n2_rehearsal.log 2>&1
rc=$?
tail -3 gpurun_out/r3a_bench_n2_rehearsal.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r3a_bench_n1.log 2>&1
rc=$?
tail -2 gpurun_out/r3a_bench_n1.log
exit $rc
