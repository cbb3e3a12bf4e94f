#!/bin/bash
# round-3 checkpoint: whole GPU suite, smoke, N = 1 bench, N = 4 rehearsal of the bench line (ranks share the GPU)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r3_full_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3_full_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
python bench.py > gpurun_out/r3_bench_n1.log 2>&1 || { tail -5 gpurun_out/r3_bench_n1.log; exit 1; }
tail -1 gpurun_out/r3_bench_n1.log | cut -c1-400
DXH_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --steps 1 --warmup 1 > gpurun_out/r3_bench_n4_rehearsal.log 2>&1 || { tail -8 gpurun_out/r3_bench_n4_rehearsal.log; exit 1; }
grep "^{" gpurun_out/r3_bench_n4_rehearsal.log | cut -c1-200; grep -o '"self_check": {[^}]*}' gpurun_out/r3_bench_n4_rehearsal.log
