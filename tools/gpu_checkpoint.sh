#!/bin/bash
# One gpurun call that says whether the tree is healthy: whole GPU suite, smoke, every fuzzer (short), the N = 1 bench line
# and a REHEARSAL of the N = 4 bench line (four ranks sharing the one GPU over the host-staged transport; not a number).
#   gpurun --timeout 1200 -- 'bash tools/gpu_checkpoint.sh'
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/checkpoint_tests.log 2>&1; rc=$?
tail -3 gpurun_out/checkpoint_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
{
for f in "fuzz_tvd.py 11 300" "fuzz_tvd2.py 12 150" "fuzz_others.py 13 150" "fuzz3.py 14 150" "fuzz_r3.py 15 600"; do
  set -- $f
  echo "== $1 seed $2, $3 cases"; python tests/fuzz/$1 $2 $3 2>&1 | tail -1
done
} > gpurun_out/checkpoint_fuzz.log 2>&1
cat gpurun_out/checkpoint_fuzz.log
grep -q MISMATCH gpurun_out/checkpoint_fuzz.log && exit 1
python bench.py > gpurun_out/checkpoint_bench_n1.log 2>&1 || { tail -5 gpurun_out/checkpoint_bench_n1.log; exit 1; }
tail -1 gpurun_out/checkpoint_bench_n1.log | cut -c1-330
DXH_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --steps 1 --warmup 1 > gpurun_out/checkpoint_bench_n4_rehearsal.log 2>&1 || { tail -8 gpurun_out/checkpoint_bench_n4_rehearsal.log; exit 1; }
grep -o '"self_check": {[^}]*}' gpurun_out/checkpoint_bench_n4_rehearsal.log
