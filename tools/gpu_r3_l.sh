#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r3l_full_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3l_full_tests.log
[ $rc -eq 0 ] || exit $rc
{
for f in "fuzz_tvd.py 11 300" "fuzz_tvd2.py 12 150" "fuzz_others.py 13 150" "fuzz3.py 14 150" "fuzz_r3.py 15 600"; do
  set -- $f
  echo "== $1 seed $2, $3 cases"; python tests/fuzz/$1 $2 $3 2>&1 | tail -3
done
} > gpurun_out/r3l_fuzz.log 2>&1
cat gpurun_out/r3l_fuzz.log
python bench.py > gpurun_out/r3l_bench.log 2>&1; tail -1 gpurun_out/r3l_bench.log | cut -c1-300
