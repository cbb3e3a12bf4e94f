#!/bin/bash
# Developer: the oracle fuzzers and the device-vs-device slab / band fuzzers on the current library (GPU box).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
S="${1:-401}"
run() { echo "== $*"; timeout -k 10 900 python "$@" 2>&1 | tail -3; }
run tests/fuzz/fuzz_r3.py $S 2500
run tests/fuzz/fuzz_r3.py $((S+1)) 2500
run tests/fuzz/fuzz_tvd.py $((S+2)) 2000
run tests/fuzz/fuzz_tvd2.py $((S+3)) 600
run tests/fuzz/fuzz_others.py $((S+4)) 400
run tests/fuzz/fuzz3.py $((S+5)) 400
run tests/fuzz/fuzz_special.py $((S+6)) 300
run tests/fuzz/fuzz_r4.py $((S+7)) 1500
run tests/fuzz/combos_r4.py
run tools/fuzz_slabs.py
run tools/fuzz_periodic_rccl.py
run tools/fuzz_banded.py
