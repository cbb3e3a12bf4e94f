#!/bin/bash
# Developer: rebuild only the objects of dxh_np2_sweeps.h (np2_sw2 .. np2_sw6) and dxh_tvd.hip, relink libdarsia_hip.so.
# Extra args -> hipcc; DXH_OUT = another output library (A/B variants).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../darsia_amd/csrc" && pwd)"
F=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function "$@")
pids=()
for k in 6 5 4 3 2; do
    for part in 0 1; do
        while [ "$(jobs -rp | wc -l)" -ge 8 ]; do sleep 0.2; done
        /opt/rocm/bin/hipcc "${F[@]}" -DDXH_SWEEPS_K=${k} -DDXH_SWEEPS_PART=${part} -c "${HERE}/dxh_np2_sweeps.hip" -o "${HERE}/_build/np2_sw${k}_${part}.o" &
        pids+=($!)
    done
done
/opt/rocm/bin/hipcc "${F[@]}" -c "${HERE}/dxh_tvd.hip" -o "${HERE}/_build/tvd.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${HERE}"/_build/*.o -o "${DXH_OUT:-${HERE}/../libdarsia_hip.so}"
echo "relinked ${DXH_OUT:-${HERE}/../libdarsia_hip.so}"
