#!/bin/bash
# Build libdarsia_hip.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libdarsia_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off: every operation rounds once, as in the NumPy evaluation the results are compared with
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wall -Wno-unused-function \
    "${HERE}/dxh_tvd.hip" "${HERE}/dxh_chambolle.hip" "${HERE}/dxh_dense.hip" -o "${OUT}" "$@"
echo "built ${OUT}"
