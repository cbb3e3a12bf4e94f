#!/bin/bash
# Developer: rebuild only the objects of dxh_np2_pipe.h (np2_s, np2_q, np2_s24, f64p) and relink libdarsia_hip.so.  Extra args -> hipcc.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../darsia_amd/csrc" && pwd)"
F=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function "$@")
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=1 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/np2_s.o" &
p1=$!
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=2 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/np2_q.o" &
p2=$!
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=1 -DDXH_MULTI_CSET=1 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/np2_s24.o" &
p3=$!
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=3 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/f64p.o" &
p4=$!
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=6 -DDXH_MULTI_NP2=4 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/np2_g6.o" &
p5=$!
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=2 -DDXH_MULTI_NP2=5 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/np2_s2.o" &
p6=$!
/opt/rocm/bin/hipcc "${F[@]}" -DDXH_MULTI_T=double -DDXH_MULTI_K=3 -DDXH_MULTI_NP2=5 -c "${HERE}/dxh_multi.hip" -o "${HERE}/_build/np2_s3.o" &
p7=$!
wait $p5
wait $p6
wait $p7
wait $p1
wait $p2
wait $p3
wait $p4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${HERE}"/_build/*.o -o "${DXH_OUT:-${HERE}/../libdarsia_hip.so}"
echo "relinked ${DXH_OUT:-${HERE}/../libdarsia_hip.so}"
