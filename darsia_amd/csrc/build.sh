#!/bin/bash
# Build libdarsia_hip.so for gfx950 in-tree (the .so is git-ignored but travels to the GPU box).
# Objects are compiled in parallel (JOBS, default 8): the multi-iteration pipelines are one object per
# (element type, K).  Extra arguments are passed to every compile (e.g. -DDXH_DEV_FAST, -DDXH_ENABLE_PAIR).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${DXH_OUT:-${HERE}/../libdarsia_hip.so}"
OBJ="${DXH_OBJ:-${HERE}/_build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
JOBS="${JOBS:-8}"
mkdir -p "${OBJ}"
# DXH_INCREMENTAL=1 (development): keep objects that are newer than their source, every header and this script, and
# were built with the same extra arguments; the default is a clean build
INCR="${DXH_INCREMENTAL:-0}"
ARGSIG="$*"
if [ "${INCR}" != "1" ] || [ "$(cat "${OBJ}/.args" 2>/dev/null)" != "${ARGSIG}" ]; then
    rm -f "${OBJ}"/*.o "${OBJ}"/*.log
fi
printf '%s' "${ARGSIG}" > "${OBJ}/.args"
NEWEST_DEP="$(ls -t "${HERE}"/*.h "${HERE}/build.sh" | head -1)"
# -ffp-contract=off: every operation rounds once, as in the NumPy evaluation the results are compared with
# -fno-slp-vectorize: v_pk_*_f32 issue at half rate on gfx950, so the SLP vectoriser's packed pairs save nothing
# and the v_mov shuffles that build them cost issue slots (-3 % on the VALU-bound pipelines, measured)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function "$@")

pids=()
names=()
run() {  # run NAME SRC [extra flags]
    local name="$1" src="$2"
    shift 2
    if [ "${INCR}" = "1" ] && [ -f "${OBJ}/${name}.o" ] && [ "${OBJ}/${name}.o" -nt "${HERE}/${src}" ] &&
       [ "${OBJ}/${name}.o" -nt "${NEWEST_DEP}" ]; then
        return 0
    fi
    rm -f "${OBJ}/${name}.o"
    while [ "$(jobs -rp | wc -l)" -ge "${JOBS}" ]; do sleep 0.2; done
    ( "${HIPCC}" "${FLAGS[@]}" "$@" -c "${HERE}/${src}" -o "${OBJ}/${name}.o" > "${OBJ}/${name}.log" 2>&1 ) &
    pids+=($!)
    names+=("${name}")
}
# heaviest first
for k in 6 5 4 3 2; do for part in 0 1; do run "np2_sw${k}_${part}" dxh_np2_sweeps.hip -DDXH_SWEEPS_K=${k} -DDXH_SWEEPS_PART=${part}; done; done   # S Jacobi sweeps per iteration, one pass each
for k in 3 4 2; do
    for cs in 0 1; do   # channel sets: {1, 3} and {2, 4}
        run "multi_f32_k${k}_c${cs}" dxh_multi.hip -DDXH_MULTI_T=float -DDXH_MULTI_K=${k} -DDXH_MULTI_CSET=${cs}
        run "multi_f64_k${k}_c${cs}" dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=${k} -DDXH_MULTI_CSET=${cs}
    done
done
run np2_s dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=1
run np2_q dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=2
run f64p dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=3
run np2_g6 dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=6 -DDXH_MULTI_NP2=4
run np2_s2 dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=2 -DDXH_MULTI_NP2=5
run np2_s3 dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=3 -DDXH_MULTI_NP2=5
run np2_s24 dxh_multi.hip -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -DDXH_MULTI_NP2=1 -DDXH_MULTI_CSET=1
run tvd dxh_tvd.hip
run vol_multi dxh_vol_multi.hip
run chambolle dxh_chambolle.hip
run dense dxh_dense.hip
run solvers dxh_solvers.hip
run bregman_gs dxh_bregman_gs.hip
run comm dxh_comm.hip
fail=0
for i in "${!pids[@]}"; do
    if ! wait "${pids[$i]}"; then
        echo "compile of ${names[$i]} failed:" >&2
        cat "${OBJ}/${names[$i]}.log" >&2
        fail=1
    elif [ -s "${OBJ}/${names[$i]}.log" ]; then
        cat "${OBJ}/${names[$i]}.log" >&2   # warnings
    fi
done
[ "${fail}" -eq 0 ] || exit 1
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC "${OBJ}"/*.o -o "${OUT}"
echo "built ${OUT}"
