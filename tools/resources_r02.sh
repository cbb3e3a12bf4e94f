#!/bin/bash
# Register / LDS / scratch / occupancy lines of the shipped kernels (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
set -euo pipefail
cd "$(dirname "$0")/../darsia_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage"
T=$(mktemp -d)
( /opt/rocm/bin/hipcc $F -DDXH_MULTI_T=float -DDXH_MULTI_K=4 -c dxh_multi.hip -o $T/a.o 2> $T/k4f.txt ) &
( /opt/rocm/bin/hipcc $F -DDXH_MULTI_T=float -DDXH_MULTI_K=3 -c dxh_multi.hip -o $T/b.o 2> $T/k3f.txt ) &
( /opt/rocm/bin/hipcc $F -DDXH_MULTI_T=double -DDXH_MULTI_K=4 -c dxh_multi.hip -o $T/c.o 2> $T/k4d.txt ) &
( /opt/rocm/bin/hipcc $F -c dxh_chambolle.hip -o $T/d.o 2> $T/ch.txt ) &
( /opt/rocm/bin/hipcc $F -c dxh_dense.hip -o $T/e.o 2> $T/dense.txt ) &
( /opt/rocm/bin/hipcc $F -c dxh_tvd.hip -o $T/f.o 2> $T/tvd.txt ) &
wait
R=../../tools/resource_usage.py
{
echo "# kernel resource usage, gfx950 (hipcc -Rpass-analysis=kernel-resource-usage); occ = waves per SIMD the registers allow"
echo "## K = 4 pipelines, float (headline: <float, 3, true, false, 4, false, ...>; isotropic: <float, 3, true, true, 4, ...>; gray: <float, 1, false, false, 4, ...>; the 6th argument is the eps variant)"
python3 $R $T/k4f.txt "k_fused_multi<float, 3, true, false, 4"; python3 $R $T/k4f.txt "k_fused_multi<float, 3, true, true, 4"; python3 $R $T/k4f.txt "k_fused_multi<float, 1, false, false, 4"
echo "## K = 3 pipelines with array-valued coefficients: general (..., true, false>) and mu-only / per-element threshold (..., false, true>)"
python3 $R $T/k3f.txt "k_fused_multi<float, 3, true, false, 3" | grep -E "true, false>|false, true>" || true
echo "## K = 4, double, gray"
python3 $R $T/k4d.txt "k_fused_multi<double, 1, false, false, 4"
echo "## Chambolle"
python3 $R $T/ch.txt "k_ch_march"
echo "## dense"
python3 $R $T/dense.txt "k_warp"; python3 $R $T/dense.txt "k_tps"; python3 $R $T/dense.txt "k_affine_color"
echo "## dim = 3"
python3 $R $T/tvd.txt "k_march3d"; python3 $R $T/tvd.txt "k_fused3d"
echo "## K = 1 sweep (float RGB)"
python3 $R $T/tvd.txt "k_fused<float, 3, true, false, false, true, true, true>"
} > ../../profiles/r02_resources.txt
rm -rf $T
cat ../../profiles/r02_resources.txt
