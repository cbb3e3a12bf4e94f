// One object per sweep count (build.sh: np2_sw2 .. np2_sw6, -DDXH_SWEEPS_K=S): the numpy2 pipeline of ONE outer iteration
// with S Jacobi sweeps (dxh_np2_sweeps.h), C in {1, 3}, scalar threshold or threshold plane (array-valued mu); float64 images: scalar threshold.
#include "dxh_multi.h"
#include "dxh_np2_sweeps.h"

#ifndef DXH_SWEEPS_K
#error "compile with -DDXH_SWEEPS_K=<sweeps per iteration>"
#endif

namespace dxh {

template <>
void launch_np2_sweeps<DXH_SWEEPS_K>(int C, bool coupled, bool kqv, bool f64, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
    constexpr int K = DXH_SWEEPS_K;
#define DXH_SW(C_, CP_)                                                                                       \
    do {                                                                                                      \
        if (f64) hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, false, double>), dim3(grid), dim3(256), 0, stream, a);   \
        else if (kqv) hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, true>), dim3(grid), dim3(256), 0, stream, a);    \
        else hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, false>), dim3(grid), dim3(256), 0, stream, a);       \
    } while (0)
    if (C == 1) { if (coupled) DXH_SW(1, true); else DXH_SW(1, false); }
    else        { if (coupled) DXH_SW(3, true); else DXH_SW(3, false); }
#undef DXH_SW
}

template <>
void launch_np2_plain_sweeps<DXH_SWEEPS_K>(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
    constexpr int K = DXH_SWEEPS_K;
#define DXH_PL(C_, CP_) hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, false, double, true>), dim3(grid), dim3(256), 0, stream, a)
    if (C == 1) { if (coupled) DXH_PL(1, true); else DXH_PL(1, false); }
    else        { if (coupled) DXH_PL(3, true); else DXH_PL(3, false); }
#undef DXH_PL
}

}  // namespace dxh
