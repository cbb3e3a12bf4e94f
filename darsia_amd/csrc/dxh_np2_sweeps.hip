// Objects per sweep count and part (build.sh: np2_sw<S>_<part>, -DDXH_SWEEPS_K=S -DDXH_SWEEPS_PART=0|1): the pipeline of ONE outer
// iteration with S Jacobi sweeps (dxh_np2_sweeps.h), C in {1, 3}.  Part 0: float32 images in the numpy2 flow, scalar threshold or
// threshold plane (array-valued mu).  Part 1: float64 images (scalar threshold) and the PLAIN form (S plain sweeps of a Jacobi
// solve on a float64 context; S <= 5).  Two objects per S because the instantiations of S = 6 alone compile for four minutes.
#include "dxh_multi.h"
#include "dxh_np2_sweeps.h"

#if !defined(DXH_SWEEPS_K) || !defined(DXH_SWEEPS_PART)
#error "compile with -DDXH_SWEEPS_K=<sweeps per iteration> -DDXH_SWEEPS_PART=<0|1>"
#endif

namespace dxh {

#if DXH_SWEEPS_PART == 0
template <>
void launch_np2_sweeps_f32<DXH_SWEEPS_K>(int C, bool coupled, bool kqv, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
    constexpr int K = DXH_SWEEPS_K;
#define DXH_SW(C_, CP_)                                                                                       \
    do {                                                                                                      \
        if (kqv) hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, true>), dim3(grid), dim3(256), 0, stream, a);    \
        else hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, false>), dim3(grid), dim3(256), 0, stream, a);       \
    } while (0)
    if (C == 1) { if (coupled) DXH_SW(1, true); else DXH_SW(1, false); }
    else        { if (coupled) DXH_SW(3, true); else DXH_SW(3, false); }
#undef DXH_SW
}
#else
template <>
void launch_np2_sweeps_f64<DXH_SWEEPS_K>(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
    constexpr int K = DXH_SWEEPS_K;
#define DXH_SW(C_, CP_) hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, false, double>), dim3(grid), dim3(256), 0, stream, a)
    if (C == 1) { if (coupled) DXH_SW(1, true); else DXH_SW(1, false); }
    else        { if (coupled) DXH_SW(3, true); else DXH_SW(3, false); }
#undef DXH_SW
}

template <>
void launch_np2_plain_sweeps<DXH_SWEEPS_K>(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#if DXH_SWEEPS_K <= 5
    constexpr int K = DXH_SWEEPS_K;
#define DXH_PL(C_, CP_) hipLaunchKernelGGL((k_sweeps_np2<C_, CP_, K, false, double, true>), dim3(grid), dim3(256), 0, stream, a)
    if (C == 1) { if (coupled) DXH_PL(1, true); else DXH_PL(1, false); }
    else        { if (coupled) DXH_PL(3, true); else DXH_PL(3, false); }
#undef DXH_PL
#else
    (void)C; (void)coupled; (void)a; (void)grid; (void)stream;   // (jacobi_group never asks for more than five plain sweeps per pass)
#endif
}
#endif

}  // namespace dxh
