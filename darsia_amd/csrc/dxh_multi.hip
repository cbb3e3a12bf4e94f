// One object per (DXH_MULTI_T, DXH_MULTI_K, DXH_MULTI_CSET): instantiates k_fused_multi<T, C, COUPLED, ISO, K, NORM> for
// C in {1, 3} (channel set 0: scalar and RGB images) or C in {2, 4} (set 1) and defines the launch switch declared in
// dxh_multi.h.  gfx950 only.
#include <type_traits>

#include "dxh_kernels.h"
#include "dxh_multi.h"

#ifndef DXH_MULTI_T
#error "compile with -DDXH_MULTI_T=float|double -DDXH_MULTI_K=2|3|4 (see build.sh)"
#endif

#ifndef DXH_MULTI_CSET
#define DXH_MULTI_CSET 0
#endif

#ifndef DXH_HET_K
#define DXH_HET_K 3  // the K for which the heterogeneous-coefficient pipeline is instantiated (dxh_multi.h)
#endif

#ifdef DXH_MULTI_NP2
// ---- the reference's float32 dtype flow (PipeNp2, dxh_np2_pipe.h): double planes, float32-valued state in float rings;
// C in {1, 3}, K = kNp2K; -DDXH_MULTI_NP2=1: scalar threshold, -DDXH_MULTI_NP2=2: per-element threshold plane (array-valued
// mu) -- one object each (build.sh: np2_s, np2_q)
#include "dxh_np2_pipe.h"
namespace dxh {
#if DXH_MULTI_NP2 == 5
// two / three iterations per pass, scalar threshold, C in {1, 3}: the remainders of an iteration count (objects np2_s2, np2_s3)
#if DXH_MULTI_K == 2
void launch_multi_kernel_np2_scalar_k2(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#else
void launch_multi_kernel_np2_scalar_k3(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#endif
    constexpr int K = DXH_MULTI_K;
    static_assert(K == 2 || K == 3, "remainder pipelines");
#define DXH_R(C_, CP_)                                                                                                       \
    do {                                                                                                                     \
        if (want_inc) hipLaunchKernelGGL((k_fused_multi_np2<C_, CP_, K, true, false>), dim3(grid), dim3(256), 0, stream, a);  \
        else hipLaunchKernelGGL((k_fused_multi_np2<C_, CP_, K, false, false>), dim3(grid), dim3(256), 0, stream, a);          \
    } while (0)
    if (C == 1) { if (coupled) DXH_R(1, true); else DXH_R(1, false); }
    else        { if (coupled) DXH_R(3, true); else DXH_R(3, false); }
#undef DXH_R
}
}  // namespace dxh
#elif DXH_MULTI_NP2 == 4
void launch_multi_kernel_np2_gray6(bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
    constexpr int K = kNp2GrayK;
#define DXH_G6(CP_)                                                                                                         \
    do {                                                                                                                    \
        if (want_inc) hipLaunchKernelGGL((k_fused_multi_np2<1, CP_, K, true, false>), dim3(grid), dim3(256), 0, stream, a);  \
        else hipLaunchKernelGGL((k_fused_multi_np2<1, CP_, K, false, false>), dim3(grid), dim3(256), 0, stream, a);          \
    } while (0)
    if (coupled) DXH_G6(true); else DXH_G6(false);
#undef DXH_G6
}
}  // namespace dxh
#else
#if DXH_MULTI_NP2 == 3
#define DXH_NP2_KQV false
#define DXH_NP2_F double
void launch_multi_kernel_f64p(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#elif DXH_MULTI_NP2 == 2
#define DXH_NP2_KQV true
void launch_multi_kernel_np2_kqv(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#elif DXH_MULTI_CSET == 1
#define DXH_NP2_KQV false
void launch_multi_kernel_np2_scalar_c24(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#else
#define DXH_NP2_KQV false
void launch_multi_kernel_np2_scalar(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
#endif
    constexpr int K = kNp2K;
#ifndef DXH_NP2_F
#define DXH_NP2_F float
#endif
#define DXH_NP2_K(C_, CP_)                                                                                                          \
    do {                                                                                                                            \
        if (want_inc) hipLaunchKernelGGL((k_fused_multi_np2<C_, CP_, K, true, DXH_NP2_KQV, true, DXH_NP2_F>), dim3(grid), dim3(256), 0, stream, a);  \
        else hipLaunchKernelGGL((k_fused_multi_np2<C_, CP_, K, false, DXH_NP2_KQV, true, DXH_NP2_F>), dim3(grid), dim3(256), 0, stream, a);          \
    } while (0)
#if DXH_MULTI_CSET == 1
    if (C == 2) { if (coupled) DXH_NP2_K(2, true); else DXH_NP2_K(2, false); }
    else        { if (coupled) DXH_NP2_K(4, true); else DXH_NP2_K(4, false); }
#else
    if (C == 1) { if (coupled) DXH_NP2_K(1, true); else DXH_NP2_K(1, false); }
    else        { if (coupled) DXH_NP2_K(3, true); else DXH_NP2_K(3, false); }
#endif
#undef DXH_NP2_K
}
}  // namespace dxh
#endif  // DXH_MULTI_NP2 == 4
#else

namespace dxh {

namespace {

template <typename T, int C, bool COUPLED, bool ISO, int K>
void launch_one(bool want_inc, bool pair, const FusedArgs<T>& a, unsigned grid, hipStream_t stream) {
    if constexpr (K == DXH_HET_K) {
#if DXH_MULTI_CSET == 0   // scalar and RGB images only (the host never sets kqv for 2 / 4 channels: compile time)
        if (a.kqv) {  // array-valued mu with scalar omega / ell: the scalar pipeline with a per-element threshold
            if (want_inc)
                hipLaunchKernelGGL((k_fused_multi<T, C, COUPLED, ISO, K, true, true, false, true>), dim3(grid), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL((k_fused_multi<T, C, COUPLED, ISO, K, false, true, false, true>), dim3(grid), dim3(256), 0, stream, a);
            return;
        }
#endif
        if (a.derive) {  // array-valued coefficients
            if (want_inc)
                hipLaunchKernelGGL((k_fused_multi<T, C, COUPLED, ISO, K, true, true, true>), dim3(grid), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL((k_fused_multi<T, C, COUPLED, ISO, K, false, true, true>), dim3(grid), dim3(256), 0, stream, a);
            return;
        }
    }
#ifdef DXH_ENABLE_PAIR
    if constexpr (K == 4 && std::is_same<T, float>::value) {
        if (pair) {
            if (want_inc)
                hipLaunchKernelGGL((k_fused_pair<T, C, COUPLED, ISO, true>), dim3(grid), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL((k_fused_pair<T, C, COUPLED, ISO, false>), dim3(grid), dim3(256), 0, stream, a);
            return;
        }
    }
#endif
    (void)pair;
    if (want_inc)
        hipLaunchKernelGGL((k_fused_multi<T, C, COUPLED, ISO, K, true>), dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((k_fused_multi<T, C, COUPLED, ISO, K, false>), dim3(grid), dim3(256), 0, stream, a);
}

template <typename T, int C, int K>
void launch_c(bool coupled, bool iso, bool want_inc, bool pair, const FusedArgs<T>& a, unsigned grid, hipStream_t stream) {
#ifdef DXH_DEV_FAST  // developer builds: one instantiation only (coupled, anisotropic)
    (void)coupled; (void)iso;
    launch_one<T, C, true, false, K>(want_inc, pair, a, grid, stream);
#else
    if (coupled) {
        if (iso) launch_one<T, C, true, true, K>(want_inc, pair, a, grid, stream);
        else launch_one<T, C, true, false, K>(want_inc, pair, a, grid, stream);
    } else {
        if (iso) launch_one<T, C, false, true, K>(want_inc, pair, a, grid, stream);
        else launch_one<T, C, false, false, K>(want_inc, pair, a, grid, stream);
    }
#endif
}

}  // namespace

#if DXH_MULTI_CSET == 0
template <typename T, int K>
void launch_multi_kernel(int C, bool coupled, bool iso, bool want_inc, bool pair, const FusedArgs<T>& a, unsigned grid,
                         hipStream_t stream) {
    if (C == 1) launch_c<T, 1, K>(coupled, iso, want_inc, pair, a, grid, stream);
    else launch_c<T, 3, K>(coupled, iso, want_inc, pair, a, grid, stream);
}

template void launch_multi_kernel<DXH_MULTI_T, DXH_MULTI_K>(int, bool, bool, bool, bool, const FusedArgs<DXH_MULTI_T>&, unsigned,
                                                            hipStream_t);
#else
template <typename T, int K>
void launch_multi_kernel_c24(int C, bool coupled, bool iso, bool want_inc, bool pair, const FusedArgs<T>& a, unsigned grid,
                             hipStream_t stream) {
    if (C == 2) launch_c<T, 2, K>(coupled, iso, want_inc, pair, a, grid, stream);
    else launch_c<T, 4, K>(coupled, iso, want_inc, pair, a, grid, stream);
}

template void launch_multi_kernel_c24<DXH_MULTI_T, DXH_MULTI_K>(int, bool, bool, bool, bool, const FusedArgs<DXH_MULTI_T>&,
                                                                unsigned, hipStream_t);
#endif

}  // namespace dxh
#endif  // DXH_MULTI_NP2
