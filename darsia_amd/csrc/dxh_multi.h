// Launch entry of the K-iterations-per-launch pipelines (k_fused_multi).  The kernels are heavy to compile
// (4 x unrolled, up to 4 stages), so each (element type, K) pair is instantiated in its own object file
// (dxh_multi.hip, compiled with -DDXH_MULTI_T / -DDXH_MULTI_K by build.sh) and the objects build in parallel.
#pragma once
#include <hip/hip_runtime.h>

namespace dxh {

template <typename T>
struct FusedArgs;

// Array-valued coefficients (a.derive != 0) are supported for K = kHetFuseK only.
constexpr int kHetFuseK = 3;

// C in {1, 3} (launch_multi_kernel) or {2, 4} (launch_multi_kernel_c24); `pair` selects the experimental wave-pair kernel (only with -DDXH_ENABLE_PAIR, float, K = 4)
template <typename T, int K>
void launch_multi_kernel(int C, bool coupled, bool iso, bool want_inc, bool pair, const FusedArgs<T>& a, unsigned grid,
                         hipStream_t stream);

// the same for C in {2, 4} (objects built with -DDXH_MULTI_CSET=1)
template <typename T, int K>
void launch_multi_kernel_c24(int C, bool coupled, bool iso, bool want_inc, bool pair, const FusedArgs<T>& a, unsigned grid,
                             hipStream_t stream);

// waves per SIMD the numpy2 pipelines (dxh_np2_pipe.h) are compiled for; the host's chunk-count model uses the same number
#ifndef DXH_NP2_WAVES
#define DXH_NP2_WAVES 3
#endif
#ifndef DXH_F64P_WAVES      // the same pipeline for float64 images (16.6 KB of LDS per wave: two blocks per CU)
#define DXH_F64P_WAVES 2
#endif
#ifndef DXH_F64P_PFD        // prefetch distance of the two-waves-per-SIMD variants: two row steps (-2 % against one, measured)
#define DXH_F64P_PFD 2
#endif
#ifndef DXH_NP2_WAVES_K6    // six iterations per pass (single-channel images): 15.7 KB of LDS per wave, two blocks per CU
#define DXH_NP2_WAVES_K6 2
#endif
#ifndef DXH_NP2_PFD_K6
#define DXH_NP2_PFD_K6 2
#endif
#ifndef DXH_NP2_PREFETCH     // dxh_np2_pipe.h: stage s fetches stage s + 1's LDS operands before its own tail
#define DXH_NP2_PREFETCH 1
#endif
#ifndef DXH_NP2_WAVES_KQV   // threshold-plane variant: three since the thresholds travel down the stages through LDS rows (152 - 167 registers;
#define DXH_NP2_WAVES_KQV 3   // with a register delay line of K x 4 registers it spilled 24 - 108 B at three and ran at two)
#endif
// ... single-channel images stay at two (HBM-bound at 60 B per element: 0.102 against 0.105 ms per iteration at three)
constexpr int np2_kqv_waves(int C) { return C == 1 ? 2 : DXH_NP2_WAVES_KQV; }

// the reference's float32 dtype flow (numpy2 pipelines, dxh_np2_pipe.h; double planes): K = kNp2K for both the scalar-threshold
// and the threshold-plane (kqv: array-valued mu) variant, C in {1, 3}; scalar threshold also for C in {2, 4}.  Three objects
// (build.sh: np2_s, np2_q, np2_s24).
constexpr int kNp2K = 4;
void launch_multi_kernel_np2_scalar(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
// ... two / three iterations per pass (the remainders of an iteration count; objects np2_s2, np2_s3)
void launch_multi_kernel_np2_scalar_k2(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
void launch_multi_kernel_np2_scalar_k3(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
// ... C in {2, 4}, scalar threshold (object np2_s24)
void launch_multi_kernel_np2_scalar_c24(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
// single-channel float32 images, scalar threshold: kNp2GrayK iterations per pass (HBM-bound at K = 4: 52 B per element and
// launch; object np2_g6)
constexpr int kNp2GrayK = 6;
void launch_multi_kernel_np2_gray6(bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
// the same K-stage pipeline with every quantity float64: float64 IMAGES, anisotropic, scalar coefficients, C in {1, 3} (object f64p)
void launch_multi_kernel_f64p(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
// one outer iteration with K Jacobi sweeps per pass: numpy2 flow, scalar omega / ell (mu: scalar or array), or a float64 image
// with scalar coefficients; C in {1, 3} (dxh_np2_sweeps.h;
// specialised in dxh_np2_sweeps.hip for K = kNp2SweepsMin .. kNp2SweepsMax, one object each)
constexpr int kNp2SweepsMin = 2, kNp2SweepsMax = 6;
// waves per SIMD the K-sweep pipeline is compiled for (registers: 128 / 168 / 256; LDS: 21 / 25 - 50 / 54 KB per block)
#ifndef DXH_SWEEPS_4W
#define DXH_SWEEPS_4W 2   // sweep counts up to this run four waves per SIMD (three sweeps: 32 - 56 B of scratch then, 0.97 vs 0.88 ms)
#endif
// the sweeps pass reduces the stopping criterion's increment itself (all but five sweeps in the numpy2 flow: Np2SweepLds::HAS_NORM)
constexpr bool np2_sweeps_fold_norm(int K, bool np2) { return !(np2 && K == 5); }
constexpr int np2_sweep_waves(int K) { return K <= DXH_SWEEPS_4W ? 4 : (K <= 5 ? 3 : 2); }
// ... of the plain-sweeps instantiation (float64 contexts, PLAIN): registers 128 / 168 / 256
constexpr int np2_plain_waves(int K) { return K <= 4 ? 4 : 3; }   // (K = 4: 101 - 111 registers)
template <int K>
void launch_np2_sweeps_f32(int C, bool coupled, bool kqv, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
template <int K>
void launch_np2_sweeps_f64(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
// K plain Jacobi sweeps per pass on a float64 context (a.q0_in = the rs plane, a.u_in / a.u_out the iterate in / out; K <= 5)
template <int K>
void launch_np2_plain_sweeps(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
#define DXH_SW_DECL(K_)                                                                                                                         \
    template <> void launch_np2_sweeps_f32<K_>(int C, bool coupled, bool kqv, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);  \
    template <> void launch_np2_sweeps_f64<K_>(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);            \
    template <> void launch_np2_plain_sweeps<K_>(int C, bool coupled, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);
DXH_SW_DECL(2) DXH_SW_DECL(3) DXH_SW_DECL(4) DXH_SW_DECL(5) DXH_SW_DECL(6)
#undef DXH_SW_DECL
template <int K>
inline void launch_np2_sweeps(int C, bool coupled, bool kqv, bool f64, const FusedArgs<double>& a, unsigned grid, hipStream_t stream) {
    if (f64) launch_np2_sweeps_f64<K>(C, coupled, a, grid, stream);
    else launch_np2_sweeps_f32<K>(C, coupled, kqv, a, grid, stream);
}
void launch_multi_kernel_np2_kqv(int C, bool coupled, bool want_inc, const FusedArgs<double>& a, unsigned grid, hipStream_t stream);

}  // namespace dxh
