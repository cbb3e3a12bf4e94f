// K Jacobi sweeps of a volume per pass (dxh_vol_multi.h), float32 and float64, K = 2, 3.
#include "dxh_vol_multi.h"

namespace dxh {

template <typename T>
void launch_jacobi_vol_multi(int K, const VolMultiArgs<T>& a, unsigned grid, hipStream_t stream) {
    if (a.dsc_plane) {   // per-element divisor: two sweeps per pass (registers)
        hipLaunchKernelGGL((k_jacobi_vol_multi<T, 2, true>), dim3(grid), dim3(VolMultiGeom<T, 2, true>::NT), 0, stream, a);
        return;
    }
    if (K == 2) hipLaunchKernelGGL((k_jacobi_vol_multi<T, 2, false>), dim3(grid), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((k_jacobi_vol_multi<T, 3, false>), dim3(grid), dim3(512), 0, stream, a);
}
template void launch_jacobi_vol_multi<float>(int, const VolMultiArgs<float>&, unsigned, hipStream_t);
template void launch_jacobi_vol_multi<double>(int, const VolMultiArgs<double>&, unsigned, hipStream_t);

}  // namespace dxh
