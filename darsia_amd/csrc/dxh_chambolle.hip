// Chambolle's TV projection algorithm (the arithmetic of DarSIA's default TVD() method, which the
// reference delegates to skimage.restoration.denoise_tv_chambolle without a channel axis,
// restoration/tvd.py:77-83: colour images are denoised as 3-D volumes).
//
// Per iteration (oracle/chambolle_numpy.py states the algorithm and its source):
//   k_ch_out : d = -div p, out = image + d, partial sums of d^2
//   k_ch_p   : g = forward differences of out, n = |g|, partial sums of n, p = (p - tau g) / (1 + n tau / w)
// Energies are reduced deterministically (per-block partials, then one block) in double; the stop test
// |E_prev - E| < eps E_0 runs on the host once per iteration, as in the reference.
#include <cmath>
#include <vector>

#include "dxh_kernels.h"

namespace dxh {

template <typename T>
struct ChArgs {
    Geom g;
    const T* f;
    T* out;
    T* p[3];
    T tau, tau_over_w;
    double* partial;
};

template <int NB>
__device__ __forceinline__ void block_sum_store(double v, double* partial) {
    __shared__ double sm[NB / kWave];
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) v += __shfl_down(v, s, kWave);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < NB / kWave; ++w) s += sm[w];
        partial[blockIdx.x] = s;
    }
}

template <typename T, int NDIM>
__global__ void __launch_bounds__(256) k_ch_out(ChArgs<T> a) {
    ElemIdx e;
    const bool in = elem_index(a.g, e);
    double sq = 0.0;
    if (in) {
        T s = a.p[0][e.off];
#pragma unroll
        for (int ax = 1; ax < NDIM; ++ax) s = s + a.p[ax][e.off];
        T d = -s;
        if (e.row > 0) d = d + a.p[0][e.off - a.g.pitch];
        if (e.i1 > 0) d = d + a.p[1][e.off - a.g.n2];
        if constexpr (NDIM == 3) {
            if (e.i2 > 0) d = d + a.p[2][e.off - 1];
        }
        a.out[e.off] = a.f[e.off] + d;
        sq = (double)d * (double)d;
    }
    block_sum_store<256>(sq, a.partial);
}

template <typename T, int NDIM>
__global__ void __launch_bounds__(256) k_ch_p(ChArgs<T> a) {
    ElemIdx e;
    const bool in = elem_index(a.g, e);
    double nsum = 0.0;
    if (in) {
        const T o = a.out[e.off];
        T g[NDIM];
        g[0] = e.row < a.g.n0 - 1 ? a.out[e.off + a.g.pitch] - o : T(0);
        g[1] = e.i1 < a.g.n1 - 1 ? a.out[e.off + a.g.n2] - o : T(0);
        if constexpr (NDIM == 3) g[2] = e.i2 < a.g.n2 - 1 ? a.out[e.off + 1] - o : T(0);
        T sq = g[0] * g[0];
#pragma unroll
        for (int ax = 1; ax < NDIM; ++ax) sq = sq + g[ax] * g[ax];
        T n = t_sqrt(sq);
        nsum = (double)n;
        n = n * a.tau_over_w;
        n = n + T(1);
#pragma unroll
        for (int ax = 0; ax < NDIM; ++ax) a.p[ax][e.off] = (a.p[ax][e.off] - a.tau * g[ax]) / n;
    }
    block_sum_store<256>(nsum, a.partial);
}

// One whole iteration in a single pass: out = image - div p is evaluated on the fly at x and at its
// upper neighbours (values come through the caches), p is double buffered.  HBM traffic per element:
// read image + ndim planes, write ndim planes (7 values for a colour image) instead of 12 for the
// two-kernel form.  `out` itself is only materialised once, after the loop (k_ch_out).
template <typename T, int NDIM>
struct ChFusedArgs {
    Geom g;
    const T* f;
    const T* pin[3];
    T* pout[3];
    T tau, tau_over_w;
    double* partial_d2;
    double* partial_n;
};

template <typename T, int NDIM>
__device__ __forceinline__ T ch_d_at(const ChFusedArgs<T, NDIM>& a, int64_t off, int64_t row, int64_t i1, int64_t i2) {
    T s = a.pin[0][off];
#pragma unroll
    for (int ax = 1; ax < NDIM; ++ax) s = s + a.pin[ax][off];
    T d = -s;
    if (row > 0) d = d + a.pin[0][off - a.g.pitch];
    if (i1 > 0) d = d + a.pin[1][off - a.g.n2];
    if constexpr (NDIM == 3) {
        if (i2 > 0) d = d + a.pin[2][off - 1];
    }
    return d;
}

template <typename T, int NDIM>
__global__ void __launch_bounds__(256) k_ch_fused(ChFusedArgs<T, NDIM> a) {
    __shared__ double sm2[256 / kWave];
    ElemIdx e;
    const bool in = elem_index(a.g, e);
    double sq = 0.0, nsum = 0.0;
    if (in) {
        const T d = ch_d_at<T, NDIM>(a, e.off, e.row, e.i1, e.i2);
        const T o = a.f[e.off] + d;
        sq = (double)d * (double)d;
        T g[NDIM];
        if (e.row < a.g.n0 - 1) {
            const int64_t off = e.off + a.g.pitch;
            g[0] = (a.f[off] + ch_d_at<T, NDIM>(a, off, e.row + 1, e.i1, e.i2)) - o;
        } else {
            g[0] = T(0);
        }
        if (e.i1 < a.g.n1 - 1) {
            const int64_t off = e.off + a.g.n2;
            g[1] = (a.f[off] + ch_d_at<T, NDIM>(a, off, e.row, e.i1 + 1, e.i2)) - o;
        } else {
            g[1] = T(0);
        }
        if constexpr (NDIM == 3) {
            if (e.i2 < a.g.n2 - 1) {
                const int64_t off = e.off + 1;
                g[2] = (a.f[off] + ch_d_at<T, NDIM>(a, off, e.row, e.i1, e.i2 + 1)) - o;
            } else {
                g[2] = T(0);
            }
        }
        T s2 = g[0] * g[0];
#pragma unroll
        for (int ax = 1; ax < NDIM; ++ax) s2 = s2 + g[ax] * g[ax];
        T n = t_sqrt(s2);
        nsum = (double)n;
        n = n * a.tau_over_w;
        n = n + T(1);
#pragma unroll
        for (int ax = 0; ax < NDIM; ++ax) a.pout[ax][e.off] = (a.pin[ax][e.off] - a.tau * g[ax]) / n;
    }
    // two deterministic block sums
    block_sum_store<256>(sq, a.partial_d2);
    __syncthreads();
    (void)sm2;
    block_sum_store<256>(nsum, a.partial_n);
}

template <typename T>
static int chambolle_t(int device, int ndim, int64_t n0, int64_t n1, int64_t n2, const void* img, void* out,
                       double weight, double eps, int max_iter, int* iters_done) {
    DXH_HIP(hipSetDevice(device));
    Geom g{};
    g.n0 = n0; g.n1 = n1; g.n2 = n2;
    g.L = n1 * n2;
    g.pitch = (g.L + 31) / 32 * 32;
    g.ndim = ndim; g.dim = ndim; g.nsum_axes = ndim;
    const size_t plane = (size_t)g.pitch * n0 * sizeof(T);
    const int64_t bpr = (g.L + 255) / 256;
    const unsigned nblocks = (unsigned)(bpr * n0);
    hipStream_t st = nullptr;
    void* bufs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // f, out, p0..p2
    double *partial = nullptr, *red = nullptr, *stage2 = nullptr;
    std::vector<void*> extra;
    int rc = DXH_OK;
    auto cleanup = [&]() {
        for (void* b : bufs) if (b) (void)hipFree(b);
        if (partial) (void)hipFree(partial);
        if (red) (void)hipFree(red);
        if (stage2) (void)hipFree(stage2);
        for (void* b : extra) if (b) (void)hipFree(b);
        if (st) (void)hipStreamDestroy(st);
    };
    auto body = [&]() -> int {
        DXH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (int i = 0; i < 2 + ndim; ++i) {
            DXH_HIP(hipMalloc(&bufs[i], plane));
            DXH_HIP(hipMemsetAsync(bufs[i], 0, plane, st));
        }
        DXH_HIP(hipMalloc((void**)&partial, sizeof(double) * nblocks));
        DXH_HIP(hipMalloc((void**)&red, sizeof(double)));
        DXH_HIP(hipMalloc((void**)&stage2, sizeof(double) * 256));
        DXH_HIP(hipMemcpy2DAsync(bufs[0], g.pitch * sizeof(T), img, g.L * sizeof(T), g.L * sizeof(T), n0,
                                 hipMemcpyHostToDevice, st));
        DXH_HIP(hipMemcpyAsync(bufs[1], bufs[0], plane, hipMemcpyDeviceToDevice, st));  // out = image (iteration 0)
        // p is double buffered: pbuf[ax][0/1]; `out` is only written after the loop
        void* p2[3] = {nullptr, nullptr, nullptr};
        for (int ax = 0; ax < ndim; ++ax) {
            DXH_HIP(hipMalloc(&p2[ax], plane));
            extra.push_back(p2[ax]);
            DXH_HIP(hipMemsetAsync(p2[ax], 0, plane, st));
        }
        double* partial_n = nullptr;
        DXH_HIP(hipMalloc((void**)&partial_n, sizeof(double) * nblocks));
        extra.push_back(partial_n);
        const double tau = 1.0 / (2.0 * ndim);
        auto reduce = [&](double* part, double* host) -> int {
            hipLaunchKernelGGL(k_sum_partials, dim3(256), dim3(256), 0, st, part, (int)nblocks, stage2);
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, stage2, 256, red);
            DXH_HIP(hipGetLastError());
            DXH_HIP(hipMemcpyAsync(host, red, sizeof(double), hipMemcpyDeviceToHost, st));
            DXH_HIP(hipStreamSynchronize(st));
            return DXH_OK;
        };
        const double size = (double)n0 * (double)n1 * (double)n2;
        double e_init = 0.0, e_prev = 0.0;
        int it = 0, cur = 0;       // p[cur] is the input of the next iteration
        int out_from = -1;         // which p buffer the returned `out` derives from (-1: out = image)
        auto pbuf = [&](int ax, int which) -> T* { return (T*)(which == 0 ? bufs[2 + ax] : p2[ax]); };
        while (it < max_iter) {
            double sum_d2 = 0.0, sum_n = 0.0;
            int r;
            if (ndim == 2) {
                ChFusedArgs<T, 2> fa;
                fa.g = g; fa.f = (const T*)bufs[0];
                for (int ax = 0; ax < 3; ++ax) { fa.pin[ax] = pbuf(ax < 2 ? ax : 0, cur); fa.pout[ax] = pbuf(ax < 2 ? ax : 0, cur ^ 1); }
                fa.tau = (T)tau; fa.tau_over_w = (T)(tau / weight); fa.partial_d2 = partial; fa.partial_n = partial_n;
                hipLaunchKernelGGL((k_ch_fused<T, 2>), dim3(nblocks), dim3(256), 0, st, fa);
            } else {
                ChFusedArgs<T, 3> fa;
                fa.g = g; fa.f = (const T*)bufs[0];
                for (int ax = 0; ax < 3; ++ax) { fa.pin[ax] = pbuf(ax, cur); fa.pout[ax] = pbuf(ax, cur ^ 1); }
                fa.tau = (T)tau; fa.tau_over_w = (T)(tau / weight); fa.partial_d2 = partial; fa.partial_n = partial_n;
                hipLaunchKernelGGL((k_ch_fused<T, 3>), dim3(nblocks), dim3(256), 0, st, fa);
            }
            DXH_HIP(hipGetLastError());
            if ((r = reduce(partial, &sum_d2)) || (r = reduce(partial_n, &sum_n))) return r;
            out_from = cur;        // the iteration just entered computed its `out` from p[cur]
            const double energy = (sum_d2 + weight * sum_n) / size;
            if (it == 0) {
                e_init = e_prev = energy;
            } else {
                if (std::fabs(e_prev - energy) < eps * e_init) break;
                e_prev = energy;
            }
            cur ^= 1;
            ++it;
        }
        if (out_from >= 0) {
            ChArgs<T> a;
            a.g = g;
            a.f = (const T*)bufs[0];
            a.out = (T*)bufs[1];
            for (int i = 0; i < 3; ++i) a.p[i] = pbuf(i < ndim ? i : 0, out_from);
            a.tau = (T)tau;
            a.tau_over_w = (T)(tau / weight);
            a.partial = partial;
            if (ndim == 2) hipLaunchKernelGGL((k_ch_out<T, 2>), dim3(nblocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_ch_out<T, 3>), dim3(nblocks), dim3(256), 0, st, a);
            DXH_HIP(hipGetLastError());
        }
        if (iters_done) *iters_done = it;
        DXH_HIP(hipMemcpy2DAsync(out, g.L * sizeof(T), bufs[1], g.pitch * sizeof(T), g.L * sizeof(T), n0,
                                 hipMemcpyDeviceToHost, st));
        DXH_HIP(hipStreamSynchronize(st));
        return DXH_OK;
    };
    rc = body();
    cleanup();
    return rc;
}

}  // namespace dxh

extern "C" int dxh_chambolle(int device, int dtype, int ndim, int64_t n0, int64_t n1, int64_t n2, const void* img,
                             void* out, double weight, double eps, int max_num_iter, int* iters_done) {
    using namespace dxh;
    DXH_REQUIRE(img && out, "null argument");
    DXH_REQUIRE(ndim == 2 || ndim == 3, "ndim must be 2 or 3 (got %d)", ndim);
    DXH_REQUIRE(n0 >= 1 && n1 >= 1 && n2 >= 1 && (ndim == 3 || n2 == 1), "bad shape");
    DXH_REQUIRE(dtype == DXH_F32 || dtype == DXH_F64, "dtype must be float32 or float64");
    DXH_REQUIRE(weight > 0.0, "weight must be positive");
    DXH_REQUIRE(max_num_iter >= 0, "max_num_iter must be >= 0");
    int ndev = 0;
    DXH_HIP(hipGetDeviceCount(&ndev));
    DXH_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d devices)", device, ndev);
    return dtype == DXH_F32 ? chambolle_t<float>(device, ndim, n0, n1, n2, img, out, weight, eps, max_num_iter, iters_done)
                            : chambolle_t<double>(device, ndim, n0, n1, n2, img, out, weight, eps, max_num_iter, iters_done);
}
