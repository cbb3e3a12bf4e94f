// Chambolle's TV projection algorithm (the arithmetic of DarSIA's default TVD() method, which the
// reference delegates to skimage.restoration.denoise_tv_chambolle without a channel axis,
// restoration/tvd.py:77-83: colour images are denoised as 3-D volumes).
//
// Per iteration (oracle/chambolle_numpy.py states the algorithm and its source):
//   k_ch_out : d = -div p, out = image + d, partial sums of d^2
//   k_ch_p   : g = forward differences of out, n = |g|, partial sums of n, p = (p - tau g) / (1 + n tau / w)
// Energies are reduced deterministically (per-block partials, then one block) in double; the stop test
// |E_prev - E| < eps E_0 runs on the host once per iteration, as in the reference.
#include <algorithm>
#include <cmath>
#include <cstdlib>
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

// --------------------------------------------------------------------------------------------------------------
// Row-marching form of the fused iteration for 2-D images (C = 1) and for colour images treated as volumes whose
// third axis is the channel axis (C = n2 = 3; that is how DarSIA's default TVD() runs, tvd.py:77-83): one wave
// owns a column strip of 64 lanes x 16 bytes and a chunk of rows and walks down the rows like the TV sweep
// (dxh_kernels.h, k_fused): the rows of p and f are read once with 16-byte loads, the column / channel neighbours
// come from DPP lane shifts, `out` of the current and the next row live in registers.  Algorithmic traffic
// (2 ndim + 1) values per element and iteration -- what the one-thread-per-element kernel above also moves from
// HBM, but through ~30 scalar cache accesses per element (1.52 ms per iteration at 4000 x 8000 x 3 against
// 0.5 ms here).  Same arithmetic, same order; the energy sums are per-wave partials in double.
// --------------------------------------------------------------------------------------------------------------
template <typename T>
struct ChMarchArgs {
    const T* f;
    const T* pin[3];
    T* pout[3];
    int64_t n0, L, pitch, rows_per_chunk;
    int nstrips, nchunks;
    T tau, tau_over_w;
    double* partial_d2;  // [nchunks * nstrips]
    double* partial_n;
};

template <typename T, int C, bool HAS3>
__global__ void __launch_bounds__(256) k_ch_march(ChMarchArgs<T> a) {
    constexpr int N = VecOf<T>::N;
    using SG = StripGeom<T, C, true>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nbx = (a.nstrips + 3) / 4;
    int bx, chunk;
    block_to_tile((int)blockIdx.x, nbx, a.nchunks, bx, chunk);
    const int strip = bx * 4 + wave;
    if (strip >= a.nstrips || chunk >= a.nchunks) return;  // wave-uniform
    const int64_t R0 = (int64_t)chunk * a.rows_per_chunk;
    const int64_t R1 = (R0 + a.rows_per_chunk < a.n0) ? R0 + a.rows_per_chunk : a.n0;
    if (R0 >= a.n0) return;
    const int64_t x0 = (int64_t)strip * SG::STRIDE + (int64_t)(lane - SG::LH) * N;
    int64_t xl = x0;
    if (xl < 0) xl = 0;
    if (xl > a.pitch - N) xl = a.pitch - N;
    const bool store_lane = lane >= SG::LH && lane < SG::LH + SG::VALID && x0 < a.L;
    bool edgeL[N], edgeR[N], ch0[N], chLast[N], inL[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int64_t x = x0 + j;
        edgeL[j] = x < C;
        edgeR[j] = x >= a.L - C;
        const int cc = (int)(((x % C) + C) % C);
        ch0[j] = cc == 0;
        chLast[j] = cc == C - 1;
        inL[j] = x >= 0 && x < a.L;
    }
    const int64_t P = a.pitch;
    const T* f = a.f + xl;
    const T* p0 = a.pin[0] + xl;
    const T* p1 = a.pin[1] + xl;
    const T* p2 = HAS3 ? a.pin[2] + xl : nullptr;
    T* q0 = a.pout[0] + xl;
    T* q1 = a.pout[1] + xl;
    T* q2 = HAS3 ? a.pout[2] + xl : nullptr;

    // d and out of one row from its p rows, the p0 row above it and f:  d = -(p0 + p1 [+ p2]) [+ p0 above]
    // [+ p1 one column left] [+ p2 one channel left]   (same order of additions as k_ch_fused / the reference)
    auto d_out = [&](const T (&a0)[N], const T (&a1)[N], const T (&a2)[N], const T (&above)[N], bool has_above, const T (&fv)[N],
                     T (&d)[N], T (&o)[N]) {
        T l1[N], l2[N];
        shifted<-C, true>(a1, l1);
        if constexpr (HAS3) shifted<-1, true>(a2, l2);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            T s = a0[j] + a1[j];
            if constexpr (HAS3) s = s + a2[j];
            T v = -s;
            if (has_above) v = v + above[j];
            if (!edgeL[j]) v = v + l1[j];
            if constexpr (HAS3) {
                if (!ch0[j]) v = v + l2[j];
            }
            d[j] = v;
            o[j] = fv[j] + v;
        }
    };

    T zero[N];
#pragma unroll
    for (int j = 0; j < N; ++j) zero[j] = T(0);
    // prologue: row R0 (needs the p0 row above it), and the loads of row R0 + 1
    T pc0[N], pc1[N], pc2[N], dc[N], oc[N];
    {
        T ab[N], fv[N];
        load_vec(p0 + R0 * P, pc0);
        load_vec(p1 + R0 * P, pc1);
        if constexpr (HAS3) load_vec(p2 + R0 * P, pc2);
        load_vec(f + R0 * P, fv);
        if (R0 > 0) load_vec(p0 + (R0 - 1) * P, ab);
        d_out(pc0, pc1, HAS3 ? pc2 : zero, R0 > 0 ? ab : zero, R0 > 0, fv, dc, oc);
    }
    T pn0[N], pn1[N], pn2[N], fn[N];
    auto clampr = [&](int64_t r) -> int64_t { return r > a.n0 - 1 ? a.n0 - 1 : r; };
    {
        const int64_t rn = clampr(R0 + 1);
        load_vec(p0 + rn * P, pn0);
        load_vec(p1 + rn * P, pn1);
        if constexpr (HAS3) load_vec(p2 + rn * P, pn2);
        load_vec(f + rn * P, fn);
    }
    double sq = 0.0, nsum = 0.0;
    for (int64_t r = R0; r < R1; ++r) {
        // prefetch row r + 2
        T pf0[N], pf1[N], pf2[N], ff[N];
        {
            const int64_t rn = clampr(r + 2);
            load_vec(p0 + rn * P, pf0);
            load_vec(p1 + rn * P, pf1);
            if constexpr (HAS3) load_vec(p2 + rn * P, pf2);
            load_vec(f + rn * P, ff);
        }
        // out of row r + 1 (its row above is row r)
        T dn[N], on[N];
#pragma unroll
        for (int j = 0; j < N; ++j) dn[j] = on[j] = T(0);
        const bool has_next = r + 1 < a.n0;
        if (has_next) d_out(pn0, pn1, HAS3 ? pn2 : zero, pc0, true, fn, dn, on);
        // gradient of out at row r
        T o1[N], o2[N];
        shifted<+C, true>(oc, o1);
        if constexpr (HAS3) shifted<+1, true>(oc, o2);
        T r0[N], r1[N], r2[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const T g0 = has_next ? on[j] - oc[j] : T(0);
            const T g1 = edgeR[j] ? T(0) : o1[j] - oc[j];
            T s2 = g0 * g0;
            s2 = s2 + g1 * g1;
            T g2 = T(0);
            if constexpr (HAS3) {
                g2 = chLast[j] ? T(0) : o2[j] - oc[j];
                s2 = s2 + g2 * g2;
            }
            T n = t_sqrt(s2);
            if (store_lane && inL[j]) {
                nsum += (double)n;
                sq += (double)dc[j] * (double)dc[j];
            }
            n = n * a.tau_over_w;
            n = n + T(1);
            r0[j] = (pc0[j] - a.tau * g0) / n;
            r1[j] = (pc1[j] - a.tau * g1) / n;
            if constexpr (HAS3) r2[j] = (pc2[j] - a.tau * g2) / n;
        }
        if (store_lane) {
            store_vec_nt(q0 + r * P, r0);
            store_vec_nt(q1 + r * P, r1);
            if constexpr (HAS3) store_vec_nt(q2 + r * P, r2);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            pc0[j] = pn0[j]; pc1[j] = pn1[j];
            pn0[j] = pf0[j]; pn1[j] = pf1[j];
            if constexpr (HAS3) { pc2[j] = pn2[j]; pn2[j] = pf2[j]; }
            fn[j] = ff[j];
            dc[j] = dn[j];
            oc[j] = on[j];
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        sq += __shfl_down(sq, off, kWave);
        nsum += __shfl_down(nsum, off, kWave);
    }
    if (lane == 0) {
        const int wid = chunk * a.nstrips + strip;
        a.partial_d2[wid] = sq;
        a.partial_n[wid] = nsum;
    }
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
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, (size_t)n0 * g.L * sizeof(T));
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
        // row-marching kernel for 2-D images and colour-images-as-volumes; everything else one thread per element
        const bool march = (ndim == 2 || (ndim == 3 && n2 == 3)) && !getenv("DXH_CHAMBOLLE_NO_MARCH");
        ChMarchArgs<T> ma{};
        unsigned march_grid = 0;
        int nparts = (int)nblocks;
        if (march) {
            const int stride = ndim == 2 ? StripGeom<T, 1, true>::STRIDE : StripGeom<T, 3, true>::STRIDE;
            ma.n0 = n0; ma.L = g.L; ma.pitch = g.pitch;
            ma.nstrips = (int)((g.L + stride - 1) / stride);
            const int nbx = (ma.nstrips + 3) / 4;
            int ncu = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
            int64_t nchunks = std::max<int64_t>(1, (int64_t)ncu * 3 / nbx);  // ~12 waves per CU
            if (nchunks >= 8) nchunks = nchunks / 8 * 8;
            const int64_t rpc = std::max<int64_t>(8, (n0 + nchunks - 1) / nchunks);
            nchunks = (n0 + rpc - 1) / rpc;
            ma.rows_per_chunk = rpc;
            ma.nchunks = (int)nchunks;
            march_grid = (unsigned)(nbx * nchunks);
            nparts = (int)(nchunks * ma.nstrips);
            if ((size_t)nparts > (size_t)nblocks) { set_error("internal: partial buffer too small"); return DXH_EHIP; }
        }
        auto reduce = [&](double* part, double* host) -> int {
            hipLaunchKernelGGL(k_sum_partials, dim3(256), dim3(256), 0, st, part, nparts, stage2);
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
            if (march) {
                ma.f = (const T*)bufs[0];
                for (int ax = 0; ax < 3; ++ax) { ma.pin[ax] = pbuf(ax < ndim ? ax : 0, cur); ma.pout[ax] = pbuf(ax < ndim ? ax : 0, cur ^ 1); }
                ma.tau = (T)tau; ma.tau_over_w = (T)(tau / weight); ma.partial_d2 = partial; ma.partial_n = partial_n;
                if (ndim == 2) hipLaunchKernelGGL((k_ch_march<T, 1, false>), dim3(march_grid), dim3(256), 0, st, ma);
                else hipLaunchKernelGGL((k_ch_march<T, 3, true>), dim3(march_grid), dim3(256), 0, st, ma);
            } else if (ndim == 2) {
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
        pf.join();
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
