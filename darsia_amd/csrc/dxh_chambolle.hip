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
#include <type_traits>
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
        T num[NDIM], quo[NDIM];
#pragma unroll
        for (int ax = 0; ax < NDIM; ++ax) num[ax] = a.p[ax][e.off] - a.tau * g[ax];
        div_shared<T, NDIM>(num, n, quo);
#pragma unroll
        for (int ax = 0; ax < NDIM; ++ax) a.p[ax][e.off] = quo[ax];
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
        T num[NDIM], quo[NDIM];
#pragma unroll
        for (int ax = 0; ax < NDIM; ++ax) num[ax] = a.pin[ax][e.off] - a.tau * g[ax];
        div_shared<T, NDIM>(num, n, quo);
#pragma unroll
        for (int ax = 0; ax < NDIM; ++ax) a.pout[ax][e.off] = quo[ax];
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
            if constexpr (HAS3) {
                const T num[3] = {pc0[j] - a.tau * g0, pc1[j] - a.tau * g1, pc2[j] - a.tau * g2};
                T quo[3];
                div_shared<T, 3>(num, n, quo);
                r0[j] = quo[0]; r1[j] = quo[1]; r2[j] = quo[2];
            } else {
                const T num[2] = {pc0[j] - a.tau * g0, pc1[j] - a.tau * g1};
                T quo[2];
                div_shared<T, 2>(num, n, quo);
                r0[j] = quo[0]; r1[j] = quo[1];
            }
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

// --------------------------------------------------------------------------------------------------------------
// Two Chambolle iterations per pass (temporal blocking as in k_fused_multi): stage 0 turns p^n into p^(n+1) row by
// row exactly like k_ch_march; stage 1 follows one row behind, takes the p^(n+1) rows straight from stage 0's
// registers and emits p^(n+2).  p^(n+1) never goes to memory: 28 B per element for TWO iterations.  Each
// iteration costs C elements of validity on either side of the strip and one extra row at either end of the row
// chunk.  Both iterations' energy sums are produced (the stop test needs every iteration's energy); when the test
// fires at the second iteration of a pass the host recomputes that single iteration from the untouched input.
// --------------------------------------------------------------------------------------------------------------
#ifndef DXH_CH_STEADY_UNROLL
#define DXH_CH_STEADY_UNROLL 1  // 3 = unroll the steady loop by the period of the row rotation (no gain measured)
#endif
template <typename T>
struct ChMarch2Args {
    ChMarchArgs<T> m;
    double* partial_d2_b;  // second iteration of the pass
    double* partial_n_b;
};

template <typename T, int C, bool HAS3>
__global__ void __launch_bounds__(256) k_ch_march2(ChMarch2Args<T> aa) {
    const ChMarchArgs<T>& a = aa.m;
    constexpr int N = VecOf<T>::N;
    using SG = StripGeomK<T, C, 2>;
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

    auto d_out = [&](const T (&a0)[N], const T (&a1)[N], const T (&a2)[N], const T (&above)[N], auto has_above, const T (&fv)[N],
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
    // p update of one row from its p, out, the next row's out (has_next) -- returns the new p rows and adds the
    // row's energy terms when `count`
    auto p_update = [&](const T (&c0)[N], const T (&c1)[N], const T (&c2)[N], const T (&dc)[N], const T (&oc)[N], const T (&on)[N],
                        auto has_next, auto count, double& sq, double& nsum, T (&r0)[N], T (&r1)[N], T (&r2)[N]) {
        T o1[N], o2[N];
        shifted<+C, true>(oc, o1);
        if constexpr (HAS3) shifted<+1, true>(oc, o2);
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
            if (count && store_lane && inL[j]) {
                nsum += (double)n;
                sq += (double)dc[j] * (double)dc[j];
            }
            n = n * a.tau_over_w;
            n = n + T(1);
            if constexpr (HAS3) {
                const T num[3] = {c0[j] - a.tau * g0, c1[j] - a.tau * g1, c2[j] - a.tau * g2};
                T quo[3];
                div_shared<T, 3>(num, n, quo);
                r0[j] = quo[0]; r1[j] = quo[1]; r2[j] = quo[2];
            } else {
                const T num[2] = {c0[j] - a.tau * g0, c1[j] - a.tau * g1};
                T quo[2];
                div_shared<T, 2>(num, n, quo);
                r0[j] = quo[0]; r1[j] = quo[1];
                r2[j] = T(0);
            }
        }
    };

    T zero[N];
#pragma unroll
    for (int j = 0; j < N; ++j) zero[j] = T(0);
    auto clampr = [&](int64_t r) -> int64_t { return r < 0 ? 0 : (r > a.n0 - 1 ? a.n0 - 1 : r); };
    // stage 0 emits p^(n+1) of rows e0 .. min(R1, n0 - 1); stage 1 emits p^(n+2) of rows R0 .. R1 - 1
    const int64_t e0 = R0 > 0 ? R0 - 1 : 0;
    // ---- stage 0 state: current row e0 of p^n with its d / out, and the loaded next row ----
    T a0[N], a1[N], a2[N], ad[N], ao[N];
    {
        T ab[N], fv[N];
        load_vec(p0 + e0 * P, a0);
        load_vec(p1 + e0 * P, a1);
        if constexpr (HAS3) load_vec(p2 + e0 * P, a2);
        load_vec(f + e0 * P, fv);
        if (e0 > 0) load_vec(p0 + (e0 - 1) * P, ab);
        d_out(a0, a1, HAS3 ? a2 : zero, e0 > 0 ? ab : zero, e0 > 0, fv, ad, ao);
    }
    T n0v[N], n1v[N], n2v[N], fn[N];
    {
        const int64_t rn = clampr(e0 + 1);
        load_vec(p0 + rn * P, n0v);
        load_vec(p1 + rn * P, n1v);
        if constexpr (HAS3) load_vec(p2 + rn * P, n2v);
        load_vec(f + rn * P, fn);
    }
    // ---- stage 1 state: current row of p^(n+1) (b*), its d / out, the p0^(n+1) row above it, f of its row ----
    T b0[N], b1[N], b2[N], bd[N], bo[N], babove[N], fcur[N];
#pragma unroll
    for (int j = 0; j < N; ++j) b0[j] = b1[j] = b2[j] = bd[j] = bo[j] = babove[j] = T(0);
    load_vec(f + e0 * P, fcur);  // f of the row stage 0 emits in this step = the row that ENTERS stage 1
    bool b_valid = false, have_above = false;
    double sqa = 0.0, nsa = 0.0, sqb = 0.0, nsb = 0.0;
    const int64_t t_last = R1 < a.n0 ? R1 : a.n0 - 1;  // last row stage 0 emits
    // One step of the march.  `steady` = std::true_type for rows where every border / start-up condition is known
    // (both stages emit, a next row and a row above exist, both rows are counted): the wave-uniform flags then fold
    // away instead of costing a v_cndmask per element; all other rows run the same code with run-time flags.
    auto pick = [](auto steady_c, bool v) {
        if constexpr (decltype(steady_c)::value) return std::true_type{};
        else return v;
    };
    auto march_step = [&](const int64_t t, auto steady) __attribute__((always_inline)) {
        // prefetch row t + 2 of p^n and f
        T f0[N], f1[N], f2[N], ff[N];
        {
            const int64_t rn = clampr(t + 2);
            load_vec(p0 + rn * P, f0);
            load_vec(p1 + rn * P, f1);
            if constexpr (HAS3) load_vec(p2 + rn * P, f2);
            load_vec(f + rn * P, ff);
        }
        T e_0[N], e_1[N], e_2[N];  // p^(n+1) of row t
        const auto emit0 = pick(steady, t <= t_last);
        if (emit0) {
            T dn[N], on[N];
#pragma unroll
            for (int j = 0; j < N; ++j) dn[j] = on[j] = T(0);
            const auto has_next = pick(steady, t + 1 < a.n0);
            if (has_next) d_out(n0v, n1v, HAS3 ? n2v : zero, a0, std::true_type{}, fn, dn, on);
            p_update(a0, a1, HAS3 ? a2 : zero, ad, ao, on, has_next, pick(steady, t >= R0 && t < R1), sqa, nsa, e_0, e_1, e_2);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                a0[j] = n0v[j]; a1[j] = n1v[j];
                if constexpr (HAS3) a2[j] = n2v[j];
                ad[j] = dn[j];
                ao[j] = on[j];
            }
        }
        // ---- stage 1: row t of p^(n+1) enters; row t - 1 is emitted ----
        if (emit0) {
            T dn[N], on[N];
            d_out(e_0, e_1, HAS3 ? e_2 : zero, babove, pick(steady, have_above), fcur, dn, on);  // d / out of row t from p^(n+1)
            if (pick(steady, b_valid)) {  // row t - 1 has its out and now the next row's out
                T r0[N], r1[N], r2[N];
                p_update(b0, b1, HAS3 ? b2 : zero, bd, bo, on, std::true_type{}, std::true_type{}, sqb, nsb, r0, r1, r2);
                if (store_lane) {
                    store_vec_nt(q0 + (t - 1) * P, r0);
                    store_vec_nt(q1 + (t - 1) * P, r1);
                    if constexpr (HAS3) store_vec_nt(q2 + (t - 1) * P, r2);
                }
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                babove[j] = e_0[j];
                b0[j] = e_0[j]; b1[j] = e_1[j];
                if constexpr (HAS3) b2[j] = e_2[j];
                bd[j] = dn[j];
                bo[j] = on[j];
            }
            have_above = true;
            b_valid = t >= R0;  // rows before R0 only serve as "row above"
        } else if (b_valid) {
            // t == n0: the image's last row leaves stage 1 without a next row
            T r0[N], r1[N], r2[N];
            p_update(b0, b1, HAS3 ? b2 : zero, bd, bo, zero, false, true, sqb, nsb, r0, r1, r2);
            if (store_lane) {
                store_vec_nt(q0 + (t - 1) * P, r0);
                store_vec_nt(q1 + (t - 1) * P, r1);
                if constexpr (HAS3) store_vec_nt(q2 + (t - 1) * P, r2);
            }
            b_valid = false;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            n0v[j] = f0[j]; n1v[j] = f1[j];
            if constexpr (HAS3) n2v[j] = f2[j];
            fcur[j] = fn[j];  // f of row t + 1: the row entering stage 1 in the next step
            fn[j] = ff[j];
        }
        };
    {
        const int64_t s_lo = R0 + 1, s_hi = (R1 - 1 < a.n0 - 2) ? R1 - 1 : a.n0 - 2;  // steady rows (may be empty)
        int64_t t = e0;
        for (; t <= R1 && t < s_lo; ++t) march_step(t, std::false_type{});
        // unrolled by 3 = the period of the row rotation (current / next / prefetched): the copies become renaming
#if DXH_CH_STEADY_UNROLL == 3
        for (; t + 2 <= s_hi; t += 3) {
            march_step(t, std::true_type{});
            march_step(t + 1, std::true_type{});
            march_step(t + 2, std::true_type{});
        }
#endif
        for (; t <= s_hi; ++t) march_step(t, std::true_type{});
        for (; t <= R1; ++t) march_step(t, std::false_type{});
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        sqa += __shfl_down(sqa, off, kWave);
        nsa += __shfl_down(nsa, off, kWave);
        sqb += __shfl_down(sqb, off, kWave);
        nsb += __shfl_down(nsb, off, kWave);
    }
    if (lane == 0) {
        const int wid = chunk * a.nstrips + strip;
        a.partial_d2[wid] = sqa;
        a.partial_n[wid] = nsa;
        aa.partial_d2_b[wid] = sqb;
        aa.partial_n_b[wid] = nsb;
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
        // the planes go back to the cache (dxh_common.h: PlanePool), which wants them idle
        if (st) (void)hipStreamSynchronize(st);
        for (void* b : bufs) pool_free(b);
        pool_free(partial);
        pool_free(red);
        pool_free(stage2);
        for (void* b : extra) pool_free(b);
    };
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, (size_t)n0 * g.L * sizeof(T), {{img, (size_t)n0 * g.L * sizeof(T)}});
    auto body = [&]() -> int {
        DXH_HIP(cached_stream(device, &st));
        for (int i = 0; i < 2 + ndim; ++i) {
            DXH_HIP(pool_malloc(&bufs[i], plane));
            DXH_HIP(hipMemsetAsync(bufs[i], 0, plane, st));
        }
        // one slot per block of the element-wise kernels or per wave of the marching kernels (strips are at least 64
        // elements wide, row chunks at least 8 rows high), whichever is more
        const size_t npart_alloc = std::max<size_t>(nblocks, (size_t)(g.L / 64 + 2) * (size_t)(n0 / 8 + 2));
        DXH_HIP(pool_malloc((void**)&partial, sizeof(double) * npart_alloc));
        DXH_HIP(pool_malloc((void**)&red, sizeof(double) * 4));
        DXH_HIP(pool_malloc((void**)&stage2, sizeof(double) * 4 * 256));
        DXH_HIP(hipMemcpy2DAsync(bufs[0], g.pitch * sizeof(T), img, g.L * sizeof(T), g.L * sizeof(T), n0,
                                 hipMemcpyDefault, st));
        DXH_HIP(hipMemcpyAsync(bufs[1], bufs[0], plane, hipMemcpyDeviceToDevice, st));  // out = image (iteration 0)
        // p is double buffered: pbuf[ax][0/1]; `out` is only written after the loop
        void* p2[3] = {nullptr, nullptr, nullptr};
        for (int ax = 0; ax < ndim; ++ax) {
            DXH_HIP(pool_malloc(&p2[ax], plane));
            extra.push_back(p2[ax]);
            DXH_HIP(hipMemsetAsync(p2[ax], 0, plane, st));
        }
        double* partial_n = nullptr;
        DXH_HIP(pool_malloc((void**)&partial_n, sizeof(double) * npart_alloc));
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
            int wpc4 = 6;  // blocks (of 4 waves) per CU the grid is sized for (measured best of 2..8)
            if (const char* e = getenv("DXH_CH_BLOCKS_PER_CU")) wpc4 = std::max(1, atoi(e));
            int64_t nchunks = std::max<int64_t>(1, (int64_t)ncu * wpc4 / nbx);
            if (nchunks >= 8) nchunks = nchunks / 8 * 8;
            const int64_t rpc = std::max<int64_t>(8, (n0 + nchunks - 1) / nchunks);
            nchunks = (n0 + rpc - 1) / rpc;
            ma.rows_per_chunk = rpc;
            ma.nchunks = (int)nchunks;
            march_grid = (unsigned)(nbx * nchunks);
            nparts = (int)(nchunks * ma.nstrips);
            if ((size_t)nparts > npart_alloc) { set_error("internal: partial buffer too small"); return DXH_EHIP; }
        }
        // two iterations per pass (k_ch_march2): its own strip geometry (twice the halo)
        const bool march2 = march && !getenv("DXH_CHAMBOLLE_NO_K2");
        ChMarch2Args<T> m2{};
        unsigned march2_grid = 0;
        int nparts2 = 0;
        double *partial_d2_b = nullptr, *partial_n_b = nullptr;
        if (march2) {
            const int stride = ndim == 2 ? StripGeomK<T, 1, 2>::STRIDE : StripGeomK<T, 3, 2>::STRIDE;
            m2.m = ma;
            m2.m.nstrips = (int)((g.L + stride - 1) / stride);
            const int nbx = (m2.m.nstrips + 3) / 4;
            int64_t nchunks = std::max<int64_t>(1, (int64_t)ma.nchunks * ((ma.nstrips + 3) / 4) / nbx);
            if (nchunks >= 8) nchunks = nchunks / 8 * 8;
            const int64_t rpc = std::max<int64_t>(8, (n0 + nchunks - 1) / nchunks);
            nchunks = (n0 + rpc - 1) / rpc;
            m2.m.rows_per_chunk = rpc;
            m2.m.nchunks = (int)nchunks;
            march2_grid = (unsigned)(nbx * nchunks);
            nparts2 = (int)(nchunks * m2.m.nstrips);
            if ((size_t)nparts2 > npart_alloc) { set_error("internal: partial buffer too small"); return DXH_EHIP; }
            DXH_HIP(pool_malloc((void**)&partial_d2_b, sizeof(double) * nparts2));
            extra.push_back(partial_d2_b);
            DXH_HIP(pool_malloc((void**)&partial_n_b, sizeof(double) * nparts2));
            extra.push_back(partial_n_b);
        }
        int reduce_count = nparts;
        // the energies' partial sums, several lists at once: one launch pair, one read-back, one synchronisation
        auto reduce_lists = [&](int nl, const double* const* parts, double* host) -> int {
            SumLists ls{};
            for (int i = 0; i < 4; ++i) ls.p[i] = parts[i < nl ? i : 0];
            hipLaunchKernelGGL(k_sum_partials_lists, dim3(256, nl), dim3(256), 0, st, ls, reduce_count, stage2);
            SumLists l2{};
            for (int i = 0; i < 4; ++i) l2.p[i] = stage2 + 256 * (i < nl ? i : 0);
            hipLaunchKernelGGL(k_sum_partials_lists, dim3(1, nl), dim3(256), 0, st, l2, 256, red);
            DXH_HIP(hipGetLastError());
            DXH_HIP(hipMemcpyAsync(host, red, sizeof(double) * nl, hipMemcpyDefault, st));
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
            if (march2 && max_iter - it >= 3) {
                // two iterations in one pass; the last iterations of a run are single ones so that the p the returned
                // `out` derives from exists in memory
                m2.m.f = (const T*)bufs[0];
                for (int ax = 0; ax < 3; ++ax) { m2.m.pin[ax] = pbuf(ax < ndim ? ax : 0, cur); m2.m.pout[ax] = pbuf(ax < ndim ? ax : 0, cur ^ 1); }
                m2.m.tau = (T)tau; m2.m.tau_over_w = (T)(tau / weight); m2.m.partial_d2 = partial; m2.m.partial_n = partial_n;
                m2.partial_d2_b = partial_d2_b; m2.partial_n_b = partial_n_b;
                if (ndim == 2) hipLaunchKernelGGL((k_ch_march2<T, 1, false>), dim3(march2_grid), dim3(256), 0, st, m2);
                else hipLaunchKernelGGL((k_ch_march2<T, 3, true>), dim3(march2_grid), dim3(256), 0, st, m2);
                DXH_HIP(hipGetLastError());
                double d2b = 0.0, nb = 0.0;
                reduce_count = nparts2;
                {
                    const double* lists[4] = {partial, partial_n, partial_d2_b, partial_n_b};
                    double sums[4] = {0, 0, 0, 0};
                    r = reduce_lists(4, lists, sums);
                    sum_d2 = sums[0]; sum_n = sums[1]; d2b = sums[2]; nb = sums[3];
                }
                reduce_count = nparts;
                if (r) return r;
                // first iteration of the pass: its `out` derives from p[cur], which is untouched
                out_from = cur;
                const double ea = (sum_d2 + weight * sum_n) / size;
                if (it == 0) {
                    e_init = e_prev = ea;
                } else {
                    if (std::fabs(e_prev - ea) < eps * e_init) break;
                    e_prev = ea;
                }
                ++it;
                // second iteration: its input p^(n+1) lived in registers only
                const double eb = (d2b + weight * nb) / size;
                if (std::fabs(e_prev - eb) < eps * e_init) {
                    // the reference stops here and returns the `out` of this iteration: recompute p^(n+1) alone
                    ma.f = (const T*)bufs[0];
                    for (int ax = 0; ax < 3; ++ax) { ma.pin[ax] = pbuf(ax < ndim ? ax : 0, cur); ma.pout[ax] = pbuf(ax < ndim ? ax : 0, cur ^ 1); }
                    ma.tau = (T)tau; ma.tau_over_w = (T)(tau / weight); ma.partial_d2 = partial; ma.partial_n = partial_n;
                    if (ndim == 2) hipLaunchKernelGGL((k_ch_march<T, 1, false>), dim3(march_grid), dim3(256), 0, st, ma);
                    else hipLaunchKernelGGL((k_ch_march<T, 3, true>), dim3(march_grid), dim3(256), 0, st, ma);
                    DXH_HIP(hipGetLastError());
                    out_from = cur ^ 1;
                    break;
                }
                e_prev = eb;
                ++it;
                cur ^= 1;  // p[cur] = p^(n+2)
                out_from = -2;  // never used: at least one single iteration follows (max_iter - it >= 1)
                continue;
            }
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
            {
                const double* lists[4] = {partial, partial_n, partial, partial};
                double sums[2] = {0, 0};
                if ((r = reduce_lists(2, lists, sums))) return r;
                sum_d2 = sums[0]; sum_n = sums[1];
            }
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
                                 hipMemcpyDefault, st));
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
