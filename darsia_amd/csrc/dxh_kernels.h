// Device kernels of the split-Bregman TV path.  All arithmetic is IEEE add/sub/mul/div/sqrt in
// the image dtype, one rounding per operation and in the reference's operation order (compile with
// -ffp-contract=off), so that results equal a NumPy evaluation in that dtype bit for bit.
//
// Storage: an array of shape (n0, n1, n2) is kept as n0 rows of L = n1*n2 elements with a row
// pitch (in elements) that is a multiple of 32.  Colour images stay INTERLEAVED (H, W*C): the
// column neighbour is C elements away, the channel neighbour 1 element.
//
// State: instead of the reference's split variables d_j, b_j (2*dim planes) only
// q_j = b_j_old + D_j(u)   ("dub", split_bregman_tvd.py:144) is stored; d_j and b_j are recomputed
// from q_j where needed (dxh_common.h: split_vars).  u and q_j are double buffered.
#pragma once

#include <type_traits>

#include "dxh_common.h"

#ifndef DXH_STRIP_ALIGN_BYTES
#define DXH_STRIP_ALIGN_BYTES 64  // k_fused_multi: a wave's store range covers whole granules of this size
#endif
#ifndef DXH_FUSE_DPP_ADDS
#define DXH_FUSE_DPP_ADDS 1  // Pipe::step: hand-fused v_add_f32_dpp for the column-neighbour sums of interior strips
#endif
#ifndef DXH_LB4
#define DXH_LB4 3  // waves per SIMD requested for the float K >= 3 scalar-coefficient pipelines (register cap 168)
#endif

namespace dxh {

template <typename T>
struct Coef {       // scalar coefficients, already rounded to T like NumPy's weak scalars
    T ell;          // Bregman regularisation / diffusion coefficient
    T diag;         // omega + ell*2*dim/h^2                               jacobi.py:72
    T dsc;          // diag / (ell/h^2)                                    jacobi.py:104-106
    T kq;           // mu / ell (shrinkage threshold)                      split_bregman_tvd.py:140,145
    T rdiag, rdsc;  // RN(1/diag), RN(1/dsc) for div_by_const; 0 = use the IEEE division
    int div3;       // 1: ONE residual correction already rounds x/diag and x/dsc correctly for every x (k_div3_check)
    int const_diag; // array-valued mu with scalar omega and ell: diag / dsc above are the (wave-invariant) divisors
};


template <typename T>
struct CoefPlanes {  // heterogeneous coefficients (nullptr = scalar)
    const T* ell;
    const T* diag;
    const T* dsc;
    const T* kq;
};

struct Geom {
    int64_t n0, n1, n2;  // array shape
    int64_t L;           // n1 * n2
    int64_t pitch;       // row pitch in elements
    int ndim, dim, nsum_axes;
};

// ============================================================================================
// Generic three-kernel path: any dim (2,3), any channel count, any number of Jacobi sweeps.
// One thread per element; neighbour values come through the caches.
// ============================================================================================

struct ElemIdx {
    int64_t row, x, i1, i2;
    int64_t off;
};

__device__ __forceinline__ bool elem_index(const Geom& g, ElemIdx& e) {
    const int64_t bpr = (g.L + blockDim.x - 1) / blockDim.x;
    const int64_t bid = blockIdx.x;
    e.row = bid / bpr;
    e.x = (bid - e.row * bpr) * blockDim.x + threadIdx.x;
    if (e.row >= g.n0 || e.x >= g.L) return false;
    e.i1 = e.x / g.n2;
    e.i2 = e.x - e.i1 * g.n2;
    e.off = e.row * g.pitch + e.x;
    return true;
}

// offset of the lower (-1) / upper (+1) neighbour along `axis`, clamped at the array ends
__device__ __forceinline__ int64_t nb_off(const Geom& g, const ElemIdx& e, int axis, int dir) {
    if (axis == 0) {
        if (dir < 0) return e.row > 0 ? e.off - g.pitch : e.off;
        return e.row < g.n0 - 1 ? e.off + g.pitch : e.off;
    } else if (axis == 1) {
        if (dir < 0) return e.i1 > 0 ? e.off - g.n2 : e.off;
        return e.i1 < g.n1 - 1 ? e.off + g.n2 : e.off;
    } else {
        if (dir < 0) return e.i2 > 0 ? e.off - 1 : e.off;
        return e.i2 < g.n2 - 1 ? e.off + 1 : e.off;
    }
}

template <typename T, int DIM>
struct QPlanes {
    const T* in[DIM];
    T* out[DIM];
};

// t_j(p) = ell * (b_j - d_j) at offset `off`, from the stored state (or from explicit d, b planes)
template <typename T, int DIM, bool ISO, bool FROM_DB>
__device__ __forceinline__ void t_at(const QPlanes<T, DIM>& q, const T* const* dpl, const T* const* bpl,
                                     int64_t off, T ell, T kq, T (&t)[DIM]) {
    T d[DIM], b[DIM];
    if constexpr (FROM_DB) {
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            d[j] = dpl[j][off];
            b[j] = bpl[j][off];
        }
    } else {
        T qq[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) qq[j] = q.in[j][off];
        split_vars<T, DIM, ISO>(qq, kq, d, b);
    }
#pragma unroll
    for (int j = 0; j < DIM; ++j) t[j] = ell * (b[j] - d[j]);
}

template <typename T>
struct DBPlanes {
    const T* d[3];
    const T* b[3];
};

// rs = (omega*img + sum_j forward_diff(ell*(b_j - d_j), j)) / diag
// split_bregman_tvd.py:84-89, jacobi.py:107
template <typename T, int DIM, bool ISO, bool FROM_DB>
__global__ void __launch_bounds__(256) k_rhs(Geom g, const T* __restrict__ of, QPlanes<T, DIM> q, DBPlanes<T> db,
                                              Coef<T> c, CoefPlanes<T> cp, T* __restrict__ rs) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const T ell_c = cp.ell ? cp.ell[e.off] : c.ell;
    const T kq_c = cp.kq ? cp.kq[e.off] : c.kq;
    T tc[DIM];
    t_at<T, DIM, ISO, FROM_DB>(q, db.d, db.b, e.off, ell_c, kq_c, tc);
    T r = of[e.off];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const int64_t o = nb_off(g, e, j, -1);
        const T ell_n = cp.ell ? cp.ell[o] : c.ell;
        const T kq_n = cp.kq ? cp.kq[o] : c.kq;
        T tn[DIM];
        t_at<T, DIM, ISO, FROM_DB>(q, db.d, db.b, o, ell_n, kq_n, tn);
        r = r + (tc[j] - tn[j]);
    }
    const T diag = cp.diag ? cp.diag[e.off] : c.diag;
    rs[e.off] = r / diag;
}

// one Jacobi sweep  x_out = rs + N(x_in) / dsc,  N = neighbour sum with replicate ghosts over the
// first nsum_axes array axes (jacobi.py:25-55,110-114)
template <typename T>
__global__ void __launch_bounds__(256) k_jacobi(Geom g, const T* __restrict__ rs, const T* __restrict__ xin,
                                                 Coef<T> c, CoefPlanes<T> cp, T* __restrict__ xout) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    T acc = T(0);
    for (int ax = 0; ax < g.nsum_axes; ++ax) {
        acc = acc + xin[nb_off(g, e, ax, -1)];
        acc = acc + xin[nb_off(g, e, ax, +1)];
    }
    const T dsc = cp.dsc ? cp.dsc[e.off] : c.dsc;
    xout[e.off] = rs[e.off] + acc / dsc;
}

// The same sweep for volumes (all three axes summed) with the index arithmetic of a 3-D grid -- block = 256 elements
// of one row of axis 2, blockIdx.y = row, blockIdx.z = plane -- instead of 64-bit divisions per element: the sweep is a
// 12-byte-per-element stream and the divisions were what it spent its time on (256^3 float32: 0.080 -> see DESIGN.md).
template <typename T>
__global__ void __launch_bounds__(256) k_jacobi_vol(Geom g, const T* __restrict__ rs, const T* __restrict__ xin, Coef<T> c,
                                                     CoefPlanes<T> cp, T* __restrict__ xout) {
    const int64_t i2 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i2 >= g.n2) return;
    const int64_t i1 = blockIdx.y, i0 = blockIdx.z;
    const int64_t off = i0 * g.pitch + i1 * g.n2 + i2;
    T acc = T(0);
    acc = acc + xin[i0 > 0 ? off - g.pitch : off];
    acc = acc + xin[i0 < g.n0 - 1 ? off + g.pitch : off];
    acc = acc + xin[i1 > 0 ? off - g.n2 : off];
    acc = acc + xin[i1 < g.n1 - 1 ? off + g.n2 : off];
    acc = acc + xin[i2 > 0 ? off - 1 : off];
    acc = acc + xin[i2 < g.n2 - 1 ? off + 1 : off];
    const T dsc = cp.dsc ? cp.dsc[off] : c.dsc;
    xout[off] = rs[off] + acc / dsc;
}

// The volume sweep again, 16 bytes per lane and XCD-aware.  k_jacobi_vol above deals the rows of the volume round-robin
// over the eight XCDs, so the four neighbour rows of a row (axis 0 and axis 1) are fetched by other XCDs' L2s: every line of
// x crosses the fabric five times per sweep.  Here XCD x (= block % 8) works through a CONTIGUOUS run of (plane, row) tiles in
// order, so the neighbour rows and planes of what it computes are what it has just read -- or will read next -- through its own
// L2 (three planes of 256 x 256 floats are 768 KB of its 4 MB); a thread owns N = 16 B / sizeof(T) consecutive elements of
// axis 2 (n2 % N == 0, which keeps every row 16-byte aligned: the pitch is a multiple of 32 elements).  Same sums in the same
// order as above.
template <typename T>
__global__ void __launch_bounds__(256) k_jacobi_vol_x(Geom g, const T* __restrict__ rs, const T* __restrict__ xin, Coef<T> c,
                                                       CoefPlanes<T> cp, T* __restrict__ xout, int nbx, int64_t tiles, int64_t per_xcd) {
    constexpr int N = VecOf<T>::N;
    const int64_t t = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || t >= tiles) return;
    const int64_t row = t / nbx;
    const int bx = (int)(t - row * nbx);
    const int64_t i0 = row / g.n1, i1 = row - i0 * g.n1;
    const int64_t i2 = ((int64_t)bx * 256 + threadIdx.x) * N;
    if (i2 >= g.n2) return;
    const int64_t off = i0 * g.pitch + i1 * g.n2 + i2;
    T xc[N], xa[N], xb[N], xl[N], xr[N], r[N], o[N];
    load_vec(xin + off, xc);
    load_vec(xin + (i0 > 0 ? off - g.pitch : off), xa);
    load_vec(xin + (i0 < g.n0 - 1 ? off + g.pitch : off), xb);
    load_vec(xin + (i1 > 0 ? off - g.n2 : off), xl);
    load_vec(xin + (i1 < g.n1 - 1 ? off + g.n2 : off), xr);
    load_vec(rs + off, r);
    const T left = i2 > 0 ? xin[off - 1] : xc[0];
    const T right = i2 + N < g.n2 ? xin[off + N] : xc[N - 1];
    T dscv[N];
    if (cp.dsc) load_vec(cp.dsc + off, dscv);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T acc = T(0);
        acc = acc + xa[j];
        acc = acc + xb[j];
        acc = acc + xl[j];
        acc = acc + xr[j];
        acc = acc + (j == 0 ? left : xc[j - 1]);
        acc = acc + (j == N - 1 ? right : xc[j + 1]);
        o[j] = r[j] + acc / (cp.dsc ? dscv[j] : c.dsc);
    }
    store_vec(xout + off, o);
}

// q_j <- backward_diff(u_new, j) + b_j       split_bregman_tvd.py:136-146 (both branches)
template <typename T, int DIM, bool ISO, bool FROM_DB>
__global__ void __launch_bounds__(256) k_shrink(Geom g, const T* __restrict__ unew, QPlanes<T, DIM> q, DBPlanes<T> db,
                                                 Coef<T> c, CoefPlanes<T> cp) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    T d[DIM], b[DIM];
    if constexpr (FROM_DB) {
#pragma unroll
        for (int j = 0; j < DIM; ++j) b[j] = db.b[j][e.off];
    } else {
        const T kq = cp.kq ? cp.kq[e.off] : c.kq;
        T qq[DIM];
#pragma unroll
        for (int j = 0; j < DIM; ++j) qq[j] = q.in[j][e.off];
        split_vars<T, DIM, ISO>(qq, kq, d, b);
    }
    const T uc = unew[e.off];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const T g_j = unew[nb_off(g, e, j, +1)] - uc;
        q.out[j][e.off] = g_j + b[j];
    }
}

// k_rhs and k_shrink for volumes (dim = 3, compressed state) in the form of k_jacobi_vol_x: N = 16 B / sizeof(T) adjacent
// columns per thread, XCD x works a contiguous run of (plane, row) tiles, no per-element index divisions -- and the
// neighbours' t = ell (b - d) are formed for the ONE component the difference needs (the generic k_rhs splits all three
// at each of its three neighbours).  These two frame the Jacobi sweeps of a multi-sweep or solver iteration
// (examples/regularization.py:8-47: Jacobi(maxiter=20)).  Same expressions, same order: same bits.
template <typename T, bool ISO>
__global__ void __launch_bounds__(256) k_rhs_vol(Geom g, const T* __restrict__ of, QPlanes<T, 3> q, Coef<T> c, CoefPlanes<T> cp,
                                                  T* __restrict__ rs, int nbx, int64_t tiles, int64_t per_xcd) {
    constexpr int N = VecOf<T>::N;
    const int64_t t = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || t >= tiles) return;
    const int64_t row = t / nbx;
    const int bx = (int)(t - row * nbx);
    const int64_t i0 = row / g.n1, i1 = row - i0 * g.n1;
    const int64_t i2 = ((int64_t)bx * 256 + threadIdx.x) * N;
    if (i2 >= g.n2) return;
    const int64_t off = i0 * g.pitch + i1 * g.n2 + i2;
    auto coef4 = [&](const T* plane, T scalar, int64_t o, T (&v)[N]) {
        if (plane) load_vec(plane + o, v);
        else {
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = scalar;
        }
    };
    // component `a` of t at N consecutive elements starting at o
    auto t_comp4 = [&](int64_t o, int a, T (&tv)[N]) {
        T ell[N], kq[N];
        coef4(cp.ell, c.ell, o, ell); coef4(cp.kq, c.kq, o, kq);
        if constexpr (ISO) {
            T q0[N], q1[N], q2[N];
            load_vec(q.in[0] + o, q0); load_vec(q.in[1] + o, q1); load_vec(q.in[2] + o, q2);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T qq[3] = {q0[j], q1[j], q2[j]}, d[3], b[3];
                split_vars<T, 3, true>(qq, kq[j], d, b);
                tv[j] = ell[j] * (b[a] - d[a]);
            }
        } else {
            T qa[N];
            load_vec(q.in[a] + o, qa);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const T qq[1] = {qa[j]};
                T d[1], b[1];
                split_vars<T, 1, false>(qq, kq[j], d, b);
                tv[j] = ell[j] * (b[0] - d[0]);
            }
        }
    };
    T q0[N], q1[N], q2[N], ofv[N], ell[N], kq[N], diag[N], tc[3][N];
    load_vec(q.in[0] + off, q0); load_vec(q.in[1] + off, q1); load_vec(q.in[2] + off, q2);
    load_vec(of + off, ofv);
    coef4(cp.ell, c.ell, off, ell); coef4(cp.kq, c.kq, off, kq); coef4(cp.diag, c.diag, off, diag);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T qq[3] = {q0[j], q1[j], q2[j]}, d[3], b[3];
        split_vars<T, 3, ISO>(qq, kq[j], d, b);
#pragma unroll
        for (int a = 0; a < 3; ++a) tc[a][j] = ell[j] * (b[a] - d[a]);
    }
    T tn0[N], tn1[N];
    if (i0 > 0) t_comp4(off - g.pitch, 0, tn0);
    else {
#pragma unroll
        for (int j = 0; j < N; ++j) tn0[j] = tc[0][j];
    }
    if (i1 > 0) t_comp4(off - g.n2, 1, tn1);
    else {
#pragma unroll
        for (int j = 0; j < N; ++j) tn1[j] = tc[1][j];
    }
    T t2_left = tc[2][0];
    if (i2 > 0) {
        const int64_t o = off - 1;
        const T ell_n = cp.ell ? cp.ell[o] : c.ell, kq_n = cp.kq ? cp.kq[o] : c.kq;
        if constexpr (ISO) {
            T qq[3] = {q.in[0][o], q.in[1][o], q.in[2][o]}, d[3], b[3];
            split_vars<T, 3, true>(qq, kq_n, d, b);
            t2_left = ell_n * (b[2] - d[2]);
        } else {
            const T qq[1] = {q.in[2][o]};
            T d[1], b[1];
            split_vars<T, 1, false>(qq, kq_n, d, b);
            t2_left = ell_n * (b[0] - d[0]);
        }
    }
    T out[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T r = ofv[j];
        r = r + (tc[0][j] - tn0[j]);
        r = r + (tc[1][j] - tn1[j]);
        r = r + (tc[2][j] - (j == 0 ? t2_left : tc[2][j - 1]));
        out[j] = r / diag[j];
    }
    store_vec(rs + off, out);
}

template <typename T, bool ISO>
__global__ void __launch_bounds__(256) k_shrink_vol(Geom g, const T* __restrict__ unew, QPlanes<T, 3> q, Coef<T> c,
                                                     CoefPlanes<T> cp, int nbx, int64_t tiles, int64_t per_xcd) {
    constexpr int N = VecOf<T>::N;
    const int64_t t = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || t >= tiles) return;
    const int64_t row = t / nbx;
    const int bx = (int)(t - row * nbx);
    const int64_t i0 = row / g.n1, i1 = row - i0 * g.n1;
    const int64_t i2 = ((int64_t)bx * 256 + threadIdx.x) * N;
    if (i2 >= g.n2) return;
    const int64_t off = i0 * g.pitch + i1 * g.n2 + i2;
    T q0[N], q1[N], q2[N], kq[N], uc[N], ua[N], ub[N];
    load_vec(q.in[0] + off, q0); load_vec(q.in[1] + off, q1); load_vec(q.in[2] + off, q2);
    if (cp.kq) load_vec(cp.kq + off, kq);
    else {
#pragma unroll
        for (int j = 0; j < N; ++j) kq[j] = c.kq;
    }
    load_vec(unew + off, uc);
    load_vec(unew + (i0 < g.n0 - 1 ? off + g.pitch : off), ua);
    load_vec(unew + (i1 < g.n1 - 1 ? off + g.n2 : off), ub);
    const T right = i2 + N < g.n2 ? unew[off + N] : uc[N - 1];
    T o0[N], o1[N], o2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T qq[3] = {q0[j], q1[j], q2[j]}, d[3], b[3];
        split_vars<T, 3, ISO>(qq, kq[j], d, b);
        o0[j] = (ua[j] - uc[j]) + b[0];
        o1[j] = (ub[j] - uc[j]) + b[1];
        o2[j] = ((j == N - 1 ? right : uc[j + 1]) - uc[j]) + b[2];
    }
    store_vec(q.out[0] + off, o0);
    store_vec(q.out[1] + off, o1);
    store_vec(q.out[2] + off, o2);
}

// ============================================================================================
// One outer iteration of a dim = 3 problem in ONE pass (examples/regularization.py:8-47 and the PET volumes of
// examples/paper/pet_simulations_comparison_block_b.py:680-713 are the reference's 3-D use): right-hand side, Jacobi
// sweep and shrinkage with the very expressions of k_rhs / k_jacobi / k_shrink above, so the bits are the same, but
// u_new never goes to memory and back: a block computes it for a (B0+1) x (B1+1) x (B2+1) brick -- its tile plus the
// forward neighbours the differences D_j u_new need -- into LDS, then updates q_j for the tile from there.  HBM traffic:
// of, u, q_0..2 in and u, q_0..2 out = 9 values per element (the three kernels: ~19); the one-element halos of
// neighbouring bricks come out of L2.  u and q are double buffered here (other blocks still read the old q next door).
// ============================================================================================
constexpr int kB0 = 8, kB1 = 8, kB2 = 64;

template <typename T, bool ISO>
__device__ __forceinline__ T t_comp_at(const QPlanes<T, 3>& q, int64_t off, int j, T ell, T kq) {
    // component j of t = ell * (b - d) at `off`; the anisotropic split is per component
    if constexpr (ISO) {
        T qq[3], d[3], b[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) qq[i] = q.in[i][off];
        split_vars<T, 3, true>(qq, kq, d, b);
        return ell * (b[j] - d[j]);
    } else {
        const T qq[1] = {q.in[j][off]};
        T d[1], b[1];
        split_vars<T, 1, false>(qq, kq, d, b);
        return ell * (b[0] - d[0]);
    }
}

template <typename T, bool ISO>
__global__ void __launch_bounds__(256) k_fused3d(Geom g, const T* __restrict__ of, QPlanes<T, 3> q,
                                                  const T* __restrict__ u_in, T* __restrict__ u_out, Coef<T> c,
                                                  CoefPlanes<T> cp, int nt1, int nt2) {
    constexpr int S2 = kB2 + 1, S1 = (kB1 + 1) * S2, NB = (kB0 + 1) * S1;
    __shared__ T un[NB];
    const int bid = blockIdx.x;
    const int tt2 = bid % nt2, tt1 = (bid / nt2) % nt1, tt0 = bid / (nt2 * nt1);
    const int64_t o0 = (int64_t)tt0 * kB0, o1 = (int64_t)tt1 * kB1, o2 = (int64_t)tt2 * kB2;
    // ---- phase 1: u_new on the brick
    for (int idx = threadIdx.x; idx < NB; idx += 256) {
        const int l0 = idx / S1, r = idx - l0 * S1, l1 = r / S2, l2 = r - l1 * S2;
        const int64_t i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
        if (i0 >= g.n0 || i1 >= g.n1 || i2 >= g.n2) continue;
        const int64_t off = i0 * g.pitch + i1 * g.n2 + i2;
        // clamped neighbour offsets (replicate ghosts, jacobi.py:43,52; one-sided ends, derivatives.py:33,57)
        const int64_t om[3] = {i0 > 0 ? off - g.pitch : off, i1 > 0 ? off - g.n2 : off, i2 > 0 ? off - 1 : off};
        const int64_t op[3] = {i0 < g.n0 - 1 ? off + g.pitch : off, i1 < g.n1 - 1 ? off + g.n2 : off,
                               i2 < g.n2 - 1 ? off + 1 : off};
        const T ell_c = cp.ell ? cp.ell[off] : c.ell;
        const T kq_c = cp.kq ? cp.kq[off] : c.kq;
        T tc[3];
        {
            T qq[3], d[3], b[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) qq[j] = q.in[j][off];
            split_vars<T, 3, ISO>(qq, kq_c, d, b);
#pragma unroll
            for (int j = 0; j < 3; ++j) tc[j] = ell_c * (b[j] - d[j]);
        }
        T rr = of[off];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int64_t o = om[j];
            const T ell_n = cp.ell ? cp.ell[o] : c.ell;
            const T kq_n = cp.kq ? cp.kq[o] : c.kq;
            rr = rr + (tc[j] - t_comp_at<T, ISO>(q, o, j, ell_n, kq_n));
        }
        const T diag = cp.diag ? cp.diag[off] : c.diag;
        const T rs = rr / diag;
        T acc = T(0);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            acc = acc + u_in[om[ax]];
            acc = acc + u_in[op[ax]];
        }
        const T dsc = cp.dsc ? cp.dsc[off] : c.dsc;
        un[idx] = rs + acc / dsc;
    }
    __syncthreads();
    // ---- phase 2: the tile's u_new goes out, q_j <- D_j u_new + b_j
    for (int idx = threadIdx.x; idx < kB0 * kB1 * kB2; idx += 256) {
        const int l0 = idx / (kB1 * kB2), r = idx - l0 * (kB1 * kB2), l1 = r / kB2, l2 = r - l1 * kB2;
        const int64_t i0 = o0 + l0, i1 = o1 + l1, i2 = o2 + l2;
        if (i0 >= g.n0 || i1 >= g.n1 || i2 >= g.n2) continue;
        const int64_t off = i0 * g.pitch + i1 * g.n2 + i2;
        const int l = l0 * S1 + l1 * S2 + l2;
        const T uc = un[l];
        u_out[off] = uc;
        const T kq = cp.kq ? cp.kq[off] : c.kq;
        T qq[3], d[3], b[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) qq[j] = q.in[j][off];
        split_vars<T, 3, ISO>(qq, kq, d, b);
        const T n0v = i0 < g.n0 - 1 ? un[l + S1] : uc;
        const T n1v = i1 < g.n1 - 1 ? un[l + S2] : uc;
        const T n2v = i2 < g.n2 - 1 ? un[l + 1] : uc;
        q.out[0][off] = (n0v - uc) + b[0];
        q.out[1][off] = (n1v - uc) + b[1];
        q.out[2][off] = (n2v - uc) + b[2];
    }
}

// ============================================================================================
// The same iteration as a 2.5-D march.  A block works on a P1 x P2 patch of the (axis 1, axis 2) plane (1024 columns,
// four per thread; 16 x 64, 8 x 128 or 4 x 256, chosen per volume) and walks down axis 0 through a chunk of planes.  What a
// column needs from its own past and future along axis 0 (u one plane up and down, t_0 one plane up, u_new one plane
// down) stays in the owning thread's registers; what it needs from its in-plane neighbours goes through two-slot LDS
// planes written ONE STEP AHEAD (u, t_1, t_2 of plane p + 1 while plane p is being computed; u_new of plane p for the
// q update of plane p at the next step), so one barrier per plane is enough.  The patch's last row and last column
// only compute u_new for the forward differences of their neighbours and store nothing: patches tile the plane with a
// stride of (P1 - 1, P2 - 1).  Columns on the rim fetch what lies
// outside from global memory (L2: neighbouring patches run on the same XCD, block_to_patch).  Every global value is read
// once per owning block: 9 values per element and iteration plus the rim.  Same expressions, same order as k_rhs /
// k_jacobi / k_shrink: same bits.
// ============================================================================================
template <int M2>
struct MarchGeom {
    static constexpr int P2 = M2, P1 = 1024 / M2;     // patch with its forward rim: 1024 columns, four per thread
    static constexpr int kM2 = P2 - 1, kM1 = P1 - 1;   // owned rows / columns = the tiling stride
    static constexpr int COLS = P1 * P2;
    static constexpr int PER = (COLS + 255) / 256;
};

template <typename T, bool ISO, int M2>
__global__ void __launch_bounds__(256) k_march3d(Geom g, const T* __restrict__ of, QPlanes<T, 3> q,
                                                  const T* __restrict__ u_in, T* __restrict__ u_out, Coef<T> c,
                                                  CoefPlanes<T> cp, int nt1, int nt2, int nchunk, int ppc) {
    using MG = MarchGeom<M2>;
    constexpr int kM1 = MG::kM1, kM2 = MG::kM2, S2 = MG::P2, COLS = MG::COLS, PER = MG::PER;
    __shared__ T s_u[2][COLS], s_t1[2][COLS], s_t2[2][COLS], s_un[2][COLS];
    // XCD x (= blockIdx % 8) takes a contiguous run of patches in (chunk, row of patches, patch) order: the patches that
    // exchange rim values through L2 sit on the same XCD
    const int total = nt1 * nt2 * nchunk;
    const int per_xcd = (total + 7) >> 3;
    const int bid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (bid >= total) return;  // block-uniform
    const int tt2 = bid % nt2, tt1 = (bid / nt2) % nt1, chunk = bid / (nt2 * nt1);
    const int64_t P0 = (int64_t)chunk * ppc;
    if (P0 >= g.n0) return;  // block-uniform
    const int64_t P1 = (P0 + ppc < g.n0) ? P0 + ppc : g.n0;
    const int64_t Pend = (P1 < g.n0) ? P1 : g.n0 - 1;  // last plane whose u_new is needed
    const int64_t r0 = (int64_t)tt1 * kM1, c0 = (int64_t)tt2 * kM2;

    // per-column geometry and state
    int col[PER], l1[PER], l2[PER];
    int64_t i1[PER], i2[PER], off[PER];
    bool in[PER], owned[PER];
    T um[PER], uc[PER], up[PER], t0m[PER], tcur[PER][3], bcur[PER][3], ofc[PER], unm[PER], bm[PER][3];

    auto coef_ell = [&](int64_t o) -> T { return cp.ell ? cp.ell[o] : c.ell; };
    auto coef_kq = [&](int64_t o) -> T { return cp.kq ? cp.kq[o] : c.kq; };
    auto split_at = [&](int64_t o, T ell, T kq, T (&t)[3], T (&b)[3]) {
        T qq[3], d[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) qq[j] = q.in[j][o];
        split_vars<T, 3, ISO>(qq, kq, d, b);
#pragma unroll
        for (int j = 0; j < 3; ++j) t[j] = ell * (b[j] - d[j]);
    };

#pragma unroll
    for (int k = 0; k < PER; ++k) {
        col[k] = (int)threadIdx.x + 256 * k;
        l1[k] = col[k] / S2;
        l2[k] = col[k] - l1[k] * S2;
        i1[k] = r0 + l1[k];
        i2[k] = c0 + l2[k];
        in[k] = col[k] < COLS && i1[k] < g.n1 && i2[k] < g.n2;
        owned[k] = in[k] && l1[k] < kM1 && l2[k] < kM2;
        off[k] = P0 * g.pitch + i1[k] * g.n2 + i2[k];
        if (in[k]) {
            const int64_t o = off[k];
            uc[k] = u_in[o];
            um[k] = u_in[P0 > 0 ? o - g.pitch : o];
            up[k] = u_in[P0 < g.n0 - 1 ? o + g.pitch : o];
            ofc[k] = of[o];
            split_at(o, coef_ell(o), coef_kq(o), tcur[k], bcur[k]);
            if (P0 > 0) {
                t0m[k] = t_comp_at<T, ISO>(q, o - g.pitch, 0, coef_ell(o - g.pitch), coef_kq(o - g.pitch));
            } else {
                t0m[k] = tcur[k][0];  // the one-sided difference at the first plane is t - t (derivatives.py:57)
            }
            s_u[P0 & 1][col[k]] = uc[k];
            s_t1[P0 & 1][col[k]] = tcur[k][1];
            s_t2[P0 & 1][col[k]] = tcur[k][2];
        }
        unm[k] = T(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) bm[k][j] = T(0);
    }
    __syncthreads();

    for (int64_t p = P0; p <= Pend; ++p) {
        const int s = (int)(p & 1);
        const bool have_next = p + 1 <= Pend;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (!in[k]) continue;
            const int64_t o = off[k];
            // prefetch the next plane of this column (a two-step prefetch was measured: the extra registers cost a
            // block per CU and 20 % of the speed)
            T nq[3] = {T(0), T(0), T(0)}, nof = T(0), nu2 = T(0), nell = T(0), nkq = T(0);
            if (have_next) {
                const int64_t on = o + g.pitch;
#pragma unroll
                for (int j = 0; j < 3; ++j) nq[j] = q.in[j][on];
                nof = of[on];
                nu2 = u_in[p + 2 < g.n0 ? on + g.pitch : on];
                nell = coef_ell(on);
                nkq = coef_kq(on);
            }
            // in-plane neighbours of u(p): the patch's LDS plane, global memory beyond the rim, the value itself at
            // the array ends (replicate ghosts, jacobi.py:43,52)
            const bool e1m = i1[k] > 0, e1p = i1[k] < g.n1 - 1, e2m = i2[k] > 0, e2p = i2[k] < g.n2 - 1;
            const T u_rm = !e1m ? uc[k] : (l1[k] > 0 ? s_u[s][col[k] - S2] : u_in[o - g.n2]);
            const T u_rp = !e1p ? uc[k] : (l1[k] < kM1 ? s_u[s][col[k] + S2] : u_in[o + g.n2]);
            const T u_cm = !e2m ? uc[k] : (l2[k] > 0 ? s_u[s][col[k] - 1] : u_in[o - 1]);
            const T u_cp = !e2p ? uc[k] : (l2[k] < kM2 ? s_u[s][col[k] + 1] : u_in[o + 1]);
            // t_1 one row up, t_2 one column left (one-sided ends: t - t, derivatives.py:57)
            T t1n, t2n;
            if (!e1m) t1n = tcur[k][1];
            else if (l1[k] > 0) t1n = s_t1[s][col[k] - S2];
            else t1n = t_comp_at<T, ISO>(q, o - g.n2, 1, coef_ell(o - g.n2), coef_kq(o - g.n2));
            if (!e2m) t2n = tcur[k][2];
            else if (l2[k] > 0) t2n = s_t2[s][col[k] - 1];
            else t2n = t_comp_at<T, ISO>(q, o - 1, 2, coef_ell(o - 1), coef_kq(o - 1));
            // right-hand side and Jacobi sweep (k_rhs, k_jacobi)
            T rr = ofc[k];
            rr = rr + (tcur[k][0] - t0m[k]);
            rr = rr + (tcur[k][1] - t1n);
            rr = rr + (tcur[k][2] - t2n);
            const T rs = rr / (cp.diag ? cp.diag[o] : c.diag);
            T acc = T(0);
            acc = acc + um[k];
            acc = acc + up[k];
            acc = acc + u_rm;
            acc = acc + u_rp;
            acc = acc + u_cm;
            acc = acc + u_cp;
            const T unp = rs + acc / (cp.dsc ? cp.dsc[o] : c.dsc);
            s_un[s][col[k]] = unp;
            // plane p - 1 is complete now: u_new(p - 1) goes out, q_j <- D_j u_new + b_j (k_shrink)
            if (p > P0 && owned[k]) {
                const int sm = s ^ 1;
                const int64_t ob = o - g.pitch;
                const T n1v = e1p ? s_un[sm][col[k] + S2] : unm[k];
                const T n2v = e2p ? s_un[sm][col[k] + 1] : unm[k];
                u_out[ob] = unm[k];
                q.out[0][ob] = (unp - unm[k]) + bm[k][0];
                q.out[1][ob] = (n1v - unm[k]) + bm[k][1];
                q.out[2][ob] = (n2v - unm[k]) + bm[k][2];
            }
            // rotate the column's window and publish plane p + 1 for the neighbours
            unm[k] = unp;
#pragma unroll
            for (int j = 0; j < 3; ++j) bm[k][j] = bcur[k][j];
            t0m[k] = tcur[k][0];
            um[k] = uc[k];
            uc[k] = up[k];
            up[k] = nu2;
            off[k] = o + g.pitch;
            if (have_next) {
                ofc[k] = nof;
                T d[3];
                split_vars<T, 3, ISO>(nq, nkq, d, bcur[k]);
#pragma unroll
                for (int j = 0; j < 3; ++j) tcur[k][j] = nell * (bcur[k][j] - d[j]);
                s_u[s ^ 1][col[k]] = uc[k];
                s_t1[s ^ 1][col[k]] = tcur[k][1];
                s_t2[s ^ 1][col[k]] = tcur[k][2];
            }
        }
        __syncthreads();
    }
    // the volume's last plane: no plane below, its axis-0 difference is zero
    if (P1 == g.n0) {
        const int sm = (int)((g.n0 - 1) & 1);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (!owned[k]) continue;
            const int64_t ob = off[k] - g.pitch;  // off has moved one plane past the last one computed
            const bool e1p = i1[k] < g.n1 - 1, e2p = i2[k] < g.n2 - 1;
            const T n1v = e1p ? s_un[sm][col[k] + S2] : unm[k];
            const T n2v = e2p ? s_un[sm][col[k] + 1] : unm[k];
            u_out[ob] = unm[k];
            q.out[0][ob] = (unm[k] - unm[k]) + bm[k][0];
            q.out[1][ob] = (n1v - unm[k]) + bm[k][1];
            q.out[2][ob] = (n2v - unm[k]) + bm[k][2];
        }
    }
}

// ============================================================================================
// The 2.5-D march with N = 16 B / sizeof(T) ADJACENT columns per thread (four floats, two doubles; n2 % N == 0).  k_march3d above gives a thread four columns
// 256 apart: every column pays its own index arithmetic, predicates and 4-byte loads / stores / LDS accesses -- ~560 VALU
// instructions per column and plane against ~36 per element in the 2-D pipelines, which is what bounds it (VALU issue 0.45,
// waves waiting 70 %, one barrier per plane).  Here a thread owns columns l2 .. l2 + 3 of one patch row: 16-byte global and
// LDS accesses, one set of predicates, and the inner column neighbours (u, t_2, u_new) are the thread's own
// registers; only the outer two come from LDS (or, on the rim, from global memory).  Patch: 256 / TPR rows x 4 TPR
// columns; the last row and the last FOUR columns are the forward rim (u_new only), so patches tile with a stride of
// (256 / TPR - 1, N TPR - N) and every vector access stays 16-byte aligned (written for N = 4 below: "four" = N).  Same expressions in the same order as k_march3d: same bits.
// ============================================================================================
template <typename T, bool ISO>
__global__ void __launch_bounds__(256, (sizeof(T) == 8 ? 2 : (ISO ? 3 : 4))) k_march3d_v(Geom g, const T* __restrict__ of, QPlanes<T, 3> q,
                                                    const T* __restrict__ u_in, T* __restrict__ u_out, Coef<T> c,
                                                    CoefPlanes<T> cp, int nt1, int nt2, int nchunk, int ppc, int TPR) {
    // TPR threads (= 4 TPR columns) per patch row, 256 / TPR rows: the host picks the TPR in 4 .. 64 whose patches waste the
    // fewest rim and overhang columns on this volume (e.g. 17 for 256 columns: four patches of 64 owned + 4 rim columns)
    constexpr int N = VecOf<T>::N, COLS = 256 * N;   // 16 bytes per thread: four floats / two doubles
    const int M2 = TPR * N, P1R = 256 / TPR, kM1 = P1R - 1, kM2 = M2 - N;
    __shared__ __align__(16) T s_u[2][COLS], s_t1[2][COLS], s_t2[2][COLS], s_un[2][COLS];
    const int total = nt1 * nt2 * nchunk;
    const int per_xcd = (total + 7) >> 3;
    const int bid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (bid >= total) return;  // block-uniform
    const int tt2 = bid % nt2, tt1 = (bid / nt2) % nt1, chunk = bid / (nt2 * nt1);
    const int64_t P0 = (int64_t)chunk * ppc;
    if (P0 >= g.n0) return;  // block-uniform
    const int64_t P1 = (P0 + ppc < g.n0) ? P0 + ppc : g.n0;
    const int64_t Pend = (P1 < g.n0) ? P1 : g.n0 - 1;  // last plane whose u_new is needed
    const int l1 = (int)threadIdx.x / TPR, lq = (int)threadIdx.x % TPR, l2 = lq * N;
    const int col = l1 * M2 + l2;                        // first of the thread's four LDS slots (< P1R * M2 <= 1024)
    const int64_t i1 = (int64_t)tt1 * kM1 + l1, i2 = (int64_t)tt2 * kM2 + l2;
    const bool in = l1 < P1R && i1 < g.n1 && i2 < g.n2;  // n2 % 4 == 0: the four columns are inside together
    const bool owned = in && l1 < kM1 && lq < TPR - 1;
    const bool e1m = i1 > 0, e1p = i1 < g.n1 - 1, e2m = i2 > 0, e2p = i2 + N < g.n2;   // e2m: column 0's left, e2p: column 3's right
    int64_t off = P0 * g.pitch + i1 * g.n2 + i2;

    auto ld4 = [](const T* p, T (&v)[N]) { load_vec(p, v); };
    auto lds_ld4 = [](const T* p, T (&v)[N]) { load_vec(p, v); };
    auto lds_st4 = [](T* p, const T (&v)[N]) { store_vec(p, v); };
    auto coef4 = [&](const T* plane, T scalar, int64_t o, T (&v)[N]) {
        if (plane) load_vec(plane + o, v);
        else {
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = scalar;
        }
    };
    // t = ell (b - d) and b of four columns from their q
    auto split4 = [&](const T (&q0)[N], const T (&q1)[N], const T (&q2)[N], const T (&ell)[N], const T (&kq)[N], T (&t)[3][N], T (&b)[3][N]) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            T qq[3] = {q0[j], q1[j], q2[j]}, d[3], bb[3];
            split_vars<T, 3, ISO>(qq, kq[j], d, bb);
#pragma unroll
            for (int a = 0; a < 3; ++a) { t[a][j] = ell[j] * (bb[a] - d[a]); b[a][j] = bb[a]; }
        }
    };
    auto coef1 = [&](const T* plane, T scalar, int64_t o) -> T { return plane ? plane[o] : scalar; };

    T um[N], uc[N], up[N], t0m[N], tcur[3][N], bcur[3][N], ofc[N], unm[N], bm[3][N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        um[j] = uc[j] = up[j] = t0m[j] = ofc[j] = unm[j] = T(0);
#pragma unroll
        for (int a = 0; a < 3; ++a) tcur[a][j] = bcur[a][j] = bm[a][j] = T(0);
    }
    if (in) {
        const int64_t o = off;
        ld4(u_in + o, uc);
        ld4(u_in + (P0 > 0 ? o - g.pitch : o), um);
        ld4(u_in + (P0 < g.n0 - 1 ? o + g.pitch : o), up);
        ld4(of + o, ofc);
        T q0[N], q1[N], q2[N], ell[N], kq[N];
        ld4(q.in[0] + o, q0); ld4(q.in[1] + o, q1); ld4(q.in[2] + o, q2);
        coef4(cp.ell, c.ell, o, ell); coef4(cp.kq, c.kq, o, kq);
        split4(q0, q1, q2, ell, kq, tcur, bcur);
        if (P0 > 0) {
            const int64_t ob = o - g.pitch;
            T r0[N], r1[N], r2[N], el[N], kk[N], tt[3][N], bb[3][N];
            ld4(q.in[0] + ob, r0);
            if constexpr (ISO) { ld4(q.in[1] + ob, r1); ld4(q.in[2] + ob, r2); }
            else {
#pragma unroll
                for (int j = 0; j < N; ++j) r1[j] = r2[j] = T(0);
            }
            coef4(cp.ell, c.ell, ob, el); coef4(cp.kq, c.kq, ob, kk);
            if constexpr (ISO) {
                split4(r0, r1, r2, el, kk, tt, bb);
#pragma unroll
                for (int j = 0; j < N; ++j) t0m[j] = tt[0][j];
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const T qq[1] = {r0[j]};
                    T d[1], b1[1];
                    split_vars<T, 1, false>(qq, kk[j], d, b1);
                    t0m[j] = el[j] * (b1[0] - d[0]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) t0m[j] = tcur[0][j];  // the one-sided difference at the first plane is t - t
        }
        lds_st4(&s_u[P0 & 1][col], uc);
        lds_st4(&s_t1[P0 & 1][col], tcur[1]);
        lds_st4(&s_t2[P0 & 1][col], tcur[2]);
    }
    __syncthreads();

    for (int64_t p = P0; p <= Pend; ++p) {
        const int s = (int)(p & 1);
        const bool have_next = p + 1 <= Pend;
        if (in) {
            const int64_t o = off;
            T nq0[N], nq1[N], nq2[N], nof[N], nu2[N], nell[N], nkq[N];
#pragma unroll
            for (int j = 0; j < N; ++j) nq0[j] = nq1[j] = nq2[j] = nof[j] = nu2[j] = nell[j] = nkq[j] = T(0);
            if (have_next) {
                const int64_t on = o + g.pitch;
                ld4(q.in[0] + on, nq0); ld4(q.in[1] + on, nq1); ld4(q.in[2] + on, nq2);
                ld4(of + on, nof);
                ld4(u_in + (p + 2 < g.n0 ? on + g.pitch : on), nu2);
                coef4(cp.ell, c.ell, on, nell); coef4(cp.kq, c.kq, on, nkq);
            }
            // in-plane neighbours of u(p)
            T u_rm[N], u_rp[N];
            if (!e1m) {
#pragma unroll
                for (int j = 0; j < N; ++j) u_rm[j] = uc[j];
            } else if (l1 > 0) lds_ld4(&s_u[s][col - M2], u_rm);
            else ld4(u_in + o - g.n2, u_rm);
            if (!e1p) {
#pragma unroll
                for (int j = 0; j < N; ++j) u_rp[j] = uc[j];
            } else if (l1 < kM1) lds_ld4(&s_u[s][col + M2], u_rp);
            else ld4(u_in + o + g.n2, u_rp);
            const T u_left = !e2m ? uc[0] : (l2 > 0 ? s_u[s][col - 1] : u_in[o - 1]);
            const T u_right = !e2p ? uc[N - 1] : (lq < TPR - 1 ? s_u[s][col + N] : u_in[o + N]);
            // t_1 one row up, t_2 one column left
            T t1n[N];
            if (!e1m) {
#pragma unroll
                for (int j = 0; j < N; ++j) t1n[j] = tcur[1][j];
            } else if (l1 > 0) lds_ld4(&s_t1[s][col - M2], t1n);
            else {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    t1n[j] = t_comp_at<T, ISO>(q, o - g.n2 + j, 1, coef1(cp.ell, c.ell, o - g.n2 + j), coef1(cp.kq, c.kq, o - g.n2 + j));
            }
            T t2_left;
            if (!e2m) t2_left = tcur[2][0];
            else if (l2 > 0) t2_left = s_t2[s][col - 1];
            else t2_left = t_comp_at<T, ISO>(q, o - 1, 2, coef1(cp.ell, c.ell, o - 1), coef1(cp.kq, c.kq, o - 1));
            T diag[N], dsc[N], unp[N];
            coef4(cp.diag, c.diag, o, diag); coef4(cp.dsc, c.dsc, o, dsc);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T rr = ofc[j];
                rr = rr + (tcur[0][j] - t0m[j]);
                rr = rr + (tcur[1][j] - t1n[j]);
                rr = rr + (tcur[2][j] - (j == 0 ? t2_left : tcur[2][j - 1]));
                const T rs = rr / diag[j];
                T acc = T(0);
                acc = acc + um[j];
                acc = acc + up[j];
                acc = acc + u_rm[j];
                acc = acc + u_rp[j];
                acc = acc + (j == 0 ? u_left : uc[j - 1]);
                acc = acc + (j == N - 1 ? u_right : uc[j + 1]);
                unp[j] = rs + acc / dsc[j];
            }
            lds_st4(&s_un[s][col], unp);
            // plane p - 1 is complete now: u_new(p - 1) goes out, q_j <- D_j u_new + b_j
            if (p > P0 && owned) {
                const int sm = s ^ 1;
                const int64_t ob = o - g.pitch;
                T n1v[N];
                if (e1p) lds_ld4(&s_un[sm][col + M2], n1v);
                else {
#pragma unroll
                    for (int j = 0; j < N; ++j) n1v[j] = unm[j];
                }
                const T un_right = e2p ? s_un[sm][col + N] : unm[N - 1];
                T o0[N], o1[N], o2[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    o0[j] = (unp[j] - unm[j]) + bm[0][j];
                    o1[j] = (n1v[j] - unm[j]) + bm[1][j];
                    o2[j] = ((j == N - 1 ? un_right : unm[j + 1]) - unm[j]) + bm[2][j];
                }
                store_vec(u_out + ob, unm);
                store_vec(q.out[0] + ob, o0);
                store_vec(q.out[1] + ob, o1);
                store_vec(q.out[2] + ob, o2);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                unm[j] = unp[j];
#pragma unroll
                for (int a = 0; a < 3; ++a) bm[a][j] = bcur[a][j];
                t0m[j] = tcur[0][j];
                um[j] = uc[j];
                uc[j] = up[j];
                up[j] = nu2[j];
            }
            off = o + g.pitch;
            if (have_next) {
#pragma unroll
                for (int j = 0; j < N; ++j) ofc[j] = nof[j];
                split4(nq0, nq1, nq2, nell, nkq, tcur, bcur);
                lds_st4(&s_u[s ^ 1][col], uc);
                lds_st4(&s_t1[s ^ 1][col], tcur[1]);
                lds_st4(&s_t2[s ^ 1][col], tcur[2]);
            }
        }
        __syncthreads();
    }
    // the volume's last plane: no plane below, its axis-0 difference is zero
    if (P1 == g.n0 && owned) {
        const int sm = (int)((g.n0 - 1) & 1);
        const int64_t ob = off - g.pitch;  // off has moved one plane past the last one computed
        T n1v[N];
        if (e1p) lds_ld4(&s_un[sm][col + M2], n1v);
        else {
#pragma unroll
            for (int j = 0; j < N; ++j) n1v[j] = unm[j];
        }
        const T un_right = e2p ? s_un[sm][col + N] : unm[N - 1];
        T o0[N], o1[N], o2[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            o0[j] = (unm[j] - unm[j]) + bm[0][j];
            o1[j] = (n1v[j] - unm[j]) + bm[1][j];
            o2[j] = ((j == N - 1 ? un_right : unm[j + 1]) - unm[j]) + bm[2][j];
        }
        store_vec(u_out + ob, unm);
        store_vec(q.out[0] + ob, o0);
        store_vec(q.out[1] + ob, o1);
        store_vec(q.out[2] + ob, o2);
    }
}

// explicit-(d, b) variant of the shrink step, used when the state cannot be kept compressed (adaptive ell:
// the threshold changes between the iteration that produced d, b and the one that consumes them)
template <typename T, int DIM, bool ISO>
__global__ void __launch_bounds__(256) k_shrink_db(Geom g, const T* __restrict__ unew, T* d0, T* d1, T* d2, T* b0, T* b1,
                                                    T* b2, Coef<T> c, CoefPlanes<T> cp) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    T* dp[3] = {d0, d1, d2};
    T* bp[3] = {b0, b1, b2};
    const T kq = cp.kq ? cp.kq[e.off] : c.kq;
    const T uc = unew[e.off];
    T q[DIM], d[DIM], b[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) q[j] = (unew[nb_off(g, e, j, +1)] - uc) + bp[j][e.off];
    split_vars<T, DIM, ISO>(q, kq, d, b);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        dp[j][e.off] = d[j];
        bp[j][e.off] = b[j];
    }
}

// state -> explicit planes (entering explicit mode)
template <typename T, int DIM, bool ISO>
__global__ void __launch_bounds__(256) k_expand_planes(Geom g, QPlanes<T, DIM> q, Coef<T> c, CoefPlanes<T> cp, T* d0, T* d1,
                                                        T* d2, T* b0, T* b1, T* b2) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    T* dp[3] = {d0, d1, d2};
    T* bp[3] = {b0, b1, b2};
    const T kq = cp.kq ? cp.kq[e.off] : c.kq;
    T qq[DIM], d[DIM], b[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) qq[j] = q.in[j][e.off];
    split_vars<T, DIM, ISO>(qq, kq, d, b);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        dp[j][e.off] = d[j];
        bp[j][e.off] = b[j];
    }
}

// explicit planes -> dense (*shape, dim) arrays
template <typename T, int DIM>
__global__ void __launch_bounds__(256) k_interleave(Geom g, const T* p0, const T* p1, const T* p2, T* __restrict__ dst) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const T* pl[3] = {p0, p1, p2};
    const int64_t o = (e.row * g.L + e.x) * DIM;
#pragma unroll
    for (int j = 0; j < DIM; ++j) dst[o + j] = pl[j][e.off];
}

// scalar coefficients -> planes with exactly the scalar path's values
template <typename T>
__global__ void __launch_bounds__(256) k_fill_coef(Geom g, Coef<T> c, T* ell_o, T* diag_o, T* dsc_o, T* kq_o) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    ell_o[e.off] = c.ell;
    diag_o[e.off] = c.diag;
    dsc_o[e.off] = c.dsc;
    kq_o[e.off] = c.kq;
}

// adaptive regularisation (split_bregman_tvd.py:149-158): ell = 1 / max(sum_j |D+_j u|, 1e-12), kq = mu / ell.
// diag / dsc are NOT touched: the reference's Jacobi keeps its cached diagonal (quirk Q3).
template <typename T, int DIM>
__global__ void __launch_bounds__(256) k_adapt_ell(Geom g, const T* __restrict__ u, T mu_s, const T* __restrict__ mu_p,
                                                    T* __restrict__ ell_o, T* __restrict__ kq_o) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const T uc = u[e.off];
    T l1 = T(0);
#pragma unroll
    for (int j = 0; j < DIM; ++j) l1 = l1 + t_abs(u[nb_off(g, e, j, +1)] - uc);
    const T lo = T(1e-12);
    const T ell = T(1) / (l1 > lo ? l1 : lo);
    ell_o[e.off] = ell;
    kq_o[e.off] = (mu_p ? mu_p[e.off] : mu_s) / ell;
}

// expand the state into the reference's split variables, layout (*shape, dim) (download / x0 round trip)
template <typename T, int DIM, bool ISO>
__global__ void __launch_bounds__(256) k_expand(Geom g, QPlanes<T, DIM> q, Coef<T> c, CoefPlanes<T> cp,
                                                 T* __restrict__ dout, T* __restrict__ bout) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const T kq = cp.kq ? cp.kq[e.off] : c.kq;
    T qq[DIM], d[DIM], b[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) qq[j] = q.in[j][e.off];
    split_vars<T, DIM, ISO>(qq, kq, d, b);
    const int64_t o = (e.row * g.L + e.x) * DIM;  // dense (no pitch)
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        if (dout) dout[o + j] = d[j];
        if (bout) bout[o + j] = b[j];
    }
}

// de-interleave a dense (*shape, dim) array into pitched planes
template <typename T, int DIM>
__global__ void __launch_bounds__(256) k_deinterleave(Geom g, const T* __restrict__ src, T* p0, T* p1, T* p2) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const int64_t o = (e.row * g.L + e.x) * DIM;
    T* pl[3] = {p0, p1, p2};
#pragma unroll
    for (int j = 0; j < DIM; ++j) pl[j][e.off] = src[o + j];
}

// out = a * s  (s scalar or plane)  -- omega * img, split_bregman_tvd.py:85
template <typename T>
__global__ void __launch_bounds__(256) k_scale(Geom g, const T* __restrict__ a, T s, const T* __restrict__ splane,
                                                T* __restrict__ out) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    out[e.off] = (splane ? splane[e.off] : s) * a[e.off];
}

// numpy2 flow: omega * img of a float32 image (held exactly in a float64 plane) is a float32 product
// out32 (optional): the same float32 products as a float32 plane of the same pitch -- what the K-stage numpy2 pipelines read
// (4 instead of 8 bytes per element and launch)
static __global__ void __launch_bounds__(256) k_scale_np2(Geom g, const double* __restrict__ a, float s, double* __restrict__ out,
                                                          float* __restrict__ out32 = nullptr) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const float v = s * (float)a[e.off];
    out[e.off] = (double)v;
    if (out32) out32[e.off] = v;
}

template <typename T>
__global__ void __launch_bounds__(256) k_divide(Geom g, const T* __restrict__ a, T s, const T* __restrict__ splane,
                                                 T* __restrict__ out) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    out[e.off] = a[e.off] / (splane ? splane[e.off] : s);
}

// coefficient planes from mu / omega / ell given as scalars or planes, with NumPy's evaluation order:
//   ell  = 2*mu                       (when not given)            split_bregman_tvd.py:62-63
//   diag = omega + ell*2*dim/h**2                                 jacobi.py:72
//   dsc  = diag / (ell/h**2)                                      jacobi.py:104-106
//   kq   = mu / ell                                               split_bregman_tvd.py:140,145
// scalar-only sub-expressions are folded in double on the host and arrive here rounded to T.
template <typename T>
struct CoefIn {
    const T* mu;
    const T* omega;
    const T* ell;
    T mu_s, omega_s, ell_s;  // scalars rounded to T
    T two_dim_over_h2_ell_s; // T(ell*2*dim/h^2) when ell is scalar
    T ell_over_h2_s;         // T(ell/h^2) when ell is scalar
    T h2;                    // T(h^2)
    T two_dim;               // unused helper
    T diag_s, dsc_s;         // omega AND ell scalar (only mu is an array): the reference's diagonal is a Python float
    T kq_s;                  // mu AND ell scalar (only omega is an array): the threshold mu / ell is a Python float, rounded ONCE
                             // (float(0.1) / float(0.3) is one ulp off float(0.1 / 0.3): found by tests/fuzz/fuzz_r4.py)
    int dim;
    int has_ell;
};

// the four derived coefficients of one element from the values of the given inputs at that element
// (mu_v / omega_v / ell_v are only read where the corresponding plane exists)
template <typename T>
__device__ __forceinline__ void derive_coefs(const CoefIn<T>& in, T mu_v, T omega_v, T ell_v, T& ell, T& diag, T& dsc, T& kq) {
    const bool ell_is_arr = in.ell != nullptr || (!in.has_ell && in.mu != nullptr);
    T lap, ell_h2;
    if (ell_is_arr) {
        ell = in.ell ? ell_v : T(2) * mu_v;
        lap = (ell * T(2)) * T(in.dim);
        ell_h2 = ell;
        if (in.h2 != T(1)) {  // x / 1 == x: skip the two divisions for the usual h = 1
            lap = lap / in.h2;
            ell_h2 = ell / in.h2;
        }
    } else {
        ell = in.ell_s;
        lap = in.two_dim_over_h2_ell_s;
        ell_h2 = in.ell_over_h2_s;
    }
    const T mu = in.mu ? mu_v : in.mu_s;
    if (!ell_is_arr && !in.omega) {
        // omega + ell*2*dim/h**2 and its quotient by ell/h**2 are folded in double by Python and rounded once
        diag = in.diag_s;
        dsc = in.dsc_s;
    } else {
        const T omega = in.omega ? omega_v : in.omega_s;
        diag = omega + lap;
        dsc = diag / ell_h2;
    }
    kq = (!in.mu && !ell_is_arr) ? in.kq_s : mu / ell;
}

template <typename T>
__global__ void __launch_bounds__(256) k_coef_planes(Geom g, CoefIn<T> in, T* __restrict__ ell_o, T* __restrict__ diag_o,
                                                      T* __restrict__ dsc_o, T* __restrict__ kq_o) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    T ell, diag, dsc, kq;
    derive_coefs<T>(in, in.mu ? in.mu[e.off] : T(0), in.omega ? in.omega[e.off] : T(0), in.ell ? in.ell[e.off] : T(0), ell,
                    diag, dsc, kq);
    ell_o[e.off] = ell;
    diag_o[e.off] = diag;
    dsc_o[e.off] = dsc;
    if (kq_o) kq_o[e.off] = kq;
}

// finite differences, utils/derivatives.py:16-59
template <typename T>
__global__ void __launch_bounds__(256) k_diff(Geom g, const T* __restrict__ in, int axis, int mode, T h, int use_h,
                                               T* __restrict__ out) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    T r;
    if (mode == 0)
        r = in[nb_off(g, e, axis, +1)] - in[e.off];
    else
        r = in[e.off] - in[nb_off(g, e, axis, -1)];
    out[e.off] = use_h ? r / h : r;
}

// ---- ConcentrationAnalysis fusion: diff -> reduce -> clean -> model, one pass over the colour frame -------
struct ModelOps {
    int n;
    int kind[8];
    int flags[8];
    double a[8], b[8];
};

template <typename T>
__device__ __forceinline__ T apply_model_ops(T x, const ModelOps& m) {
    for (int i = 0; i < m.n; ++i) {
        if (m.kind[i] == 0) {
            x = (T)m.a[i] * x + (T)m.b[i];
        } else if (m.kind[i] == 1) {
            x = (T)m.a[i] * x;
        } else {
            if ((m.flags[i] & 1) && x < (T)m.a[i]) x = (T)m.a[i];  // np.clip: maximum(x, lo) then minimum(., hi)
            if ((m.flags[i] & 2) && x > (T)m.b[i]) x = (T)m.b[i];
        }
    }
    return x;
}

template <typename T>
__device__ __forceinline__ T diff_value(T v, T bse, bool has_base, int option) {
    // concentrationanalysis.py:328-364
    const T d = has_base ? v - bse : v;
    switch (option) {
        case 1: return has_base ? t_max0(d) : t_max0(v);                   // positive: clip(img - base, 0, None)
        case 2: return has_base ? t_max0(bse - v) : t_max0(-v);            // negative: clip(base - img, 0, None)
        case 3: return t_abs(d);                                           // absolute: |img - base|
        default: return d;                                                 // plain
    }
}

// Element of a probe frame as the float value the reference subtracts the baseline from: float frames as they are, integer
// frames the scikit-image way (concentrationanalysis.py:241-246 -> img_as(float): np.multiply(x, 1. / imax, dtype=float64)).
template <typename S> struct SigLoad { using F = S; static __device__ __forceinline__ F get(S v) { return v; } };
template <> struct SigLoad<unsigned char> {
    using F = double;
    static __device__ __forceinline__ double get(unsigned char v) { return (double)v * (1.0 / 255); }
};
template <> struct SigLoad<unsigned short> {
    using F = double;
    static __device__ __forceinline__ double get(unsigned short v) { return (double)v * (1.0 / 65535); }
};

// S: stored type of the probe frame, B: of the baseline, T: of the scalar signal (the context).  Difference and reduction run
// in NumPy's promotion D of probe and baseline; the signal is widened to T where the cleaning filter promotes it
// (concentrationanalysis.py:328-395).
// O: element type of the plane the signal goes into -- T, or double for a float32 signal in a numpy2 context (the reference's
// float32 flow keeps its iterate in float64 planes; the signal itself is formed in float32 as NumPy does)
template <typename T, int C, typename S, typename B, typename O = T>
__global__ void __launch_bounds__(256) k_signal(Geom g, const S* __restrict__ probe, const B* __restrict__ base,
                                                 const T* __restrict__ filt, int option, int reduction, ModelOps pre,
                                                 O* __restrict__ out) {
    using F = typename SigLoad<S>::F;
    using D = std::conditional_t<(sizeof(F) > sizeof(B)), F, B>;
    static_assert(sizeof(T) >= sizeof(D), "the signal cannot be narrower than probe - baseline");
    // g describes the SCALAR image (n2 == 1); probe/base are dense (H, W, C)
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const int64_t pix = e.row * g.L + e.x;
    D d[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
        d[c] = diff_value<D>((D)SigLoad<S>::get(probe[pix * C + c]), base ? (D)base[pix * C + c] : D(0), base != nullptr, option);
    D sig;
    if constexpr (C == 1) {
        sig = d[0];
    } else {
        switch (reduction) {
            case 1: sig = d[0]; break;
            case 2: sig = d[1]; break;
            case 3: sig = d[2]; break;
            case 4: sig = d[0] + d[1]; break;
            case 5: {  // cv2.COLOR_RGB2GRAY on float32
                const float gr = (float)d[0] * 0.299f + (float)d[1] * 0.587f + (float)d[2] * 0.114f;
                sig = (D)gr;
            } break;
            default: {  // negative-key: 1 - min(1 - rgb)
                const D c0 = D(1) - d[0], c1 = D(1) - d[1], c2 = D(1) - d[2];
                D key = c0 < c1 ? c0 : c1;
                key = key < c2 ? key : c2;
                sig = D(1) - key;
            }
        }
    }
    T s = (T)sig;
    if (filt) s = t_max0(s - filt[pix]);  // np.clip(img - filter, 0, None)
    out[e.off] = (O)apply_model_ops<T>(s, pre);
}

// I: element type of the plane -- T, or double in a numpy2 context: convert_dtype(img_iter, float32) (split_bregman_tvd.py:177)
// rounds the float64 iterate first, the model runs on the float32 array
template <typename T, typename I = T>
__global__ void __launch_bounds__(256) k_model(Geom g, const I* __restrict__ in, ModelOps post, T* __restrict__ out_dense) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    out_dense[e.row * g.L + e.x] = apply_model_ops<T>((T)in[e.off], post);
}

// energy functional 0.5 ||omega (x - img)||^2 + sum_j sum |mu D+_j x|  (split_bregman_tvd.py:71-77, verbose only).
// Diagnostic: evaluated in double from the stored planes (omega*img instead of img, mu = kq*ell).
template <typename T, int DIM>
__global__ void __launch_bounds__(256) k_energy(Geom g, const T* __restrict__ x, const T* __restrict__ of, double omega_s,
                                                 const T* __restrict__ omega_p, Coef<T> c, CoefPlanes<T> cp,
                                                 double* __restrict__ partial) {
    __shared__ double sm[256 / kWave];
    ElemIdx e;
    double v = 0.0;
    if (elem_index(g, e)) {
        const double om = omega_p ? (double)omega_p[e.off] : omega_s;
        const double fid = om * (double)x[e.off] - (double)of[e.off];
        const double mu = (double)(cp.kq ? cp.kq[e.off] : c.kq) * (double)(cp.ell ? cp.ell[e.off] : c.ell);
        v = 0.5 * fid * fid;
#pragma unroll
        for (int j = 0; j < DIM; ++j) v += fabs(mu * ((double)x[nb_off(g, e, j, +1)] - (double)x[e.off]));
    }
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) v += __shfl_down(v, s, kWave);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) s += sm[w];
        partial[blockIdx.x] = s;
    }
}

// ---- reductions: sum (a - b)^2 (b may be null) over rows [row0, row0 + nrows) -----------------
// deterministic two-stage tree in double (per-block partials, then one block)
template <typename T>
__global__ void __launch_bounds__(256) k_sqdiff_partial(Geom g, const T* __restrict__ a, const T* __restrict__ b,
                                                         int64_t row0, int64_t nrows, double* __restrict__ partial) {
    __shared__ double sm[256 / kWave];
    double acc = 0.0;
    const int64_t total = nrows * g.L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / g.L;
        const int64_t x = i - r * g.L;
        const int64_t off = (row0 + r) * g.pitch + x;
        const double v = b ? (double)a[off] - (double)b[off] : (double)a[off];
        acc += v * v;
    }
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) acc += __shfl_down(acc, s, kWave);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) sm[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) s += sm[w];
        partial[blockIdx.x] = s;
    }
}

static __global__ void __launch_bounds__(256) k_sum_partials(const double* __restrict__ partial, int n, double* __restrict__ out) {
    // block b sums the slice [b*chunk, (b+1)*chunk) of `partial` into out[b]; a fixed tree, so the result does
    // not depend on timing.  Long lists are reduced in two launches (many blocks, then one).
    __shared__ double sm[256];
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) acc += partial[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

// k_sum_partials over up to four lists of the same length in one launch: blockIdx.y selects the list, the tree of every
// list is the one k_sum_partials builds (same values); out is [4][gridDim.x].
struct SumLists { const double* p[4]; };
static __global__ void __launch_bounds__(256) k_sum_partials_lists(SumLists lists, int n, double* __restrict__ out) {
    __shared__ double sm[256];
    const double* __restrict__ partial = lists.p[blockIdx.y];
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double acc = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += 256) acc += partial[i];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}

// ============================================================================================
// Fused sweep (dim = 2, one Jacobi sweep): rhs + Jacobi + shrink of one outer iteration in a single
// pass.  Algorithmic traffic per element: read of, u, q0, q1; write u, q0, q1  = 7 values.
//
// Decomposition: one WAVE owns a column strip of 64 lanes x N elements (16 B per lane) and a chunk of
// rows, and marches down the rows keeping a rolling window in registers:
//     um, uc, up   u at rows r-1, r, r+1            t0m   ell*(b0-d0) at row r-1
//     upm          u_new at row r-1                  b0m   b0 at row r-1
// Row neighbours are registers; column (+-C) and channel (+-1) neighbours come from adjacent lanes by
// DPP wave shifts.  Strips overlap by LH lanes on the left and RH lanes on the right so that no wave
// ever needs another wave's data: no LDS, no barriers.  The chunk's last step computes u_new of the
// first row of the next chunk (not stored) to finish q0 of the chunk's last row.
//
// Image borders are handled by clamping: row indices are clamped in the address, column / channel
// neighbours by selecting the element itself (loop-invariant lane masks).  This reproduces the
// reference's replicate ghosts (jacobi.py:43,52) and zero one-sided differences
// (derivatives.py:33,57) with the same arithmetic (x - x = 0).
// ============================================================================================

template <typename T>
struct FusedArgs {
    const T* of;
    const T* u_in;
    T* u_out;
    const T* q0_in;
    const T* q1_in;
    T* q0_out;
    T* q1_out;
    int64_t n0, L, pitch;
    // Row chunks: up to three runs of equally long chunks.  Run s covers rows [seg_row[s], seg_row[s + 1]) with chunks of
    // seg_rpc[s] rows, the first of which has index seg_chunk[s] (seg_chunk[3] = nchunks).  An ordinary launch uses run
    // 1 only; a slab launch that is split for the halo exchange puts short chunks next to the slab interfaces (runs 0 and
    // 2) so that they finish early (chunking never changes results).
    int64_t seg_row[4];
    int64_t seg_rpc[3];
    int seg_chunk[4];
    int nstrips;       // strips per row
    int nchunks;
    int remap;         // XCD-aware block -> tile mapping (block_to_tile)
    // Slab launches that precede a halo exchange: the chunks of runs 0 and 2 (next to the slab interfaces) are "class A".
    // Their blocks come first in the grid on every XCD, they are short, and each of their waves bumps *signal when its
    // rows are in memory, so that the exchange can start while the long interior chunks (class B) are still running.
    // Ordinary launches have no class A chunks (seg_chunk[1] = 0, seg_chunk[2] = nchunks) and signal = nullptr.
    unsigned* signal;
    Coef<T> c;
    CoefPlanes<T> cp;
    // heterogeneous coefficients: when `derive` is set the kernel reads only the planes the caller gave (mu, omega,
    // ell -- usually just mu) and recomputes ell / diag / dsc / kq per element (derive_coefs: the very expressions that
    // filled cp) instead of reading four derived planes: 32 instead of 44 bytes per element for an array-valued mu
    CoefIn<T> ci;
    int derive;
    int kqv;   // array-valued mu only (scalar omega, ell): the scalar pipeline with the shrink threshold mu / ell read per
               // element from cp.kq (K = kHetFuseK; k_fused_multi<..., KQV>)
    // optional (k_fused_multi): per-wave partial sums of (u_{n+s+1} - u_{n+s})^2 for each fused iteration s,
    // layout [K][nwaves_total]; nullptr = off.  Lets the eps test of split_bregman_tvd.py:161-175 run on
    // K-iteration launches without any extra pass over the data.
    double* inc_partial;
    int nwaves_total;
    int np2_first;     // NP2 pipelines (the reference's float32 dtype flow): u still holds the float32 image, so the neighbour
                       // sums of the launch's first iteration (stage 0) are float32 sums
    int np2_div3;      // NP2 pipelines: the three-operation float32 division rhs / diag is exact for this diag (k_div3_check)
    const float* of32; // NP2 pipelines: omega * img as a float32 plane (same pitch in elements as `of`, whose values it holds)
    int iso_lean, iso_div3;   // isotropic float pipelines, scalar threshold: relaxations verified for this threshold (iso_split_vec)
};

// ALIGN: the lanes that store cover whole 128-byte lines (strip stride rounded down to a multiple of
// 128 B; row pitch is a multiple of 128 B), so no line is ever written by two waves.  Measured on
// MI355X with the 4-read/3-write marching stream (tools/membench.hip): +11..14 % over the tight
// packing (61 of 64 lanes storing), although 5 more lanes per wave only load.
template <typename T, int C, bool ALIGN = true>
struct StripGeom {
    static constexpr int N = VecOf<T>::N;
    static constexpr int LINE = 128 / (int)sizeof(T);   // elements per 128-byte line
    static constexpr int LH = (C + N - 1) / N;          // left halo lanes
    static constexpr int RH = (2 * C + N - 1) / N;      // right halo lanes (minimum)
    static constexpr int TIGHT = (kWave - LH - RH) * N;
    static constexpr int STRIDE = ALIGN ? (TIGHT / LINE) * LINE : TIGHT;  // elements advanced per strip
    static constexpr int VALID = STRIDE / N;            // lanes that store: [LH, LH + VALID)
};

// blockIdx -> (column block bx, row chunk).  Blocks are dealt round-robin over the 8 XCDs
// (MI355X_MICROARCH.md, workgroup dispatch), so with nchunks a multiple of 8 the mapping
// chunk = id % nchunks puts a whole chunk row -- all column strips that share halo columns -- on one
// XCD's L2.  Placement only affects speed, never results.
__device__ __forceinline__ void block_to_tile(int bid, int nbx, int nchunks, int& bx, int& chunk) {
    chunk = bid % nchunks;
    bx = bid / nchunks;
}
// The same for any number of chunks: XCD x (= bid % 8) works on a contiguous run of tiles in (chunk, column block)
// order, so neighbouring strips -- and neighbouring chunk rows, which share their fill rows -- meet in one L2.  Class A
// chunks (FusedArgs::signal) are dealt the same way among themselves and come first on every XCD.  The grid has
// 8 * (ceil(tilesA / 8) + ceil(tilesB / 8)) blocks (tile_grid on the host); returns false for the surplus ones.
template <typename A>
__device__ __forceinline__ bool block_to_tile_sub(int bid, int nbx, const A& a, int& bx, int& chunk) {
    const int c0 = a.seg_chunk[1], c1 = a.seg_chunk[2] - a.seg_chunk[1], c2 = a.seg_chunk[3] - a.seg_chunk[2];
    const int tiles_a = nbx * (c0 + c2), tiles_b = nbx * c1;
    const int per_a = (tiles_a + 7) >> 3, per_b = (tiles_b + 7) >> 3;
    const int xcd = bid & 7;
    int idx = bid >> 3;
    if (idx < per_a) {
        const int t = xcd * per_a + idx;
        if (t >= tiles_a) return false;
        const int ca = t / nbx;
        bx = t - ca * nbx;
        chunk = ca < c0 ? ca : ca + c1;
    } else {
        idx -= per_a;
        const int t = xcd * per_b + idx;
        if (t >= tiles_b) return false;
        const int cb = t / nbx;
        bx = t - cb * nbx;
        chunk = c0 + cb;
    }
    return true;
}
// A class A wave has stored its rows: make them visible device-wide (the reader is a later kernel on another stream,
// started by a stream wait on the counter), then count the wave.  Recipe of MI355X_MICROARCH.md (inter-workgroup
// visibility): drain the stores, agent-scope release (writes back this XCD's L2), drain again -- as inline asm, because
// the compiler drops the wait after buffer_wbl2 when it believes the scoreboard empty --, agent-scope atomic add.
template <typename A>
__device__ __forceinline__ void signal_chunk_done(const A& a, int chunk, int lane) {
    if (a.signal == nullptr) return;                                        // uniform
    if (chunk >= a.seg_chunk[1] && chunk < a.seg_chunk[2]) return;          // class B (wave-uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename A>
__device__ __forceinline__ void chunk_rows(const A& a, int chunk, int64_t& R0, int64_t& R1) {
    const int s = (chunk >= a.seg_chunk[1] ? 1 : 0) + (chunk >= a.seg_chunk[2] ? 1 : 0);
    R0 = a.seg_row[s] + (int64_t)(chunk - a.seg_chunk[s]) * a.seg_rpc[s];
    R1 = (R0 + a.seg_rpc[s] < a.seg_row[s + 1]) ? R0 + a.seg_rpc[s] : a.seg_row[s + 1];
}

template <typename T, int C, bool COUPLED, bool ISO, bool HET, bool DPP, bool ALIGN = true, bool NTS = true>
__global__ void __launch_bounds__(256) k_fused(FusedArgs<T> a) {
    constexpr int N = VecOf<T>::N;
    using SG = StripGeom<T, C, ALIGN>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int WPB = 4;
    const int nbx = (a.nstrips + WPB - 1) / WPB;
    int bx, chunk;
    if (a.remap) {
        if (!block_to_tile_sub((int)blockIdx.x, nbx, a, bx, chunk)) return;  // block-uniform
    } else {
        bx = blockIdx.x % nbx;
        chunk = blockIdx.x / nbx;
    }
    const int strip = bx * WPB + wave;
    if (strip >= a.nstrips || chunk >= a.nchunks) return;  // wave-uniform

    int64_t R0, R1;
    chunk_rows(a, chunk, R0, R1);
    if (R0 >= a.n0) return;

    // element range of this lane: [x0, x0 + N)
    const int64_t x0 = (int64_t)strip * SG::STRIDE + (int64_t)(lane - SG::LH) * N;
    // lanes outside the padded row read a clamped address (their values are never used for stores)
    int64_t xl = x0;
    if (xl < 0) xl = 0;
    if (xl > a.pitch - N) xl = a.pitch - N;
    const bool store_lane = lane >= SG::LH && lane < SG::LH + SG::VALID && x0 < a.L;

    // loop-invariant per-element border predicates
    bool edgeL[N], edgeR[N], ch0[N], chLast[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int64_t x = x0 + j;
        edgeL[j] = x < C;
        edgeR[j] = x >= a.L - C;
        const int cc = (int)(((x % C) + C) % C);
        ch0[j] = cc == 0;
        chLast[j] = cc == C - 1;
    }

    const T* of = a.of + xl;
    const T* u_in = a.u_in + xl;
    const T* q0_in = a.q0_in + xl;
    const T* q1_in = a.q1_in + xl;
    T* u_out = a.u_out + xl;
    T* q0_out = a.q0_out + xl;
    T* q1_out = a.q1_out + xl;
    const T* p_ell = HET ? a.cp.ell + xl : nullptr;
    const T* p_diag = HET ? a.cp.diag + xl : nullptr;
    const T* p_dsc = HET ? a.cp.dsc + xl : nullptr;
    const T* p_kq = HET ? a.cp.kq + xl : nullptr;
    const bool derive = HET && a.derive != 0;  // uniform
    // heterogeneous coefficients of one row: the loads (four derived planes, or the given input planes) are issued
    // early like the other operands; `finish_coefs` turns them into ell / diag / dsc / kq when the row becomes current
    auto load_raw = [&](int64_t row, T (&c0)[N], T (&c1)[N], T (&c2)[N], T (&c3)[N]) {
        if (derive) {
#pragma unroll
            for (int j = 0; j < N; ++j) c0[j] = c1[j] = c2[j] = c3[j] = T(0);
            if (a.ci.mu) load_vec(a.ci.mu + xl + row * a.pitch, c0);
            if (a.ci.omega) load_vec(a.ci.omega + xl + row * a.pitch, c1);
            if (a.ci.ell) load_vec(a.ci.ell + xl + row * a.pitch, c2);
        } else {
            load_vec(p_ell + row * a.pitch, c0);
            load_vec(p_diag + row * a.pitch, c1);
            load_vec(p_dsc + row * a.pitch, c2);
            load_vec(p_kq + row * a.pitch, c3);
        }
    };
    auto finish_coefs = [&](const T (&c0)[N], const T (&c1)[N], const T (&c2)[N], const T (&c3)[N], T (&ell_o)[N],
                            T (&diag_o)[N], T (&dsc_o)[N], T (&kq_o)[N]) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (derive) {
                derive_coefs<T>(a.ci, c0[j], c1[j], c2[j], ell_o[j], diag_o[j], dsc_o[j], kq_o[j]);
            } else {
                ell_o[j] = c0[j];
                diag_o[j] = c1[j];
                dsc_o[j] = c2[j];
                kq_o[j] = c3[j];
            }
        }
    };
    const int64_t P = a.pitch;
    auto clampr = [&](int64_t r) -> int64_t { return r < 0 ? 0 : (r > a.n0 - 1 ? a.n0 - 1 : r); };

    T ell[N], diag[N], dsc[N], kq[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        ell[j] = a.c.ell;
        diag[j] = a.c.diag;
        dsc[j] = a.c.dsc;
        kq[j] = a.c.kq;
    }

    // ---- prologue: rows R0-1 and R0 ---------------------------------------------------------
    T um[N], uc[N], up[N], t0m[N], upm[N], b0m[N];
    load_vec(u_in + clampr(R0 - 1) * P, um);
    load_vec(u_in + R0 * P, uc);
    {
        const int64_t rm = clampr(R0 - 1);
        T q0v[N], q1v[N], ellm[N], kqm[N];
        load_vec(q0_in + rm * P, q0v);
        load_vec(q1_in + rm * P, q1v);
        if constexpr (HET) {
            T r0[N], r1[N], r2[N], r3[N], dg[N], ds[N];
            load_raw(rm, r0, r1, r2, r3);
            finish_coefs(r0, r1, r2, r3, ellm, dg, ds, kqm);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            T qq[2] = {q0v[j], q1v[j]}, d[2], b[2];
            split_vars<T, 2, ISO, !HET>(qq, HET ? kqm[j] : kq[j], d, b);
            t0m[j] = (HET ? ellm[j] : ell[j]) * (b[0] - d[0]);
            upm[j] = T(0);
            b0m[j] = T(0);
        }
    }

    // software prefetch of the first step's operands
    T fr[N], q0r[N], q1r[N];
    load_vec(u_in + clampr(R0 + 1) * P, up);
    load_vec(of + R0 * P, fr);
    load_vec(q0_in + R0 * P, q0r);
    load_vec(q1_in + R0 * P, q1r);
    if constexpr (HET) {
        T r0[N], r1[N], r2[N], r3[N];
        load_raw(R0, r0, r1, r2, r3);
        finish_coefs(r0, r1, r2, r3, ell, diag, dsc, kq);
    }

    for (int64_t r = R0; r <= R1; ++r) {
        // issue next step's loads early (clamped rows keep the addresses valid on the last steps)
        T upn[N], frn[N], q0n[N], q1n[N], elln[N], diagn[N], dscn[N], kqn[N];
        {
            const int64_t rn = clampr(r + 1);
            load_vec(u_in + clampr(r + 2) * P, upn);
            load_vec(of + rn * P, frn);
            load_vec(q0_in + rn * P, q0n);
            load_vec(q1_in + rn * P, q1n);
            if constexpr (HET) load_raw(rn, elln, diagn, dscn, kqn);  // raw planes; finished at the rotation
        }

        T un[N], b0c[N], t0c[N];
        if (r < a.n0) {
            // split variables and t = ell*(b - d) of row r
            T b1c[N], t1c[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T qq[2] = {q0r[j], q1r[j]}, d[2], b[2];
                split_vars<T, 2, ISO, !HET>(qq, kq[j], d, b);
                b0c[j] = b[0];
                b1c[j] = b[1];
                t0c[j] = ell[j] * (b[0] - d[0]);
                t1c[j] = ell[j] * (b[1] - d[1]);
            }
            T t1l[N], ul[N], ur[N];
            shifted<-C, DPP>(t1c, t1l);
            shifted<-C, DPP>(uc, ul);
            shifted<+C, DPP>(uc, ur);
            T ucl[N], ucr[N];
            if constexpr (COUPLED) {
                shifted<-1, DPP>(uc, ucl);
                shifted<+1, DPP>(uc, ucr);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                // rhs = omega*img + fd_0 + fd_1 ; rs = rhs / diag
                T rhs = fr[j] + (t0c[j] - t0m[j]);
                rhs = rhs + (t1c[j] - (edgeL[j] ? t1c[j] : t1l[j]));
                const T rs = rhs / diag[j];
                // neighbour sum in the reference's order: axis 0 (rows), axis 1 (columns), axis 2 (channels)
                T acc = um[j] + up[j];
                acc = acc + (edgeL[j] ? uc[j] : ul[j]);
                acc = acc + (edgeR[j] ? uc[j] : ur[j]);
                if constexpr (COUPLED) {
                    acc = acc + (ch0[j] ? uc[j] : ucl[j]);
                    acc = acc + (chLast[j] ? uc[j] : ucr[j]);
                }
                un[j] = rs + acc / dsc[j];
            }
            // q1 of row r:  (u_new[x+C] - u_new[x]) + b1
            T unr[N], q1o[N];
            shifted<+C, DPP>(un, unr);
#pragma unroll
            for (int j = 0; j < N; ++j) q1o[j] = ((edgeR[j] ? un[j] : unr[j]) - un[j]) + b1c[j];
            if (store_lane && r < R1) {
                if constexpr (NTS) {
                    store_vec_nt(q1_out + r * P, q1o);
                    store_vec_nt(u_out + r * P, un);
                } else {
                    store_vec(q1_out + r * P, q1o);
                    store_vec(u_out + r * P, un);
                }
            }
        } else {
            // r == n0: below the last row, replicate u_new so that the row difference is 0
#pragma unroll
            for (int j = 0; j < N; ++j) {
                un[j] = upm[j];
                b0c[j] = T(0);
                t0c[j] = T(0);
            }
        }
        // q0 of row r-1:  (u_new[r] - u_new[r-1]) + b0[r-1]
        if (r > R0) {
            T q0o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) q0o[j] = (un[j] - upm[j]) + b0m[j];
            if (store_lane) {
                if constexpr (NTS)
                    store_vec_nt(q0_out + (r - 1) * P, q0o);
                else
                    store_vec(q0_out + (r - 1) * P, q0o);
            }
        }
        // rotate the window
#pragma unroll
        for (int j = 0; j < N; ++j) {
            um[j] = uc[j];
            uc[j] = up[j];
            up[j] = upn[j];
            t0m[j] = t0c[j];
            upm[j] = un[j];
            b0m[j] = b0c[j];
            fr[j] = frn[j];
            q0r[j] = q0n[j];
            q1r[j] = q1n[j];
        }
        if constexpr (HET) finish_coefs(elln, diagn, dscn, kqn, ell, diag, dsc, kq);
    }
    signal_chunk_done(a, chunk, lane);
}


// ============================================================================================
// The reference's OWN dtype flow for a float32 image under NumPy >= 2 ("numpy2" promotion), one outer iteration per
// pass.  With scalar omega / ell the reference's cached const_diag_scaled is an np.float64 (jacobi.py:101-106), so
//   rhs        = omega*img + sum_j fd_j(ell (b_j - d_j))       float32   (split_bregman_tvd.py:84-89, in-place +=)
//   rhs_scaled = rhs / diag                                    float32   (jacobi.py:107)
//   u_new      = rhs_scaled + N(u) / diag_scaled               float64   (jacobi.py:110-114; N(u) is summed in float32
//                                                                         while u is still the float32 image, i.e. in
//                                                                         the first iteration of a call, in float64 after)
//   dub_j      = D_j u_new + b_j                               float64   (:143)
//   d_j        = shrink(dub_j, mu / ell)  -> stored float32              (:144; mu / ell: Python float, or a float64
//                                                                         array when mu is one -- the workflow's weights)
//   b_j        = dub_j - d_j              -> stored float32              (:145)
// The context is a DOUBLE context (planes, pitch, halo plumbing of the float64 path): u and the compressed split
// state q_j = dub_j are float64 planes, omega*img is a float64 plane that holds float32 values.  d, b, t = ell (b - d)
// and the right-hand side are recomputed / evaluated in float32 exactly as above, so the result equals the
// reference's bit for bit (tests/golden/tvd_mixed_golden.npz and the float32 cases of tvd_golden.npz).
// Strip / chunk decomposition, border rules and lane shifts are those of k_fused<double>.
// ============================================================================================
struct Np2Coef {
    double dsc;      // const_diag_scaled (np.float64)
    double kq;       // mu / ell as a Python float (scalar mu)
    float ell, diag; // weak Python floats meeting float32 arrays
    const double* kq_plane;  // mu / ell per element (float64 array-valued mu), or nullptr
    int first;       // u still holds the float32 image: neighbour sums in float32 (first iteration of a call)
    const double* ell_plane = nullptr;  // adaptive schedule (explicit d, b): ell per element, float32 values (k_adapt_ell_np2)
    int q32 = 0;     // the stored q was formed in float32 (np2_dub with u32): a scalar threshold then met a float32 array in the
                     // reference, i.e. its shrinkage ran in float32 arithmetic
    double rdsc = 0.0;   // k_shrink_rhs_np2: RN(1 / dsc) and RN(1 / diag) for the reciprocal divisions (0: IEEE divisions)
    float rdiag = 0.0f;
};

// FAST (callers guarantee k >= 0: the scalar threshold of a fused launch): q - clamp(q, -k, k), which is
// max(|q| - k, 0) sign(q) bit for bit up to the sign of a zero that no later operation can tell apart (split_vars).
template <bool FAST = false>
__device__ __forceinline__ void np2_split(double q, double kq, float& d, float& b) {
    double d64;
    if constexpr (FAST) d64 = q - t_clamp_sym(q, kq);
    else {
        // np.maximum(np.abs(x) - k, 0) * np.sign(x), float64 -- NaN included (np.maximum and np.sign hand a NaN on; a
        // diverged adaptive run reaches inf - inf)
        const double m = fabs(q) - kq;
        d64 = (m != m || q != q) ? (m + q) : t_max0(m) * sgn(q);
    }
    d = (float)d64;                                       // d[..., j] = ...   (float32 array)
    b = (float)(q - (double)d);                           // b[..., j] = dub - d[..., j]
}

// d, b from a stored q in the generic kernels: as np2_split, or -- q32 with a scalar threshold -- in float32 arithmetic
template <bool KQARR>
__device__ __forceinline__ void np2_split_q(double q, double kq, int q32, float& d, float& b) {
    if (!KQARR && q32) {
        const float x = (float)q, kf = (float)kq;
        const float m = fabsf(x) - kf;
        d = (m != m || x != x) ? (m + x) : t_max0(m) * sgn(x);
        b = x - d;
    } else {
        np2_split(q, kq, d, b);
    }
}

// ---- the same flow as three kernels (any dim, any shape, several Jacobi sweeps): the generic path of a numpy2 context ----
// One thread per element like k_rhs / k_jacobi / k_shrink; the expressions and their order are those of k_fused_np2.
//   k_rhs_np2:          rs = float32( float32(omega img + sum_j fd_j(ell (b_j - d_j))) / diag ), stored in the double plane
//   k_jacobi_np2_first: the FIRST sweep of a call, whose iterate still is the float32 image: N(u) is summed in float32
//                       (jacobi.py:25-55 on a float32 array), then / np.float64(diag_scaled) -> float64.  Every later sweep
//                       is the double context's own sweep (float64 N(u); rs holds float32 values, which float64 + float32
//                       -> float64 adds exactly as stored).
//   k_shrink_np2:       q_j <- D_j u_new + float64(b_j)  with b_j the float32 value re-derived from the stored q_j
// FROM_DB: the first iteration of a warm start -- d and b are the caller's float32 arrays, held as explicit planes
template <int DIM, bool KQARR, bool FROM_DB = false>
__global__ void __launch_bounds__(256) k_rhs_np2(Geom g, const double* __restrict__ of, QPlanes<double, DIM> q, DBPlanes<double> db,
                                                  Np2Coef k, double* __restrict__ rs) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const double kq_c = KQARR ? k.kq_plane[e.off] : k.kq;
    float r = (float)of[e.off];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        float d, b;
        const int64_t o = nb_off(g, e, j, -1);
        if constexpr (FROM_DB) { d = (float)db.d[j][e.off]; b = (float)db.b[j][e.off]; }
        else np2_split_q<KQARR>(q.in[j][e.off], kq_c, k.q32, d, b);
        const float tc = (FROM_DB && k.ell_plane ? (float)k.ell_plane[e.off] : k.ell) * (b - d);
        if constexpr (FROM_DB) { d = (float)db.d[j][o]; b = (float)db.b[j][o]; }
        else np2_split_q<KQARR>(q.in[j][o], KQARR ? k.kq_plane[o] : k.kq, k.q32, d, b);
        const float tn = (FROM_DB && k.ell_plane ? (float)k.ell_plane[o] : k.ell) * (b - d);
        r = r + (tc - tn);
    }
    rs[e.off] = (double)(r / k.diag);
}

static __global__ void __launch_bounds__(256) k_jacobi_np2_first(Geom g, const double* __restrict__ rs, const double* __restrict__ xin,
                                                                 double dsc, const double* __restrict__ dsc_p,
                                                                 double* __restrict__ xout) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    float acc = 0.f;
    for (int ax = 0; ax < g.nsum_axes; ++ax) {
        acc = acc + (float)xin[nb_off(g, e, ax, -1)];
        acc = acc + (float)xin[nb_off(g, e, ax, +1)];
    }
    xout[e.off] = rs[e.off] + (double)acc / (dsc_p ? dsc_p[e.off] : dsc);
}

// Array-valued omega and / or ell (float64 arrays, or ell = 2 mu of a float64 mu) next to the float32 image: what NumPy's
// promotion makes of split_bregman_tvd.py:84-89 and jacobi.py:107 then --
//   ell an array:   t = ell (b - d) is float64 (b - d is formed in float32 first); rhs += fd_j(t) adds in float64 and
//                   stores float32 (in-place ufunc, same-kind cast) unless the right-hand side itself is float64
//   omega an array: rhs = omega img is float64 and stays so
//   diag an array (either of the two): rhs / diag is float64 whatever rhs is
struct Np2Het {
    const double *ell_p, *diag_p, *kq_p;  // planes or nullptr (ell_p only when ell is an array)
    double kq, diag_d;
    float ell_f, diag_f;
    int rhs64;
};
// FROM_DB: the first iteration of a warm start -- d, b are the caller's float32 arrays, held as explicit planes
template <int DIM, bool FROM_DB = false>
__global__ void __launch_bounds__(256) k_rhs_np2_het(Geom g, const double* __restrict__ of, QPlanes<double, DIM> q, DBPlanes<double> db,
                                                      Np2Het k, double* __restrict__ rs) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const double kq_c = k.kq_p ? k.kq_p[e.off] : k.kq;
    double r = of[e.off];   // a float32 value unless rhs64
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        float d, b;
        const int64_t o = nb_off(g, e, j, -1);
        float bd_c, bd_n;
        if constexpr (FROM_DB) {
            bd_c = (float)db.b[j][e.off] - (float)db.d[j][e.off];
            bd_n = (float)db.b[j][o] - (float)db.d[j][o];
        } else {
            np2_split(q.in[j][e.off], kq_c, d, b);
            bd_c = b - d;
            np2_split(q.in[j][o], k.kq_p ? k.kq_p[o] : k.kq, d, b);
            bd_n = b - d;
        }
        if (k.ell_p) {
            const double fd = k.ell_p[e.off] * (double)bd_c - k.ell_p[o] * (double)bd_n;
            r = k.rhs64 ? r + fd : (double)(float)(r + fd);
        } else {
            const float fd = k.ell_f * bd_c - k.ell_f * bd_n;
            r = k.rhs64 ? r + (double)fd : (double)((float)r + fd);
        }
    }
    rs[e.off] = k.diag_p ? r / k.diag_p[e.off] : (k.rhs64 ? r / k.diag_d : (double)((float)r / k.diag_f));
}

// u32: the solver handed the float32 iterate itself back (a tolerance met by the very first sweep of a call, jacobi.py:
// 123-126): the reference's D u + b is a float32 sum then
__device__ __forceinline__ double np2_dub(double un, double uc, float b, int u32) {
    return u32 ? (double)(((float)un - (float)uc) + b) : (un - uc) + (double)b;
}

template <int DIM, bool KQARR, bool FROM_DB = false>
__global__ void __launch_bounds__(256) k_shrink_np2(Geom g, const double* __restrict__ unew, QPlanes<double, DIM> q, DBPlanes<double> db,
                                                     Np2Coef k, int u32) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const double kq = KQARR ? k.kq_plane[e.off] : k.kq;
    const double uc = unew[e.off];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        float d, b;
        if constexpr (FROM_DB) b = (float)db.b[j][e.off];
        else np2_split_q<KQARR>(q.in[j][e.off], kq, k.q32, d, b);
        q.out[j][e.off] = np2_dub(unew[nb_off(g, e, j, +1)], uc, b, u32);
    }
}

// ---- shrinkage of iteration n, right-hand side of iteration n + 1 and its first Jacobi sweep in ONE pass ---------------------
// The three-launch flow streams ~19 values of 8 B per element and iteration; most of them are q and u read twice: by the
// shrinkage (q_j <- D_j u_new + b_j) and again by the next right-hand side (t_j = ell (b_j - d_j) of that q_j at the element
// and at its lower neighbour along j), whose first sweep x_1 = rs + N(u_new) / diag_scaled then reads u_new a third time.
// Here a thread forms the new q_j at its element AND at the lower neighbour along each axis (the same expression, so the
// value the neighbour's thread stores is the value used here), the float32 right-hand side, rs = rhs / diag, and x_1 --
// k_shrink_np2, k_rhs_np2 and the first sweep with the operations and their order unchanged.  q is read from the `in`
// buffers and written to the `out` buffers (other threads still read the old values of this thread's element).
// Scalar omega and ell (mu: scalar or the threshold plane), float64 iterate (not the first iteration of a call), no warm
// start planes: what every iteration of a numpy2 context after the first looks like.  rs is written only when more sweeps
// follow (WRITE_RS); with Jacobi(maxiter=1) x_1 already is the next iterate.
//   bytes per element, dim = 2: u 8 + q 16 + omega*img 4 read, q 16 + x_1 8 (+ rs 8) written = 52 (60); the three launches
//   it replaces: 32 (rhs) + 24 (sweep) + 40 (shrinkage) = 96
// FASTK: a scalar threshold k >= 0 (checked on the host): the clamp form of the shrinkage (np2_split<true>)
template <int DIM, bool KQARR, bool WRITE_RS, bool FASTK>
__global__ void __launch_bounds__(256) k_shrink_rhs_np2(Geom g, const double* __restrict__ unew, QPlanes<double, DIM> q,
                                                         const float* __restrict__ of32, Np2Coef k, double* __restrict__ rs,
                                                         double* __restrict__ x1) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const double kq_c = KQARR ? k.kq_plane[e.off] : k.kq;
    const double uc = unew[e.off];
    float r = of32[e.off];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        float d, b;
        np2_split<FASTK>(q.in[j][e.off], kq_c, d, b);                                // the old b_j here (k_shrink_np2)
        const double qe = np2_dub(unew[nb_off(g, e, j, +1)], uc, b, 0);
        q.out[j][e.off] = qe;
        np2_split<FASTK>(qe, kq_c, d, b);                                            // k_rhs_np2 on the new state
        const float tc = k.ell * (b - d);
        const int64_t o = nb_off(g, e, j, -1);
        float tn = tc;
        if (o != e.off) {   // the lower neighbour's new q_j: its upper neighbour along j is this element
            const double kq_o = KQARR ? k.kq_plane[o] : k.kq;
            np2_split<FASTK>(q.in[j][o], kq_o, d, b);
            const double qo = np2_dub(uc, unew[o], b, 0);
            np2_split<FASTK>(qo, kq_o, d, b);
            tn = k.ell * (b - d);
        }
        r = r + (tc - tn);
    }
    double acc = 0.0;   // k_jacobi<double> on the float64 iterate
    for (int ax = 0; ax < g.nsum_axes; ++ax) {
        acc = acc + unew[nb_off(g, e, ax, -1)];
        acc = acc + unew[nb_off(g, e, ax, +1)];
    }
    // both quotients are correctly rounded either way: through the host-rounded reciprocals with two residual corrections
    // (dxh_common.h, div_by_const) unless a numerator is non-zero and tiny, or the divisor was outside the comfortable range
    float rsv;
    double qa;
    if (k.rdsc != 0.0 && !div_unsafe(r) && !div_unsafe(acc)) {
        rsv = div_by_const<float>(r, k.diag, k.rdiag);
        qa = div_by_const<double>(acc, k.dsc, k.rdsc);
    } else {
        rsv = r / k.diag;
        qa = acc / k.dsc;
    }
    if constexpr (WRITE_RS) rs[e.off] = (double)rsv;
    x1[e.off] = (double)rsv + qa;
}

// ---- explicit split variables in the numpy2 flow (the adaptive schedule, split_bregman_tvd.py:149-158): d and b are planes
// of float32 values, the shrinkage writes them, the right-hand side reads them (k_rhs_np2<.., FROM_DB>) ----
template <int DIM, bool KQARR>
__global__ void __launch_bounds__(256) k_expand_planes_np2(Geom g, QPlanes<double, DIM> q, Np2Coef k, double* d0, double* d1, double* d2,
                                                            double* b0, double* b1, double* b2) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    double* dp[3] = {d0, d1, d2};
    double* bp[3] = {b0, b1, b2};
    const double kq = KQARR ? k.kq_plane[e.off] : k.kq;
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        float d, b;
        np2_split_q<KQARR>(q.in[j][e.off], kq, k.q32, d, b);
        dp[j][e.off] = (double)d;
        bp[j][e.off] = (double)b;
    }
}

template <int DIM, bool KQARR>
__global__ void __launch_bounds__(256) k_shrink_db_np2(Geom g, const double* __restrict__ unew, double* d0, double* d1, double* d2,
                                                        double* b0, double* b1, double* b2, Np2Coef k, int u32) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    double* dp[3] = {d0, d1, d2};
    double* bp[3] = {b0, b1, b2};
    const double kq = KQARR ? k.kq_plane[e.off] : k.kq;
    const double uc = unew[e.off];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        const double dub = np2_dub(unew[nb_off(g, e, j, +1)], uc, (float)bp[j][e.off], u32);   // float64 + float32
        float d, b;
        if (u32 && !KQARR) {   // a float32 dub meets the Python-float threshold: the shrinkage itself is float32 arithmetic
            const float x = (float)dub, kf = (float)kq;
            const float m = fabsf(x) - kf;
            d = (m != m || x != x) ? (m + x) : t_max0(m) * sgn(x);
            b = x - d;
        } else {
            np2_split(dub, kq, d, b);
        }
        dp[j][e.off] = (double)d;
        bp[j][e.off] = (double)b;
    }
}

// ell <- 1 / max(|grad u|_1, 1e-12) with grad a float32 array (the float64 differences rounded on assignment), hence ell
// float32; the threshold mu / ell: Python float / float32 array -> float32, float64 array / float32 array -> float64
template <int DIM>
__global__ void __launch_bounds__(256) k_adapt_ell_np2(Geom g, const double* __restrict__ u, double mu_s, const double* __restrict__ mu_p,
                                                        double* __restrict__ ell_o, double* __restrict__ kq_o) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const double uc = u[e.off];
    float l1 = 0.f;
#pragma unroll
    for (int j = 0; j < DIM; ++j) l1 = l1 + fabsf((float)(u[nb_off(g, e, j, +1)] - uc));
    const float lo = (float)1e-12;
    const float ell = 1.0f / ((l1 > lo || l1 != l1) ? l1 : lo);   // np.maximum hands a NaN on
    ell_o[e.off] = (double)ell;
    kq_o[e.off] = mu_p ? mu_p[e.off] / (double)ell : (double)((float)mu_s / ell);
}

// d, b of a numpy2 context as the reference holds them (float32, (..., dim) interleaved): re-derived from the stored q
template <int DIM, bool KQARR>
__global__ void __launch_bounds__(256) k_expand_np2(Geom g, QPlanes<double, DIM> q, Np2Coef k, float* __restrict__ dout,
                                                     float* __restrict__ bout) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const double kq = KQARR ? k.kq_plane[e.off] : k.kq;
    const int64_t o = (e.row * g.L + e.x) * DIM;  // dense (no pitch)
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
        float d, b;
        np2_split_q<KQARR>(q.in[j][e.off], kq, k.q32, d, b);
        if (dout) dout[o + j] = d;
        if (bout) bout[o + j] = b;
    }
}

// (..., dim) interleaved arrays of one float type <-> explicit planes of the other
template <typename TS, typename TD, int DIM>
__global__ void __launch_bounds__(256) k_deinterleave_cvt(Geom g, const TS* __restrict__ src, TD* p0, TD* p1, TD* p2) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const int64_t o = (e.row * g.L + e.x) * DIM;
    TD* pl[3] = {p0, p1, p2};
#pragma unroll
    for (int j = 0; j < DIM; ++j) pl[j][e.off] = (TD)src[o + j];
}
template <typename TS, typename TD, int DIM>
__global__ void __launch_bounds__(256) k_interleave_cvt(Geom g, const TS* p0, const TS* p1, const TS* p2, TD* __restrict__ dst) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const int64_t o = (e.row * g.L + e.x) * DIM;
    const TS* pl[3] = {p0, p1, p2};
#pragma unroll
    for (int j = 0; j < DIM; ++j) dst[o + j] = (TD)pl[j][e.off];
}

template <int C, bool COUPLED, bool KQARR>
__global__ void __launch_bounds__(256) k_fused_np2(FusedArgs<double> a, Np2Coef k) {
    using T = double;
    constexpr int N = VecOf<T>::N;
    constexpr bool DPP = true;
    using SG = StripGeom<T, C, true>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int WPB = 4;
    const int nbx = (a.nstrips + WPB - 1) / WPB;
    int bx, chunk;
    if (!block_to_tile_sub((int)blockIdx.x, nbx, a, bx, chunk)) return;  // block-uniform
    const int strip = bx * WPB + wave;
    if (strip >= a.nstrips || chunk >= a.nchunks) return;  // wave-uniform
    int64_t R0, R1;
    chunk_rows(a, chunk, R0, R1);
    if (R0 >= a.n0) return;

    const int64_t x0 = (int64_t)strip * SG::STRIDE + (int64_t)(lane - SG::LH) * N;
    int64_t xl = x0;
    if (xl < 0) xl = 0;
    if (xl > a.pitch - N) xl = a.pitch - N;
    const bool store_lane = lane >= SG::LH && lane < SG::LH + SG::VALID && x0 < a.L;
    bool edgeL[N], edgeR[N], ch0[N], chLast[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int64_t x = x0 + j;
        edgeL[j] = x < C;
        edgeR[j] = x >= a.L - C;
        const int cc = (int)(((x % C) + C) % C);
        ch0[j] = cc == 0;
        chLast[j] = cc == C - 1;
    }
    const T* of = a.of + xl;
    const T* u_in = a.u_in + xl;
    const T* q0_in = a.q0_in + xl;
    const T* q1_in = a.q1_in + xl;
    T* u_out = a.u_out + xl;
    T* q0_out = a.q0_out + xl;
    T* q1_out = a.q1_out + xl;
    const T* p_kq = KQARR ? k.kq_plane + xl : nullptr;
    const int64_t P = a.pitch;
    auto clampr = [&](int64_t r) -> int64_t { return r < 0 ? 0 : (r > a.n0 - 1 ? a.n0 - 1 : r); };
    const bool first = k.first != 0;  // uniform

    T kq[N];
#pragma unroll
    for (int j = 0; j < N; ++j) kq[j] = k.kq;

    // ---- prologue: rows R0-1 and R0
    T um[N], uc[N], up[N], upm[N];
    float t0m[N], b0m[N];
    load_vec(u_in + clampr(R0 - 1) * P, um);
    load_vec(u_in + R0 * P, uc);
    {
        const int64_t rm = clampr(R0 - 1);
        T q0v[N], kqm[N];
        load_vec(q0_in + rm * P, q0v);
        if constexpr (KQARR) load_vec(p_kq + rm * P, kqm);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float d, b;
            np2_split<!KQARR>(q0v[j], KQARR ? kqm[j] : kq[j], d, b);
            t0m[j] = k.ell * (b - d);
            upm[j] = T(0);
            b0m[j] = 0.f;
        }
    }
    T fr[N], q0r[N], q1r[N];
    load_vec(u_in + clampr(R0 + 1) * P, up);
    load_vec(of + R0 * P, fr);
    load_vec(q0_in + R0 * P, q0r);
    load_vec(q1_in + R0 * P, q1r);
    if constexpr (KQARR) load_vec(p_kq + R0 * P, kq);

    for (int64_t r = R0; r <= R1; ++r) {
        T upn[N], frn[N], q0n[N], q1n[N], kqn[N];
        {
            const int64_t rn = clampr(r + 1);
            load_vec(u_in + clampr(r + 2) * P, upn);
            load_vec(of + rn * P, frn);
            load_vec(q0_in + rn * P, q0n);
            load_vec(q1_in + rn * P, q1n);
            if constexpr (KQARR) load_vec(p_kq + rn * P, kqn);
        }
        T un[N];
        float b0c[N], t0c[N];
        if (r < a.n0) {
            float b1c[N];
            T t1c[N];  // float32 values carried in the lane-shift type
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float d0, b0, d1, b1;
                np2_split<!KQARR>(q0r[j], kq[j], d0, b0);
                np2_split<!KQARR>(q1r[j], kq[j], d1, b1);
                b0c[j] = b0;
                b1c[j] = b1;
                t0c[j] = k.ell * (b0 - d0);
                t1c[j] = (T)(k.ell * (b1 - d1));
            }
            T t1l[N], ul[N], ur[N], ucl[N], ucr[N];
            shifted<-C, DPP>(t1c, t1l);
            shifted<-C, DPP>(uc, ul);
            shifted<+C, DPP>(uc, ur);
            if constexpr (COUPLED) {
                shifted<-1, DPP>(uc, ucl);
                shifted<+1, DPP>(uc, ucr);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                // float32: rhs = omega*img; rhs += fd_0; rhs += fd_1; rhs_scaled = rhs / diag
                float rhs = (float)fr[j] + (t0c[j] - t0m[j]);
                rhs = rhs + ((float)t1c[j] - (edgeL[j] ? (float)t1c[j] : (float)t1l[j]));
                const float rs = rhs / k.diag;
                const T nl = edgeL[j] ? uc[j] : ul[j], nr = edgeR[j] ? uc[j] : ur[j];
                T acc;
                if (first) {  // u is the float32 image: the reference's accumulator is float32 too
                    float af = (float)um[j] + (float)up[j];
                    af = af + (float)nl;
                    af = af + (float)nr;
                    if constexpr (COUPLED) {
                        af = af + (float)(ch0[j] ? uc[j] : ucl[j]);
                        af = af + (float)(chLast[j] ? uc[j] : ucr[j]);
                    }
                    acc = (T)af;
                } else {
                    acc = um[j] + up[j];
                    acc = acc + nl;
                    acc = acc + nr;
                    if constexpr (COUPLED) {
                        acc = acc + (ch0[j] ? uc[j] : ucl[j]);
                        acc = acc + (chLast[j] ? uc[j] : ucr[j]);
                    }
                }
                un[j] = (T)rs + acc / k.dsc;
            }
            T unr[N], q1o[N];
            shifted<+C, DPP>(un, unr);
#pragma unroll
            for (int j = 0; j < N; ++j) q1o[j] = ((edgeR[j] ? un[j] : unr[j]) - un[j]) + (T)b1c[j];
            if (store_lane && r < R1) {
                store_vec_nt(q1_out + r * P, q1o);
                store_vec_nt(u_out + r * P, un);
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                un[j] = upm[j];
                b0c[j] = 0.f;
                t0c[j] = 0.f;
            }
        }
        if (r > R0) {
            T q0o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) q0o[j] = (un[j] - upm[j]) + (T)b0m[j];
            if (store_lane) store_vec_nt(q0_out + (r - 1) * P, q0o);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            um[j] = uc[j];
            uc[j] = up[j];
            up[j] = upn[j];
            t0m[j] = t0c[j];
            upm[j] = un[j];
            b0m[j] = b0c[j];
            fr[j] = frn[j];
            q0r[j] = q0n[j];
            q1r[j] = q1n[j];
            if constexpr (KQARR) kq[j] = kqn[j];
        }
    }
    signal_chunk_done(a, chunk, lane);
}


// ============================================================================================
// K outer iterations in ONE pass (temporal blocking of the fused sweep; scalar coefficients).
//
// The wave still marches down its column strip, but now runs a K-stage pipeline per step: stage s
// applies iteration n+s+1 to row r-s, taking its operands (level-s rows of u, q0, q1) from stage
// s-1's outputs of this and the previous step, which live in registers.  Only level 0 is read from
// memory and only level K is written: 7 values per element per K iterations.
//
//   stage s at row p = r - s needs          comes from
//     U_s[p-1], U_s[p]                      rolling registers um[s], uc[s]
//     U_s[p+1]                              stage s-1, this step (s = 0: loaded)
//     Q0_s[p]                               stage s-1, this step (it emits q0 one row behind)
//     Q1_s[p]                               stage s-1, previous step (held in q1h[s])
//     t0_s[p-1], b0_s[p-1]                  rolling registers
//     of[p]                                 s-deep delay line of the loaded rows
//
// Each stage loses C elements of validity on the left and 2C on the right, so a wave stores lanes
// [LH, LH + VALID) with LH = ceil(K C / N), RH = ceil(2 K C / N); for float32 RGB the 128-byte aligned
// stride stays 224 elements up to K = 3.  A chunk starts K-1 rows early and ends K-1 rows late so that
// every stage reaches its first stored row with valid carried state (the leading rows are computed
// and discarded).  Image borders: rows above 0 / below n0-1 are replicated per stage (p == 0: the
// carried row-above state is replaced by the current row; p == n0: the new iterate is replicated),
// which is the same arithmetic as clamped addressing in the single-iteration kernel.
// ============================================================================================

template <typename T, int C, int K, bool ALIGN = true>
struct StripGeomK {
    static constexpr int N = VecOf<T>::N;
    static constexpr int LINE = DXH_STRIP_ALIGN_BYTES / (int)sizeof(T);  // store-range granule in elements
    static constexpr int LH = (K * C + N - 1) / N;
    static constexpr int RH = (2 * K * C + N - 1) / N;
    static constexpr int TIGHT = (kWave - LH - RH) * N;
    static constexpr int STRIDE = ALIGN ? (TIGHT / LINE) * LINE : TIGHT;
    static constexpr int VALID = STRIDE / N;
    static_assert(STRIDE > 0, "too many fused iterations for this channel count");
};

// --------------------------------------------------------------------------------------------------
// Implementation: every carried value lives in a small ring whose slot index is a compile-time
// function of the step's phase (step number mod 4), and the row loop is unrolled by 4 (one
// instantiation of Pipe::step per phase), so the pipeline advances by register renaming instead of
// register-to-register copies.  EDGE: the strip contains elements within C of the left or right image
// border (column replicate selects needed); interior strips skip them.  Rings:
//   U[l][4]   rows of level l (l = 0: loaded, l = s+1: written by stage s), slot = (row - rs) & 3;
//             a stage reads slots c-1, c, c+1 and the producer overwrites the dead slot c-3 = c+1
//   U[K][2]   the last level only needs its previous row
//   t0, b0, q1, q0l [..][2]   one-step carries, alternate by phase parity
//   of[4]     omega*f rows r .. r-3 (the prefetched row r+1 is moved in at the end of the step)
// (A first version with rolling variables spent ~20 % of its VALU instructions on v_mov rotation.)
// --------------------------------------------------------------------------------------------------
template <int V>
using IC = std::integral_constant<int, V>;

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// accumulator type of the per-iteration squared increments: the element type (NumPy squares in the array dtype too;
// a float accumulator for double images would lose small magnitudes entirely)
template <typename T>
using IncT = T;

// Occupancy the pipelines are compiled for (waves per SIMD) and their prefetch distance.  float, scalar
// coefficients: 3 waves (<= 168 VGPRs; the kernel is VALU-issue bound and a third wave fills issue slots);
// that budget has no room for the staging registers of a two-step prefetch, and with three waves one step
// of distance is enough.  double and the array-coefficient pipelines keep 2 waves (they would spill).
#ifndef DXH_ISO_WAVES
#define DXH_ISO_WAVES DXH_LB4   // developer A/B: waves per SIMD of the isotropic float pipelines (K >= 3)
#endif
template <typename T, int K, bool HET, bool ISO = false>
constexpr int pipe_waves() { return HET ? 2 : (K >= 3 ? (sizeof(T) == 4 ? (ISO ? DXH_ISO_WAVES : DXH_LB4) : 2) : 3); }
template <typename T, int K, bool HET, bool ISO = false>
constexpr int pipe_pfd() { return (K >= 4 && pipe_waves<T, K, HET, ISO>() <= 2) ? 2 : 1; }
template <typename T, int K>
struct PipeState {
    static constexpr int N = VecOf<T>::N;
    T U[K + 1][4][N];
    T t0[K][2][N];
    T b0[K][2][N];
    T q1l[2][N];   // loaded level-0 q1 rows (prefetch target / current)
    T q0l[2][N];   // loaded level-0 q0 rows
    T of0[N];      // omega*f of stage 0's row
    T pf[2][4][N]; // prefetch distance 2: rows in flight two steps ahead (u, omega*f, q0, q1), slot = phase parity
    T coef[K][4][N];  // heterogeneous coefficients (ell, diag, dsc, kq) of each stage's current row
    T kqv[K][N];      // KQV: the shrink threshold mu / ell of each stage's current row
    T kqn[N];         // KQV: prefetched threshold row (stage 0's next row)
    T craw[2][3][N];  // prefetched raw coefficient rows (mu, omega, ell as given), slot = phase parity
    T ofr[K][N];   // pair roles: omega*f delay line in registers (ofr[s] = row of local stage s)
    T q1r[K][2][N];// pair roles: inter-stage Q1 hand-off in registers
};

// LDS rings of one wave (register relief, not communication: every lane reads back only what it wrote
// itself, so no barrier is involved).  Layout per wave: [slot][lane] vectors of 16 bytes.
//   slots 0..3            omega*f rows r .. r-3 (slot = (row - rs) & 3), written by stage 0, read by stages 1..K-1
//   slots 4 + 2(s-1) + p  Q1 of level s handed from stage s-1 (written with parity p = next step's) to stage s
template <int K>
struct PipeLds {
    static constexpr int SLOTS = 4 + 2 * (K - 1);
};

// step<PH, ROLE>: ROLE 0 = SOLO (one wave runs all K stages: HBM in, HBM out), 1 = HEAD (first K stages of a
// wave pair: HBM in, the pair's shared LDS rings out), 2 = TAIL (last K stages of a pair: shared LDS rings in,
// HBM out).
// HET: array-valued coefficients.  Each stage carries the derived coefficients (ell, diag, dsc, kq) of its current
// row in registers; stage 0 derives them from the prefetched rows of the given planes (derive_coefs) and they move
// down one stage per step.  Divisions are IEEE divisions (the divisors differ per element).
// KQV (scalar pipelines only): an array-valued mu with scalar omega and ell -- what the TVD operator and the workflow's
// porosity weights produce.  ell, diag, dsc stay the wave-invariant scalars of the scalar pipeline (verified-reciprocal
// division), only the shrink threshold kq = mu / ell is per element: one more plane read per launch (cp.kq, filled by
// k_coef_planes with derive_coefs' expression) and K x N registers, instead of the 4 x K x N coefficient registers and
// the per-row derivation of the general array pipeline -- three waves per SIMD instead of two.
// (The reference's mixed float32 / float64 flow for float32 images -- "numpy2" -- has its own pipeline: dxh_np2_pipe.h.)
template <typename T, int C, bool COUPLED, bool ISO, int K, bool NTS, bool EDGE, bool NORM, bool HET = false, bool KQV = false>
struct Pipe {
    static_assert(!(HET && KQV), "KQV is a variant of the scalar pipeline");
    const T* kq_p;
    static constexpr int N = VecOf<T>::N;
    static constexpr bool DPP = true;
    CoefIn<T> ci;
    T* xlds;            // pair roles: this lane's vector in slot 0 of the pair's shared rings (U, Q0, Q1, of: 4 slots each)
    __device__ __forceinline__ void xput(int slot, const T (&v)[N]) const { store_vec(xlds + slot * (kWave * N), v); }
    __device__ __forceinline__ void xget(int slot, T (&v)[N]) const { load_vec(xlds + slot * (kWave * N), v); }
    const T* of_p;
    const T* u_in;
    const T* q0_in;
    const T* q1_in;
    T* u_out;
    T* q0_out;
    T* q1_out;
    int64_t P, nrow, R0, R1;
    int nrow32, R0_32, R1_32;  // the same rows as 32-bit integers: s_cmp_*_i32 for the per-row tests (there is no
                               // s_cmp_lt_i64, 64-bit tests become v_mov_b64 + v_cmp_*_i64 on the VALU); n0 < 2^31 - 64
    T ell, diag, dsc, kq, rdiag, rdsc;
    bool div3;          // Coef::div3
    bool iso_lean, iso_div3;  // FusedArgs::iso_lean / iso_div3
    bool cdiag;         // Coef::const_diag
    bool store_lane;
    uint32_t edgeL[N], edgeR[N], ch0[N], chLast[N];  // lane patterns as 0 / ~0 words (sel_bits)
    bool inL[N];        // element inside the row (x < L)
    uint32_t voff;      // this lane's byte offset inside a row: addresses are (uniform row base in SGPRs) + voff
    T* lds;             // this lane's vector in slot 0 of the wave's LDS rings; slot stride = kWave * N elements
    // eps instantiation: the K running sums of squared increments of this lane live in LDS (slot stride kWave
    // elements), not in registers -- sixteen more live registers do not fit under the 168 of three waves per SIMD (the
    // register version spilled 15 dwords to scratch, whose traffic shares vmcnt with the row prefetches: +33 % time)
    T* linc;
#ifdef DXH_ENABLE_PAIR
    static constexpr bool LINC = false;
#else
    static constexpr bool LINC = NORM;
#endif

#ifdef DXH_LDS_OPAQUE
    __device__ __forceinline__ void lds_put(int slot, const T (&v)[N]) const { store_vec_opaque(lds + slot * (kWave * N), v); }
    __device__ __forceinline__ void lds_get(int slot, T (&v)[N]) const { load_vec_opaque(lds + slot * (kWave * N), v); }
#else
    __device__ __forceinline__ void lds_put(int slot, const T (&v)[N]) const { store_vec(lds + slot * (kWave * N), v); }
    __device__ __forceinline__ void lds_get(int slot, T (&v)[N]) const { load_vec(lds + slot * (kWave * N), v); }
#endif

    __device__ __forceinline__ int64_t clampr(int64_t r) const { return r < 0 ? 0 : (r > nrow - 1 ? nrow - 1 : r); }
    // row pointer (wave-uniform) + 32-bit lane offset -> global_load/store with an SGPR base
    __device__ __forceinline__ const T* at(const T* plane, int64_t row) const {
        return reinterpret_cast<const T*>(reinterpret_cast<const char*>(plane + row * P) + voff);
    }
    __device__ __forceinline__ T* at(T* plane, int64_t row) const {
        return reinterpret_cast<T*>(reinterpret_cast<char*>(plane + row * P) + voff);
    }

    // raw rows of the given coefficient planes (absent planes leave zeros that derive_coefs never reads)
    __device__ __forceinline__ void load_craw(int64_t row, T (&raw)[3][N]) const {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < N; ++j) raw[q][j] = T(0);
        if (ci.mu) load_vec(at(ci.mu, row), raw[0]);
        if (ci.omega) load_vec(at(ci.omega, row), raw[1]);
        if (ci.ell) load_vec(at(ci.ell, row), raw[2]);
    }
    __device__ __forceinline__ void finish_coef(const T (&raw)[3][N], T (&cf)[4][N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) derive_coefs<T>(ci, raw[0][j], raw[1][j], raw[2][j], cf[0][j], cf[1][j], cf[2][j], cf[3][j]);
    }

    // un = rhs / diag + acc / dsc with wave-invariant divisors: reciprocal refinement (div_by_const), the second
    // correction only when the coefficient set needs it, the IEEE division for lanes' numerators that are tiny
    __device__ __forceinline__ void div_const(const T (&rhs)[N], const T (&acc)[N], T (&un)[N]) const {
        const T lo = abs_min_all<T, N>(rhs, acc);
        T qa[N], qb[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            qa[j] = div_by_const3(rhs[j], diag, rdiag);
            qb[j] = div_by_const3(acc[j], dsc, rdsc);
        }
        if (!div3) {  // divisors whose quotients need the second correction for some x
            DXH_UNIFORM_BRANCH();
#pragma unroll
            for (int j = 0; j < N; ++j) {
                qa[j] = div_refine(qa[j], rhs[j], diag, rdiag);
                qb[j] = div_refine(qb[j], acc[j], dsc, rdsc);
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) un[j] = qa[j] + qb[j];
        if (__builtin_expect(__any(lo < DivSafe<T>::tiny), 0)) {
            bool unsafe = false;
#pragma unroll
            for (int j = 0; j < N; ++j) unsafe = unsafe || div_unsafe(rhs[j]) || div_unsafe(acc[j]);
            if (__any(unsafe)) {
#pragma unroll
                for (int j = 0; j < N; ++j) un[j] = rhs[j] / diag + acc[j] / dsc;
            }
        }
    }

    // ACT: stages 0 .. ACT - 1 run, the others are skipped (the first steps of a chunk, fused_pipe_body: a stage has nothing
    // valid to work on before its predecessors have produced two rows)
    template <int PH, int ROLE = 0, int ACT = K>
    __device__ __forceinline__ void step(PipeState<T, K>& st, IncT<T> (&inc)[NORM ? K : 1][N], const int64_t r) const {
        constexpr int PFD = ROLE == 0 ? pipe_pfd<T, K, HET, ISO>() : 1;  // prefetch distance in steps
        constexpr bool INTRA_LDS = ROLE == 0;               // of / q1 hand-off between own stages through LDS
        // prefetch the level-0 rows of a later step: distance 1 straight into their ring slots, distance 2
        // (K >= 4: a step is long, the registers are there at 2 waves/SIMD; +2.6 % measured) into staging
        // registers that are moved into the rings at the end of the next step, when the loads are two steps old
        T frn[N];
        if constexpr (ROLE == 2) {
            // level-0 rows of THIS step come from the pair's shared rings (written by the head wave at
            // least one barrier ago): U[row+1], Q0[row], Q1[row], omega*f[row]
            xget(0 + ((PH + 1) & 3), st.U[0][(PH + 1) & 3]);
            xget(4 + (PH & 3), st.q0l[PH & 1]);
            xget(8 + (PH & 3), st.q1l[PH & 1]);
            xget(12 + (PH & 3), st.of0);
        } else if constexpr (ROLE == 1) {
            // head wave: only u is fetched a full step ahead; the q / omega*f rows of the next step are
            // loaded into the SAME registers right after stage 0 has consumed the current ones (register
            // budget of 128 for 4 waves per SIMD; the other waves of the SIMD cover the shorter distance)
            load_vec(at(u_in, clampr(r + 2)), st.U[0][(PH + 2) & 3]);
        } else if constexpr (PFD == 1) {
            const int64_t rn = clampr(r + 1);
            load_vec(at(u_in, clampr(r + 2)), st.U[0][(PH + 2) & 3]);
            load_vec(at(of_p, rn), frn);
            load_vec(at(q0_in, rn), st.q0l[(PH + 1) & 1]);
            load_vec(at(q1_in, rn), st.q1l[(PH + 1) & 1]);
            if constexpr (HET) load_craw(rn, st.craw[(PH + 1) & 1]);
            if constexpr (KQV) load_vec(at(kq_p, rn), st.kqn);
        } else {
            static_assert(!KQV, "the per-element threshold is prefetched at distance 1");
            const int64_t rn = clampr(r + 2);
            load_vec(at(u_in, clampr(r + 3)), st.pf[PH & 1][0]);
            load_vec(at(of_p, rn), st.pf[PH & 1][1]);
            load_vec(at(q0_in, rn), st.pf[PH & 1][2]);
            load_vec(at(q1_in, rn), st.pf[PH & 1][3]);
        }
        if constexpr (INTRA_LDS) {
            if constexpr (K > 1) lds_put(PH & 3, st.of0);  // omega*f of row r for the later stages
        } else if constexpr (ROLE == 2) {
            // delay line in registers: ofr[s] must hold row r - s
#pragma unroll
            for (int s = K - 1; s >= 1; --s)
#pragma unroll
                for (int j = 0; j < N; ++j) st.ofr[s][j] = st.ofr[s - 1][j];
#pragma unroll
            for (int j = 0; j < N; ++j) st.ofr[0][j] = st.of0[j];
        }
        if constexpr (ROLE == 1) xput(12 + (PH & 3), st.of0);  // forward omega*f[row r] to the tail wave
        T q0h[N];  // Q0 of the next level at the next stage's row, handed from stage to stage
        static_for<0, ACT>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int c = (PH - s) & 3;          // ring slot of this stage's row
            constexpr int cm = (c + 3) & 3, cp = (c + 1) & 3;
            constexpr int cur = PH & 1, prv = cur ^ 1;
            constexpr int lcur = (ROLE == 1) ? 0 : cur;  // slot of the loaded q rows (head: single buffer)
            const int64_t p = r - s;
            const int p32 = (int)p;
            T(&um)[N] = st.U[s][cm];
            T(&uc)[N] = st.U[s][c];
            T(&up)[N] = st.U[s][cp];
            T(&t0m)[N] = st.t0[s][prv];
            T(&t0c)[N] = st.t0[s][cur];
            T(&b0m)[N] = st.b0[s][prv];
            T(&b0c)[N] = st.b0[s][cur];
            T q1c[N], ofv[N];
            if constexpr (s == 0) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    q1c[j] = st.q1l[lcur][j];
                    ofv[j] = st.of0[j];
                }
            } else if constexpr (INTRA_LDS) {
                lds_get(4 + 2 * (s - 1) + cur, q1c);
                lds_get(c, ofv);
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) q1c[j] = st.q1r[s][cur][j];
                if constexpr (ROLE == 1) {
                    xget(12 + c, ofv);  // the head's own forward of row r - s, written s steps ago
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) ofv[j] = st.ofr[s][j];
                }
            }
            // destination of the new iterate (level s+1, row p) and that level's previous row
            constexpr bool last = (s == K - 1);
            constexpr bool ring_out = last && ROLE == 1;  // head: level K lives in the pair's shared U ring only
            T un_tmp[N], unm_tmp[N];
            T(&un)[N] = ring_out ? un_tmp : (last ? st.U[K][cur] : st.U[s + 1][c]);
            const T(&unm)[N] = ring_out ? unm_tmp : (last ? st.U[K][prv] : st.U[s + 1][cm]);
            if constexpr (ring_out) xget(0 + cm, unm_tmp);  // written by this wave one step ago

            T b1c[N], t1c[N];
            if constexpr (ISO && !HET && !KQV && std::is_same<T, float>::value) {
                // scalar threshold, float: the lane's N elements at once (one wave-level test, verified relaxations)
                T q0v[N], d0v[N], d1v[N];
#pragma unroll
                for (int j = 0; j < N; ++j) q0v[j] = (s == 0) ? st.q0l[lcur][j] : q0h[j];
                iso_split_vec<N>(q0v, q1c, kq, iso_lean, iso_div3, d0v, b0c, d1v, b1c);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    t0c[j] = ell * (b0c[j] - d0v[j]);
                    t1c[j] = ell * (b1c[j] - d1v[j]);
                }
            } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T qq[2] = {(s == 0) ? st.q0l[lcur][j] : q0h[j], q1c[j]}, d[2], b[2];
                if constexpr (HET) {
                    split_vars<T, 2, ISO, false>(qq, st.coef[s][3][j], d, b);
                    t0c[j] = st.coef[s][0][j] * (b[0] - d[0]);
                    t1c[j] = st.coef[s][0][j] * (b[1] - d[1]);
                } else if constexpr (KQV) {
                    split_vars<T, 2, ISO, false>(qq, st.kqv[s][j], d, b);
                    t0c[j] = ell * (b[0] - d[0]);
                    t1c[j] = ell * (b[1] - d[1]);
                } else {
                    split_vars<T, 2, ISO, true>(qq, kq, d, b);
                    t0c[j] = ell * (b[0] - d[0]);
                    t1c[j] = ell * (b[1] - d[1]);
                }
                b0c[j] = b[0];
                b1c[j] = b[1];
            }
            }
            if (p == 0) {  // replicate the row above the image
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    um[j] = uc[j];
                    t0m[j] = t0c[j];
                }
            }
            T t1l[N], ul[N], ur[N], ucl[N], ucr[N];
            shifted<-C, DPP>(t1c, t1l);
            shifted<-C, DPP>(uc, ul);
            shifted<+C, DPP>(uc, ur);
            if constexpr (COUPLED) {
                shifted<-1, DPP>(uc, ucl);
                shifted<+1, DPP>(uc, ucr);
            }
            T rhs[N], acc[N];
            if constexpr (!EDGE && DXH_FUSE_DPP_ADDS) {
                // interior strips: the lane shifts ride on the adds themselves (add_shifted_elem); neighbour elements
                // that the channel selects need as values anyway keep their shared v_mov_dpp
                static_for<0, N>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    rhs[j] = ofv[j] + (t0c[j] - t0m[j]);
                    rhs[j] = rhs[j] + add_shifted_elem<-C, j, true>(t1c[j], t1c);
                    acc[j] = um[j] + up[j];
                    constexpr bool shl = COUPLED && same_shift_source<-C, j, -1, 0, N>();
                    constexpr bool shr = COUPLED && same_shift_source<+C, j, +1, N - 1, N>();
                    if constexpr (shl)
                        acc[j] = acc[j] + ul[j];
                    else
                        acc[j] = add_shifted_elem<-C, j, false>(acc[j], uc);
                    if constexpr (shr)
                        acc[j] = acc[j] + ur[j];
                    else
                        acc[j] = add_shifted_elem<+C, j, false>(acc[j], uc);
                    if constexpr (COUPLED) {
                        acc[j] = acc[j] + sel_bits(ch0[j], uc[j], ucl[j]);
                        acc[j] = acc[j] + sel_bits(chLast[j], uc[j], ucr[j]);
                    }
                });
            } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                rhs[j] = ofv[j] + (t0c[j] - t0m[j]);
                if constexpr (EDGE)
                    rhs[j] = rhs[j] + (t1c[j] - sel_bits(edgeL[j], t1c[j], t1l[j]));
                else
                    rhs[j] = rhs[j] + (t1c[j] - t1l[j]);
                acc[j] = um[j] + up[j];
                if constexpr (EDGE) {
                    acc[j] = acc[j] + sel_bits(edgeL[j], uc[j], ul[j]);
                    acc[j] = acc[j] + sel_bits(edgeR[j], uc[j], ur[j]);
                } else {
                    acc[j] = acc[j] + ul[j];
                    acc[j] = acc[j] + ur[j];
                }
                if constexpr (COUPLED) {
                    acc[j] = acc[j] + sel_bits(ch0[j], uc[j], ucl[j]);
                    acc[j] = acc[j] + sel_bits(chLast[j], uc[j], ucr[j]);
                }
            }
            }
            if constexpr (ROLE == 1 && s == 0) {
                const int64_t rn = clampr(r + 1);
                load_vec(at(of_p, rn), st.of0);
                load_vec(at(q0_in, rn), st.q0l[0]);
                load_vec(at(q1_in, rn), st.q1l[0]);
            }
            if constexpr (HET) {
                if (cdiag) {  // only mu varies: the divisors are wave-invariant like in the scalar pipeline
                    DXH_UNIFORM_BRANCH();
                    div_const(rhs, acc, un);
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) un[j] = rhs[j] / st.coef[s][1][j] + acc[j] / st.coef[s][2][j];
                }
            } else {
                div_const(rhs, acc, un);
            }
            if (p32 >= nrow32) {  // below the image: replicate the new iterate
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) un[j] = unm[j];
            }
            if (NORM && p32 >= R0_32 && p32 < R1_32) {  // wave-uniform; rows of this chunk only (each element once)
                if constexpr (LINC) {
                    // one running sum per lane and stage, kept in LDS through volatile accesses (the optimiser would pull
                    // it back into a register): the row's elements are summed in order, then added to it.  The lane mask
                    // (strip halo) does not depend on the row and is applied when the sums are read back (fused_pipe_body)
                    IncT<T> t = IncT<T>(0);
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const IncT<T> dlt = un[j] - uc[j];
                        const IncT<T> sq = inL[j] ? dlt * dlt : IncT<T>(0);
                        t = j == 0 ? sq : t + sq;
                    }
                    volatile T* slot = linc + s * kWave;
                    *slot = *slot + t;
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const IncT<T> dlt = un[j] - uc[j];
                        inc[NORM ? s : 0][j] += (store_lane && inL[j]) ? dlt * dlt : IncT<T>(0);
                    }
                }
            }
            T unr[N], q1o[N], q0o[N];
            shifted<+C, DPP>(un, unr);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if constexpr (EDGE)
                    q1o[j] = (sel_bits(edgeR[j], un[j], unr[j]) - un[j]) + b1c[j];
                else
                    q1o[j] = (unr[j] - un[j]) + b1c[j];
                q0o[j] = (un[j] - unm[j]) + b0m[j];
            }
            if constexpr (last && ROLE == 1) {
                // hand level K over to the tail wave: U and Q1 of row p, Q0 of row p-1
                xput(0 + c, un);
                xput(8 + c, q1o);
                xput(4 + cm, q0o);
            } else if constexpr (last) {
                if (store_lane) {
                    if (p32 >= R0_32 && p32 < R1_32) {
                        if constexpr (NTS) {
                            store_vec_nt(at(q1_out, p), q1o);
                            store_vec_nt(at(u_out, p), un);
                        } else {
                            store_vec(at(q1_out, p), q1o);
                            store_vec(at(u_out, p), un);
                        }
                    }
                    if (p32 - 1 >= R0_32 && p32 - 1 < R1_32) {
                        if constexpr (NTS)
                            store_vec_nt(at(q0_out, p - 1), q0o);
                        else
                            store_vec(at(q0_out, p - 1), q0o);
                    }
                }
            } else {
                if constexpr (INTRA_LDS) {
                    lds_put(4 + 2 * s + prv, q1o);  // read by stage s+1 at the next step (its parity "cur")
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) st.q1r[s + 1][prv][j] = q1o[j];
                }
#pragma unroll
                for (int j = 0; j < N; ++j) q0h[j] = q0o[j];
            }
        });
        if constexpr (HET) {
            // the coefficients follow their rows down the pipeline; stage 0 gets the prefetched next row
#pragma unroll
            for (int s2 = K - 1; s2 >= 1; --s2)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < N; ++j) st.coef[s2][q][j] = st.coef[s2 - 1][q][j];
            finish_coef(st.craw[(PH + 1) & 1], st.coef[0]);
        }
        if constexpr (KQV) {
#pragma unroll
            for (int s2 = K - 1; s2 >= 1; --s2)
#pragma unroll
                for (int j = 0; j < N; ++j) st.kqv[s2][j] = st.kqv[s2 - 1][j];
#pragma unroll
            for (int j = 0; j < N; ++j) st.kqv[0][j] = st.kqn[j];
        }
        if constexpr (ROLE != 0) {
            return;  // tail: the next step reads its rows from the shared rings; head: loaded in place above
        }
        if constexpr (PFD == 2) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                st.U[0][(PH + 2) & 3][j] = st.pf[(PH + 1) & 1][0][j];
                frn[j] = st.pf[(PH + 1) & 1][1][j];
                st.q0l[(PH + 1) & 1][j] = st.pf[(PH + 1) & 1][2][j];
                st.q1l[(PH + 1) & 1][j] = st.pf[(PH + 1) & 1][3][j];
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) st.of0[j] = frn[j];
    }
};

// Fill the strip / chunk geometry and the plane pointers of a pipeline object.  KG = number of fused
// iterations the strip halo is sized for (the pipeline's own K for a solo wave, the pair's total for a wave pair).
template <int KG, typename T, int C, typename PipeT>
__device__ __forceinline__ bool pipe_setup(PipeT& pp, const FusedArgs<T>& a, const int strip, const int chunk, const int lane) {
    constexpr int N = VecOf<T>::N;
    using SG = StripGeomK<T, C, KG>;
    chunk_rows(a, chunk, pp.R0, pp.R1);
    if (pp.R0 >= a.n0) return false;
    const int64_t x0 = (int64_t)strip * SG::STRIDE + (int64_t)(lane - SG::LH) * N;
    int64_t xl = x0;
    if (xl < 0) xl = 0;
    if (xl > a.pitch - N) xl = a.pitch - N;
    pp.store_lane = lane >= SG::LH && lane < SG::LH + SG::VALID && x0 < a.L;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int64_t x = x0 + j;
        pp.edgeL[j] = lane_mask_bits(x < C);
        pp.edgeR[j] = lane_mask_bits(x >= a.L - C);
        const int cc = (int)(((x % C) + C) % C);
        pp.ch0[j] = lane_mask_bits(cc == 0);
        pp.chLast[j] = lane_mask_bits(cc == C - 1);
        pp.inL[j] = x >= 0 && x < a.L;
    }
    pp.of_p = a.of;
    pp.u_in = a.u_in;
    pp.q0_in = a.q0_in;
    pp.q1_in = a.q1_in;
    pp.u_out = a.u_out;
    pp.q0_out = a.q0_out;
    pp.q1_out = a.q1_out;
    pp.voff = (uint32_t)(xl * (int64_t)sizeof(T));
    pp.P = a.pitch;
    pp.nrow = a.n0;
    pp.nrow32 = (int)a.n0;  // the host keeps n0 below 2^31 - 64 for these kernels (multi_supported)
    pp.R0_32 = (int)pp.R0;
    pp.R1_32 = (int)pp.R1;
    pp.ell = a.c.ell;
    pp.diag = a.c.diag;
    pp.dsc = a.c.dsc;
    pp.kq = a.c.kq;
    pp.rdiag = a.c.rdiag;
    pp.rdsc = a.c.rdsc;
    pp.div3 = a.c.div3 != 0;
    pp.iso_lean = a.iso_lean != 0;
    pp.iso_div3 = a.iso_div3 != 0;
    pp.cdiag = a.c.const_diag != 0;
    pp.ci = a.ci;
    pp.kq_p = a.cp.kq;
    return true;
}

template <typename T, int K>
__device__ __forceinline__ void pipe_state_zero(PipeState<T, K>& st) {
    constexpr int N = VecOf<T>::N;
#pragma unroll
    for (int l = 0; l <= K; ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) st.U[l][i][j] = T(0);
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) st.t0[l][i][j] = st.b0[l][i][j] = st.q1r[l][i][j] = T(0);
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int j = 0; j < N; ++j) st.ofr[l][j] = T(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) st.q0l[i][j] = st.q1l[i][j] = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) st.of0[j] = T(0);
    // heterogeneous coefficients of the not yet started stages: 1 keeps the discarded leading rows finite
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < N; ++j) st.coef[l][q][j] = T(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < N; ++j) st.craw[i][q][j] = T(0);
    // thresholds of the not yet started stages: 0 (shrink by nothing) keeps the discarded leading rows finite
#pragma unroll
    for (int l = 0; l < K; ++l)
#pragma unroll
        for (int j = 0; j < N; ++j) st.kqv[l][j] = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) st.kqn[j] = T(0);
}

// Prologue of a wave that reads level 0 from HBM (solo / head): phase 0 at row rs, rows rs-1, rs, rs+1 of u in
// ring slots 3, 0, 1, the carried (t0, b0) of row rs-1 recomputed from the stored q.
template <int PFD, bool ISO, bool HET = false, bool KQV = false, typename T, int K, typename PipeT>
__device__ __forceinline__ void pipe_prologue(const PipeT& pp, PipeState<T, K>& st, const int64_t rs) {
    constexpr int N = VecOf<T>::N;
    load_vec(pp.at(pp.u_in, pp.clampr(rs - 1)), st.U[0][3]);
    load_vec(pp.at(pp.u_in, pp.clampr(rs)), st.U[0][0]);
    load_vec(pp.at(pp.u_in, pp.clampr(rs + 1)), st.U[0][1]);
    load_vec(pp.at(pp.of_p, pp.clampr(rs)), st.of0);
    load_vec(pp.at(pp.q0_in, pp.clampr(rs)), st.q0l[0]);
    load_vec(pp.at(pp.q1_in, pp.clampr(rs)), st.q1l[0]);
    if constexpr (PFD == 2) {  // prefetch distance 2: rows consumed at the end of the first step
        load_vec(pp.at(pp.u_in, pp.clampr(rs + 2)), st.pf[1][0]);
        load_vec(pp.at(pp.of_p, pp.clampr(rs + 1)), st.pf[1][1]);
        load_vec(pp.at(pp.q0_in, pp.clampr(rs + 1)), st.pf[1][2]);
        load_vec(pp.at(pp.q1_in, pp.clampr(rs + 1)), st.pf[1][3]);
    }
    const int64_t rm = pp.clampr(rs - 1);
    T q0v[N], q1v[N];
    load_vec(pp.at(pp.q0_in, rm), q0v);
    load_vec(pp.at(pp.q1_in, rm), q1v);
    T cfm[4][N];
    if constexpr (HET) {
        T raw[3][N];
        pp.load_craw(rm, raw);
        pp.finish_coef(raw, cfm);
        pp.load_craw(pp.clampr(rs), raw);
        pp.finish_coef(raw, st.coef[0]);
    }
    T kqm[N];
    if constexpr (KQV) {
        load_vec(pp.at(pp.kq_p, rm), kqm);
        load_vec(pp.at(pp.kq_p, pp.clampr(rs)), st.kqv[0]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        T qq[2] = {q0v[j], q1v[j]}, d[2], b[2];
        if constexpr (HET) {
            split_vars<T, 2, ISO, false>(qq, cfm[3][j], d, b);
            st.t0[0][1][j] = cfm[0][j] * (b[0] - d[0]);
        } else if constexpr (KQV) {
            split_vars<T, 2, ISO, false>(qq, kqm[j], d, b);
            st.t0[0][1][j] = pp.ell * (b[0] - d[0]);
        } else {
            split_vars<T, 2, ISO, true>(qq, pp.kq, d, b);
            st.t0[0][1][j] = pp.ell * (b[0] - d[0]);  // parity "prv" of phase 0
        }
        st.b0[0][1][j] = b[0];
    }
}

template <int KLOC, int G0, typename T>
__device__ __forceinline__ void pipe_write_partials(const FusedArgs<T>& a, IncT<T> (&inc)[KLOC][VecOf<T>::N], const int wave_id,
                                                    const int lane) {
    constexpr int N = VecOf<T>::N;
#pragma unroll
    for (int s = 0; s < KLOC; ++s) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) v += (double)inc[s][j];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
        if (lane == 0) a.inc_partial[(size_t)(G0 + s) * a.nwaves_total + wave_id] = v;
    }
}

template <typename T, int C, bool COUPLED, bool ISO, int K, bool NTS, bool EDGE, bool NORM, bool HET = false, bool KQV = false>
__device__ __forceinline__ void fused_pipe_body(const FusedArgs<T>& a, const int strip, const int chunk, const int lane,
                                                T* lds_lane, T* linc_lane) {
    constexpr int N = VecOf<T>::N;
    static_assert(!HET || K <= 3, "heterogeneous pipelines prefetch at distance 1 (K <= 3)");
    using PipeT = Pipe<T, C, COUPLED, ISO, K, NTS, EDGE, NORM, HET, KQV>;
    PipeT pp;
    pp.lds = lds_lane;
    pp.xlds = lds_lane;
    pp.linc = linc_lane;
    if (!pipe_setup<K, T, C>(pp, a, strip, chunk, lane)) return;

    PipeState<T, K> st;
    pipe_state_zero(st);
    {
        // LDS rings start zeroed: the leading (discarded) rows must not see NaN bit patterns that would
        // trip the division screen for the whole wave
        T z[N];
#pragma unroll
        for (int j = 0; j < N; ++j) z[j] = T(0);
#pragma unroll
        for (int i = 0; i < PipeLds<K>::SLOTS; ++i) pp.lds_put(i, z);
    }
    const int64_t rs = pp.R0 - (K - 1);
    pipe_prologue<pipe_pfd<T, K, HET, ISO>(), ISO, HET, KQV>(pp, st, rs);
    // the last stage must reach row R1: steps rs .. R1 + K - 1, rounded up to whole groups of 4
    // (extra steps touch clamped rows only and store nothing)
    const int64_t r_end = pp.R1 + (K - 1);
    IncT<T> inc[NORM ? K : 1][N];
#pragma unroll
    for (int s = 0; s < (NORM ? K : 1); ++s)
#pragma unroll
        for (int j = 0; j < N; ++j) inc[s][j] = IncT<T>(0);
    if constexpr (PipeT::LINC) {
#pragma unroll
        for (int i = 0; i < K; ++i) *(volatile T*)(linc_lane + i * kWave) = T(0);
    }
    // Start-up.  Stage s first matters at step 2 s - 1 of the chunk: its first VALID row is rs + s (step 2 s), and the
    // step before builds the (t0, b0) it carries from the row above; earlier it would only turn zeros into garbage that
    // nothing reads.  So the first groups of four steps run with act(t) = min(K, (t + 1) / 2 + 1) stages -- 2.25 of a K = 4
    // chunk's row steps less (3 % of an 80-row chunk).  -DDXH_NO_STAGE_SKIP: all stages from the first step (A/B).
    int64_t r = rs;
#ifndef DXH_NO_STAGE_SKIP
    // (not in the isotropic and array-coefficient instantiations: they sit at their register limit, and the extra code
    // paths cost them scratch -- isotropic RGB K = 4: 20 bytes per lane and +6.5 % time, profiles/r03_stage_skip_ab.txt)
    if constexpr (!ISO && !HET) {
        constexpr int NG = (2 * K - 4) / 4 + 1;   // groups that contain a step with fewer than K active stages
        static_for<0, NG>([&](auto gc) {
            constexpr int t0 = 4 * decltype(gc)::value;
            constexpr auto act = [](int t) { return (t + 1) / 2 + 1 < K ? (t + 1) / 2 + 1 : K; };
            pp.template step<0, 0, act(t0)>(st, inc, r);
            pp.template step<1, 0, act(t0 + 1)>(st, inc, r + 1);
            pp.template step<2, 0, act(t0 + 2)>(st, inc, r + 2);
            pp.template step<3, 0, act(t0 + 3)>(st, inc, r + 3);
            r += 4;
        });
    }
#endif
    for (; r <= r_end; r += 4) {
        pp.template step<0>(st, inc, r);
        pp.template step<1>(st, inc, r + 1);
        pp.template step<2>(st, inc, r + 2);
        pp.template step<3>(st, inc, r + 3);
    }
    if constexpr (PipeT::LINC) {
        // a wave's LDS operations complete in order: the sums are final here.  Lanes / elements outside the strip's
        // stored part or outside the row drop out now (their sums may be anything, NaN included)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            const T v = *(volatile T*)(linc_lane + s * kWave);
            inc[s][0] = pp.store_lane ? v : IncT<T>(0);   // the lane's whole sum in the first slot, the others stay 0
        }
    }
    if constexpr (NORM) pipe_write_partials<K, 0>(a, inc, chunk * a.nstrips + strip, lane);
}

#ifdef DXH_ENABLE_PAIR
// --------------------------------------------------------------------------------------------------
// EXPERIMENT, compiled only with -DDXH_ENABLE_PAIR (bit-exact, not faster; measured notes in DESIGN.md).
// Wave-pair variant of the K = 4 pipeline.  A solo wave that runs four stages needs ~200 VGPRs, i.e. two
// waves per SIMD, and the kernel is bound by VALU issue at that occupancy.  Here a strip is owned by a PAIR of
// waves: the head wave runs global stages 0-1 (HBM loads in, level-2 rows out to LDS), the tail wave runs stages
// 2-3 (LDS in, HBM stores out) three row-steps behind the head.  The pair's shared rings (U, Q0, Q1, omega*f)
// have 4 slots each, slot = (row - rs) & 3.  At iteration t the head writes U[t-1], Q1[t-1], Q0[t-2], of[t]
// (rows relative to rs) and the tail reads U[t-2], Q0[t-3], Q1[t-3], of[t-3]: always different slots inside
// one iteration, and every slot is read one to three barriers after it was written and overwritten at the
// earliest one barrier after it was read.  One __syncthreads() per iteration, the same trip count for every
// wave of the block (the count depends on the chunk only, and all waves of a block share the chunk).
// --------------------------------------------------------------------------------------------------
#ifndef DXH_PAIR_LB
#define DXH_PAIR_LB 4
#endif
constexpr int kPairSlots = 16;  // U 0-3, Q0 4-7, Q1 8-11, omega*f 12-15
constexpr int kPairLag = 3;

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), which would drain
// the head wave's prefetched HBM loads and the tail wave's stores at every row step.  The rings live in LDS, so
// lgkmcnt(0) before the barrier is all the hand-over needs.
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int C, bool COUPLED, bool ISO, bool NTS, bool EDGE, bool NORM>
__device__ __forceinline__ void fused_pair_body(const FusedArgs<T>& a, const int strip, const int chunk, const int lane,
                                                const bool tail, const bool active, T* xlds_lane) {
    constexpr int N = VecOf<T>::N;
    constexpr int KH = 2, KT = 4;
    Pipe<T, C, COUPLED, ISO, KH, NTS, EDGE, NORM> pp;
    pp.lds = xlds_lane;
    pp.xlds = xlds_lane;
    // an inactive pair (strip past the end in the last block) runs the same barriers on a clamped strip and
    // stores nothing
    const bool ok = pipe_setup<KT, T, C>(pp, a, strip, chunk, lane);
    if (!ok) return;  // chunk past the end: uniform over the block
    if (!active) pp.store_lane = false;
    PipeState<T, KH> st;
    pipe_state_zero(st);
    const int64_t rs = pp.R0 - (KT - 1);
    if (!tail) {
        T z[N];
#pragma unroll
        for (int j = 0; j < N; ++j) z[j] = T(0);
#pragma unroll
        for (int i = 0; i < kPairSlots; ++i) pp.xput(i, z);
        pipe_prologue<1, ISO>(pp, st, rs);
    }
    __syncthreads();
    IncT<T> inc[NORM ? KH : 1][N];
#pragma unroll
    for (int s = 0; s < (NORM ? KH : 1); ++s)
#pragma unroll
        for (int j = 0; j < N; ++j) inc[s][j] = IncT<T>(0);
    // head steps rs .. R1 + KT - 1, tail steps rs .. R1 + 1 at a lag of kPairLag iterations
    const int64_t r_end = pp.R1 + (KT - KH - 1) + kPairLag;
    for (int64_t r = rs; r <= r_end; r += 4) {
        if (!tail) pp.template step<0, 1>(st, inc, r); else pp.template step<1, 2>(st, inc, r - kPairLag);
        pair_barrier();
        if (!tail) pp.template step<1, 1>(st, inc, r + 1); else pp.template step<2, 2>(st, inc, r + 1 - kPairLag);
        pair_barrier();
        if (!tail) pp.template step<2, 1>(st, inc, r + 2); else pp.template step<3, 2>(st, inc, r + 2 - kPairLag);
        pair_barrier();
        if (!tail) pp.template step<3, 1>(st, inc, r + 3); else pp.template step<0, 2>(st, inc, r + 3 - kPairLag);
        pair_barrier();
    }
    if constexpr (NORM) {
        if (active) {
            const int wave_id = chunk * a.nstrips + strip;
            if (!tail)
                pipe_write_partials<KH, 0>(a, inc, wave_id, lane);
            else
                pipe_write_partials<KH, KH>(a, inc, wave_id, lane);
        }
    }
}

template <typename T, int C, bool COUPLED, bool ISO, bool NORM = false, bool NTS = true>
__global__ void __launch_bounds__(256, DXH_PAIR_LB) k_fused_pair(FusedArgs<T> a) {
    constexpr int N = VecOf<T>::N;
    using SG = StripGeomK<T, C, 4>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int PPB = 2;  // pairs (strips) per block
    const int pair = wave >> 1;
    // waves land on SIMD (wave & 3): alternate the roles from block to block so that every SIMD hosts heads
    // (HBM loads) and tails (HBM stores) alike
    const bool tail = ((wave ^ (int)blockIdx.x) & 1) != 0;
    const int nbx = (a.nstrips + PPB - 1) / PPB;
    int bx, chunk;
    if (!block_to_tile_sub((int)blockIdx.x, nbx, a, bx, chunk)) return;
    if (chunk >= a.nchunks) return;  // block-uniform
    int strip = bx * PPB + pair;
    const bool active = strip < a.nstrips;
    if (!active) strip = a.nstrips - 1;
    // border handling is chosen per BLOCK so that every wave of the block meets the same barriers
    const int64_t wlo = (int64_t)(bx * PPB) * SG::STRIDE - SG::LH * N;
    const int last_strip = (bx * PPB + PPB - 1 < a.nstrips) ? bx * PPB + PPB - 1 : a.nstrips - 1;
    const int64_t whi = (int64_t)last_strip * SG::STRIDE - SG::LH * N + kWave * N;
    const bool edge = (wlo < C) || (whi > a.L - C);
    __shared__ typename VecOf<T>::type pair_rings[PPB * kPairSlots * kWave];
    T* xl = reinterpret_cast<T*>(pair_rings + (size_t)pair * kPairSlots * kWave + lane);
    if (edge)
        fused_pair_body<T, C, COUPLED, ISO, NTS, true, NORM>(a, strip, chunk, lane, tail, active, xl);
    else
        fused_pair_body<T, C, COUPLED, ISO, NTS, false, NORM>(a, strip, chunk, lane, tail, active, xl);
    if (active && tail) signal_chunk_done(a, chunk, lane);
}
#endif  // DXH_ENABLE_PAIR

template <typename T, int C, bool COUPLED, bool ISO, int K, bool NORM = false, bool NTS = true, bool HET = false, bool KQV = false>
__global__ void __launch_bounds__(256, (pipe_waves<T, K, HET, ISO>())) k_fused_multi(FusedArgs<T> a) {
    constexpr int N = VecOf<T>::N;
    using SG = StripGeomK<T, C, K>;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    constexpr int WPB = 4;
    const int nbx = (a.nstrips + WPB - 1) / WPB;
    int bx, chunk;
    if (!block_to_tile_sub((int)blockIdx.x, nbx, a, bx, chunk)) return;  // block-uniform
    const int strip = bx * WPB + wave;
    if (strip >= a.nstrips || chunk >= a.nchunks) return;  // wave-uniform
    const int64_t w0 = (int64_t)strip * SG::STRIDE - SG::LH * N;  // first element of the wave's window
    const bool edge = (w0 < C) || (w0 + kWave * N > a.L - C);     // wave-uniform
    __shared__ typename VecOf<T>::type lds_rings[WPB * PipeLds<K>::SLOTS * kWave];
    T* lds_lane = reinterpret_cast<T*>(lds_rings + (size_t)wave * PipeLds<K>::SLOTS * kWave + lane);
    // running sums of squared increments of the eps instantiation: [wave][stage][lane]
    __shared__ T lds_inc[NORM ? WPB * K * kWave : 1];
    T* linc_lane = NORM ? lds_inc + (size_t)wave * (K * kWave) + lane : lds_inc;
    if (edge)
        fused_pipe_body<T, C, COUPLED, ISO, K, NTS, true, NORM, HET, KQV>(a, strip, chunk, lane, lds_lane, linc_lane);
    else
        fused_pipe_body<T, C, COUPLED, ISO, K, NTS, false, NORM, HET, KQV>(a, strip, chunk, lane, lds_lane, linc_lane);
    signal_chunk_done(a, chunk, lane);
}

// ============================================================================================
// K Jacobi sweeps in one pass (Jacobi(maxiter > 1), H1_regularization; jacobi.py:110-114):
//     x_{s+1} = rs + N(x_s) / dsc        rs = rhs / diag is the same for every sweep
// Row-marching waves as in the fused TV sweep; stage s applies sweep s to row r - s and takes the row below it
// from stage s - 1's output of the same step.  Only x_0 and rs are read and x_K written: 12 B per element for K
// sweeps instead of 12 B per sweep.  Scalar coefficients, dim = 2, no tolerance test.
// ============================================================================================
template <typename T>
struct JacobiMultiArgs {
    const T* rs;
    const T* x_in;
    T* x_out;
    int64_t n0, L, pitch, rows_per_chunk;
    int nstrips, nchunks;
    T dsc;
    T rdsc;    // RN(1/dsc), 0 = divide the IEEE way
    int div3;  // Coef::div3 for dsc
};

template <typename T, int C, bool COUPLED, int K>
__global__ void __launch_bounds__(256) k_jacobi_multi(JacobiMultiArgs<T> a) {
    constexpr int N = VecOf<T>::N;
    using SG = StripGeomK<T, C, K>;
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
    bool edgeL[N], edgeR[N], ch0[N], chLast[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int64_t x = x0 + j;
        edgeL[j] = x < C;
        edgeR[j] = x >= a.L - C;
        const int cc = (int)(((x % C) + C) % C);
        ch0[j] = cc == 0;
        chLast[j] = cc == C - 1;
    }
    const int64_t P = a.pitch;
    const T* rs = a.rs + xl;
    const T* xin = a.x_in + xl;
    T* xout = a.x_out + xl;
    auto clampr = [&](int64_t r) -> int64_t { return r < 0 ? 0 : (r > a.n0 - 1 ? a.n0 - 1 : r); };

    // stage s: um[s], uc[s] = rows p - 1, p of x_s (p = r - s); rsd[s] = rs of row r - s
    T um[K][N], uc[K][N], rsd[K][N];
#pragma unroll
    for (int s2 = 0; s2 < K; ++s2)
#pragma unroll
        for (int j = 0; j < N; ++j) um[s2][j] = uc[s2][j] = rsd[s2][j] = T(0);
    const int64_t rstart = R0 - (K - 1);  // the last stage reaches R0 with valid rows above it
    load_vec(xin + clampr(rstart - 1) * P, um[0]);
    load_vec(xin + clampr(rstart) * P, uc[0]);
    T up0[N], rsn[N];
    load_vec(xin + clampr(rstart + 1) * P, up0);
    load_vec(rs + clampr(rstart) * P, rsn);
    for (int64_t r = rstart; r <= R1 - 1 + (K - 1); ++r) {
        T upn[N], rsf[N];
        load_vec(xin + clampr(r + 2) * P, upn);
        load_vec(rs + clampr(r + 1) * P, rsf);
        // shift the rs delay line: rsd[s] = rs of row r - s
#pragma unroll
        for (int s2 = K - 1; s2 >= 1; --s2)
#pragma unroll
            for (int j = 0; j < N; ++j) rsd[s2][j] = rsd[s2 - 1][j];
#pragma unroll
        for (int j = 0; j < N; ++j) rsd[0][j] = rsn[j];
        T below[N];  // row p + 1 of the stage's input level: loaded for stage 0, the previous stage's output otherwise
#pragma unroll
        for (int j = 0; j < N; ++j) below[j] = up0[j];
#pragma unroll
        for (int s2 = 0; s2 < K; ++s2) {
            const int64_t p = r - s2;
            T xn[N];
            T ul[N], ur[N], ucl[N], ucr[N];
            shifted<-C, true>(uc[s2], ul);
            shifted<+C, true>(uc[s2], ur);
            if constexpr (COUPLED) {
                shifted<-1, true>(uc[s2], ucl);
                shifted<+1, true>(uc[s2], ucr);
            }
            const bool top = p <= 0, bottom = p >= a.n0 - 1;  // uniform: replicate ghosts (jacobi.py:43,52)
            T accv[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T acc = (top ? uc[s2][j] : um[s2][j]) + (bottom ? uc[s2][j] : below[j]);
                acc = acc + (edgeL[j] ? uc[s2][j] : ul[j]);
                acc = acc + (edgeR[j] ? uc[s2][j] : ur[j]);
                if constexpr (COUPLED) {
                    acc = acc + (ch0[j] ? uc[s2][j] : ucl[j]);
                    acc = acc + (chLast[j] ? uc[s2][j] : ucr[j]);
                }
                accv[j] = acc;
            }
            // acc / dsc through the host-rounded reciprocal (div_by_const: correctly rounded; the second correction
            // is skipped when the exhaustive float check allows it); tiny numerators and odd divisors: IEEE division
            bool ieee = a.rdsc == T(0);
            if (!ieee) {
                T q[N];
#pragma unroll
                for (int j = 0; j < N; ++j) q[j] = div_by_const3(accv[j], a.dsc, a.rdsc);
                if (!a.div3) {
                    DXH_UNIFORM_BRANCH();
#pragma unroll
                    for (int j = 0; j < N; ++j) q[j] = div_refine(q[j], accv[j], a.dsc, a.rdsc);
                }
#pragma unroll
                for (int j = 0; j < N; ++j) xn[j] = rsd[s2][j] + q[j];
                const T lo = abs_min_all<T, N>(accv, accv);
                if (__builtin_expect(__any(lo < DivSafe<T>::tiny), 0)) {
                    bool unsafe = false;
#pragma unroll
                    for (int j = 0; j < N; ++j) unsafe = unsafe || div_unsafe(accv[j]);
                    ieee = __any(unsafe);
                }
            }
            if (ieee) {
                DXH_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) xn[j] = rsd[s2][j] + accv[j] / a.dsc;
            }
            if (s2 == K - 1) {
                if (store_lane && p >= R0 && p < R1) store_vec_nt(xout + p * P, xn);
            }
            // rotate this stage's window; its output is the row below for the next stage (which sits one row up)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                um[s2][j] = uc[s2][j];
                uc[s2][j] = below[j];
                below[j] = xn[j];
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            up0[j] = upn[j];
            rsn[j] = rsf[j];
        }
    }
}

}  // namespace dxh
