// Multigrid (V-cycle) and conjugate-gradient solvers on device planes -- see dxh_solvers.h.
// Every kernel evaluates the reference's NumPy expressions operation by operation (-ffp-contract=off), so the
// multigrid results are bit-identical to the reference's (oracle/solvers_numpy.py states the algorithm and its
// quirks; tests/golden/solvers_golden.npz holds the reference's own outputs).  gfx950 only.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "dxh_solvers.h"

namespace dxh {

namespace {

dim3 grid_of(const Geom& g) {
    const int64_t bpr = (g.L + 255) / 256;
    return dim3((unsigned)(bpr * g.n0));
}

__device__ __forceinline__ int64_t axis_len(const Geom& g, int ax) { return ax == 0 ? g.n0 : (ax == 1 ? g.n1 : g.n2); }
__device__ __forceinline__ int64_t axis_idx(const ElemIdx& e, int ax) { return ax == 0 ? e.row : (ax == 1 ? e.i1 : e.i2); }
__device__ __forceinline__ int64_t axis_stride(const Geom& g, int ax) { return ax == 0 ? g.pitch : (ax == 1 ? g.n2 : 1); }

// laplace(x, dim, h) of utils/derivatives.py:62-121 at one element, for a diffusion coefficient that is either 1
// (MG: multiplied afterwards, mg.py:105-107) or a scalar / plane INSIDE the divergence (CG, cg.py:41-44):
//   0.5 * ( backward_diff(c * forward_diff(x)) + forward_diff(c * backward_diff(x)) )   summed over the axes,
// forward_diff = x[i] - x[i-1] (0 at i = 0), backward_diff = x[i+1] - x[i] (0 at the end), each divided by h.
template <typename T, bool INSIDE>
__device__ __forceinline__ T laplace_at(const Geom& g, const ElemIdx& e, const T* __restrict__ x, T c_s,
                                        const T* __restrict__ c_p, T h, bool use_h) {
    T lap = T(0);
    const T xc = x[e.off];
    for (int ax = 0; ax < g.dim; ++ax) {
        const int64_t n = axis_len(g, ax), i = axis_idx(e, ax), st = axis_stride(g, ax);
        const bool has_m = i > 0, has_p = i < n - 1;
        T fm = T(0), fp = T(0);  // forward_diff at i and i+1 (== backward_diff at i-1 and i)
        if (has_m) { fm = xc - x[e.off - st]; if (use_h) fm = fm / h; }
        if (has_p) { fp = x[e.off + st] - xc; if (use_h) fp = fp / h; }
        T t1, t2;
        if constexpr (INSIDE) {
            const T cc = c_p ? c_p[e.off] : c_s;
            const T cm = (c_p && has_m) ? c_p[e.off - st] : c_s;
            const T cpl = (c_p && has_p) ? c_p[e.off + st] : c_s;
            // term 1: backward_diff(c * F)[i] = c[i+1] F[i+1] - c[i] F[i];  F[i] = fm, F[i+1] = fp
            t1 = has_p ? (cpl * fp - cc * fm) : T(0);
            // term 2: forward_diff(c * B)[i] = c[i] B[i] - c[i-1] B[i-1];  B[i] = fp, B[i-1] = fm
            t2 = has_m ? (cc * fp - cm * fm) : T(0);
        } else {
            t1 = has_p ? (fp - fm) : T(0);
            t2 = has_m ? (fp - fm) : T(0);
        }
        if (use_h) { t1 = t1 / h; t2 = t2 / h; }
        lap = lap + T(0.5) * (t1 + t2);
    }
    return lap;
}

// r = rhs - (mass * x - diffusion * laplace(x))            mg.py:93-107, 217
template <typename T>
__global__ void __launch_bounds__(256) k_mg_residual(Geom g, const T* __restrict__ x, const T* __restrict__ rhs, T mass_s,
                                                      const T* __restrict__ mass_p, T diff_s, const T* __restrict__ diff_p,
                                                      T h, T* __restrict__ r) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const T lap = laplace_at<T, false>(g, e, x, T(1), nullptr, h, true);
    const T m = mass_p ? mass_p[e.off] : mass_s;
    const T d = diff_p ? diff_p[e.off] : diff_s;
    const T op = m * x[e.off] - d * lap;
    r[e.off] = rhs[e.off] - op;
}

// out = mass * x - laplace(x, diffusion_coeff=diffusion)    cg.py:37-44
template <typename T>
__global__ void __launch_bounds__(256) k_cg_apply(Geom g, const T* __restrict__ x, T mass_s, const T* __restrict__ mass_p,
                                                   T diff_s, const T* __restrict__ diff_p, T h, int use_h,
                                                   T* __restrict__ out) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    const T lap = laplace_at<T, true>(g, e, x, diff_s, diff_p, h, use_h != 0);
    const T m = mass_p ? mass_p[e.off] : mass_s;
    out[e.off] = m * x[e.off] - lap;
}

// restriction, mg.py:109-130: pair averages along axis 0, then 1 (, then 2); a trailing odd element is dropped
template <typename T>
__global__ void __launch_bounds__(256) k_mg_restrict(Geom gf, Geom gc, const T* __restrict__ xf, T* __restrict__ xc) {
    ElemIdx e;
    if (!elem_index(gc, e)) return;
    const int64_t base = 2 * e.row * gf.pitch + 2 * e.i1 * gf.n2 + (gf.dim == 3 ? 2 * e.i2 : e.i2);
    const int nk = gf.dim == 3 ? 2 : 1;
    T b[2];
    for (int k = 0; k < nk; ++k) {
        T a[2];
        for (int j = 0; j < 2; ++j) {
            const int64_t o = base + j * gf.n2 + k;
            a[j] = (xf[o] + xf[o + gf.pitch]) / T(2);
        }
        b[k] = (a[0] + a[1]) / T(2);
    }
    xc[e.off] = nk == 2 ? (b[0] + b[1]) / T(2) : b[0];
}

// prolongation + edge pad, mg.py:139-153, 231-239:  xf (+)= repeat(xc, 2)[clamped]
template <typename T, bool ADD>
__global__ void __launch_bounds__(256) k_mg_prolong(Geom gf, Geom gc, const T* __restrict__ xc, T* __restrict__ xf) {
    ElemIdx e;
    if (!elem_index(gf, e)) return;
    int64_t i0 = e.row >> 1, i1 = e.i1 >> 1, i2 = gf.dim == 3 ? (e.i2 >> 1) : e.i2;
    if (i0 > gc.n0 - 1) i0 = gc.n0 - 1;
    if (i1 > gc.n1 - 1) i1 = gc.n1 - 1;
    if (i2 > gc.n2 - 1) i2 = gc.n2 - 1;
    const T v = xc[i0 * gc.pitch + i1 * gc.n2 + i2];
    if constexpr (ADD)
        xf[e.off] = xf[e.off] + v;
    else
        xf[e.off] = v;
}

// CG vector updates
template <typename T>
__global__ void __launch_bounds__(256) k_axpy(Geom g, T a, const T* __restrict__ x, T* __restrict__ y) {  // y += a x
    ElemIdx e;
    if (!elem_index(g, e)) return;
    y[e.off] = y[e.off] + a * x[e.off];
}
template <typename T>
__global__ void __launch_bounds__(256) k_xpay(Geom g, const T* __restrict__ x, T a, T* __restrict__ y) {  // y = x + a y
    ElemIdx e;
    if (!elem_index(g, e)) return;
    y[e.off] = x[e.off] + a * y[e.off];
}
template <typename T>
__global__ void __launch_bounds__(256) k_sub(Geom g, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out) {
    ElemIdx e;
    if (!elem_index(g, e)) return;
    out[e.off] = a[e.off] - b[e.off];
}
template <typename T>
__global__ void __launch_bounds__(256) k_dot_partial(Geom g, const T* __restrict__ a, const T* __restrict__ b,
                                                      double* __restrict__ partial) {
    // grid-stride over rows; deterministic per-block partial sums in double
    double acc = 0.0;
    for (int64_t row = blockIdx.x; row < g.n0; row += gridDim.x)
        for (int64_t x = threadIdx.x; x < g.L; x += blockDim.x) {
            const int64_t o = row * g.pitch + x;
            acc += (double)a[o] * (double)b[o];
        }
    __shared__ double sm[256 / kWave];
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

struct Reducer {
    double* partial = nullptr;
    double* out = nullptr;
    int nb = 512;
    hipStream_t stream = nullptr;
    int init(hipStream_t s) {
        stream = s;
        DXH_HIP(hipMalloc((void**)&partial, sizeof(double) * nb));
        DXH_HIP(hipMalloc((void**)&out, sizeof(double) * 2));
        return DXH_OK;
    }
    void release() {
        if (partial) (void)hipFree(partial);
        if (out) (void)hipFree(out);
        partial = out = nullptr;
    }
    template <typename T>
    int dot(const Geom& g, const void* a, const void* b, double* host) {
        hipLaunchKernelGGL((k_dot_partial<T>), dim3(nb), dim3(256), 0, stream, g, (const T*)a, (const T*)b, partial);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, stream, partial, nb, out);
        DXH_HIP(hipGetLastError());
        DXH_HIP(hipMemcpyAsync(host, out, sizeof(double), hipMemcpyDefault, stream));
        DXH_HIP(hipStreamSynchronize(stream));
        return DXH_OK;
    }
    // sum (a - b)^2, b may be null
    template <typename T>
    int sqdiff(const Geom& g, const void* a, const void* b, double* host) {
        hipLaunchKernelGGL((k_sqdiff_partial<T>), dim3(nb), dim3(256), 0, stream, g, (const T*)a, (const T*)b, (int64_t)0,
                           g.n0, partial);
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, stream, partial, nb, out);
        DXH_HIP(hipGetLastError());
        DXH_HIP(hipMemcpyAsync(host, out, sizeof(double), hipMemcpyDefault, stream));
        DXH_HIP(hipStreamSynchronize(stream));
        return DXH_OK;
    }
};

size_t plane_bytes_of(const Geom& g, size_t esz) { return (size_t)g.n0 * (size_t)g.pitch * esz; }

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct MGSolver final : DeviceSolver {
    struct Level {
        Geom g{};
        size_t bytes = 0;
        double h = 1.0;
        void *xa = nullptr, *t0 = nullptr, *t1 = nullptr, *r = nullptr, *rs = nullptr;  // work planes
        void *x0 = nullptr, *rhs = nullptr, *sol = nullptr;                              // coarse levels only
        void *mass = nullptr, *diff = nullptr;                                           // heterogeneous: current arrays
        void *diag = nullptr, *dsc = nullptr, *scratch = nullptr;                        // heterogeneous: smoother planes
    };
    std::vector<Level> lv;
    int depth, smooth_it, maxiter;
    double tol;
    bool het;
    bool mass_is_arr = false, diff_is_arr = false;
    double mass_s = 0, diff_s = 0;
    bool have_params = false;
    T c_diag = 0, c_dsc = 0;  // scalar coefficients: the smoother's cached diagonal (first call: h = 1)
    hipStream_t stream;
    Reducer red;
    void* top[2] = {nullptr, nullptr};  // iterates of successive cycles

    MGSolver(int depth_, int smooth_, int maxiter_, double tol_, bool het_, hipStream_t s)
        : depth(depth_), smooth_it(smooth_), maxiter(maxiter_), tol(tol_), het(het_), stream(s) {}

    int alloc(void** p, size_t bytes) {
        DXH_HIP(hipMalloc(p, bytes));
        DXH_HIP(hipMemsetAsync(*p, 0, bytes, stream));
        return DXH_OK;
    }

    int init(const Geom& g0) {
        DXH_REQUIRE(g0.ndim == g0.dim, "MG: arrays must have exactly dim axes (the reference pads one pair per spatial axis, mg.py:231-239)");
        DXH_REQUIRE(depth >= 0 && smooth_it >= 1 && maxiter >= 0, "MG: depth >= 0, smoother_iterations >= 1, maxiter >= 0");
        int rc;
        if ((rc = red.init(stream))) return rc;
        lv.resize(depth + 2);
        Geom g = g0;
        for (int l = 0; l <= depth + 1; ++l) {
            Level& L = lv[l];
            if (l > 0) {
                const Geom& f = lv[l - 1].g;
                g.n0 = f.n0 / 2;
                g.n1 = f.n1 / 2;
                g.n2 = g0.dim == 3 ? f.n2 / 2 : f.n2;
                DXH_REQUIRE(g.n0 >= 1 && g.n1 >= 1 && g.n2 >= 1, "MG: level %d would be empty (shape too small for depth %d)", l, depth);
                g.L = g.n1 * g.n2;
                g.pitch = (g.L + 31) / 32 * 32;
            }
            L.g = g;
            L.h = std::ldexp(1.0, l);
            L.bytes = plane_bytes_of(g, sizeof(T));
            for (void** p : {&L.xa, &L.t0, &L.t1, &L.r, &L.rs})
                if ((rc = alloc(p, L.bytes))) return rc;
            if (l > 0)
                for (void** p : {&L.x0, &L.rhs, &L.sol})
                    if ((rc = alloc(p, L.bytes))) return rc;
            if (het)
                for (void** p : {&L.mass, &L.diff, &L.diag, &L.dsc, &L.scratch})
                    if ((rc = alloc(p, L.bytes))) return rc;
        }
        for (void** p : {&top[0], &top[1]})
            if ((rc = alloc(p, lv[0].bytes))) return rc;
        return DXH_OK;
    }

    ~MGSolver() override {
        for (Level& L : lv)
            for (void* p : {L.xa, L.t0, L.t1, L.r, L.rs, L.x0, L.rhs, L.sol, L.mass, L.diff, L.diag, L.dsc, L.scratch})
                if (p) (void)hipFree(p);
        for (void* p : top)
            if (p) (void)hipFree(p);
        red.release();
    }

    // smoother planes of level l from its current coefficient arrays, for grid spacing h (jacobi.py:57-72,101-106)
    int build_coef(int l, double h) {
        Level& L = lv[l];
        CoefIn<T> in{};
        in.mu = nullptr;
        in.omega = mass_is_arr ? (const T*)L.mass : nullptr;
        in.ell = diff_is_arr ? (const T*)L.diff : nullptr;
        in.mu_s = T(0);
        in.omega_s = (T)mass_s;
        in.ell_s = (T)diff_s;
        const double h2 = h * h;
        in.two_dim_over_h2_ell_s = (T)(diff_s * 2 * L.g.dim / h2);
        in.ell_over_h2_s = (T)(diff_s / h2);
        in.diag_s = (T)(mass_s + diff_s * 2 * L.g.dim / h2);
        in.dsc_s = (T)((mass_s + diff_s * 2 * L.g.dim / h2) / (diff_s / h2));
        in.kq_s = T(0);   // (no threshold in a linear solve)
        in.h2 = (T)h2;
        in.two_dim = (T)(2 * L.g.dim);
        in.dim = L.g.dim;
        in.has_ell = 1;
        hipLaunchKernelGGL((k_coef_planes<T>), grid_of(L.g), dim3(256), 0, stream, L.g, in, (T*)L.scratch, (T*)L.diag,
                           (T*)L.dsc, (T*)nullptr);
        ++launches;
        DXH_HIP(hipGetLastError());
        return DXH_OK;
    }

    int set_params(double mass, const void* mass_plane, double diffusion, const void* diffusion_plane) override {
        mass_is_arr = mass_plane != nullptr;
        diff_is_arr = diffusion_plane != nullptr;
        DXH_REQUIRE(het == (mass_is_arr || diff_is_arr),
                    het ? "MG: created as heterogeneous but no coefficient array was given"
                        : "MG: coefficient arrays need a solver created with array coefficients (the reference decides "
                          "'heterogeneous' in the constructor, mg.py:72-75, and fails to broadcast otherwise)");
        mass_s = mass;
        diff_s = diffusion;
        if (!het) {
            if (!have_params) {
                // the reference's single smoother caches its diagonal at its first call, i.e. for h = 1, and keeps
                // it for every level and every later update_params (jacobi.py:101-106): same here
                const double diag = mass + diffusion * 2 * lv[0].g.dim / 1.0;
                c_diag = (T)diag;
                c_dsc = (T)(diag / (diffusion / 1.0));
            }
        } else {
            if (mass_is_arr) DXH_HIP(hipMemcpyAsync(lv[0].mass, mass_plane, lv[0].bytes, hipMemcpyDeviceToDevice, stream));
            if (diff_is_arr) DXH_HIP(hipMemcpyAsync(lv[0].diff, diffusion_plane, lv[0].bytes, hipMemcpyDeviceToDevice, stream));
            int rc = build_coef(0, lv[0].h);
            if (rc) return rc;
        }
        have_params = true;
        return DXH_OK;
    }

    int get_params(void* mass_plane, void* diffusion_plane) override {
        DXH_REQUIRE(het, "MG: only a heterogeneous solver holds coefficient arrays");
        DXH_REQUIRE((!mass_plane || mass_is_arr) && (!diffusion_plane || diff_is_arr), "MG: that coefficient is a scalar");
        if (mass_plane) DXH_HIP(hipMemcpyAsync(mass_plane, lv[0].mass, lv[0].bytes, hipMemcpyDeviceToDevice, stream));
        if (diffusion_plane) DXH_HIP(hipMemcpyAsync(diffusion_plane, lv[0].diff, lv[0].bytes, hipMemcpyDeviceToDevice, stream));
        return DXH_OK;
    }

    // `smooth_it` Jacobi sweeps from `in` into `out` (out != in, neither is t0 / t1 / rs of the level)
    int smooth(int l, const void* in, const void* rhs, void* out) {
        Level& L = lv[l];
        const dim3 grid = grid_of(L.g), blk(256);
        hipLaunchKernelGGL((k_divide<T>), grid, blk, 0, stream, L.g, (const T*)rhs, c_diag, het ? (const T*)L.diag : nullptr,
                           (T*)L.rs);
        ++launches;
        Coef<T> co{};
        co.dsc = c_dsc;
        CoefPlanes<T> cp{};
        cp.dsc = het ? (const T*)L.dsc : nullptr;
        const void* src = in;
        int sweeps = smooth_it;
        if (!het && L.g.dim == 2 && L.g.n2 == 1 && sweeps >= 2 && !getenv("DXH_MG_NO_MULTI")) {
            // several sweeps per pass (k_jacobi_multi, bit-identical to sweep by sweep); a remainder of one sweep and
            // the final placement in `out` are handled by the loop below
            int left = sweeps;
            int flip = 0;
            while (left >= 2) {
                const int k = left == 5 ? 3 : std::min(4, left);
                void* dst = (left - k == 0) ? out : (flip ? L.t1 : L.t0);
                JacobiMultiArgs<T> ja{};
                ja.rs = (const T*)L.rs; ja.x_in = (const T*)src; ja.x_out = (T*)dst;
                ja.n0 = L.g.n0; ja.L = L.g.L; ja.pitch = L.g.pitch; ja.dsc = c_dsc;
                // reciprocal division (five operations, correctly rounded); the exhaustive three-operation check
                // belongs to the TVD context and is not consulted for the smoother
                ja.rdsc = (std::isfinite((double)c_dsc) && std::fabs((double)c_dsc) >= 0x1p-20 && std::fabs((double)c_dsc) <= 0x1p+20) ? T(1) / c_dsc : T(0);
                ja.div3 = 0;
#define DXH_MGJ(K_)                                                                                                        \
    {                                                                                                                      \
        using SG = StripGeomK<T, 1, K_>;                                                                                  \
        ja.nstrips = (int)((L.g.L + SG::STRIDE - 1) / SG::STRIDE);                                                        \
        const int nbx = (ja.nstrips + 3) / 4;                                                                             \
        int64_t nchunks = std::max<int64_t>(1, (int64_t)1024 / nbx);                                                      \
        if (nchunks >= 8) nchunks = nchunks / 8 * 8;                                                                      \
        const int64_t rpc = std::max<int64_t>(8 * K_, (L.g.n0 + nchunks - 1) / nchunks);                                  \
        nchunks = (L.g.n0 + rpc - 1) / rpc;                                                                               \
        ja.rows_per_chunk = rpc;                                                                                          \
        ja.nchunks = (int)nchunks;                                                                                        \
        hipLaunchKernelGGL((k_jacobi_multi<T, 1, false, K_>), dim3((unsigned)(nbx * nchunks)), dim3(256), 0, stream, ja); \
    }
                if (k == 2) DXH_MGJ(2) else if (k == 3) DXH_MGJ(3) else DXH_MGJ(4)
#undef DXH_MGJ
                ++launches;
                src = dst;
                flip ^= 1;
                left -= k;
            }
            sweeps = left;  // 0 or 1
            if (sweeps == 1) {
                hipLaunchKernelGGL((k_jacobi<T>), grid, blk, 0, stream, L.g, (const T*)L.rs, (const T*)src, co, cp, (T*)out);
                ++launches;
            }
            DXH_HIP(hipGetLastError());
            return DXH_OK;
        }
        for (int s = 0; s < smooth_it; ++s) {
            void* dst = (s == smooth_it - 1) ? out : (s & 1 ? L.t1 : L.t0);
            hipLaunchKernelGGL((k_jacobi<T>), grid, blk, 0, stream, L.g, (const T*)L.rs, (const T*)src, co, cp, (T*)dst);
            ++launches;
            src = dst;
        }
        DXH_HIP(hipGetLastError());
        return DXH_OK;
    }

    int restrict_plane(int l, const void* fine, void* coarse) {
        hipLaunchKernelGGL((k_mg_restrict<T>), grid_of(lv[l + 1].g), dim3(256), 0, stream, lv[l].g, lv[l + 1].g, (const T*)fine,
                           (T*)coarse);
        ++launches;
        DXH_HIP(hipGetLastError());
        return DXH_OK;
    }

    // mg.py:195-247
    int v_cycle(int l, const void* x_in, const void* rhs, void* x_out, int depth_left) {
        Level& L = lv[l];
        Level& C = lv[l + 1];
        int rc;
        if ((rc = smooth(l, x_in, rhs, L.xa))) return rc;
        hipLaunchKernelGGL((k_mg_residual<T>), grid_of(L.g), dim3(256), 0, stream, L.g, (const T*)L.xa, (const T*)rhs, (T)mass_s,
                           (het && mass_is_arr) ? (const T*)L.mass : nullptr, (T)diff_s,
                           (het && diff_is_arr) ? (const T*)L.diff : nullptr, (T)L.h, (T*)L.r);
        ++launches;
        if ((rc = restrict_plane(l, L.r, C.rhs))) return rc;
        if (het) {
            if (mass_is_arr && (rc = restrict_plane(l, L.mass, C.mass))) return rc;
            if (diff_is_arr && (rc = restrict_plane(l, L.diff, C.diff))) return rc;
            if ((rc = build_coef(l + 1, C.h))) return rc;  // fresh smoother, first called with the coarse h
        }
        DXH_HIP(hipMemsetAsync(C.x0, 0, C.bytes, stream));
        if (depth_left == 0) {
            if ((rc = smooth(l + 1, C.x0, C.rhs, C.sol))) return rc;
        } else {
            if ((rc = v_cycle(l + 1, C.x0, C.rhs, C.sol, depth_left - 1))) return rc;
        }
        hipLaunchKernelGGL((k_mg_prolong<T, true>), grid_of(L.g), dim3(256), 0, stream, L.g, C.g, (const T*)C.sol, (T*)L.xa);
        ++launches;
        if (het) {
            // the reference REPLACES the level's coefficients by the padded prolongation of the coarse ones
            if (mass_is_arr) {
                hipLaunchKernelGGL((k_mg_prolong<T, false>), grid_of(L.g), dim3(256), 0, stream, L.g, C.g, (const T*)C.mass, (T*)L.mass);
                ++launches;
            }
            if (diff_is_arr) {
                hipLaunchKernelGGL((k_mg_prolong<T, false>), grid_of(L.g), dim3(256), 0, stream, L.g, C.g, (const T*)C.diff, (T*)L.diff);
                ++launches;
            }
            if ((rc = build_coef(l, L.h))) return rc;
        }
        DXH_HIP(hipGetLastError());
        return smooth(l, L.xa, rhs, x_out);
    }

    // mg.py:249-270
    int solve(const void* x0, const void* rhs, void* out, int* iters) override {
        DXH_REQUIRE(have_params, "MG: set_params first");
        const Geom& g = lv[0].g;
        const bool use_tol = tol >= 0.0 && !std::isnan(tol);
        double nrm0 = 0.0;
        int rc;
        if (use_tol) {
            if ((rc = red.template sqdiff<T>(g, x0, nullptr, &nrm0))) return rc;
            nrm0 = std::sqrt(nrm0);
        }
        const void* x = x0;
        int w = 0, done = 0;
        for (int i = 0; i < maxiter; ++i) {
            void* xn = top[w];
            if ((rc = v_cycle(0, x, rhs, xn, depth))) return rc;
            if (use_tol) {
                double inc = 0.0;
                if ((rc = red.template sqdiff<T>(g, xn, x, &inc))) return rc;
                if (std::sqrt(inc) / nrm0 < tol) break;  // the reference keeps the previous iterate
            }
            x = xn;
            w ^= 1;
            ++done;
        }
        DXH_HIP(hipMemcpyAsync(out, x, lv[0].bytes, hipMemcpyDeviceToDevice, stream));
        if (iters) *iters = done;
        return DXH_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct CGSolver final : DeviceSolver {
    Geom g{};
    size_t bytes = 0;
    int maxiter;
    double tol, h;
    hipStream_t stream;
    Reducer red;
    void *r = nullptr, *p = nullptr, *ap = nullptr, *xw = nullptr, *mass = nullptr, *diff = nullptr;
    bool mass_is_arr = false, diff_is_arr = false, have_params = false;
    double mass_s = 0, diff_s = 0;
    double last_residual = 0;

    CGSolver(int maxiter_, double tol_, double h_, hipStream_t s) : maxiter(maxiter_), tol(tol_), h(h_), stream(s) {}

    int init(const Geom& g0) {
        DXH_REQUIRE(g0.ndim == g0.dim, "CG: arrays must have exactly dim axes (cg.py:58 reshapes to x0.shape[:dim])");
        DXH_REQUIRE(maxiter >= 0, "CG: maxiter >= 0");
        DXH_REQUIRE(tol >= 0.0 && !std::isnan(tol), "CG: a tolerance is required (scipy.sparse.linalg.cg(tol=None) fails)");
        g = g0;
        bytes = plane_bytes_of(g, sizeof(T));
        int rc;
        if ((rc = red.init(stream))) return rc;
        for (void** q : {&r, &p, &ap, &xw, &mass, &diff}) {
            DXH_HIP(hipMalloc(q, bytes));
            DXH_HIP(hipMemsetAsync(*q, 0, bytes, stream));
        }
        return DXH_OK;
    }
    ~CGSolver() override {
        for (void* q : {r, p, ap, xw, mass, diff})
            if (q) (void)hipFree(q);
        red.release();
    }
    int set_params(double mass_v, const void* mass_plane, double diffusion, const void* diffusion_plane) override {
        mass_is_arr = mass_plane != nullptr;
        diff_is_arr = diffusion_plane != nullptr;
        mass_s = mass_v;
        diff_s = diffusion;
        if (mass_is_arr) DXH_HIP(hipMemcpyAsync(mass, mass_plane, bytes, hipMemcpyDeviceToDevice, stream));
        if (diff_is_arr) DXH_HIP(hipMemcpyAsync(diff, diffusion_plane, bytes, hipMemcpyDeviceToDevice, stream));
        have_params = true;
        return DXH_OK;
    }
    int get_params(void* mass_plane, void* diffusion_plane) override {
        DXH_REQUIRE((!mass_plane || mass_is_arr) && (!diffusion_plane || diff_is_arr), "CG: that coefficient is a scalar");
        if (mass_plane) DXH_HIP(hipMemcpyAsync(mass_plane, mass, bytes, hipMemcpyDeviceToDevice, stream));
        if (diffusion_plane) DXH_HIP(hipMemcpyAsync(diffusion_plane, diff, bytes, hipMemcpyDeviceToDevice, stream));
        return DXH_OK;
    }
    void apply(const void* x, void* out) {
        hipLaunchKernelGGL((k_cg_apply<T>), grid_of(g), dim3(256), 0, stream, g, (const T*)x, (T)mass_s,
                           mass_is_arr ? (const T*)mass : nullptr, (T)diff_s, diff_is_arr ? (const T*)diff : nullptr,
                           (T)(h > 0 ? h : 1.0), h > 0 ? 1 : 0, (T*)out);
        ++launches;
    }
    // textbook CG with SciPy's legacy stop rule |r| <= tol |b| (scipy < 1.14, the only versions cg.py runs on)
    int solve(const void* x0, const void* rhs, void* out, int* iters) override {
        DXH_REQUIRE(have_params, "CG: set_params first");
        const dim3 grid = grid_of(g), blk(256);
        int rc;
        double bb = 0.0, rr = 0.0;
        if ((rc = red.template sqdiff<T>(g, rhs, nullptr, &bb))) return rc;
        const double atol = tol * std::sqrt(bb);
        DXH_HIP(hipMemcpyAsync(xw, x0, bytes, hipMemcpyDeviceToDevice, stream));
        apply(xw, ap);
        hipLaunchKernelGGL((k_sub<T>), grid, blk, 0, stream, g, (const T*)rhs, (const T*)ap, (T*)r);
        DXH_HIP(hipMemcpyAsync(p, r, bytes, hipMemcpyDeviceToDevice, stream));
        if ((rc = red.template dot<T>(g, r, r, &rr))) return rc;
        int done = 0;
        for (int i = 0; i < maxiter; ++i) {
            if (std::sqrt(rr) <= atol) break;
            apply(p, ap);
            double pap = 0.0;
            if ((rc = red.template dot<T>(g, p, ap, &pap))) return rc;
            if (!(pap > 0.0)) break;  // breakdown: the operator is not positive definite for these coefficients
            const double alpha = rr / pap;
            hipLaunchKernelGGL((k_axpy<T>), grid, blk, 0, stream, g, (T)alpha, (const T*)p, (T*)xw);
            hipLaunchKernelGGL((k_axpy<T>), grid, blk, 0, stream, g, (T)(-alpha), (const T*)ap, (T*)r);
            double rr_new = 0.0;
            if ((rc = red.template dot<T>(g, r, r, &rr_new))) return rc;
            hipLaunchKernelGGL((k_xpay<T>), grid, blk, 0, stream, g, (const T*)r, (T)(rr_new / rr), (T*)p);
            launches += 3;
            rr = rr_new;
            ++done;
        }
        DXH_HIP(hipGetLastError());
        last_residual = std::sqrt(rr);
        DXH_HIP(hipMemcpyAsync(out, xw, bytes, hipMemcpyDeviceToDevice, stream));
        if (iters) *iters = done;
        return DXH_OK;
    }
};

}  // namespace

DeviceSolver* make_mg_solver(int dtype, const Geom& g0, int depth, int smoother_iterations, int maxiter, double tol,
                             bool heterogeneous, hipStream_t stream, int* rc) {
    *rc = DXH_OK;
    if (dtype == DXH_F32) {
        auto* s = new (std::nothrow) MGSolver<float>(depth, smoother_iterations, maxiter, tol, heterogeneous, stream);
        if (!s) { set_error("out of host memory"); *rc = DXH_ENOMEM; return nullptr; }
        if ((*rc = s->init(g0))) { delete s; return nullptr; }
        return s;
    }
    auto* s = new (std::nothrow) MGSolver<double>(depth, smoother_iterations, maxiter, tol, heterogeneous, stream);
    if (!s) { set_error("out of host memory"); *rc = DXH_ENOMEM; return nullptr; }
    if ((*rc = s->init(g0))) { delete s; return nullptr; }
    return s;
}

DeviceSolver* make_cg_solver(int dtype, const Geom& g0, int maxiter, double tol, double h, hipStream_t stream, int* rc) {
    *rc = DXH_OK;
    if (dtype == DXH_F32) {
        auto* s = new (std::nothrow) CGSolver<float>(maxiter, tol, h, stream);
        if (!s) { set_error("out of host memory"); *rc = DXH_ENOMEM; return nullptr; }
        if ((*rc = s->init(g0))) { delete s; return nullptr; }
        return s;
    }
    auto* s = new (std::nothrow) CGSolver<double>(maxiter, tol, h, stream);
    if (!s) { set_error("out of host memory"); *rc = DXH_ENOMEM; return nullptr; }
    if ((*rc = s->init(g0))) { delete s; return nullptr; }
    return s;
}

}  // namespace dxh

// ---- C ABI: a solver object over dense host arrays (include/darsia_hip.h) -------------------------------------
namespace {

struct SolverObj {
    dxh_solver_desc desc{};
    dxh::Geom g{};
    size_t esz = 4, bytes = 0;
    hipStream_t stream = nullptr;
    dxh::DeviceSolver* impl = nullptr;
    void *x0 = nullptr, *rhs = nullptr, *out = nullptr, *m = nullptr, *d = nullptr;
};

int up(SolverObj* o, void* dst, const void* src) {
    DXH_HIP(hipMemcpy2DAsync(dst, o->g.pitch * o->esz, src, o->g.L * o->esz, o->g.L * o->esz, o->g.n0,
                             hipMemcpyDefault, o->stream));
    return DXH_OK;
}
int down(SolverObj* o, void* dst, const void* src) {
    DXH_HIP(hipMemcpy2DAsync(dst, o->g.L * o->esz, src, o->g.pitch * o->esz, o->g.L * o->esz, o->g.n0,
                             hipMemcpyDefault, o->stream));
    return DXH_OK;
}

}  // namespace

extern "C" {

int dxh_solver_destroy(dxh_solver_t h) {
    auto* o = reinterpret_cast<SolverObj*>(h);
    if (!o) return DXH_OK;
    (void)hipSetDevice(o->desc.device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    delete o->impl;
    for (void* p : {o->x0, o->rhs, o->out, o->m, o->d})
        if (p) (void)hipFree(p);
    delete o;
    return DXH_OK;
}

int dxh_solver_create(dxh_solver_t* out, const dxh_solver_desc* d) {
    using namespace dxh;
    DXH_REQUIRE(out && d, "null argument");
    DXH_REQUIRE(d->struct_size == (int32_t)sizeof(dxh_solver_desc), "dxh_solver_desc size mismatch (%d != %zu)", d->struct_size,
                sizeof(dxh_solver_desc));
    DXH_REQUIRE(d->dtype == DXH_F32 || d->dtype == DXH_F64, "dtype must be DXH_F32 or DXH_F64");
    DXH_REQUIRE(d->dim == 2 || d->dim == 3, "dim must be 2 or 3");
    DXH_REQUIRE(d->n0 >= 1 && d->n1 >= 1 && d->n2 >= 1 && (d->dim == 3 || d->n2 == 1), "bad shape for dim=%d", d->dim);
    DXH_REQUIRE(d->kind == DXH_SOLVER_MG || d->kind == DXH_SOLVER_CG, "kind must be DXH_SOLVER_MG or DXH_SOLVER_CG");
    DXH_HIP(hipSetDevice(d->device));
    auto* o = new (std::nothrow) SolverObj();
    DXH_REQUIRE(o, "out of host memory");
    o->desc = *d;
    o->esz = d->dtype == DXH_F32 ? 4 : 8;
    o->stream = (hipStream_t)d->stream;
    Geom g{};
    g.n0 = d->n0; g.n1 = d->n1; g.n2 = d->n2;
    g.L = g.n1 * g.n2;
    g.pitch = (g.L + 31) / 32 * 32;
    g.ndim = d->dim; g.dim = d->dim; g.nsum_axes = d->dim;
    o->g = g;
    o->bytes = (size_t)g.n0 * g.pitch * o->esz;
    auto fail = [&](int rc) { dxh_solver_destroy(reinterpret_cast<dxh_solver_t>(o)); return rc; };
    for (void** p : {&o->x0, &o->rhs, &o->out, &o->m, &o->d}) {
        if (hipMalloc(p, o->bytes) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", o->bytes); return fail(DXH_ENOMEM); }
        if (hipMemsetAsync(*p, 0, o->bytes, o->stream) != hipSuccess) { set_error("hipMemsetAsync failed"); return fail(DXH_EHIP); }
    }
    int rc = DXH_OK;
    if (d->kind == DXH_SOLVER_MG)
        o->impl = make_mg_solver(d->dtype, g, d->depth, d->smoother_iterations, d->maxiter, d->tol, d->heterogeneous != 0,
                                 o->stream, &rc);
    else
        o->impl = make_cg_solver(d->dtype, g, d->maxiter, d->tol, d->h, o->stream, &rc);
    if (!o->impl) return fail(rc ? rc : DXH_EHIP);
    *out = reinterpret_cast<dxh_solver_t>(o);
    return DXH_OK;
}

int dxh_solver_set_params(dxh_solver_t h, double mass, const void* mass_arr, double diffusion, const void* diffusion_arr) {
    auto* o = reinterpret_cast<SolverObj*>(h);
    DXH_REQUIRE(o, "null solver");
    DXH_HIP(hipSetDevice(o->desc.device));
    int rc;
    if (mass_arr && (rc = up(o, o->m, mass_arr))) return rc;
    if (diffusion_arr && (rc = up(o, o->d, diffusion_arr))) return rc;
    rc = o->impl->set_params(mass, mass_arr ? o->m : nullptr, diffusion, diffusion_arr ? o->d : nullptr);
    if (rc) return rc;
    DXH_HIP(hipStreamSynchronize(o->stream));
    return DXH_OK;
}

int dxh_solver_solve(dxh_solver_t h, const void* x0, const void* rhs, void* out, int32_t* iters) {
    auto* o = reinterpret_cast<SolverObj*>(h);
    DXH_REQUIRE(o && x0 && rhs && out, "null argument");
    DXH_HIP(hipSetDevice(o->desc.device));
    int rc, n = 0;
    dxh::HostPrefault pf;  // map the output pages while the device works
    {
        const size_t nb = (size_t)o->g.n0 * (size_t)o->g.L * o->esz;
        pf.start(out, nb, {{x0, nb}, {rhs, nb}});
    }
    if ((rc = up(o, o->x0, x0)) || (rc = up(o, o->rhs, rhs))) return rc;
    if ((rc = o->impl->solve(o->x0, o->rhs, o->out, &n))) return rc;
    pf.join();
    if ((rc = down(o, out, o->out))) return rc;
    DXH_HIP(hipStreamSynchronize(o->stream));
    if (iters) *iters = n;
    return DXH_OK;
}

int dxh_solver_get_params(dxh_solver_t h, void* mass_out, void* diffusion_out) {
    auto* o = reinterpret_cast<SolverObj*>(h);
    DXH_REQUIRE(o, "null solver");
    DXH_HIP(hipSetDevice(o->desc.device));
    int rc = o->impl->get_params(mass_out ? o->m : nullptr, diffusion_out ? o->d : nullptr);
    if (rc) return rc;
    if (mass_out && (rc = down(o, mass_out, o->m))) return rc;
    if (diffusion_out && (rc = down(o, diffusion_out, o->d))) return rc;
    DXH_HIP(hipStreamSynchronize(o->stream));
    return DXH_OK;
}

}  // extern "C"
