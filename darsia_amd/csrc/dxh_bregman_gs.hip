// Split-Bregman TV denoising with scikit-image's in-place Gauss-Seidel u-update: the arithmetic behind DarSIA's
// TVD(method = "anisotropic bregman" | "isotropic bregman") (restoration/tvd.py:85-101) and the "tvd" smoothing of
// utils/segmentation.py:117-120.  oracle/bregman_gs_numpy.py states the algorithm and how it was pinned.
//
// The raster sweep has a sequential dependence (u[r-1,c], u[r,c-1], d and b of those cells are the NEW values), but
// cells on one anti-diagonal r + c = const are independent, so a sweep is rows + cols - 1 wavefront launches.  To
// keep those launches coalesced all state lives in a SKEWED layout: element (r, c) of the framed (rows+2, cols+2)
// array is stored at [k][d = r + c][r]; the five neighbours of a cell are then at [d-1][r-1], [d-1][r], [d+1][r],
// [d+1][r+1] -- contiguous in r for a whole wavefront.  Channels are independent planes (the reference sweeps them
// one after the other).  The stop test sqrt(sum (u_new-u_old)^2 / size) > eps is evaluated once per sweep from a
// plane of increments (fixed reduction tree, double).  gfx950 only.
#include <cmath>

#include "dxh_kernels.h"

namespace dxh {

namespace {

template <typename T>
struct GsArgs {
    T* u;        // [dims][nd][rs] skewed, framed
    T* dx;
    T* dy;
    T* bx;
    T* by;
    T* inc;      // u_new - u_old of the last sweep (0 outside the image)
    const T* f;  // image, skewed like u (frame unused)
    int64_t rows, cols, dims;
    int64_t rs;     // positions per diagonal (>= rows + 2, multiple of 32)
    int64_t plane;  // nd * rs
    T weight, lam, norm;
    double th;      // 1.0 / lam evaluated in double (the float32 build of the reference does the same)
};

// dense (rows, cols, dims) image -> skewed u and f, with the reference's one-time frame:
// top = image[1], left = image[:, 1], bottom = image[-1], right = image[:, -1], corners 0
template <typename T>
__global__ void __launch_bounds__(256) k_gs_init(GsArgs<T> a, const T* __restrict__ img) {
    const int64_t n = (a.rows + 2) * (a.cols + 2) * a.dims;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i % a.dims, rc = i / a.dims;
        const int64_t c = rc % (a.cols + 2), r = rc / (a.cols + 2);
        int64_t sr = r - 1, sc = c - 1;  // source pixel
        const bool fr = r == 0 || r == a.rows + 1, fc = c == 0 || c == a.cols + 1;
        T v = T(0);
        bool have = true;
        if (fr && fc) {
            have = false;
        } else if (r == 0) {
            if (a.rows > 1) sr = 1; else have = false;
        } else if (r == a.rows + 1) {
            sr = a.rows - 1;
        } else if (c == 0) {
            if (a.cols > 1) sc = 1; else have = false;
        } else if (c == a.cols + 1) {
            sc = a.cols - 1;
        }
        if (have) v = img[(sr * a.cols + sc) * a.dims + k];
        const int64_t o = k * a.plane + (r + c) * a.rs + r;
        a.u[o] = v;
        if (!fr && !fc) const_cast<T*>(a.f)[o] = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_gs_final(GsArgs<T> a, T* __restrict__ out) {
    const int64_t n = a.rows * a.cols * a.dims;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i % a.dims, rc = i / a.dims;
        const int64_t c = rc % a.cols + 1, r = rc / a.cols + 1;
        out[i] = a.u[k * a.plane + (r + c) * a.rs + r];
    }
}

// one wavefront: all cells with r + c = d, r in [r_lo, r_hi], every channel
template <typename T, bool ISO>
__global__ void __launch_bounds__(256) k_gs_diag(GsArgs<T> a, int64_t d, int64_t r_lo, int64_t r_hi) {
    const int64_t r = r_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > r_hi) return;
    const int64_t k = blockIdx.y;
    const int64_t o = k * a.plane + d * a.rs + r;  // (r, c)
    const int64_t om = o - a.rs;                   // (r, c-1);  om - 1 = (r-1, c)
    const int64_t op = o + a.rs;                   // (r, c+1);  op + 1 = (r+1, c)
    const T uprev = a.u[o];
    const T u_cp = a.u[op], u_rp = a.u[op + 1];
    const T ux = u_cp - uprev;
    const T uy = u_rp - uprev;
    const T bxx = a.bx[o], byy = a.by[o];
    T acc = u_rp + a.u[om - 1];
    acc = acc + u_cp;
    acc = acc + a.u[om];
    acc = acc + a.dx[om];
    acc = acc - a.dx[o];
    acc = acc + a.dy[om - 1];
    acc = acc - a.dy[o];
    acc = acc - a.bx[om];
    acc = acc + bxx;
    acc = acc - a.by[om - 1];
    acc = acc + byy;
    const T unew = (a.lam * acc + a.weight * a.f[o]) / a.norm;
    a.u[o] = unew;
    a.inc[o] = unew - uprev;
    T dxx, dyy;
    if constexpr (ISO) {
        const T tx = ux + bxx, ty = uy + byy;
        const T s = (T)sqrt((double)(tx * tx + ty * ty));
        const T sl = s * a.lam;
        const double den = (double)sl + 1.0;
        dxx = (T)((double)(sl * tx) / den);
        dyy = (T)((double)(sl * ty) / den);
    } else {
        const double sx = (double)(ux + bxx), sy = (double)(uy + byy);
        dxx = sx > a.th ? (T)(sx - a.th) : (sx < -a.th ? (T)(sx + a.th) : T(0));
        dyy = sy > a.th ? (T)(sy - a.th) : (sy < -a.th ? (T)(sy + a.th) : T(0));
    }
    a.dx[o] = dxx;
    a.dy[o] = dyy;
    a.bx[o] = bxx + (ux - dxx);
    a.by[o] = byy + (uy - dyy);
}

template <typename T>
int run_gs(int64_t rows, int64_t cols, int64_t dims, const void* image, void* out, double weight, int max_iter,
           double eps, int isotropic, int* iters, double* rmse_out) {
    GsArgs<T> a{};
    a.rows = rows; a.cols = cols; a.dims = dims;
    a.rs = (rows + 2 + 31) / 32 * 32;
    const int64_t nd = rows + cols + 3;  // diagonals 0 .. rows + cols + 2
    a.plane = nd * a.rs;
    a.weight = (T)weight;
    a.lam = T(2) * a.weight;
    a.norm = a.weight + T(4) * a.lam;
    a.th = 1.0 / (double)a.lam;
    const size_t bytes = (size_t)a.plane * dims * sizeof(T);
    const size_t dense = (size_t)rows * cols * dims * sizeof(T);
    void* bufs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void *d_img = nullptr, *d_out = nullptr;
    double *partial = nullptr, *red = nullptr;
    const int nb = 1024;
    auto cleanup = [&]() {
        for (void* p : bufs) if (p) (void)hipFree(p);
        if (d_img) (void)hipFree(d_img);
        if (d_out) (void)hipFree(d_out);
        if (partial) (void)hipFree(partial);
        if (red) (void)hipFree(red);
    };
    auto body = [&]() -> int {
        for (void*& p : bufs) {
            DXH_HIP(hipMalloc(&p, bytes));
            DXH_HIP(hipMemsetAsync(p, 0, bytes, nullptr));
        }
        DXH_HIP(hipMalloc(&d_img, dense));
        DXH_HIP(hipMalloc(&d_out, dense));
        DXH_HIP(hipMalloc((void**)&partial, sizeof(double) * nb));
        DXH_HIP(hipMalloc((void**)&red, sizeof(double) * 2));
        DXH_HIP(hipMemcpyAsync(d_img, image, dense, hipMemcpyHostToDevice, nullptr));
        a.u = (T*)bufs[0]; a.dx = (T*)bufs[1]; a.dy = (T*)bufs[2]; a.bx = (T*)bufs[3]; a.by = (T*)bufs[4];
        a.inc = (T*)bufs[5]; a.f = (const T*)bufs[6];
        hipLaunchKernelGGL((k_gs_init<T>), dim3(2048), dim3(256), 0, nullptr, a, (const T*)d_img);
        DXH_HIP(hipGetLastError());
        // the increment plane as a (dims * nd, rs) array for the reduction
        Geom gi{};
        gi.n0 = dims * nd; gi.n1 = a.rs; gi.n2 = 1; gi.L = a.rs; gi.pitch = a.rs; gi.ndim = 2; gi.dim = 2; gi.nsum_axes = 2;
        const double total = (double)rows * (double)cols * (double)dims;
        int i = 0;
        double rmse = INFINITY;
        while (i < max_iter && rmse > eps) {
            for (int64_t d = 2; d <= rows + cols; ++d) {
                const int64_t r_lo = d - cols > 1 ? d - cols : 1, r_hi = d - 1 < rows ? d - 1 : rows;
                const dim3 grid((unsigned)((r_hi - r_lo + 1 + 255) / 256), (unsigned)dims);
                if (isotropic)
                    hipLaunchKernelGGL((k_gs_diag<T, true>), grid, dim3(256), 0, nullptr, a, d, r_lo, r_hi);
                else
                    hipLaunchKernelGGL((k_gs_diag<T, false>), grid, dim3(256), 0, nullptr, a, d, r_lo, r_hi);
            }
            hipLaunchKernelGGL((k_sqdiff_partial<T>), dim3(nb), dim3(256), 0, nullptr, gi, (const T*)a.inc, (const T*)nullptr,
                               (int64_t)0, gi.n0, partial);
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, nullptr, partial, nb, red);
            DXH_HIP(hipGetLastError());
            double sq = 0.0;
            DXH_HIP(hipMemcpy(&sq, red, sizeof(double), hipMemcpyDeviceToHost));
            rmse = std::sqrt(sq / total);
            ++i;
        }
        hipLaunchKernelGGL((k_gs_final<T>), dim3(2048), dim3(256), 0, nullptr, a, (T*)d_out);
        DXH_HIP(hipGetLastError());
        DXH_HIP(hipMemcpy(out, d_out, dense, hipMemcpyDeviceToHost));
        if (iters) *iters = i;
        if (rmse_out) *rmse_out = rmse;
        return DXH_OK;
    };
    const int rc = body();
    (void)hipDeviceSynchronize();
    cleanup();
    return rc;
}

}  // namespace
}  // namespace dxh

extern "C" int dxh_tv_bregman_gs(int device, int dtype, int64_t rows, int64_t cols, int64_t dims, const void* image,
                                 void* out, double weight, int max_iter, double eps, int isotropic, int* iters,
                                 double* rmse) {
    using namespace dxh;
    DXH_REQUIRE(image && out, "null argument");
    DXH_REQUIRE(dtype == DXH_F32 || dtype == DXH_F64, "dtype must be DXH_F32 or DXH_F64");
    DXH_REQUIRE(rows >= 1 && cols >= 1 && dims >= 1, "empty image");
    DXH_REQUIRE(max_iter >= 0, "max_iter must be >= 0");
    DXH_REQUIRE(dims <= 65535, "too many channels");
    DXH_HIP(hipSetDevice(device));
    return dtype == DXH_F32 ? run_gs<float>(rows, cols, dims, image, out, weight, max_iter, eps, isotropic, iters, rmse)
                            : run_gs<double>(rows, cols, dims, image, out, weight, max_iter, eps, isotropic, iters, rmse);
}
