// Split-Bregman TV denoising with scikit-image's in-place Gauss-Seidel u-update: the arithmetic behind DarSIA's
// TVD(method = "anisotropic bregman" | "isotropic bregman") (restoration/tvd.py:85-101) and the "tvd" smoothing of
// utils/segmentation.py:117-120.  oracle/bregman_gs_numpy.py states the algorithm and how it was pinned.
//
// The raster sweep has a sequential dependence (u[r-1,c], u[r,c-1], d and b of those cells are the NEW values), but
// cells on one anti-diagonal r + c = const are independent, so a sweep is rows + cols - 1 wavefront launches.  To
// keep those launches coalesced all state lives in a SKEWED layout: element (r, c) of the framed (rows+2, cols+2)
// array is stored at [k][d = r + c][r]; the five neighbours of a cell are then at [d-1][r-1], [d-1][r], [d+1][r],
// [d+1][r+1] -- contiguous in r for a whole wavefront.  Channels are independent planes (the reference sweeps them
// one after the other).
//
// A wavefront is tiny (<= min(rows, cols) cells per channel) and the launches of one sweep depend on each other, so
// a sweep alone is latency-bound (~4 us per wavefront).  Consecutive SWEEPS are therefore pipelined: sweep s + 1 may
// update diagonal d as soon as sweep s has finished diagonal d + 1, so one launch advances up to S sweeps at once,
// sweep s working on diagonal t - 2 s -- S sweeps cost rows + cols + 2 S launches instead of S (rows + cols), with
// exactly the sequential algorithm's data dependences (bit-identical results).  The stop test
// sqrt(sum (u_new - u_old)^2 / size) > eps needs every sweep's own sum: the blocks write per-sweep partial sums
// (fixed tree, double), the host looks at them after a batch, and if the criterion fired inside the batch the state
// is restored from a snapshot taken before the batch and exactly that many sweeps are redone.  gfx950 only.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

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
    double* part;  // per-sweep partial sums of (u_new - u_old)^2: [sweep in batch][diagonal][channel][block]
    int64_t nd;    // diagonals
    int nbx;       // blocks per (diagonal, channel)
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

// one pipeline step t: sweep z of the batch (blockIdx.z) updates its wavefront d = t - 2 z (if that is a diagonal of
// the image), all cells r in [r_lo, r_hi] of it, channel blockIdx.y
template <typename T, bool ISO>
__global__ void __launch_bounds__(256) k_gs_step(GsArgs<T> a, int64_t t) {
    const int64_t d = t - 2 * (int64_t)blockIdx.z;
    if (d < 2 || d > a.rows + a.cols) return;  // block-uniform
    const int64_t r_lo = d - a.cols > 1 ? d - a.cols : 1, r_hi = d - 1 < a.rows ? d - 1 : a.rows;
    const int64_t r = r_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = blockIdx.y;
    double t2 = 0.0;
    if (r <= r_hi) {
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
        const T inc = unew - uprev;
        t2 = (double)inc * (double)inc;
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
    // deterministic block sum of the squared increments
    __shared__ double sm[256 / kWave];
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) t2 += __shfl_down(t2, s, kWave);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) sm[wid] = t2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) sum += sm[w];
        a.part[(((int64_t)blockIdx.z * a.nd + d) * a.dims + k) * a.nbx + blockIdx.x] = sum;
    }
}

template <typename T>
int run_gs(int64_t rows, int64_t cols, int64_t dims, const void* image, void* out, double weight, int max_iter,
           double eps, int isotropic, int* iters, double* rmse_out) {
    GsArgs<T> a{};
    a.rows = rows; a.cols = cols; a.dims = dims;
    a.rs = (rows + 2 + 31) / 32 * 32;
    const int64_t nd = rows + cols + 3;  // diagonals 0 .. rows + cols + 2
    a.nd = nd;
    a.plane = nd * a.rs;
    a.weight = (T)weight;
    a.lam = T(2) * a.weight;
    a.norm = a.weight + T(4) * a.lam;
    a.th = 1.0 / (double)a.lam;
    const int64_t longest = rows < cols ? rows : cols;
    a.nbx = (int)((longest + 255) / 256);
    // sweeps in flight: enough to hide the wavefront latency, few enough that the pipeline fill (2 S steps) stays small
    int batch = 64;
    if (const char* e = getenv("DXH_GS_BATCH")) batch = atoi(e);
    batch = std::max(1, std::min<int>(batch, (int)std::max<int64_t>(1, (rows + cols) / 8)));
    const size_t bytes = (size_t)a.plane * dims * sizeof(T);
    const size_t dense = (size_t)rows * cols * dims * sizeof(T);
    const size_t part_per_sweep = (size_t)nd * dims * a.nbx;
    void* bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // u, dx, dy, bx, by, f
    void* snap[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    void *d_img = nullptr, *d_out = nullptr;
    double *part = nullptr, *stage = nullptr, *red = nullptr;
    auto cleanup = [&]() {
        for (void* p : bufs) if (p) (void)hipFree(p);
        for (void* p : snap) if (p) (void)hipFree(p);
        for (void* p : {d_img, d_out, (void*)part, (void*)stage, (void*)red}) if (p) (void)hipFree(p);
    };
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, dense, {{image, dense}});
    auto body = [&]() -> int {
        for (void*& p : bufs) {
            DXH_HIP(hipMalloc(&p, bytes));
            DXH_HIP(hipMemsetAsync(p, 0, bytes, nullptr));
        }
        if (max_iter > 1 && batch > 1)
            for (void*& p : snap) DXH_HIP(hipMalloc(&p, bytes));
        DXH_HIP(hipMalloc(&d_img, dense));
        DXH_HIP(hipMalloc(&d_out, dense));
        DXH_HIP(hipMalloc((void**)&part, sizeof(double) * part_per_sweep * batch));
        DXH_HIP(hipMalloc((void**)&stage, sizeof(double) * 256));
        DXH_HIP(hipMalloc((void**)&red, sizeof(double) * batch));
        DXH_HIP(hipMemcpyAsync(d_img, image, dense, hipMemcpyDefault, nullptr));
        a.u = (T*)bufs[0]; a.dx = (T*)bufs[1]; a.dy = (T*)bufs[2]; a.bx = (T*)bufs[3]; a.by = (T*)bufs[4];
        a.f = (const T*)bufs[5];
        a.part = part;
        hipLaunchKernelGGL((k_gs_init<T>), dim3(2048), dim3(256), 0, nullptr, a, (const T*)d_img);
        DXH_HIP(hipGetLastError());
        const double total = (double)rows * (double)cols * (double)dims;
        // S pipelined sweeps; sq[s] = sum of squared increments of sweep s
        auto run_batch = [&](int S, double* sq) -> int {
            DXH_HIP(hipMemsetAsync(part, 0, sizeof(double) * part_per_sweep * S, nullptr));
            const dim3 grid((unsigned)a.nbx, (unsigned)dims, (unsigned)S);
            for (int64_t t = 2; t <= rows + cols + 2 * (int64_t)(S - 1); ++t) {
                if (isotropic) hipLaunchKernelGGL((k_gs_step<T, true>), grid, dim3(256), 0, nullptr, a, t);
                else hipLaunchKernelGGL((k_gs_step<T, false>), grid, dim3(256), 0, nullptr, a, t);
            }
            for (int s2 = 0; s2 < S; ++s2) {
                hipLaunchKernelGGL(k_sum_partials, dim3(256), dim3(256), 0, nullptr, part + (size_t)s2 * part_per_sweep,
                                   (int)part_per_sweep, stage);
                hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, nullptr, stage, 256, red + s2);
            }
            DXH_HIP(hipGetLastError());
            DXH_HIP(hipMemcpy(sq, red, sizeof(double) * S, hipMemcpyDefault));
            return DXH_OK;
        };
        int done = 0;
        double rmse = INFINITY;
        std::vector<double> sq((size_t)batch);
        while (done < max_iter && rmse > eps) {
            const int S = std::min(batch, max_iter - done);
            const bool guard = S > 1;  // even eps = 0 stops the reference when a sweep changes nothing
            if (guard)
                for (int q = 0; q < 5; ++q) DXH_HIP(hipMemcpyAsync(snap[q], bufs[q], bytes, hipMemcpyDeviceToDevice, nullptr));
            int rc2 = run_batch(S, sq.data());
            if (rc2) return rc2;
            int stop = -1;  // first sweep of the batch after which the reference would have left its loop
            for (int s2 = 0; s2 < S; ++s2)
                if (!(std::sqrt(sq[s2] / total) > eps)) { stop = s2; break; }
            if (stop >= 0 && stop < S - 1) {
                // overshoot: back to the state before the batch, then exactly stop + 1 sweeps
                DXH_REQUIRE(guard, "internal: stop inside an unguarded batch");
                for (int q = 0; q < 5; ++q) DXH_HIP(hipMemcpyAsync(bufs[q], snap[q], bytes, hipMemcpyDeviceToDevice, nullptr));
                if ((rc2 = run_batch(stop + 1, sq.data()))) return rc2;
                done += stop + 1;
                rmse = std::sqrt(sq[stop] / total);
                break;
            }
            done += S;
            rmse = std::sqrt(sq[S - 1] / total);
        }
        hipLaunchKernelGGL((k_gs_final<T>), dim3(2048), dim3(256), 0, nullptr, a, (T*)d_out);
        DXH_HIP(hipGetLastError());
        pf.join();
        DXH_HIP(hipMemcpy(out, d_out, dense, hipMemcpyDefault));
        if (iters) *iters = done;
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
