// Dense per-pixel kernels around the TV path: piecewise perspective warp (registration), thin-plate
// spline displacement field, affine colour balance.  gfx950 only.
#include <algorithm>
#include <array>
#include <atomic>
#include <memory>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "dxh_common.h"

namespace dxh {

// ---- piecewise perspective warp -------------------------------------------------------------------
// One thread per canvas pixel (all channels).  It visits the patches in the reference's loop order
// (piecewiseperspective.py:71-133) and, for every patch whose effective destination rectangle covers
// the pixel, evaluates OpenCV's float32 bilinear perspective sample (see oracle/warp_numpy.py for the
// restated algorithm) and adds it to the pixel in float32 -- the same order of additions as the
// reference's  canvas[roi] += tmp , hence deterministic and free of atomics.
// A block covers kWarpBW x kWarpRows pixels and looks only at the patches whose rectangle meets its
// kWarpBW x kWarpBH bin of the canvas (CSR lists built on the host, table order kept): one to four
// candidates instead of the whole table (200 records for 10 x 20 patches), and the list is block-uniform,
// so the records come in through scalar loads.
struct WarpPatch {
    double minv[9];              // inverse homography: destination (patch-local) -> source (patch-local)
    int sr0, sc0, sh, sw;        // source patch: origin in the base image and size
    int ox, oy;                  // origin of the warped tile in the canvas (x = column, y = row)
    int w_eff, h_eff;            // size of the warped tile (cv2 dsize)
    int nx, ny;                  // part of the tile added to the canvas: canvas[oy:oy+ny, ox:ox+nx]
    int bw0;                     // OpenCV's block width for this tile size
    int pad;                     // set by dxh_warp_patches: ceil(2^32 / bw0) for the division by bw0, 0 = divide
};

// saturate_cast<int>(max(INT_MIN, min(INT_MAX, v))) with cvRound's round-half-to-even: the clamped value is an int.
// fmax / fmin return the other operand for a NaN (so NaN -> INT_MIN like the comparison chain of the host code), and the
// clamped v plus 1.5 * 2^52 is rounded to an integer by the addition itself (ulp = 1, ties to even), whose two's complement
// sits in the low word: three full-rate operations instead of compares, selects, v_rndne_f64 and v_cvt_i32_f64.
__device__ __forceinline__ int round_half_even_clamped(double v) {
    v = fmin(fmax(v, -2147483648.0), 2147483647.0);
    return __double2loint(v + 6755399441055744.0);
}

// W != 0 ? 32.0 / W : 0.0  (OpenCV's perspective scale).  The compiler's correctly rounded division is
// v_div_scale x 2, v_rcp_f64, two Newton steps, q0 = n r, e = n - d q0, v_div_fmas, v_div_fixup; for a denominator whose
// exponent is far from both ends nothing is scaled, v_div_fmas is a plain FMA and v_div_fixup passes the quotient through,
// so the same operations without the three quarter-rate helpers give the same bits (dxh_warp_div_probe compares the two
// on the device over random and edge denominators).  Everything else -- zero, denormals, huge values, NaN -- divides.
__device__ __forceinline__ double warp_scale_ieee(double W) { return W != 0.0 ? 32.0 / W : 0.0; }
__device__ __forceinline__ double warp_scale(double W) {
#ifdef DXH_WARP_IEEE
    return warp_scale_ieee(W);
#else
    const unsigned ex = ((unsigned)__double2hiint(W) >> 20) & 0x7ffu;
    if (ex - 523u < 1001u) {   // 2^-500 <= |W| < 2^501
        double r = __builtin_amdgcn_rcp(W);
        double e = fma(-W, r, 1.0);
        r = fma(r, e, r);
        e = fma(-W, r, 1.0);
        r = fma(r, e, r);
        const double q0 = 32.0 * r;
        return fma(fma(-W, q0, 32.0), r, q0);
    }
    return warp_scale_ieee(W);
#endif
}

// Device test of warp_scale against the compiler's division: every thread walks `per_thread` pseudo-random denominators
// (splitmix64 of a global counter) -- any sign, mantissa and exponent in round 0, exponents within 2^+-16 of 1 (the
// values a homography produces) in round 1, and the boundary exponents of the screen in round 2.
__global__ void __launch_bounds__(256) k_warp_div_check(unsigned long long seed, int per_thread, int round,
                                                         unsigned long long* __restrict__ bad) {
    unsigned long long n = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * (unsigned long long)per_thread;
    unsigned mism = 0;
    for (int i = 0; i < per_thread; ++i, ++n) {
        unsigned long long z = seed + (n + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        unsigned long long bits = z;
        if (round == 1) {
            const unsigned long long ex = 1023ull - 16ull + ((z >> 52) & 31ull);
            bits = (z & 0x800FFFFFFFFFFFFFull) | (ex << 52);
        } else if (round == 2) {
            const unsigned long long edges[8] = {522, 523, 524, 1522, 1523, 1524, 0, 2047};
            bits = (z & 0x800FFFFFFFFFFFFFull) | (edges[(z >> 52) & 7ull] << 52);
        }
        const double W = __longlong_as_double((long long)bits);
        const double a = warp_scale(W), b = warp_scale_ieee(W);
        const bool same = __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
        mism += same ? 0u : 1u;
    }
    if (mism) atomicAdd(bad, (unsigned long long)mism);
}

constexpr int kWarpBW = 256, kWarpBH = 16;

// The float32 canvas value in the image's own type, as ``canvas.astype(dtype)`` gives it on the host
// (piecewiseperspective.py:133): exact for float64; integer types truncate towards zero through int32 and wrap.
template <typename S> __device__ __forceinline__ S warp_store(float v) { return (S)v; }
template <> __device__ __forceinline__ unsigned char warp_store<unsigned char>(float v) { return (unsigned char)(int)v; }
template <> __device__ __forceinline__ unsigned short warp_store<unsigned short>(float v) { return (unsigned short)(int)v; }

// S: stored type of image and canvas; the patch is read as ``patch.astype(float32)`` (piecewiseperspective.py:121) and the
// sum over the patches is a float32 canvas, whatever S.
// A thread produces kWarpRows vertically adjacent pixels (same bin, hence the same candidate records): the chain of scalar
// loads that fetches a record is paid once for them, the products with the column are shared and the independent
// per-row chains interleave.
constexpr int kWarpRows = 4;
static_assert(kWarpBH % kWarpRows == 0, "the rows of a thread must lie in one bin");

template <int C, typename S>
__global__ void __launch_bounds__(256) k_warp(const S* __restrict__ base, int H, int W, const WarpPatch* __restrict__ pt,
                                               const int* __restrict__ bin_off, const int* __restrict__ bin_idx,
                                               S* __restrict__ canvas) {
    constexpr int R = kWarpRows;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * R;
    if (x >= W || y0 >= H) return;
    float acc[R][C];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[r][c] = 0.f;
    const int bin = (y0 / kWarpBH) * gridDim.x + blockIdx.x;
    const int k0 = bin_off[bin], k1 = bin_off[bin + 1];
    for (int kk = k0; kk < k1; ++kk) {
        const WarpPatch& p = pt[bin_idx[kk]];
        // (one batch of scalar loads for the integers of the record, no short-circuit in between)
        const int nx = p.nx, ny = p.ny, bw0 = p.bw0;
        const unsigned magic = (unsigned)p.pad;
        const int tx = x - p.ox, ty0 = y0 - p.oy;
        bool in[R];
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            in[r] = !(((tx | (ty0 + r)) < 0) | (tx >= nx) | (ty0 + r >= ny) | (y0 + r >= H));
            any |= in[r];
        }
        if (!any) continue;
        // tx < 2^16 and bw0 < 2^16: the high word of tx * ceil(2^32 / bw0) is the exact quotient
        const int bx = (magic ? (int)__umulhi((unsigned)tx, magic) : tx / bw0) * bw0;
        const double x1 = (double)(tx - bx), fbx = (double)bx;
        const int sh = p.sh, sw = p.sw;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // a row outside the tile runs along on whatever its coordinates give (the taps are clamped into the patch) and
            // is dropped by the select at the end
            const double fy_ = (double)(ty0 + r);
            const double X0 = p.minv[0] * fbx + p.minv[1] * fy_ + p.minv[2];
            const double Y0 = p.minv[3] * fbx + p.minv[4] * fy_ + p.minv[5];
            const double W0 = p.minv[6] * fbx + p.minv[7] * fy_ + p.minv[8];
            const double Wd = warp_scale(W0 + p.minv[6] * x1);
            const int X = round_half_even_clamped((X0 + p.minv[0] * x1) * Wd);
            const int Y = round_half_even_clamped((Y0 + p.minv[3] * x1) * Wd);
            int sx = X >> 5, sy = Y >> 5;   // arithmetic shifts; saturate_cast<short> below
            sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
            sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
            const float fx = (float)(X & 31) * (1.0f / 32.0f), fy = (float)(Y & 31) * (1.0f / 32.0f);
            const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy};
            // the four taps, loaded unconditionally from clamped positions (no divergent branches around the loads; the
            // patch size stays in scalar registers) and zeroed by a select where the tap lies outside the patch
            const bool oky[2] = {(unsigned)sy < (unsigned)sh, (unsigned)(sy + 1) < (unsigned)sh};
            const bool okx[2] = {(unsigned)sx < (unsigned)sw, (unsigned)(sx + 1) < (unsigned)sw};
            const S* row[2];
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) row[ky] = base + ((size_t)(p.sr0 + (oky[ky] ? sy + ky : 0)) * W + p.sc0) * C;
            const int col[2] = {okx[0] ? sx : 0, okx[1] ? sx + 1 : 0};
            S raw[2][2][C];
#pragma unroll
            for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx)
#pragma unroll
                    for (int c = 0; c < C; ++c) raw[ky][kx][c] = row[ky][(size_t)col[kx] * C + c];
            float v[C];
#pragma unroll
            for (int c = 0; c < C; ++c) v[c] = 0.f;
            bool first = true;
#pragma unroll
            for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    const bool ok = oky[ky] && okx[kx];
                    const float wgt = wy[ky] * wx[kx];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float term = (ok ? (float)raw[ky][kx][c] : 0.f) * wgt;
                        v[c] = first ? term : v[c] + term;
                    }
                    first = false;
                }
#pragma unroll
            for (int c = 0; c < C; ++c) acc[r][c] = in[r] ? acc[r][c] + v[c] : acc[r][c];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (y0 + r >= H) break;
        S* o = canvas + ((size_t)(y0 + r) * W + x) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = warp_store<S>(acc[r][c]);
    }
}

// ---- thin-plate-spline field ------------------------------------------------------------------------
// out[n][d] = sum_i coef[i][d] * r^2 log r  +  polynomial(xhat) with r = |x_n - y_i|, SciPy's
// RBFInterpolator evaluation for kernel 'thin_plate_spline', epsilon = 1 (translationanalysis.py:285-301,
// imageregistration.py:265-306).  Points are either given explicitly or are the pixel grid
// (x = column + 0 , y = row) scaled by an affine map, so that the dense field needs no point upload.
// log for the spline kernel.  v = 2^e * m, m in [1, 2); the top 7 mantissa bits pick c_j ~ 1 / m from a 128-entry table
// held in LDS together with -log(c_j) (rounded from long double), r = m c_j - 1 is one FMA with |r| <= 2^-8, and
// log v = e ln 2 - log c_j + log1p(r) with the series of log1p up to r^7 (next term: 2^-59 relative to r).  Absolute error
// ~1e-16 (a few ulp of the result away from v ~ 1, where the result itself is tiny and enters r^2 log r with r^2 ~ 1):
// 11 double-precision operations instead of the ~50 of the library routine.  Denormals, inf, NaN take the library path.
constexpr int kLogTab = 128;
__device__ __forceinline__ double tps_log(double v, const double2* __restrict__ tab) {
    const int hi = __double2hiint(v);
    const unsigned ex = (unsigned)hi >> 20;                  // v > 0: no sign bit
    if (ex == 0u || ex >= 0x7ffu) return log(v);
    const double2 t = tab[(hi >> 13) & (kLogTab - 1)];
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(v));
    const double r = fma(m, t.x, -1.0);
    double q = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    q = fma(q, r, 1.0 / 5.0);
    q = fma(q, r, -1.0 / 4.0);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -0.5);
    q = fma(q * r, r, r);                                    // r - r^2/2 + ...
    const double e = (double)((int)ex - 1023);
    return fma(e, 0.6931471805599453094, t.y + q);
}

// The same for arguments known to be positive, normal and finite (k_tps_pair clamps r^2 from below; pixel coordinates are
// finite): no special-case branch, and the exponent becomes a double by assembling 2^52 + ex from its bits and subtracting
// (one full-rate addition instead of v_cvt_f64_i32).
__device__ __forceinline__ double tps_log_pos(double v, const double2* __restrict__ tab) {
    const int hi = __double2hiint(v);
    const double2 t = tab[(hi >> 13) & (kLogTab - 1)];
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(v));
    const double r = fma(m, t.x, -1.0);            // |r| <= 2^-8: the series up to r^6 (next term r^7 / 7 < 4e-18)
    double q = fma(r, -1.0 / 6.0, 1.0 / 5.0);
    q = fma(q, r, -1.0 / 4.0);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -0.5);
    q = fma(q * r, r, r);
    const double e = __hiloint2double(0x43300000, (int)((unsigned)hi >> 20)) - (4503599627370496.0 + 1023.0);
    return fma(e, 0.6931471805599453094, t.y + q);
}

static const double* tps_log_table() {
    static const std::array<double, 2 * kLogTab> tab = [] {
        std::array<double, 2 * kLogTab> t{};
        for (int j = 0; j < kLogTab; ++j) {
            const double c = 1.0 / (1.0 + (j + 0.5) / kLogTab);
            t[2 * j] = c;
            t[2 * j + 1] = (double)(-logl((long double)c));
        }
        return t;
    }();
    return tab.data();
}

struct TpsArgs {
    const double* logtab;   // kLogTab x (c, -log c)
    const double* centers;  // (p, 2)
    const double* coeffs;   // (p + 3, dout)
    const double* points;   // (n, 2) or null for the grid
    double shift[2], scale[2];
    int powers[3][2];
    int p, dout;
    long long n;
    int grid_w;             // grid mode: point n = (gx0 + gsx * (n % grid_w), gy0 + gsy * (n / grid_w))
    double gx0, gy0, gsx, gsy;
    double* out;            // (n, dout)
};

__global__ void __launch_bounds__(256) k_tps(TpsArgs a) {
    extern __shared__ __align__(16) double sm_all[];  // log table, then centers (2p), then coeffs ((p+3) * dout)
    const double2* ltab = reinterpret_cast<const double2*>(sm_all);
    double* sm = sm_all + 2 * kLogTab;
    const int nc = 2 * a.p, nk = (a.p + 3) * a.dout;
    for (int i = threadIdx.x; i < 2 * kLogTab; i += blockDim.x) sm_all[i] = a.logtab[i];
    for (int i = threadIdx.x; i < nc; i += blockDim.x) sm[i] = a.centers[i];
    for (int i = threadIdx.x; i < nk; i += blockDim.x) sm[nc + i] = a.coeffs[i];
    __syncthreads();
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n) return;
    double x, y;
    if (a.points) {
        x = a.points[2 * n];
        y = a.points[2 * n + 1];
    } else {
        x = a.gx0 + a.gsx * (double)(n % a.grid_w);
        y = a.gy0 + a.gsy * (double)(n / a.grid_w);
    }
    double acc0 = 0.0, acc1 = 0.0;
    const double* cf = sm + nc;
    {
        // fused multiply-adds and two centres per trip, as in k_tps_pair (compared with SciPy to a tolerance)
#pragma clang fp contract(fast)
#pragma unroll 2
        for (int i = 0; i < a.p; ++i) {
            const double dx = x - sm[2 * i], dy = y - sm[2 * i + 1];
            // thin-plate spline r^2 log r = 0.5 r^2 log(r^2): no square root (SciPy evaluates r^2 log r; the two forms
            // differ in the last bits only, tests compare to 1e-10)
            const double r2 = dx * dx + dy * dy;
            // (branch-free logarithm on the clamped argument as in k_tps_pair: exact 0 at a centre, NaN / inf coordinates
            // still give NaN / inf; the factor 0.5 once at the end -- a power of two, the same bits)
            const double phi = r2 * tps_log_pos(fmax(r2, 1e-300), ltab);
            acc0 += phi * cf[i * a.dout];
            if (a.dout > 1) acc1 += phi * cf[i * a.dout + 1];
        }
    }
    acc0 *= 0.5;
    acc1 *= 0.5;
    const double xh = (x - a.shift[0]) / a.scale[0], yh = (y - a.shift[1]) / a.scale[1];
    for (int j = 0; j < 3; ++j) {
        const double mono = pow(xh, (double)a.powers[j][0]) * pow(yh, (double)a.powers[j][1]);
        acc0 += mono * cf[(a.p + j) * a.dout];
        if (a.dout > 1) acc1 += mono * cf[(a.p + j) * a.dout + 1];
    }
    a.out[n * a.dout] = acc0;
    if (a.dout > 1) a.out[n * a.dout + 1] = acc1;
}

// Two splines at once on the pixel grid -- the x and y translation maps of a TranslationAnalysis share the patch centres
// and differ in their boundary points (translationanalysis.py:285-294) -- over the UNION of their centres (a spline
// that does not have a centre carries a zero weight there): one r^2 log r per centre and pixel serves both.  Every
// spline keeps its own polynomial tail (SciPy normalises it by the spline's own point cloud).  The result is written as
// (out_scale[0] * s0, out_scale[1] * s1) per pixel: the (x, y) metric displacement of imageregistration.py:265-306.
struct TpsPairArgs {
    const double* logtab;
    const double* table;      // pu x (cx, cy, w0, w1)
    int pu;
    double poly[2][3];        // polynomial coefficients of spline 0 / 1
    double shift[2][2], scale[2][2];
    int powers[2][3][2];
    long long n;
    int grid_w;
    double gx0, gy0, gsx, gsy;
    double out_scale[2];
    double* out;              // (n, 2)
};

// kTpsPts pixels of one row per thread (n, n + 1, ...: adjacent in x): the centre record comes out of LDS once for all of
// them and their chains are independent (the loop is bound by dependent double-precision operations)
constexpr int kTpsPts = 4;
__global__ void __launch_bounds__(256) k_tps_pair(TpsPairArgs a) {
    extern __shared__ __align__(16) double sm_all[];  // log table, then the centre table
    const double2* ltab = reinterpret_cast<const double2*>(sm_all);
    double* sm = sm_all + 2 * kLogTab;
    for (int i = threadIdx.x; i < 2 * kLogTab; i += blockDim.x) sm_all[i] = a.logtab[i];
    for (int i = threadIdx.x; i < 4 * a.pu; i += blockDim.x) sm[i] = a.table[i];
    __syncthreads();
    constexpr int P = kTpsPts;
    const long long n0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * P;
    if (n0 >= a.n) return;
    double x[P], y[P], acc0[P], acc1[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const long long n = n0 + k < a.n ? n0 + k : a.n - 1;
        x[k] = a.gx0 + a.gsx * (double)(n % a.grid_w);
        y[k] = a.gy0 + a.gsy * (double)(n / a.grid_w);
        acc0[k] = acc1[k] = 0.0;
    }
    const double2* tb = reinterpret_cast<const double2*>(sm);
    {
        // compared with SciPy to a tolerance, not bit for bit: fused multiply-adds are welcome here (the library is built with
        // -ffp-contract=off for the NumPy-exact kernels)
#pragma clang fp contract(fast)
        for (int i = 0; i < a.pu; ++i) {
            const double2 c = tb[2 * i], w = tb[2 * i + 1];
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const double dx = x[k] - c.x, dy = y[k] - c.y;
                const double r2 = dx * dx + dy * dy;
                // r^2 log r = 0.5 r^2 log r^2, the 0.5 folded into the weights on the host; at a centre (r^2 = 0) the
                // clamped argument keeps the logarithm finite and the product an exact 0
                const double phi = r2 * tps_log_pos(fmax(r2, 1e-300), ltab);
                acc0[k] += phi * w.x;
                acc1[k] += phi * w.y;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (n0 + k >= a.n) break;
        double acc[2] = {acc0[k], acc1[k]};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const double xh = (x[k] - a.shift[o][0]) / a.scale[o][0], yh = (y[k] - a.shift[o][1]) / a.scale[o][1];
            for (int j = 0; j < 3; ++j)
                acc[o] += pow(xh, (double)a.powers[o][j][0]) * pow(yh, (double)a.powers[o][j][1]) * a.poly[o][j];
        }
        a.out[2 * (n0 + k)] = a.out_scale[0] * acc[0];
        a.out[2 * (n0 + k) + 1] = a.out_scale[1] * acc[1];
    }
}

// ---- affine colour balance -----------------------------------------------------------------------------
// out = img_as_float(img) @ A + b, optional clip to [0, 1], float32 result
// (corrections/color/colorbalance.py:186-196, colorcorrection.py:388-394); arithmetic in double.
template <typename TIN>
__global__ void __launch_bounds__(256) k_affine_color(const TIN* __restrict__ in, long long npix, double in_scale,
                                                       const double* __restrict__ ab, int clip, float* __restrict__ out) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= npix) return;
    double x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) x[k] = (double)in[3 * n + k] / in_scale;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double v = x[0] * ab[0 * 3 + c];
        v = v + x[1] * ab[1 * 3 + c];
        v = v + x[2] * ab[2 * 3 + c];
        v = v + ab[9 + c];
        if (clip) v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        out[3 * n + c] = (float)v;
    }
}

struct Dev {
    std::vector<void*> bufs;
    hipStream_t st = nullptr;
    ~Dev() {
        // buffers go back to the plane cache (dxh_common.h), which wants them idle: every entry point synchronises its
        // stream before it returns, an early error return has not
        if (st) (void)hipStreamSynchronize(st);
        for (void* b : bufs) pool_free(b);
        // st is the thread's cached stream (cached_stream): not destroyed
    }
    int alloc(void** p, size_t n) {
        DXH_HIP(pool_malloc(p, n ? n : 1));
        bufs.push_back(*p);
        return DXH_OK;
    }
};

static int check_device(int device) {
    int ndev = 0;
    DXH_HIP(hipGetDeviceCount(&ndev));
    DXH_REQUIRE(device >= 0 && device < ndev, "device %d not available (%d devices)", device, ndev);
    DXH_HIP(hipSetDevice(device));
    return DXH_OK;
}


// ---- scipy.ndimage.map_coordinates(channel, grid, order = 0 | 1) with the default mode "constant" ---------------
// CurvatureCorrection._transform_image (corrections/shape/curvature.py:921-957).  SciPy's arithmetic (third-party,
// ndimage/src/ni_interpolation.c; restated and checked against SciPy in oracle/curvature_numpy.py):
// a point with a coordinate < 0 or > n-1 yields cval; otherwise order 1 takes start = floor(c), weights
// (1 - t, t) in double, out-of-range taps mirrored (they carry weight 0), and sums the four taps row-major as
// ((v * w_row) * w_col); order 0 takes floor(c + 0.5).  Integer outputs round half away from zero (unsigned: +0.5).
template <typename T>
__device__ __forceinline__ T mc_cast(double t);
template <> __device__ __forceinline__ float mc_cast<float>(double t) { return (float)t; }
template <> __device__ __forceinline__ double mc_cast<double>(double t) { return t; }
template <> __device__ __forceinline__ unsigned char mc_cast<unsigned char>(double t) { return (unsigned char)(t > 0 ? t + 0.5 : 0.0); }
template <> __device__ __forceinline__ unsigned short mc_cast<unsigned short>(double t) { return (unsigned short)(t > 0 ? t + 0.5 : 0.0); }

__device__ __forceinline__ long long mc_mirror(long long i, long long n) {
    if (n <= 1) return 0;
    const long long s2 = 2 * n - 2;
    if (i < 0) i = -i;
    i %= s2;
    return i >= n ? s2 - i : i;
}

template <typename T, typename CT>
__global__ void __launch_bounds__(256) k_map_coords(const T* __restrict__ img, long long H, long long W, int C,
                                                     const CT* __restrict__ rows, const CT* __restrict__ cols, long long n,
                                                     int order, double cval, T* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double cr = (double)rows[i], cc = (double)cols[i];
    T* o = out + i * C;
    if (!(cr >= 0.0 && cr <= (double)(H - 1) && cc >= 0.0 && cc <= (double)(W - 1))) {  // also NaN
        for (int k = 0; k < C; ++k) o[k] = mc_cast<T>(cval);
        return;
    }
    if (order == 0) {
        const long long r = mc_mirror((long long)floor(cr + 0.5), H), c = mc_mirror((long long)floor(cc + 0.5), W);
        const T* p = img + (r * W + c) * C;
        for (int k = 0; k < C; ++k) o[k] = mc_cast<T>((double)p[k]);
        return;
    }
    const double fr = floor(cr), fc = floor(cc);
    const double tr = cr - fr, tc = cc - fc;
    const double wr[2] = {1.0 - tr, tr}, wc[2] = {1.0 - tc, tc};
    const long long r0 = (long long)fr, c0 = (long long)fc;
    const long long rr[2] = {mc_mirror(r0, H), mc_mirror(r0 + 1, H)}, cl[2] = {mc_mirror(c0, W), mc_mirror(c0 + 1, W)};
    for (int k = 0; k < C; ++k) {
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                double v = (double)img[(rr[a] * W + cl[b]) * C + k];
                v = v * wr[a];
                v = v * wc[b];
                t = t + v;
            }
        o[k] = mc_cast<T>(t);
    }
}


// ---- cv2.resize per channel (restoration/resize.py:150-176), OpenCV's generic code paths restated in
// oracle/resize_numpy.py (third-party arithmetic, parity unpinned: no OpenCV in this image).  The tap / area
// tables are built on the host (a few thousand entries) exactly as OpenCV builds xofs / alpha / beta / xtab / ytab;
// one thread per destination pixel and channel applies them with OpenCV's order of operations. ---------------
struct ResizePlanDev {
    int mode;  // 0 nearest, 1 bilinear, 2 area with integer scales, 3 area with weight tables
    int isx, isy;
    const int* xi; const float* xw; const int* yi; const float* yw;
    const int* xptr; const int* xsrc; const float* xalpha;
    const int* yptr; const int* ysrc; const float* yalpha;
};

template <typename T> struct ResizeWT { typedef float type; };
template <> struct ResizeWT<double> { typedef double type; };
template <typename T> __device__ __forceinline__ T rs_cast(float v) { return (T)v; }
template <> __device__ __forceinline__ unsigned char rs_cast<unsigned char>(float v) {
    const float r = rintf(v);  // cvRound: half to even
    return (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}
template <typename T> __device__ __forceinline__ T rs_cast_d(double v) { return (T)v; }

template <typename T>
__global__ void __launch_bounds__(256) k_resize(const T* __restrict__ src, int H, int W, int C, ResizePlanDev p, int DH, int DW,
                                                 T* __restrict__ dst) {
    typedef typename ResizeWT<T>::type WT;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)DH * DW * C) return;
    const int k = (int)(i % C);
    const int dx = (int)((i / C) % DW), dy = (int)(i / ((long long)C * DW));
    auto S = [&](int y, int x) -> T { return src[((long long)y * W + x) * C + k]; };
    if (p.mode == 0) {
        dst[i] = S(p.yi[dy], p.xi[dx]);
    } else if (p.mode == 1) {
        const int x0 = p.xi[dx], x1 = x0 + 1 < W ? x0 + 1 : W - 1, y0 = p.yi[dy], y1 = y0 + 1 < H ? y0 + 1 : H - 1;
        const float fx = p.xw[dx], fy = p.yw[dy];
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        if constexpr (sizeof(T) == 1) {
            auto fix = [](float w) -> int { float r = rintf(w * 2048.f); r = r < -32768.f ? -32768.f : (r > 32767.f ? 32767.f : r); return (int)r; };
            const int ia0 = fix(a0), ia1 = fix(a1), ib0 = fix(b0), ib1 = fix(b1);
            const int r0 = (int)S(y0, x0) * ia0 + (int)S(y0, x1) * ia1, r1 = (int)S(y1, x0) * ia0 + (int)S(y1, x1) * ia1;
            int v = (((ib0 * (r0 >> 4)) >> 16) + ((ib1 * (r1 >> 4)) >> 16) + 2) >> 2;
            dst[i] = (T)(v < 0 ? 0 : (v > 255 ? 255 : v));
        } else {
            const WT r0 = (WT)S(y0, x0) * (WT)a0 + (WT)S(y0, x1) * (WT)a1;
            const WT r1 = (WT)S(y1, x0) * (WT)a0 + (WT)S(y1, x1) * (WT)a1;
            dst[i] = (T)(r0 * (WT)b0 + r1 * (WT)b1);
        }
    } else if (p.mode == 2) {
        const int y0 = dy * p.isy, x0 = dx * p.isx;
        const int ny = y0 + p.isy <= H ? p.isy : (H - y0 > 0 ? H - y0 : 0), nx = x0 + p.isx <= W ? p.isx : (W - x0 > 0 ? W - x0 : 0);
        const float scl = 1.f / (float)(p.isx * p.isy);
        if (ny == 0 || nx == 0) {
            dst[i] = (T)0;
        } else if constexpr (sizeof(T) == 1) {
            int s = 0;
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) s += (int)S(y0 + y, x0 + x);
            if (ny * nx < p.isx * p.isy) dst[i] = rs_cast<T>((float)s / (float)(ny * nx));
            else if (p.isx == 2 && p.isy == 2) dst[i] = (T)((s + 2) >> 2);
            else dst[i] = rs_cast<T>((float)s * scl);
        } else {
            WT acc = WT(0);
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) acc = acc + (WT)S(y0 + y, x0 + x);
            if (ny * nx < p.isx * p.isy) dst[i] = (T)((float)acc / (float)(ny * nx));
            else dst[i] = (T)(acc * (WT)scl);
        }
    } else {
        WT sum = WT(0);
        for (int j = p.yptr[dy]; j < p.yptr[dy + 1]; ++j) {
            const int sy = p.ysrc[j];
            WT buf = WT(0);
            for (int q = p.xptr[dx]; q < p.xptr[dx + 1]; ++q) buf = buf + (WT)S(sy, p.xsrc[q]) * (WT)p.xalpha[q];
            const WT t = (WT)p.yalpha[j] * buf;
            sum = j == p.yptr[dy] ? t : sum + t;
        }
        if constexpr (sizeof(T) == 1) dst[i] = rs_cast<T>((float)sum);
        else dst[i] = (T)sum;
    }
}

}  // namespace dxh

using namespace dxh;

// ---- patch translation by phase correlation ----------------------------------------------------------------------
// The estimator darsia_amd's registration ships in place of the reference's OpenCV ORB + RANSAC one (no counterpart in
// the reference; algorithm stated in oracle/phase_correlation_numpy.py): per patch pair, Hann-windowed mean-free patches
// zero-padded to (P0, P1), cross-power spectrum normalised to unit modulus, inverse transform, peak + parabolic fit.
// All patches of a call go through the device together, float64 throughout.  The transforms are radix-2 in LDS, one
// row or column per workgroup: forward passes decimate in frequency (natural order in, bit-reversed out), inverse
// passes decimate in time (bit-reversed in, natural out) -- the cross-power product in between is elementwise, so no
// permutation is ever needed.
struct PcRoi { int r0, r1, c0, c1; };

template <typename S>
__device__ __forceinline__ double pc_pixel(const S* __restrict__ img, int W, int C, int r, int c) {
    const S* p = img + ((size_t)r * W + c) * C;
    if (C == 1) return (double)p[0];
    double s = (double)p[0];        // ndarray.mean(axis=2) of three channels: ((x0 + x1) + x2) / 3
    s += (double)p[1];
    s += (double)p[2];
    return s / 3.0;
}

// a patch larger than the longest transform is read in cells of dec x dec pixels (their mean); dec = 1 otherwise
template <typename S>
__device__ __forceinline__ double pc_cell(const S* __restrict__ img, int W, int C, int r, int c, int dec) {
    if (dec == 1) return pc_pixel(img, W, C, r, c);
    double s = 0.0;
    for (int i = 0; i < dec; ++i)
        for (int j = 0; j < dec; ++j) s += pc_pixel(img, W, C, r + i, c + j);
    return s / (double)(dec * dec);
}

__device__ __forceinline__ double2 pc_cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// np.hanning(M)[k] = 0.5 + 0.5 cos(pi n / (M - 1)), n = 2 k - (M - 1)
__device__ __forceinline__ double pc_hann(int k, int M) {
    if (M == 1) return 1.0;
    return 0.5 + 0.5 * cos(3.14159265358979323846 * (double)(2 * k - (M - 1)) / (double)(M - 1));
}

// forward: decimation in frequency, twiddles tw[k] = exp(-2 pi i k / N), output bit-reversed; `batch` independent
// transforms lie one after the other in buf
__device__ __forceinline__ void pc_fft_dif(double2* buf, int N, int logN, const double2* __restrict__ tw, int batch = 1) {
    for (int s = logN; s >= 1; --s) {
        const int half = 1 << (s - 1);
        const int stride = N >> s;
        for (int t = threadIdx.x; t < batch * (N / 2); t += blockDim.x) {
            const int tt = t & (N / 2 - 1);
            const int k = tt & (half - 1);
            const int j = (t >> (logN - 1)) * N + ((tt >> (s - 1)) << s) + k;
            const double2 u = buf[j], v = buf[j + half];
            buf[j] = make_double2(u.x + v.x, u.y + v.y);
            buf[j + half] = pc_cmul(make_double2(u.x - v.x, u.y - v.y), tw[k * stride]);
        }
        __syncthreads();
    }
}

// inverse (unnormalised): decimation in time on bit-reversed input, conjugate twiddles, natural output
__device__ __forceinline__ void pc_ifft_dit(double2* buf, int N, int logN, const double2* __restrict__ tw, int batch = 1) {
    for (int s = 1; s <= logN; ++s) {
        const int half = 1 << (s - 1);
        const int stride = N >> s;
        for (int t = threadIdx.x; t < batch * (N / 2); t += blockDim.x) {
            const int tt = t & (N / 2 - 1);
            const int k = tt & (half - 1);
            const int j = (t >> (logN - 1)) * N + ((tt >> (s - 1)) << s) + k;
            double2 w = tw[k * stride];
            w.y = -w.y;
            const double2 u = buf[j], v = pc_cmul(buf[j + half], w);
            buf[j] = make_double2(u.x + v.x, u.y + v.y);
            buf[j + half] = make_double2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

// mean of every patch of both images: grid (npatch, 2)
template <typename S>
__global__ void __launch_bounds__(256) k_pc_mean(const S* __restrict__ src, const S* __restrict__ dst, int W, int C, int dec,
                                                  const PcRoi* __restrict__ rois, double* __restrict__ means) {
    __shared__ double sm[256 / kWave];
    const PcRoi q = rois[blockIdx.x];
    const S* img = blockIdx.y ? dst : src;
    const int ph = (q.r1 - q.r0) / dec, pw = (q.c1 - q.c0) / dec;
    double acc = 0.0;
    for (int i = threadIdx.x; i < ph * pw; i += blockDim.x) acc += pc_cell(img, W, C, q.r0 + (i / pw) * dec, q.c0 + (i % pw) * dec, dec);
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) acc += __shfl_down(acc, s, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) sm[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) s += sm[w];
        means[blockIdx.y * gridDim.x + blockIdx.x] = s / ((double)ph * (double)pw);
    }
}

// windowed, padded rows -> spectrum along the row.  grid (max patch height, npatch, 2); F[image][patch][P0][P1]
template <typename S>
__global__ void __launch_bounds__(256) k_pc_rows(const S* __restrict__ src, const S* __restrict__ dst, int W, int C, int dec,
                                                  const PcRoi* __restrict__ rois, const double* __restrict__ means, int nmeans,
                                                  int P0, int P1, int logP1, const double2* __restrict__ tw1, double2* __restrict__ FA,
                                                  double2* __restrict__ FB) {
    extern __shared__ double2 pc_buf[];
    const PcRoi q = rois[blockIdx.y];
    const int ph = (q.r1 - q.r0) / dec, pw = (q.c1 - q.c0) / dec, r = blockIdx.x;
    if (r >= ph) return;
    const S* img = blockIdx.z ? dst : src;
    const double mean = means[(size_t)blockIdx.z * nmeans + blockIdx.y], wy = pc_hann(r, ph);
    for (int c = threadIdx.x; c < P1; c += blockDim.x) {
        double v = 0.0;
        if (c < pw) v = ((pc_cell(img, W, C, q.r0 + r * dec, q.c0 + c * dec, dec) - mean) * wy) * pc_hann(c, pw);
        pc_buf[c] = make_double2(v, 0.0);
    }
    __syncthreads();
    pc_fft_dif(pc_buf, P1, logP1, tw1);
    double2* out = (blockIdx.z ? FB : FA) + ((size_t)blockIdx.y * P0 + r) * P1;
    for (int c = threadIdx.x; c < P1; c += blockDim.x) out[c] = pc_buf[c];
}

// columns: spectrum of both images, unit-modulus cross-power, inverse along the column.  A workgroup takes CB adjacent
// columns (CB x 16 B contiguous per row: 8 columns are one 128-byte line), column cb at pc_buf[cb * P0 ...].  grid (P1 / CB, npatch)
__global__ void __launch_bounds__(256) k_pc_cols(const PcRoi* __restrict__ rois, int dec, int P0, int P1, int logP0, int CB,
                                                  const double2* __restrict__ tw0, double2* __restrict__ FA,
                                                  double2* __restrict__ FB) {
    extern __shared__ double2 pc_buf[];
    const PcRoi q = rois[blockIdx.y];
    const int ph = (q.r1 - q.r0) / dec, c = blockIdx.x * CB;
    double2* a = FA + (size_t)blockIdx.y * P0 * P1 + c;
    double2* b = FB + (size_t)blockIdx.y * P0 * P1 + c;
    const int n = P0 * CB;   // element i: row i / CB, column i % CB (CB is a power of two)
    const int cm = CB - 1, cs = __ffs(CB) - 1;
    const double2 zero = make_double2(0.0, 0.0);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i >> cs, cb = i & cm;
        pc_buf[cb * P0 + r] = r < ph ? a[(size_t)r * P1 + cb] : zero;
    }
    __syncthreads();
    pc_fft_dif(pc_buf, P0, logP0, tw0, CB);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {   // re-read below by the same thread
        const int r = i >> cs, cb = i & cm;
        a[(size_t)r * P1 + cb] = pc_buf[cb * P0 + r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i >> cs, cb = i & cm;
        pc_buf[cb * P0 + r] = r < ph ? b[(size_t)r * P1 + cb] : zero;
    }
    __syncthreads();
    pc_fft_dif(pc_buf, P0, logP0, tw0, CB);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i >> cs, cb = i & cm;
        const double2 fa = a[(size_t)r * P1 + cb], fb = pc_buf[cb * P0 + r];
        const double2 x = pc_cmul(fb, make_double2(fa.x, -fa.y));
        const double mag = fmax(hypot(x.x, x.y), 1e-12);
        pc_buf[cb * P0 + r] = make_double2(x.x / mag, x.y / mag);
    }
    __syncthreads();
    pc_ifft_dit(pc_buf, P0, logP0, tw0, CB);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i >> cs, cb = i & cm;
        b[(size_t)r * P1 + cb] = pc_buf[cb * P0 + r];
    }
}

// rows back: correlation surface (unnormalised; only its peak and ratios of its values are used).  grid (P0, npatch)
__global__ void __launch_bounds__(256) k_pc_rows_inv(int P0, int P1, int logP1, const double2* __restrict__ tw1,
                                                      const double2* __restrict__ FB, double* __restrict__ corr) {
    extern __shared__ double2 pc_buf[];
    const size_t row = (size_t)blockIdx.y * P0 + blockIdx.x;
    const double2* in = FB + row * P1;
    for (int c = threadIdx.x; c < P1; c += blockDim.x) pc_buf[c] = in[c];
    __syncthreads();
    pc_ifft_dit(pc_buf, P1, logP1, tw1);
    double* out = corr + row * P1;
    for (int c = threadIdx.x; c < P1; c += blockDim.x) out[c] = pc_buf[c].x;
}

// peak (first maximum in row-major order, like np.argmax), parabolic refinement, acceptance.  grid (npatch)
__global__ void __launch_bounds__(1024) k_pc_peak(const PcRoi* __restrict__ rois, int dec, int P0, int P1, const double* __restrict__ corr,
                                                   double max_shift_fraction, double min_peak, double* __restrict__ shifts,
                                                   int* __restrict__ ok, double* __restrict__ peaks) {
    __shared__ double sv[1024 / kWave];
    __shared__ long long si[1024 / kWave];
    const double* cp = corr + (size_t)blockIdx.x * P0 * P1;
    const long long n = (long long)P0 * P1;
    double best = -INFINITY;
    long long at = n;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = cp[i];
        if (v > best) { best = v; at = i; }        // ascending i per thread: the first maximum stays
    }
#pragma unroll
    for (int s = kWave / 2; s > 0; s >>= 1) {
        const double ov = __shfl_down(best, s, kWave);
        const long long oi = __shfl_down(at, s, kWave);
        if (ov > best || (ov == best && oi < at)) { best = ov; at = oi; }
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { sv[threadIdx.x / kWave] = best; si[threadIdx.x / kWave] = at; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < (int)(blockDim.x / kWave); ++w)
        if (sv[w] > best || (sv[w] == best && si[w] < at)) { best = sv[w]; at = si[w]; }
    const PcRoi q = rois[blockIdx.x];
    const int ph = (q.r1 - q.r0) / dec, pw = (q.c1 - q.c0) / dec;
    if (at >= n) at = 0;   // all NaN
    const int py = (int)(at / P1), px = (int)(at % P1);
    auto refine = [](double cm, double c0, double cq) {
        const double den = cm - 2 * c0 + cq;
        return den == 0 ? 0.0 : 0.5 * (cm - cq) / den;
    };
    const double c0 = cp[(size_t)py * P1 + px];
    double dy = py + refine(cp[(size_t)((py + P0 - 1) % P0) * P1 + px], c0, cp[(size_t)((py + 1) % P0) * P1 + px]);
    double dx = px + refine(cp[(size_t)py * P1 + (px + P1 - 1) % P1], c0, cp[(size_t)py * P1 + (px + 1) % P1]);
    if (dy > P0 / 2) dy -= P0;
    if (dx > P1 / 2) dx -= P1;
    shifts[2 * blockIdx.x] = dx * dec;       // cells -> pixels
    shifts[2 * blockIdx.x + 1] = dy * dec;
    // accepted: a finite shift within the window AND a peak that stands out.  The correlation surface of a unit-modulus
    // cross-power spectrum is 1 at the shift of two equal patterns, 0 everywhere for a flat patch (its mean-free
    // transform is 0) and, for unrelated patterns, noise of standard deviation ~ 1 / sqrt(2 N) whose maximum over the N
    // cells is ~ sqrt(ln N / N): min_peak is a multiple of that (dxh_phase_correlation_q).  !(x <= y) also rejects NaN.
    const bool within = fabs(dy) <= max_shift_fraction * ph && fabs(dx) <= max_shift_fraction * pw;
    const double peak = c0 / (double)n;   // the inverse transforms of this file do not scale by 1 / N
    ok[blockIdx.x] = within && peak >= min_peak && isfinite(peak);
    if (peaks) peaks[blockIdx.x] = peak;
}

namespace {
template <typename S>
int phase_correlation_t(Dev& d, int H, int W, int C, const S* src, const S* dst, int npatch, const PcRoi* rois_h,
                        double max_shift_fraction, double min_peak_ratio, double* shifts, int32_t* ok, double* peaks, int* dec_out) {
    int maxh = 0, maxw = 0;
    for (int k = 0; k < npatch; ++k) {
        maxh = std::max(maxh, rois_h[k].r1 - rois_h[k].r0);
        maxw = std::max(maxw, rois_h[k].c1 - rois_h[k].c0);
    }
    // patches longer than the longest transform (8192 points: 128 KB of LDS) are read in cells of dec x dec pixels
    int dec = 1;
    while (maxh / dec > 8192 || maxw / dec > 8192) dec *= 2;
    *dec_out = dec;
    maxh /= dec;
    maxw /= dec;
    int P0 = 8, P1 = 8, l0 = 3, l1 = 3;
    while (P0 < maxh) { P0 *= 2; ++l0; }
    while (P1 < maxw) { P1 *= 2; ++l1; }
    // twiddles exp(-2 pi i k / N), k < N / 2, for both lengths
    std::vector<double> tw((size_t)P0 + P1);
    for (int k = 0; k < P0 / 2; ++k) { tw[2 * k] = std::cos(2.0 * M_PI * k / P0); tw[2 * k + 1] = -std::sin(2.0 * M_PI * k / P0); }
    for (int k = 0; k < P1 / 2; ++k) { tw[P0 + 2 * k] = std::cos(2.0 * M_PI * k / P1); tw[P0 + 2 * k + 1] = -std::sin(2.0 * M_PI * k / P1); }
    int rc;
    void *dtw = nullptr, *drois = nullptr, *dmeans = nullptr, *dshift = nullptr, *dok = nullptr, *FA = nullptr, *FB = nullptr, *dpeak = nullptr;
    const double ncell = (double)P0 * (double)P1;
    const double min_peak = min_peak_ratio > 0.0 ? min_peak_ratio * std::sqrt(std::log(ncell) / ncell) : -INFINITY;
    // patches per pass: two complex planes of P0 x P1 each, kept below a quarter of the free device memory
    size_t free_b = 0, total_b = 0;
    DXH_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t per_patch = (size_t)P0 * P1 * sizeof(double2);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)npatch, free_b / 8 / per_patch));
    if (const char* e = std::getenv("DXH_PC_CHUNK")) chunk = std::max(1, std::min(npatch, std::atoi(e)));   // developer knob
    if ((rc = d.alloc(&dtw, sizeof(double) * tw.size())) || (rc = d.alloc(&drois, sizeof(PcRoi) * (size_t)npatch)) ||
        (rc = d.alloc(&dmeans, sizeof(double) * 2 * (size_t)npatch)) || (rc = d.alloc(&dshift, sizeof(double) * 2 * (size_t)npatch)) ||
        (rc = d.alloc(&dok, sizeof(int) * (size_t)npatch)) || (rc = d.alloc(&FA, per_patch * chunk)) || (rc = d.alloc(&FB, per_patch * chunk)) ||
        (rc = d.alloc(&dpeak, sizeof(double) * (size_t)npatch)))
        return rc;
    DXH_HIP(hipMemcpyAsync(dtw, tw.data(), sizeof(double) * tw.size(), hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(drois, rois_h, sizeof(PcRoi) * (size_t)npatch, hipMemcpyDefault, d.st));
    const double2* tw0 = (const double2*)dtw;
    const double2* tw1 = tw0 + P0 / 2;
    // columns per workgroup of the column pass: four (64 contiguous bytes per row, the other half of the line goes to the
    // neighbouring workgroup through L2) -- measured on 200 patches of 512 x 512: 8 columns 2.27 ms, 4: 1.92, 2: 2.21, 1: 3.69;
    // the smaller LDS image lets more workgroups share a CU, and the pass is latency-bound (a barrier per radix-2 stage)
    int CB = std::max(1, std::min(4, 4096 / P0));
    if (const char* e = std::getenv("DXH_PC_CB")) {   // developer knob: fewer columns per workgroup (power of two)
        const int want = std::atoi(e);
        if (want >= 1 && want <= CB && (want & (want - 1)) == 0) CB = want;
    }
    const size_t lds0 = (size_t)P0 * CB * sizeof(double2), lds1 = (size_t)P1 * sizeof(double2);
    if (lds1 > 64 * 1024) {
        DXH_HIP(hipFuncSetAttribute((const void*)k_pc_rows<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        DXH_HIP(hipFuncSetAttribute((const void*)k_pc_rows_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    }
    if (lds0 > 64 * 1024) DXH_HIP(hipFuncSetAttribute((const void*)k_pc_cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
    hipLaunchKernelGGL((k_pc_mean<S>), dim3(npatch, 2), dim3(256), 0, d.st, src, dst, W, C, dec, (const PcRoi*)drois, (double*)dmeans);
    for (int p0 = 0; p0 < npatch; p0 += chunk) {
        const int np = std::min(chunk, npatch - p0);
        const PcRoi* r = (const PcRoi*)drois + p0;
        // means are laid out [image][patch of the whole call]: the chunk's slice starts at p0, the image stride stays npatch
        hipLaunchKernelGGL((k_pc_rows<S>), dim3(maxh, np, 2), dim3(256), lds1, d.st, src, dst, W, C, dec, r,
                           (const double*)dmeans + p0, npatch, P0, P1, l1, tw1, (double2*)FA, (double2*)FB);
        hipLaunchKernelGGL(k_pc_cols, dim3(P1 / CB, np), dim3(256), lds0, d.st, r, dec, P0, P1, l0, CB, tw0, (double2*)FA, (double2*)FB);
        hipLaunchKernelGGL(k_pc_rows_inv, dim3(P0, np), dim3(256), lds1, d.st, P0, P1, l1, tw1, (const double2*)FB, (double*)FA);
        hipLaunchKernelGGL(k_pc_peak, dim3(np), dim3(1024), 0, d.st, r, dec, P0, P1, (const double*)FA, max_shift_fraction, min_peak,
                           (double*)dshift + 2 * p0, (int*)dok + p0, (double*)dpeak + p0);
        DXH_HIP(hipGetLastError());
    }
    DXH_HIP(hipMemcpyAsync(shifts, dshift, sizeof(double) * 2 * (size_t)npatch, hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(ok, dok, sizeof(int) * (size_t)npatch, hipMemcpyDefault, d.st));
    if (peaks) DXH_HIP(hipMemcpyAsync(peaks, dpeak, sizeof(double) * (size_t)npatch, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));
    return DXH_OK;
}
}  // namespace

extern "C" {

int dxh_warp_patches_typed(int device, int H, int W, int C, int dtype, const void* base, const dxh_warp_patch* patches,
                           int npatch, void* canvas) {
    DXH_REQUIRE(base && canvas && (patches || npatch == 0), "null argument");
    DXH_REQUIRE(dtype == DXH_F32 || dtype == DXH_F64 || dtype == DXH_U8 || dtype == DXH_U16,
                "dtype %d not supported (float32, float64, uint8, uint16)", dtype);
    static const size_t width[4] = {4, 8, 1, 2};
    const size_t esz = width[dtype];
    DXH_REQUIRE(H >= 1 && W >= 1 && (C == 1 || C == 3), "shape (%d, %d, %d) not supported (C must be 1 or 3)", H, W, C);
    static_assert(sizeof(dxh_warp_patch) == sizeof(WarpPatch), "patch struct mismatch");
    int rc = check_device(device);
    if (rc) return rc;
    for (int k = 0; k < npatch; ++k) {
        const dxh_warp_patch& p = patches[k];
        DXH_REQUIRE(p.sr0 >= 0 && p.sc0 >= 0 && p.sh >= 1 && p.sw >= 1 && p.sr0 + p.sh <= H && p.sc0 + p.sw <= W,
                    "patch %d: source rectangle outside the image", k);
        DXH_REQUIRE(p.ox >= 0 && p.oy >= 0 && p.nx >= 0 && p.ny >= 0 && p.ox + p.nx <= W && p.oy + p.ny <= H,
                    "patch %d: destination rectangle outside the canvas", k);
        DXH_REQUIRE(p.bw0 >= 1, "patch %d: bad block width", k);
    }
    HostPrefault pf;  // map the output pages while the device works
    pf.start(canvas, (size_t)H * W * C * esz, {{base, (size_t)H * W * C * esz}});
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    const size_t bytes = (size_t)H * W * C * esz;
    // candidate lists per kWarpBW x kWarpBH bin of the canvas, patches in table order (= the reference's loop order)
    const int nbx = (W + kWarpBW - 1) / kWarpBW, nby = (H + kWarpBH - 1) / kWarpBH;
    std::vector<int> off((size_t)nbx * nby + 1, 0), idx;
    for (int pass = 0; pass < 2; ++pass) {
        for (int k = 0; k < npatch; ++k) {
            const dxh_warp_patch& p = patches[k];
            if (p.nx <= 0 || p.ny <= 0) continue;
            const int bx0 = p.ox / kWarpBW, bx1 = (p.ox + p.nx - 1) / kWarpBW;
            const int by0 = p.oy / kWarpBH, by1 = (p.oy + p.ny - 1) / kWarpBH;
            for (int by = by0; by <= by1; ++by)
                for (int bx = bx0; bx <= bx1; ++bx) {
                    const size_t b = (size_t)by * nbx + bx;
                    if (pass == 0) ++off[b + 1];
                    else idx[off[b]++] = k;
                }
        }
        if (pass == 0) {
            for (size_t b = 0; b < (size_t)nbx * nby; ++b) off[b + 1] += off[b];
            idx.resize((size_t)off.back() + 1);
        } else {
            // the fill pass advanced every offset to the end of its list: shift back
            for (size_t b = (size_t)nbx * nby; b > 0; --b) off[b] = off[b - 1];
            off[0] = 0;
        }
    }
    void *dbase = nullptr, *dcanvas = nullptr, *dpt = nullptr, *doff = nullptr, *didx = nullptr;
    const bool base_on_device = is_device_ptr(base), canvas_on_device = is_device_ptr(canvas);
    DXH_REQUIRE(!(base_on_device && canvas_on_device && (const void*)base == (const void*)canvas),
                "a device-resident canvas must not be the base image itself");
    if ((rc = d.alloc(&dpt, sizeof(WarpPatch) * (size_t)npatch)) || (rc = d.alloc(&doff, sizeof(int) * off.size())) ||
        (rc = d.alloc(&didx, sizeof(int) * idx.size())))
        return rc;
    if (base_on_device) dbase = const_cast<void*>(base);
    else if ((rc = d.alloc(&dbase, bytes))) return rc;
    if (canvas_on_device) dcanvas = canvas;
    else if ((rc = d.alloc(&dcanvas, bytes))) return rc;
    if (!base_on_device) DXH_HIP(hipMemcpyAsync(dbase, base, bytes, hipMemcpyDefault, d.st));
    // the table with the division constants filled in (kept alive until the synchronisation below)
    std::vector<WarpPatch> table((size_t)npatch);
    for (int k = 0; k < npatch; ++k) {
        std::memcpy(&table[k], &patches[k], sizeof(WarpPatch));
        const int64_t bw = patches[k].bw0;
        table[k].pad = (bw >= 2 && bw < 65536 && patches[k].nx <= 65536) ? (int)(uint32_t)(((int64_t(1) << 32) + bw - 1) / bw) : 0;
    }
    if (npatch) DXH_HIP(hipMemcpyAsync(dpt, table.data(), sizeof(WarpPatch) * (size_t)npatch, hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(doff, off.data(), sizeof(int) * off.size(), hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(didx, idx.data(), sizeof(int) * idx.size(), hipMemcpyDefault, d.st));
    const dim3 grid(nbx, (H + kWarpRows - 1) / kWarpRows), blk(kWarpBW);
#define DXH_WARP(CH, S) hipLaunchKernelGGL((k_warp<CH, S>), grid, blk, 0, d.st, (const S*)dbase, H, W, (const WarpPatch*)dpt, \
                                           (const int*)doff, (const int*)didx, (S*)dcanvas)
#define DXH_WARP_C(S) do { if (C == 1) DXH_WARP(1, S); else DXH_WARP(3, S); } while (0)
    switch (dtype) {
        case DXH_F32: DXH_WARP_C(float); break;
        case DXH_F64: DXH_WARP_C(double); break;
        case DXH_U8: DXH_WARP_C(unsigned char); break;
        default: DXH_WARP_C(unsigned short); break;
    }
#undef DXH_WARP_C
#undef DXH_WARP
    DXH_HIP(hipGetLastError());
    pf.join();
    if (!canvas_on_device) DXH_HIP(hipMemcpyAsync(canvas, dcanvas, bytes, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));  // also keeps the host-side tables alive until their copies are done
    return DXH_OK;
}

int dxh_warp_div_probe(int device, unsigned long long seed, unsigned long long* mismatches, unsigned long long* tested) {
    DXH_REQUIRE(mismatches && tested, "null argument");
    int rc = check_device(device);
    if (rc) return rc;
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    void* dbad = nullptr;
    if ((rc = d.alloc(&dbad, sizeof(unsigned long long)))) return rc;
    DXH_HIP(hipMemsetAsync(dbad, 0, sizeof(unsigned long long), d.st));
    constexpr int kBlocks = 4096, kPer = 512;   // 2^29 denominators per round
    for (int round = 0; round < 3; ++round)
        hipLaunchKernelGGL(k_warp_div_check, dim3(kBlocks), dim3(256), 0, d.st, seed + 0x1234567ull * (unsigned)round, kPer, round,
                           (unsigned long long*)dbad);
    DXH_HIP(hipGetLastError());
    DXH_HIP(hipMemcpyAsync(mismatches, dbad, sizeof(unsigned long long), hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));
    *tested = 3ull * kBlocks * 256ull * kPer;
    return DXH_OK;
}

int dxh_warp_patches(int device, int H, int W, int C, const float* base, const dxh_warp_patch* patches, int npatch,
                     float* canvas) {
    return dxh_warp_patches_typed(device, H, W, C, DXH_F32, base, patches, npatch, canvas);
}

int dxh_phase_correlation(int device, int H, int W, int C, int dtype, const void* src, const void* dst, int npatch,
                          const int32_t* rois, double max_shift_fraction, double* shifts, int32_t* ok) {
    return dxh_phase_correlation_q(device, H, W, C, dtype, src, dst, npatch, rois, max_shift_fraction, 0.0, shifts, ok, nullptr);
}

int dxh_phase_correlation_q(int device, int H, int W, int C, int dtype, const void* src, const void* dst, int npatch,
                            const int32_t* rois, double max_shift_fraction, double min_peak_ratio, double* shifts, int32_t* ok,
                            double* peaks) {
    DXH_REQUIRE(src && dst && rois && shifts && ok, "null argument");
    DXH_REQUIRE(min_peak_ratio >= 0.0, "min_peak_ratio must be >= 0");
    DXH_REQUIRE(H >= 1 && W >= 1 && (C == 1 || C == 3) && npatch >= 1, "shape (%d, %d, %d) with %d patches not supported", H, W, C, npatch);
    DXH_REQUIRE(dtype == DXH_F32 || dtype == DXH_F64 || dtype == DXH_U8 || dtype == DXH_U16,
                "dtype %d not supported (float32, float64, uint8, uint16)", dtype);
    static_assert(sizeof(PcRoi) == 4 * sizeof(int32_t), "roi layout");
    const PcRoi* q = reinterpret_cast<const PcRoi*>(rois);
    for (int k = 0; k < npatch; ++k)
        DXH_REQUIRE(q[k].r0 >= 0 && q[k].r0 < q[k].r1 && q[k].r1 <= H && q[k].c0 >= 0 && q[k].c0 < q[k].c1 && q[k].c1 <= W,
                    "patch %d: rectangle outside the image", k);
    int rc = check_device(device);
    if (rc) return rc;
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    static const size_t width[4] = {4, 8, 1, 2};
    const size_t bytes = (size_t)H * W * C * width[dtype];
    const void* dimg[2] = {src, dst};
    for (int i = 0; i < 2; ++i) {
        if (is_device_ptr(dimg[i])) continue;
        void* p = nullptr;
        if ((rc = d.alloc(&p, bytes))) return rc;
        DXH_HIP(hipMemcpyAsync(p, dimg[i], bytes, hipMemcpyDefault, d.st));
        dimg[i] = p;
    }
    int dec = 1;
    switch (dtype) {
        case DXH_F32: rc = phase_correlation_t<float>(d, H, W, C, (const float*)dimg[0], (const float*)dimg[1], npatch, q, max_shift_fraction, min_peak_ratio, shifts, ok, peaks, &dec); break;
        case DXH_F64: rc = phase_correlation_t<double>(d, H, W, C, (const double*)dimg[0], (const double*)dimg[1], npatch, q, max_shift_fraction, min_peak_ratio, shifts, ok, peaks, &dec); break;
        case DXH_U8: rc = phase_correlation_t<unsigned char>(d, H, W, C, (const unsigned char*)dimg[0], (const unsigned char*)dimg[1], npatch, q, max_shift_fraction, min_peak_ratio, shifts, ok, peaks, &dec); break;
        default: rc = phase_correlation_t<unsigned short>(d, H, W, C, (const unsigned short*)dimg[0], (const unsigned short*)dimg[1], npatch, q, max_shift_fraction, min_peak_ratio, shifts, ok, peaks, &dec); break;
    }
    if (rc) return rc;
    for (int k = 0; k < npatch; ++k)
        if (std::min(q[k].r1 - q[k].r0, q[k].c1 - q[k].c0) / dec < 8) ok[k] = 0;   // too small to say anything
    return DXH_OK;
}

int dxh_tps_eval(int device, int p, int dout, const double* centers, const double* coeffs, const double* shift,
                 const double* scale, const int* powers, int64_t n, const double* points, int grid_w, double gx0,
                 double gy0, double gsx, double gsy, double* out) {
    DXH_REQUIRE(centers && coeffs && shift && scale && powers && out, "null argument");
    DXH_REQUIRE(p >= 1 && (dout == 1 || dout == 2), "p >= 1 and dout in {1, 2} required");
    DXH_REQUIRE(n >= 0 && (points || grid_w >= 1), "bad point set");
    const size_t smem = sizeof(double) * (2 * (size_t)kLogTab + 2 * (size_t)p + (size_t)(p + 3) * dout);
    DXH_REQUIRE(smem <= 64 * 1024, "too many centres (%d) for the LDS-resident table", p);
    int rc = check_device(device);
    if (rc) return rc;
    if (n == 0) return DXH_OK;
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, sizeof(double) * (size_t)n * dout, {{points, points ? sizeof(double) * 2 * (size_t)n : 0}});
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    void *dc = nullptr, *dk = nullptr, *dp = nullptr, *dout_ = nullptr, *dl = nullptr;
    if ((rc = d.alloc(&dc, sizeof(double) * 2 * p)) || (rc = d.alloc(&dk, sizeof(double) * (p + 3) * dout)) ||
        (rc = d.alloc(&dout_, sizeof(double) * (size_t)n * dout)) || (rc = d.alloc(&dl, sizeof(double) * 2 * kLogTab)))
        return rc;
    DXH_HIP(hipMemcpyAsync(dl, tps_log_table(), sizeof(double) * 2 * kLogTab, hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(dc, centers, sizeof(double) * 2 * p, hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(dk, coeffs, sizeof(double) * (p + 3) * dout, hipMemcpyDefault, d.st));
    if (points) {
        if ((rc = d.alloc(&dp, sizeof(double) * 2 * (size_t)n))) return rc;
        DXH_HIP(hipMemcpyAsync(dp, points, sizeof(double) * 2 * (size_t)n, hipMemcpyDefault, d.st));
    }
    TpsArgs a;
    a.logtab = (const double*)dl;
    a.centers = (const double*)dc;
    a.coeffs = (const double*)dk;
    a.points = (const double*)dp;
    for (int i = 0; i < 2; ++i) { a.shift[i] = shift[i]; a.scale[i] = scale[i]; }
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 2; ++i) a.powers[j][i] = powers[2 * j + i];
    a.p = p; a.dout = dout; a.n = n;
    a.grid_w = grid_w > 0 ? grid_w : 1;
    a.gx0 = gx0; a.gy0 = gy0; a.gsx = gsx; a.gsy = gsy;
    a.out = (double*)dout_;
    hipLaunchKernelGGL(k_tps, dim3((unsigned)((n + 255) / 256)), dim3(256), smem, d.st, a);
    DXH_HIP(hipGetLastError());
    pf.join();
    DXH_HIP(hipMemcpyAsync(out, dout_, sizeof(double) * (size_t)n * dout, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));
    return DXH_OK;
}

int dxh_tps_eval_pair(int device, int p0, const double* centers0, const double* coeffs0, const double* shift0,
                      const double* scale0, const int* powers0, int p1, const double* centers1, const double* coeffs1,
                      const double* shift1, const double* scale1, const int* powers1, int64_t grid_h, int64_t grid_w,
                      double gx0, double gy0, double gsx, double gsy, double out_scale0, double out_scale1, double* out) {
    DXH_REQUIRE(centers0 && coeffs0 && shift0 && scale0 && powers0 && centers1 && coeffs1 && shift1 && scale1 && powers1 && out,
                "null argument");
    DXH_REQUIRE(p0 >= 1 && p1 >= 1 && grid_h >= 0 && grid_w >= 1, "bad sizes");
    int rc = check_device(device);
    if (rc) return rc;
    const int64_t n = grid_h * grid_w;
    if (n == 0) return DXH_OK;
    // union of the centres, first spline's order first; equal centres are merged
    std::vector<double> table;
    table.reserve(4 * (size_t)(p0 + p1));
    for (int i = 0; i < p0; ++i) {
        table.push_back(centers0[2 * i]); table.push_back(centers0[2 * i + 1]);
        table.push_back(coeffs0[i]); table.push_back(0.0);
    }
    std::vector<char> merged((size_t)p0, 0);
    for (int j = 0; j < p1; ++j) {
        int hit = -1;
        for (int i = 0; i < p0 && hit < 0; ++i)
            if (!merged[i] && centers0[2 * i] == centers1[2 * j] && centers0[2 * i + 1] == centers1[2 * j + 1]) hit = i;
        if (hit >= 0) {
            merged[hit] = 1;
            table[4 * hit + 3] = coeffs1[j];
        } else {
            table.push_back(centers1[2 * j]); table.push_back(centers1[2 * j + 1]);
            table.push_back(0.0); table.push_back(coeffs1[j]);
        }
    }
    const int pu = (int)(table.size() / 4);
    // r^2 log r = 0.5 r^2 log r^2: the factor goes into the weights (a power of two: the same bits as multiplying afterwards)
    for (int i = 0; i < pu; ++i) { table[4 * (size_t)i + 2] *= 0.5; table[4 * (size_t)i + 3] *= 0.5; }
    const size_t smem = sizeof(double) * (2 * (size_t)kLogTab + 4 * (size_t)pu);
    DXH_REQUIRE(smem <= 64 * 1024, "too many centres (%d) for the LDS-resident table", pu);
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, sizeof(double) * 2 * (size_t)n);
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    void *dt = nullptr, *dl = nullptr, *dout_ = nullptr;
    const bool out_on_device = is_device_ptr(out);
    if ((rc = d.alloc(&dt, sizeof(double) * table.size())) || (rc = d.alloc(&dl, sizeof(double) * 2 * kLogTab))) return rc;
    if (out_on_device) dout_ = out;
    else if ((rc = d.alloc(&dout_, sizeof(double) * 2 * (size_t)n))) return rc;
    DXH_HIP(hipMemcpyAsync(dt, table.data(), sizeof(double) * table.size(), hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(dl, tps_log_table(), sizeof(double) * 2 * kLogTab, hipMemcpyDefault, d.st));
    TpsPairArgs a;
    a.logtab = (const double*)dl;
    a.table = (const double*)dt;
    a.pu = pu;
    const double* sh[2] = {shift0, shift1};
    const double* sc[2] = {scale0, scale1};
    const int* pw[2] = {powers0, powers1};
    const double* cf[2] = {coeffs0 + p0, coeffs1 + p1};
    for (int o = 0; o < 2; ++o) {
        for (int i = 0; i < 2; ++i) { a.shift[o][i] = sh[o][i]; a.scale[o][i] = sc[o][i]; }
        for (int j = 0; j < 3; ++j) {
            a.poly[o][j] = cf[o][j];
            for (int i = 0; i < 2; ++i) a.powers[o][j][i] = pw[o][2 * j + i];
        }
    }
    a.n = n;
    a.grid_w = (int)grid_w;
    a.gx0 = gx0; a.gy0 = gy0; a.gsx = gsx; a.gsy = gsy;
    a.out_scale[0] = out_scale0; a.out_scale[1] = out_scale1;
    a.out = (double*)dout_;
    hipLaunchKernelGGL(k_tps_pair, dim3((unsigned)(((n + kTpsPts - 1) / kTpsPts + 255) / 256)), dim3(256), smem, d.st, a);
    DXH_HIP(hipGetLastError());
    pf.join();
    if (!out_on_device) DXH_HIP(hipMemcpyAsync(out, dout_, sizeof(double) * 2 * (size_t)n, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));  // also keeps `table` alive until its copy is done
    return DXH_OK;
}

int dxh_mem_alloc(int device, size_t bytes, void** ptr) {
    DXH_REQUIRE(ptr, "null argument");
    int rc = check_device(device);
    if (rc) return rc;
    // image-sized arrays come from / go back to the plane cache (dxh_common.h: PlanePool): a chain of operators allocates
    // and releases a few of them per frame, always of the same sizes
    DXH_HIP(pool_malloc(ptr, bytes ? bytes : 1));
    return DXH_OK;
}

int dxh_mem_free(int device, void* ptr) {
    if (!ptr) return DXH_OK;
    int rc = check_device(device);
    if (rc) return rc;
    DXH_HIP(hipDeviceSynchronize());   // what hipFree would have waited for: nothing may still use the block
    pool_free(ptr);
    return DXH_OK;
}

int dxh_mem_trim(size_t* bytes) {
    const size_t n = pool_trim();
    if (bytes) *bytes = n;
    return DXH_OK;
}

int dxh_mem_copy(int device, void* dst, const void* src, size_t bytes) {
    DXH_REQUIRE((dst && src) || bytes == 0, "null argument");
    int rc = check_device(device);
    if (rc) return rc;
    if (bytes == 0) return DXH_OK;
    hipStream_t st;
    DXH_HIP(cached_stream(device, &st));
    const bool to_host = !is_device_ptr(dst), from_host = !is_device_ptr(src);
    constexpr size_t kChunk = size_t(32) << 20;
    if (to_host && !from_host && bytes >= 2 * kChunk) {
        // Device -> a (probably fresh) host array: a copy into never-touched pages is slowed down by the page faults
        // (384 MB: 22-36 ms instead of 7-8), touching them first costs ~15 ms.  Do both at once: a helper thread touches the
        // pages chunk by chunk, the copy of a chunk starts when its pages are mapped.
        // kTouchers threads share every chunk (page faults of one process scale over a few threads).
        constexpr int kTouchers = HostPrefault::kMaxThreads;
        const size_t nchunks = (bytes + kChunk - 1) / kChunk;
        std::unique_ptr<std::atomic<int>[]> mapped(new std::atomic<int>[nchunks]);
        for (size_t k = 0; k < nchunks; ++k) mapped[k].store(0, std::memory_order_relaxed);
        std::thread touchers[kTouchers];
        for (int t = 0; t < kTouchers; ++t)
            touchers[t] = std::thread([&, t]() {
                volatile char* c = static_cast<volatile char*>(dst);
                for (size_t k = 0; k < nchunks; ++k) {
                    const size_t off = k * kChunk, len = std::min(kChunk, bytes - off), pages = (len + 4095) / 4096;
                    const size_t b = off + pages * t / kTouchers * 4096, e = std::min(off + len, off + pages * (t + 1) / kTouchers * 4096);
                    for (size_t i = b; i < e; i += 4096) c[i] = 0;
                    if (e > b) c[e - 1] = 0;
                    mapped[k].fetch_add(1, std::memory_order_release);
                }
            });
        hipError_t e = hipSuccess;
        for (size_t k = 0; k < nchunks && e == hipSuccess; ++k) {
            const size_t off = k * kChunk, end = std::min(off + kChunk, bytes);
            while (mapped[k].load(std::memory_order_acquire) < kTouchers) std::this_thread::yield();
            e = hipMemcpyAsync(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, end - off, hipMemcpyDefault, st);
            if (e == hipSuccess && k + 1 == nchunks) e = hipStreamSynchronize(st);
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(st);
        for (auto& t : touchers) t.join();
        if (e != hipSuccess) { set_error("device-to-host copy failed: %s", hipGetErrorString(e)); return DXH_EHIP; }
        return DXH_OK;
    }
    DXH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
    DXH_HIP(hipStreamSynchronize(st));
    return DXH_OK;
}

int dxh_affine_color(int device, int in_dtype, int64_t npix, const void* img, const double* A, const double* b,
                     int clip, float* out) {
    DXH_REQUIRE(img && A && b && out, "null argument");
    DXH_REQUIRE(npix >= 0, "npix must be >= 0");
    DXH_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "in_dtype: 0 float32, 1 float64, 2 uint8, 3 uint16");
    int rc = check_device(device);
    if (rc) return rc;
    if (npix == 0) return DXH_OK;
    const size_t esz = in_dtype == 0 ? 4 : in_dtype == 1 ? 8 : in_dtype == 2 ? 1 : 2;
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, sizeof(float) * 3 * (size_t)npix, {{img, (size_t)(in_dtype == 0 ? 4 : in_dtype == 1 ? 8 : in_dtype == 2 ? 1 : 2) * 3 * (size_t)npix}});
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    void *din = nullptr, *dab = nullptr, *dout_ = nullptr;
    if ((rc = d.alloc(&din, esz * 3 * (size_t)npix)) || (rc = d.alloc(&dab, sizeof(double) * 12)) ||
        (rc = d.alloc(&dout_, sizeof(float) * 3 * (size_t)npix)))
        return rc;
    double ab[12];
    for (int i = 0; i < 9; ++i) ab[i] = A[i];
    for (int i = 0; i < 3; ++i) ab[9 + i] = b[i];
    DXH_HIP(hipMemcpyAsync(din, img, esz * 3 * (size_t)npix, hipMemcpyDefault, d.st));
    DXH_HIP(hipMemcpyAsync(dab, ab, sizeof ab, hipMemcpyDefault, d.st));
    const dim3 grid((unsigned)((npix + 255) / 256)), blk(256);
    switch (in_dtype) {
        case 0: hipLaunchKernelGGL((k_affine_color<float>), grid, blk, 0, d.st, (const float*)din, (long long)npix, 1.0, (const double*)dab, clip, (float*)dout_); break;
        case 1: hipLaunchKernelGGL((k_affine_color<double>), grid, blk, 0, d.st, (const double*)din, (long long)npix, 1.0, (const double*)dab, clip, (float*)dout_); break;
        case 2: hipLaunchKernelGGL((k_affine_color<unsigned char>), grid, blk, 0, d.st, (const unsigned char*)din, (long long)npix, 255.0, (const double*)dab, clip, (float*)dout_); break;
        default: hipLaunchKernelGGL((k_affine_color<unsigned short>), grid, blk, 0, d.st, (const unsigned short*)din, (long long)npix, 65535.0, (const double*)dab, clip, (float*)dout_); break;
    }
    DXH_HIP(hipGetLastError());
    pf.join();
    DXH_HIP(hipMemcpyAsync(out, dout_, sizeof(float) * 3 * (size_t)npix, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));
    return DXH_OK;
}

int dxh_map_coordinates(int device, int dtype, int64_t H, int64_t W, int C, const void* img, int coord_dtype,
                        const void* rows, const void* cols, int64_t n, int order, double cval, void* out) {
    DXH_REQUIRE(img && rows && cols && out, "null argument");
    DXH_REQUIRE(dtype >= 0 && dtype <= 3, "dtype: 0 float32, 1 float64, 2 uint8, 3 uint16");
    DXH_REQUIRE(coord_dtype == 0 || coord_dtype == 1, "coord_dtype: 0 float32, 1 float64");
    DXH_REQUIRE(order == 0 || order == 1, "interpolation order %d is not available on the device (0 or 1)", order);
    DXH_REQUIRE(H >= 1 && W >= 1 && C >= 1 && n >= 0, "bad shape");
    int rc = check_device(device);
    if (rc) return rc;
    if (n == 0) return DXH_OK;
    const size_t esz = dtype == 0 ? 4 : dtype == 1 ? 8 : dtype == 2 ? 1 : 2;
    const size_t csz = coord_dtype == 0 ? 4 : 8;
    HostPrefault pf;  // map the output pages while the device works
    pf.start(out, (dtype == 0 ? 4 : dtype == 1 ? 8 : dtype == 2 ? 1 : 2) * (size_t)n * C,
             {{img, (size_t)(dtype == 0 ? 4 : dtype == 1 ? 8 : dtype == 2 ? 1 : 2) * (size_t)H * W * C},
              {rows, (size_t)(coord_dtype == 0 ? 4 : 8) * (size_t)n}, {cols, (size_t)(coord_dtype == 0 ? 4 : 8) * (size_t)n}});
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    // device-resident operands are used where they are (a cached coordinate grid is 2 x 4 B per output pixel: 256 MB for
    // a 4000 x 8000 image, 5 ms over PCIe per call otherwise); host operands are staged
    void *dimg = nullptr, *dr = nullptr, *dc = nullptr, *dout_ = nullptr;
    auto stage = [&](const void* src, size_t bytes, void** dev) -> int {
        if (is_device_ptr(src)) { *dev = const_cast<void*>(src); return DXH_OK; }
        int r = d.alloc(dev, bytes);
        if (r) return r;
        DXH_HIP(hipMemcpyAsync(*dev, src, bytes, hipMemcpyDefault, d.st));
        return DXH_OK;
    };
    if ((rc = stage(img, esz * (size_t)H * W * C, &dimg)) || (rc = stage(rows, csz * (size_t)n, &dr)) ||
        (rc = stage(cols, csz * (size_t)n, &dc)))
        return rc;
    const bool out_on_device = is_device_ptr(out);
    DXH_REQUIRE(!(out_on_device && out == img), "a device-resident output must not be the input image");
    if (out_on_device) dout_ = out;
    else if ((rc = d.alloc(&dout_, esz * (size_t)n * C))) return rc;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
#define DXH_MC(T_, CT_) hipLaunchKernelGGL((k_map_coords<T_, CT_>), grid, blk, 0, d.st, (const T_*)dimg, (long long)H, (long long)W, C, \
                                           (const CT_*)dr, (const CT_*)dc, (long long)n, order, cval, (T_*)dout_)
    if (coord_dtype == 0) {
        switch (dtype) { case 0: DXH_MC(float, float); break; case 1: DXH_MC(double, float); break;
                         case 2: DXH_MC(unsigned char, float); break; default: DXH_MC(unsigned short, float); break; }
    } else {
        switch (dtype) { case 0: DXH_MC(float, double); break; case 1: DXH_MC(double, double); break;
                         case 2: DXH_MC(unsigned char, double); break; default: DXH_MC(unsigned short, double); break; }
    }
#undef DXH_MC
    DXH_HIP(hipGetLastError());
    pf.join();
    if (!out_on_device) DXH_HIP(hipMemcpyAsync(out, dout_, esz * (size_t)n * C, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));
    return DXH_OK;
}

int dxh_resize(int device, int dtype, int64_t H, int64_t W, int C, const void* src, int64_t DH, int64_t DW,
               const dxh_resize_plan* plan, void* dst) {
    DXH_REQUIRE(src && dst && plan, "null argument");
    DXH_REQUIRE(plan->struct_size == (int32_t)sizeof(dxh_resize_plan), "dxh_resize_plan size mismatch");
    DXH_REQUIRE(dtype == 0 || dtype == 1 || dtype == 2, "dtype: 0 float32, 1 float64, 2 uint8");
    DXH_REQUIRE(H >= 1 && W >= 1 && C >= 1 && DH >= 1 && DW >= 1, "bad shape");
    DXH_REQUIRE(H < (1 << 30) && W < (1 << 30) && DH < (1 << 30) && DW < (1 << 30), "image too large");
    DXH_REQUIRE(plan->mode >= 0 && plan->mode <= 3, "unknown resize mode %d", plan->mode);
    int rc = check_device(device);
    if (rc) return rc;
    const size_t esz = dtype == 0 ? 4 : dtype == 1 ? 8 : 1;
    HostPrefault pf;  // map the output pages while the device works
    pf.start(dst, (dtype == 0 ? 4 : dtype == 1 ? 8 : 1) * (size_t)DH * DW * C, {{src, (size_t)(dtype == 0 ? 4 : dtype == 1 ? 8 : 1) * (size_t)H * W * C}});
    Dev d;
    DXH_HIP(cached_stream(device, &d.st));
    ResizePlanDev p{};
    p.mode = plan->mode; p.isx = plan->isx; p.isy = plan->isy;
    auto up = [&](const void* host, size_t bytes, const void** devp) -> int {
        *devp = nullptr;
        if (!host || !bytes) return DXH_OK;
        void* q = nullptr;
        int r = d.alloc(&q, bytes);
        if (r) return r;
        DXH_HIP(hipMemcpyAsync(q, host, bytes, hipMemcpyDefault, d.st));
        *devp = q;
        return DXH_OK;
    };
    if (plan->mode == 0 || plan->mode == 1) {
        DXH_REQUIRE(plan->xi && plan->yi && (plan->mode == 0 || (plan->xw && plan->yw)), "tap tables missing");
        for (int64_t x = 0; x < DW; ++x) DXH_REQUIRE(plan->xi[x] >= 0 && plan->xi[x] < W, "x tap %lld out of range", (long long)x);
        for (int64_t y = 0; y < DH; ++y) DXH_REQUIRE(plan->yi[y] >= 0 && plan->yi[y] < H, "y tap %lld out of range", (long long)y);
        if ((rc = up(plan->xi, sizeof(int) * DW, (const void**)&p.xi)) || (rc = up(plan->yi, sizeof(int) * DH, (const void**)&p.yi)) ||
            (rc = up(plan->xw, sizeof(float) * DW, (const void**)&p.xw)) || (rc = up(plan->yw, sizeof(float) * DH, (const void**)&p.yw)))
            return rc;
    } else if (plan->mode == 2) {
        DXH_REQUIRE(plan->isx >= 1 && plan->isy >= 1, "block size must be positive");
    } else {
        DXH_REQUIRE(plan->xptr && plan->xsrc && plan->xalpha && plan->yptr && plan->ysrc && plan->yalpha, "area tables missing");
        const int nxe = plan->xptr[DW], nye = plan->yptr[DH];
        for (int64_t x = 0; x < DW; ++x) DXH_REQUIRE(plan->xptr[x] <= plan->xptr[x + 1], "x table not monotone");
        for (int64_t y = 0; y < DH; ++y) DXH_REQUIRE(plan->yptr[y] <= plan->yptr[y + 1], "y table not monotone");
        for (int q = 0; q < nxe; ++q) DXH_REQUIRE(plan->xsrc[q] >= 0 && plan->xsrc[q] < W, "x table entry %d out of range", q);
        for (int q = 0; q < nye; ++q) DXH_REQUIRE(plan->ysrc[q] >= 0 && plan->ysrc[q] < H, "y table entry %d out of range", q);
        if ((rc = up(plan->xptr, sizeof(int) * (DW + 1), (const void**)&p.xptr)) || (rc = up(plan->xsrc, sizeof(int) * nxe, (const void**)&p.xsrc)) ||
            (rc = up(plan->xalpha, sizeof(float) * nxe, (const void**)&p.xalpha)) || (rc = up(plan->yptr, sizeof(int) * (DH + 1), (const void**)&p.yptr)) ||
            (rc = up(plan->ysrc, sizeof(int) * nye, (const void**)&p.ysrc)) || (rc = up(plan->yalpha, sizeof(float) * nye, (const void**)&p.yalpha)))
            return rc;
    }
    void *dsrc = nullptr, *ddst = nullptr;
    const size_t nsrc = esz * (size_t)H * W * C, ndst = esz * (size_t)DH * DW * C;
    if ((rc = d.alloc(&dsrc, nsrc)) || (rc = d.alloc(&ddst, ndst))) return rc;
    DXH_HIP(hipMemcpyAsync(dsrc, src, nsrc, hipMemcpyDefault, d.st));
    const long long total = (long long)DH * DW * C;
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL((k_resize<float>), grid, blk, 0, d.st, (const float*)dsrc, (int)H, (int)W, C, p, (int)DH, (int)DW, (float*)ddst); break;
        case 1: hipLaunchKernelGGL((k_resize<double>), grid, blk, 0, d.st, (const double*)dsrc, (int)H, (int)W, C, p, (int)DH, (int)DW, (double*)ddst); break;
        default: hipLaunchKernelGGL((k_resize<unsigned char>), grid, blk, 0, d.st, (const unsigned char*)dsrc, (int)H, (int)W, C, p, (int)DH, (int)DW, (unsigned char*)ddst); break;
    }
    DXH_HIP(hipGetLastError());
    pf.join();
    DXH_HIP(hipMemcpyAsync(dst, ddst, ndst, hipMemcpyDefault, d.st));
    DXH_HIP(hipStreamSynchronize(d.st));
    return DXH_OK;
}

}  // extern "C"
