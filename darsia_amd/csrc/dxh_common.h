// Shared helpers of libdarsia_hip: error reporting, device-side vector access,
// cross-lane shifts.  gfx950 only (wave64, DPP wave shifts).
#pragma once

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <cstdlib>
#include <vector>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>

#include "../../include/darsia_hip.h"

namespace dxh {

void set_error(const char* fmt, ...);

#define DXH_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ::dxh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? DXH_ENOMEM : DXH_EHIP;                     \
        }                                                                                   \
    } while (0)

#define DXH_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            ::dxh::set_error(__VA_ARGS__); \
            return DXH_EINVAL;            \
        }                                 \
    } while (0)

// A non-blocking stream per (host thread, device), created on first use and kept: hipStreamCreate costs 10-15 ms on
// this stack (tools/alloc_probe.hip: more than the 11 device allocations, memsets and frees of a whole call together),
// and the synchronous entry points need a stream only for the duration of a call.
inline hipError_t cached_stream(int device, hipStream_t* out) {
    constexpr int kMax = 64;
    thread_local hipStream_t streams[kMax] = {};
    if (device < 0 || device >= kMax) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    if (!streams[device]) {
        hipError_t e = hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    *out = streams[device];
    return hipSuccess;
}

// Plane cache: hipFree of a few hundred MB costs 0.1 - 0.3 ms and waits for the device, a whole call releases 10 - 30
// planes.  Freed planes are kept (exact-size reuse, newest first) up to a budget of 1/16 of the device memory
// (DXH_POOL_MB overrides; 0 turns the cache off); pool_trim() returns everything to the runtime.  Blocks come back with
// their old contents: callers clear what they need cleared.
struct PlanePool {
    struct Blk { void* p; size_t n; int dev; };
    std::mutex m;
    std::vector<Blk> idle;
    std::vector<Blk> live;   // blocks handed out (a few dozen at most: linear search)
    size_t cached = 0;
    long long budget = -1;   // bytes; -1: not yet determined
};
inline PlanePool& plane_pool() {
    static PlanePool* p = new PlanePool();   // never destroyed: contexts may be released during process teardown
    return *p;
}
inline hipError_t pool_malloc(void** out, size_t n) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    PlanePool& pl = plane_pool();
    {
        std::lock_guard<std::mutex> lock(pl.m);
        for (size_t i = pl.idle.size(); i-- > 0;)
            if (pl.idle[i].n == n && pl.idle[i].dev == dev) {
                *out = pl.idle[i].p;
                pl.live.push_back(pl.idle[i]);
                pl.cached -= n;
                pl.idle.erase(pl.idle.begin() + i);
                return hipSuccess;
            }
    }
    e = hipMalloc(out, n);
    if (e == hipErrorOutOfMemory) {
        // give the cache back and try once more
        std::vector<PlanePool::Blk> drop;
        {
            std::lock_guard<std::mutex> lock(pl.m);
            drop.swap(pl.idle);
            pl.cached = 0;
        }
        for (auto& b : drop) (void)hipFree(b.p);
        (void)hipGetLastError();
        e = hipMalloc(out, n);
    }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(pl.m);
    pl.live.push_back({*out, n, dev});
    return hipSuccess;
}
// the caller guarantees that no work still uses the block
inline void pool_free(void* p) {
    if (!p) return;
    PlanePool& pl = plane_pool();
    std::vector<PlanePool::Blk> drop;
    bool known = false;
    {
        std::lock_guard<std::mutex> lock(pl.m);
        if (pl.budget < 0) {
            size_t fr = 0, tot = 0;
            pl.budget = hipMemGetInfo(&fr, &tot) == hipSuccess ? (long long)(tot / 16) : 0;
            if (const char* e = getenv("DXH_POOL_MB")) pl.budget = atoll(e) * (1ll << 20);
        }
        for (size_t i = 0; i < pl.live.size(); ++i)
            if (pl.live[i].p == p) {
                PlanePool::Blk b = pl.live[i];
                pl.live.erase(pl.live.begin() + i);
                known = true;
                if ((long long)b.n > pl.budget) { drop.push_back(b); break; }
                pl.idle.push_back(b);
                pl.cached += b.n;
                while ((long long)pl.cached > pl.budget) {   // oldest out
                    drop.push_back(pl.idle.front());
                    pl.cached -= pl.idle.front().n;
                    pl.idle.erase(pl.idle.begin());
                }
                break;
            }
    }
    if (!known) { (void)hipFree(p); return; }
    for (auto& b : drop) (void)hipFree(b.p);
}
inline size_t pool_trim() {
    PlanePool& pl = plane_pool();
    std::vector<PlanePool::Blk> drop;
    {
        std::lock_guard<std::mutex> lock(pl.m);
        drop.swap(pl.idle);
        pl.cached = 0;
    }
    size_t n = 0;
    for (auto& b : drop) {
        (void)hipFree(b.p);
        n += b.n;
    }
    return n;
}

// Image arrays handed to the C ABI may live on the host or on the device (dxh_mem_alloc): every copy of caller data
// uses hipMemcpyDefault, which resolves the direction from the pointers.  This tells the two apart where host-only
// work (page touching) is concerned; an address the runtime does not know is ordinary host memory.
inline bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

// Maps the pages of a (large) host output buffer on a helper thread while the device works.  A device-to-host
// copy into never-touched pages is slowed down by the page faults (384 MB: 22-36 ms instead of 7 ms, measured); the
// buffer is going to be overwritten completely, so writing one zero per page beforehand is harmless.  join() before
// the copy is enqueued.
struct HostPrefault {
    static constexpr int kMaxThreads = 4;  // page faults of one process scale over a few threads, then the mm lock binds
    std::thread th[kMaxThreads];
    // `inputs`: host buffers the call still has to read -- an output that overlaps one of them (in-place use of the
    // C ABI) must not be touched
    void start(void* p, size_t n, std::initializer_list<std::pair<const void*, size_t>> inputs = {}) {
        if (!p || n < (size_t(8) << 20) || is_device_ptr(p)) return;
        const char* lo = static_cast<const char*>(p);
        for (const auto& in : inputs) {
            const char* a = static_cast<const char*>(in.first);
            if (a && a < lo + n && lo < a + in.second) return;
        }
        join();
        const size_t pages = (n + 4095) / 4096;
        const int nt = (int)std::min<size_t>(kMaxThreads, std::max<size_t>(1, n >> 25));  // one thread per 32 MB
        for (int t = 0; t < nt; ++t) {
            const size_t b = pages * t / nt * 4096, e = std::min(n, pages * (t + 1) / nt * 4096);
            th[t] = std::thread([p, b, e]() {
                volatile char* c = static_cast<volatile char*>(p);
                for (size_t i = b; i < e; i += 4096) c[i] = 0;
                if (e > b) c[e - 1] = 0;
            });
        }
    }
    void join() {
        for (auto& t : th)
            if (t.joinable()) t.join();
    }
    ~HostPrefault() { join(); }
};

// ---- roctx ranges for rocprofv3 --marker-trace timelines --------------------------------------------------------
// Named host-side ranges around the TV loop, the halo exchange and the band copies.  Off unless DXH_ROCTX is set (the marker
// library is then resolved with dlopen once: librocprofiler-sdk-roctx, else the older libroctx64); without it a range is two
// predictable branches.
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    static const RoctxApi& get() {
        static const RoctxApi api = [] {
            RoctxApi a;
            if (!getenv("DXH_ROCTX")) return a;
            for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
                void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (!h) continue;
                *(void**)(&a.push) = dlsym(h, "roctxRangePushA");
                *(void**)(&a.pop) = dlsym(h, "roctxRangePop");
                if (a.push && a.pop) break;
                a.push = nullptr;
                a.pop = nullptr;
            }
            return a;
        }();
        return api;
    }
};
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(RoctxApi::get().push != nullptr) { if (on) RoctxApi::get().push(name); }
    RoctxRange(const char* fmt, long long a, long long b = 0) : on(RoctxApi::get().push != nullptr) {
        if (on) {
            char buf[96];
            snprintf(buf, sizeof buf, fmt, a, b);
            RoctxApi::get().push(buf);
        }
    }
    ~RoctxRange() { if (on) RoctxApi::get().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

constexpr int kWave = 64;

// ---- per-lane vectors: 16 bytes per lane for both dtypes ---------------------------------
template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
    static constexpr int N = 4;
    using type = float4;
};
template <>
struct VecOf<double> {
    static constexpr int N = 2;
    using type = double2;
};

template <typename T>
__device__ __forceinline__ void load_vec(const T* p, T (&v)[VecOf<T>::N]) {
    using V = typename VecOf<T>::type;
    V t = *reinterpret_cast<const V*>(p);
    const T* s = reinterpret_cast<const T*>(&t);
#pragma unroll
    for (int j = 0; j < VecOf<T>::N; ++j) v[j] = s[j];
}

template <typename T>
__device__ __forceinline__ void store_vec(T* p, const T (&v)[VecOf<T>::N]) {
    using V = typename VecOf<T>::type;
    V t;
    T* s = reinterpret_cast<T*>(&t);
#pragma unroll
    for (int j = 0; j < VecOf<T>::N; ++j) s[j] = v[j];
    *reinterpret_cast<V*>(p) = t;
}

// opaque (volatile) 16-byte accesses: LDS traffic the optimiser must not forward into registers
template <typename T>
__device__ __forceinline__ void load_vec_opaque(const T* p, T (&v)[VecOf<T>::N]) {
    typedef T EV __attribute__((ext_vector_type(VecOf<T>::N)));
    EV t = *reinterpret_cast<const volatile EV*>(p);
#pragma unroll
    for (int j = 0; j < VecOf<T>::N; ++j) v[j] = t[j];
}
template <typename T>
__device__ __forceinline__ void store_vec_opaque(T* p, const T (&v)[VecOf<T>::N]) {
    typedef T EV __attribute__((ext_vector_type(VecOf<T>::N)));
    EV t;
#pragma unroll
    for (int j = 0; j < VecOf<T>::N; ++j) t[j] = v[j];
    *reinterpret_cast<volatile EV*>(p) = t;
}

// streaming (non-temporal) 16-byte store: the written lines are not read again by this launch
template <typename T>
__device__ __forceinline__ void store_vec_nt(T* p, const T (&v)[VecOf<T>::N]) {
    typedef T EV __attribute__((ext_vector_type(VecOf<T>::N)));
    EV t;
#pragma unroll
    for (int j = 0; j < VecOf<T>::N; ++j) t[j] = v[j];
    __builtin_nontemporal_store(t, reinterpret_cast<EV*>(p));
}

// ---- cross-lane moves -----------------------------------------------------------------------
// lane_from<K>(x): value of x held by lane (lane + K), K in {-3..3}; lanes whose source is outside
// the wave get an unspecified value (callers never use it for a stored result).
// DPP form: wave_shr:1 (0x138) moves data to the next higher lane, wave_shl:1 (0x130) to the next
// lower one; one v_mov_b32_dpp per 32-bit word and step, no LDS crossbar involved.
template <bool DPP>
__device__ __forceinline__ int lane_step_up(int x) {  // result(lane) = x(lane - 1)
    if constexpr (DPP)  // old = 0 with bound_ctrl: lane 0 reads 0, no copy of x is needed before the move
        return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, true);
    else
        return __shfl_up(x, 1, kWave);
}
template <bool DPP>
__device__ __forceinline__ int lane_step_down(int x) {  // result(lane) = x(lane + 1)
    if constexpr (DPP)
        return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, true);
    else
        return __shfl_down(x, 1, kWave);
}

template <int K, bool DPP>
__device__ __forceinline__ int lane_from_i32(int x) {
    if constexpr (K == 0) {
        return x;
    } else if constexpr (K < 0) {
        return lane_from_i32<K + 1, DPP>(lane_step_up<DPP>(x));
    } else {
        return lane_from_i32<K - 1, DPP>(lane_step_down<DPP>(x));
    }
}

template <int K, bool DPP>
__device__ __forceinline__ float lane_from(float x) {
    return __builtin_bit_cast(float, lane_from_i32<K, DPP>(__builtin_bit_cast(int, x)));
}
template <int K, bool DPP>
__device__ __forceinline__ double lane_from(double x) {
    long long b = __builtin_bit_cast(long long, x);
    int lo = lane_from_i32<K, DPP>(static_cast<int>(b & 0xffffffffLL));
    int hi = lane_from_i32<K, DPP>(static_cast<int>(b >> 32));
    long long r = (static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo);
    return __builtin_bit_cast(double, r);
}

constexpr int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Element J of the lane's vector shifted by DELTA elements along the row:
// w[J] = row[x0 + J + DELTA] where the lane holds row[x0 .. x0+N).
template <int DELTA, int J, bool DPP, typename T, int N>
__device__ __forceinline__ T shifted_elem(const T (&v)[N]) {
    constexpr int idx = J + DELTA;
    constexpr int lo = floor_div(idx, N);
    constexpr int e = idx - lo * N;
    return lane_from<lo, DPP>(v[e]);
}

template <int DELTA, bool DPP, typename T>
__device__ __forceinline__ void shifted(const T (&v)[VecOf<T>::N], T (&w)[VecOf<T>::N]) {
    if constexpr (VecOf<T>::N == 4) {
        w[0] = shifted_elem<DELTA, 0, DPP>(v);
        w[1] = shifted_elem<DELTA, 1, DPP>(v);
        w[2] = shifted_elem<DELTA, 2, DPP>(v);
        w[3] = shifted_elem<DELTA, 3, DPP>(v);
    } else {
        w[0] = shifted_elem<DELTA, 0, DPP>(v);
        w[1] = shifted_elem<DELTA, 1, DPP>(v);
    }
}

// a + row[x0 + J + DELTA] (SUB: a - ...) where the lane holds row[x0 .. x0+N) in v.  When the element lives in
// a neighbour lane (float), the DPP move is folded into the add itself -- LLVM's DPP combiner leaves most of
// these as v_mov_b32_dpp + v_add_f32 -- one VALU instead of two.  The leading s_nop covers the two wait states
// a DPP read needs after a VALU write of its source, which the hazard recogniser cannot see inside an asm.
template <int DELTA, int J, bool SUB, typename T, int N>
__device__ __forceinline__ T add_shifted_elem(T a, const T (&v)[N]) {
    constexpr int idx = J + DELTA;
    constexpr int lo = floor_div(idx, N);
    constexpr int e = idx - lo * N;
    if constexpr (sizeof(T) == 4 && (lo == -1 || lo == 1)) {
        float r;
        if constexpr (lo == -1 && !SUB)
            asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v[e]), "v"(a));
        else if constexpr (lo == -1 && SUB)
            asm("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v[e]), "v"(a));
        else if constexpr (lo == 1 && !SUB)
            asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v[e]), "v"(a));
        else
            asm("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v[e]), "v"(a));
        return r;
    } else {
        const T w = lane_from<lo, true>(v[e]);
        return SUB ? a - w : a + w;
    }
}
// does element J of shifted<DELTA> come from the same (lane, element) as element J2 of shifted<DELTA2>?
template <int DELTA, int J, int DELTA2, int J2, int N>
constexpr bool same_shift_source() {
    return floor_div(J + DELTA, N) == floor_div(J2 + DELTA2, N) && (J + DELTA) - floor_div(J + DELTA, N) * N == (J2 + DELTA2) - floor_div(J2 + DELTA2, N) * N;
}

// ---- arithmetic shared by all kernels -------------------------------------------------------
template <typename T>
__device__ __forceinline__ T sgn(T x) {
    return static_cast<T>((x > T(0)) - (x < T(0)));
}
template <typename T>
__device__ __forceinline__ T t_max0(T x) {  // np.maximum(x, 0) for non-NaN x
    return x > T(0) ? x : T(0);
}
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return fabs(x); }
__device__ __forceinline__ float t_copysign(float m, float s) { return copysignf(m, s); }
__device__ __forceinline__ double t_copysign(double m, double s) { return copysign(m, s); }
__device__ __forceinline__ float t_clamp_sym(float x, float k) { return __builtin_amdgcn_fmed3f(x, -k, k); }
__device__ __forceinline__ double t_clamp_sym(double x, double k) { return fmin(fmax(x, -k), k); }
// Correctly rounded square root of a sum of squares (x >= 0 or NaN).  float: the compiler's IEEE expansion is ~16
// instructions (scaling for denormals, v_sqrt_f32, two neighbour residuals, selects).  For x in [2^-100, inf) the
// five-instruction chain  y = rsq(x); g = x y; h = y / 2; r = fma(-g, g, x); s = fma(r, h, g)  gives the SAME float for
// every input -- decided by trying all of them on the device (k_sqrt_check / dxh_sqrt_probe, 1.9e9 values, a few
// milliseconds; tests/test_tvd_gpu.py runs it).  x = 0 falls out right (rsq sees max(x, 2^-100): g = 0 y = 0); inputs
// in (0, 2^-100) -- |q| below 1e-15 -- and +inf take the IEEE path on a rarely executed branch.
constexpr float kSqrtFastMin = 0x1p-100f;
__device__ __forceinline__ float t_sqrt_chain(float x) {
    const float y = __builtin_amdgcn_rsqf(fmaxf(x, kSqrtFastMin));
    const float g = x * y, h = 0.5f * y;
    const float r = fmaf(-g, g, x);
    return fmaf(r, h, g);
}
__device__ __forceinline__ float t_sqrt(float x) {
#ifdef DXH_SQRT_IEEE  // A/B builds: the compiler's expansion everywhere
    return sqrtf(x);
#endif
    float s = t_sqrt_chain(x);
    const unsigned u = __float_as_uint(x);
    // outside [2^-100, inf) and not zero (NaN and negative bit patterns included): the IEEE square root
    if (__builtin_expect((u - 0x0D800000u) >= (0x7F800000u - 0x0D800000u) && u != 0u, 0)) s = sqrtf(x);
    return s;
}
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }

// Inside a rarely taken, wave-uniform `if` block: keeps the block a real s_cbranch instead of a run of
// v_cndmask (if-conversion) that every wave would pay on every row.
#define DXH_UNIFORM_BRANCH() asm volatile("")

// ---- per-lane selects through bit masks held in VGPRs ----------------------------------------------
// A lane pattern that never changes (channel position, strip edge) kept as an SGPR-pair mask for v_cndmask
// costs scalar registers the long pipelines do not have (the masks get spilled to VGPR lanes and come back
// through v_readlane, one VALU each); as 0 / ~0 words in a VGPR the same select is one v_bfi_b32 per word.
__device__ __forceinline__ uint32_t lane_mask_bits(bool c) {
    uint32_t m = c ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));  // opaque: keeps the optimiser from turning the mask back into an i1 select
    return m;
}
__device__ __forceinline__ float sel_bits(uint32_t m, float a, float b) {  // m ? a : b
    float r;  // spelled out: the optimiser splits (m & a) | (~m & b) into and / and_or with a hoisted ~m
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double sel_bits(uint32_t m, double a, double b) {
    const uint64_t ua = __builtin_bit_cast(uint64_t, a), ub = __builtin_bit_cast(uint64_t, b);
    uint32_t lo, hi;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(m), "v"(static_cast<uint32_t>(ua)), "v"(static_cast<uint32_t>(ub)));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(m), "v"(static_cast<uint32_t>(ua >> 32)), "v"(static_cast<uint32_t>(ub >> 32)));
    return __builtin_bit_cast(double, (static_cast<uint64_t>(hi) << 32) | lo);
}

// ---- division by a wave-invariant divisor ----------------------------------------------------------
// x / d with r = RN(1/d) supplied by the host: the refinement chain of correctly rounded division
//     q0 = x r;  e0 = x - d q0;  q1 = q0 + e0 r;  e1 = x - d q1;  q = q1 + e1 r        (fma: exact residuals)
// q1 is within half an ulp (+2^-47 relative) of x/d, hence faithful, and one more residual correction
// with a reciprocal of relative error < 2^-p then yields RN(x/d) (Markstein's theorem; it assumes
// that no intermediate under- or overflows).  The residuals are exact as long as ulp(x)*2^-p is
// representable, i.e. |x| >= 2^(emin+2p): numerators that are non-zero and smaller than DivSafe::tiny,
// are detected by the caller (cheap screen: min |x| over the lane's numerators; exact zeros are harmless
// but trip the screen, so a precise test follows before the step is redone with the IEEE division).
// Non-finite numerators give NaN here where the IEEE division would give +-inf; an image with an inf
// turns into NaNs within one iteration in the reference as well (inf - inf in the differences).
// 5 VALU instead of the 10-instruction v_div_scale/v_rcp/v_div_fmas/v_div_fixup sequence.
// For most divisors q1 is already RN(x/d) for EVERY x; in float that is decidable by trying all 2^23
// mantissas (k_div3_check, run once per coefficient set), and the pipelines then stop after three operations.
template <typename T>
struct DivSafe;
template <>
struct DivSafe<float> {
    static constexpr float tiny = 0x1p-100f;  // > 2^(-149 + 46)
    static constexpr float huge = 0x1p+100f;
};
template <>
struct DivSafe<double> {
    static constexpr double tiny = 0x1p-960;  // > 2^(-1074 + 104)
    static constexpr double huge = 0x1p+960;
};
__device__ __forceinline__ float t_fma(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return fma(a, b, c); }

template <typename T>
__device__ __forceinline__ T div_by_const3(T x, T d, T r) {  // first three operations: faithful, mostly exact
    const T q0 = x * r;
    const T e0 = t_fma(-d, q0, x);
    return t_fma(e0, r, q0);
}
template <typename T>
__device__ __forceinline__ T div_refine(T q1, T x, T d, T r) {  // the second residual correction
    const T e1 = t_fma(-d, q1, x);
    return t_fma(e1, r, q1);
}
template <typename T>
__device__ __forceinline__ T div_by_const(T x, T d, T r) {
    return div_refine(div_by_const3(x, d, r), x, d, r);
}
__device__ __forceinline__ float t_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return fmin(a, b); }
// min(|a|, |b|, |c|) in one instruction (float): the screen below runs once per row and stage
__device__ __forceinline__ float abs_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ double abs_min3(double a, double b, double c) { return fmin(fmin(fabs(a), fabs(b)), fabs(c)); }
// smallest magnitude among the 2 N numerators of a lane
template <typename T, int N>
__device__ __forceinline__ T abs_min_all(const T (&x)[N], const T (&y)[N]) {
    T lo = abs_min3(x[0], y[0], x[N - 1]);
    lo = abs_min3(lo, y[N - 1], y[N - 1]);
    if constexpr (N == 4) {
        lo = abs_min3(lo, x[1], y[1]);
        lo = abs_min3(lo, x[2], y[2]);
    }
    return lo;
}
// numerator x is outside div_by_const's proven range: non-zero but tiny (x == 0 is exact: every term is 0)
template <typename T>
__device__ __forceinline__ bool div_unsafe(T x) {
    return t_abs(x) < DivSafe<T>::tiny && x != T(0);
}

// ---- several quotients with one per-element divisor (Chambolle's p / norm) -----------------------------
// float: r = rcp(d) refined once,  r = fma(fma(-d, rcp(d), 1), rcp(d), rcp(d)),  IS the correctly rounded reciprocal
// RN(1/d) for every float d in [2^-100, 2^100] -- decided by trying all of them on the device (k_rcp_check /
// dxh_rcp_probe; the test suite runs it).  With an exact RN(1/d) the five-operation refinement of div_by_const above
// yields RN(x/d) (same theorem, same provisos: x zero or not tiny, nothing under- or overflows), so NQ quotients cost
// 3 + 5 NQ instructions instead of ~14 NQ for the IEEE sequence.  The proven range used here: d and every non-zero |x|
// within [2^-62, 2^62), hence quotients within (2^-124, 2^124) -- all normal.  The screen lets exact zeros through
// (bits - 1 wraps a zero to the largest unsigned); anything else takes the IEEE division on a rarely executed branch.
// A single quotient gains nothing over the IEEE sequence once the screen is paid (the isotropic shrinkage keeps it);
// double keeps the IEEE division (no exhaustive proof there).
constexpr float kDivLo = 0x1p-62f, kDivHi = 0x1p+62f;
__device__ __forceinline__ float t_rcp_rn(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ bool div_elem_safe(float x, float d) {
    const float ax = fabsf(x);
    return (x == 0.0f || (ax >= kDivLo && ax < kDivHi)) && d >= kDivLo && d < kDivHi;
}
// q[i] = x[i] / d for NQ numerators and one positive divisor
template <typename T, int NQ>
__device__ __forceinline__ void div_shared(const T (&x)[NQ], T d, T (&q)[NQ]) {
#ifdef DXH_DIV_IEEE  // A/B builds
    constexpr bool fast = false;
#else
    constexpr bool fast = std::is_same<T, float>::value && NQ >= 2;
#endif
    if constexpr (fast) {
        const float r = t_rcp_rn(d);
        // screen: every numerator zero or >= 2^-62 in magnitude (bits << 1 drops the sign, - 1 wraps a zero to the
        // largest unsigned, so zeros never lower the minimum); their magnitudes sum to less than 2^62 (a NaN or inf
        // numerator makes the sum fail the comparison); the divisor within [2^-62, 2^62)
        unsigned lo = 0xFFFFFFFFu;
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            q[i] = div_by_const<float>(x[i], d, r);
            lo = min(lo, (__float_as_uint(x[i]) << 1) - 1u);
            sum = sum + fabsf(x[i]);
        }
        const bool ok = lo >= (__float_as_uint(kDivLo) << 1) - 1u && sum < kDivHi && d >= kDivLo && d < kDivHi;
        if (__builtin_expect(!ok, 0)) {
#pragma unroll
            for (int i = 0; i < NQ; ++i)
                if (!div_elem_safe(x[i], d)) q[i] = x[i] / d;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NQ; ++i) q[i] = x[i] / d;
    }
}

// (d, b) of one pixel from its stored state q = b_old + D(u) ("dub" in the reference):
//   anisotropic  d = shrink(q, k), b = q - d                    split_bregman_tvd.py:80-81,143-146
//   isotropic    s = |q|_2, f = max(s-k,0)/(s+1e-18), d = q f, b = q (1-f)       :107-126
// FASTSIGN (callers guarantee k >= 0): max(|q| - k, 0) * sign(q) == q - clamp(q, -k, k) bit for bit up to
// the sign of a zero result (which no later operation can tell apart: b = q - d, t = ell (b - d) and the
// downloaded d compare equal) -- q > k: RN(q - k); q < -k: RN(q + k) = -RN(|q| - k); else q - q = 0.  One
// v_med3_f32 and a (packable) subtraction instead of and / sub / max / bfi.  For k < 0 or NaN the forms differ.
// ---- the isotropic shrink factor  max(s - k, 0) / (s + 1e-18),  s = sqrt(x)  (split_bregman_tvd.py:119-121) --------------
// float: the square-root chain above starts from y = rsq(x) ~ 1 / s, so the reciprocal of the divisor is one Newton step
// away:  rho = fma(fma(-s, y, 1), y, y).  For x in [2^-60, 2^60) the divisor s + 1e-18f IS s (1e-18 < ulp(s) / 2 from
// s = 2^-30 on) and rho IS RN(1 / s) for every such x but the 120 whose root has an all-ones significand (screened out
// below) -- decided by trying all of them on the device (k_isofac_rcp_check /
// dxh_isofac_probe; the test suite runs it).  With an exact RN(1 / s) the five-operation refinement (div_by_const, the
// same theorem and provisos as div_shared: numerator zero or not tiny, nothing under- or overflows -- the numerator
// s - k is zero or at least ulp(s) / 2 >= 2^-54 here, the quotient lies in [0, 1]) yields RN(num / s): 2 + 5 operations
// instead of the ~13 issue slots of the IEEE sequence (v_div_scale x 2, quarter-rate v_rcp, 4 fma, v_div_fmas,
// v_div_fixup).  x == 0 (flat regions: common) goes the same way: the numerator is 0 and so is every term.
// Anything else -- x outside the range, k < 0 or NaN -- takes the IEEE expressions on a rarely executed branch.
// The whole routine is compared with the IEEE expressions on ~1e10 pseudo-random (x, k) pairs by the same probe.
constexpr float kIsoLo = 0x1p-60f, kIsoHi = 0x1p+60f;
__device__ __forceinline__ void iso_s_fac_fast(float x, float k, float& s, float& fac, bool& ok) {
    const float y = __builtin_amdgcn_rsqf(fmaxf(x, kSqrtFastMin));
    const float g = x * y, h = 0.5f * y;
    s = fmaf(fmaf(-g, g, x), h, g);                      // t_sqrt_chain(x)
    const float rho = fmaf(fmaf(-s, y, 1.0f), y, y);     // RN(1 / s) on [2^-60, 2^60) (exhaustive check)
    const float num = (s - k) > 0.0f ? (s - k) : 0.0f;   // t_max0
    fac = div_by_const<float>(num, s, rho);
    // (the one s per binade whose significand is all ones is the classical exception of the Newton step: there rho is
    // one ulp short of RN(1 / s) -- 120 of the 1.0e9 inputs, found by the exhaustive check; they take the IEEE branch)
    // k >= 0 (what mu / ell is for any sensible input; false for NaN) keeps the numerator within [0, s]: nothing overflows
    ok = k >= 0.0f && (((x >= kIsoLo && x < kIsoHi) && (__float_as_uint(s) & 0x7FFFFFu) != 0x7FFFFFu) || x == 0.0f);
}
template <typename T>
__device__ __forceinline__ void iso_s_fac(T x, T k, T& s, T& fac) {
#ifndef DXH_ISO_IEEE  // A/B builds: the IEEE expressions everywhere
    if constexpr (std::is_same<T, float>::value) {
        bool ok;
        iso_s_fac_fast(x, k, s, fac, ok);
        if (__builtin_expect(__any(!ok), 0)) {
            if (!ok) {
                s = t_sqrt(x);
                fac = t_max0(s - k) / (s + 1e-18f);
            }
        }
        return;
    }
#endif
    s = t_sqrt(x);
    fac = t_max0(s - k) / (s + T(1e-18));
}

// The isotropic (d, b) of the N elements a lane holds, scalar threshold k (the K-stage float pipelines): the fast chain of
// iso_s_fac_fast for all of them, ONE wave-level test for the rarely taken IEEE branch instead of one per element, and two
// wave-uniform relaxations that the host grants per threshold after trying every float x in [2^-60, 2^60) on the device
// (k_isolean_check, dxh_tvd.hip: the whole expression max(s - k, 0) / (s + 1e-18f) against its IEEE evaluation):
//   lean   the 120 square roots with an all-ones significand need no IEEE detour for THIS k (their reciprocal is one ulp
//          short, which the quotient's rounding absorbs unless s - k lands badly) -- no per-element significand test
//   div3   the quotient is correctly rounded after ONE residual correction for every x -- three operations instead of five
// Same operations on the same values as split_vars<float, 2, true> otherwise, so the results are bit-identical to it.
template <int N>
__device__ __forceinline__ void iso_split_vec(const float (&q0)[N], const float (&q1)[N], float k, bool lean, bool div3,
                                              float (&d0)[N], float (&b0)[N], float (&d1)[N], float (&b1)[N]) {
    float fac[N];
    bool okv[N];
    bool bad = !(k >= 0.0f);   // (false for NaN too) keeps the numerator within [0, s]: wave-uniform
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float xx = 0.0f;
        xx = xx + q0[j] * q0[j];
        xx = xx + q1[j] * q1[j];
        const float y = __builtin_amdgcn_rsqf(fmaxf(xx, kSqrtFastMin));
        const float g = xx * y, h = 0.5f * y;
        const float s = fmaf(fmaf(-g, g, xx), h, g);               // t_sqrt_chain(x)
        const float rho = fmaf(fmaf(-s, y, 1.0f), y, y);           // RN(1 / s) on [2^-60, 2^60) but for the all-ones significands
        const float num = (s - k) > 0.0f ? (s - k) : 0.0f;         // t_max0
        float f = div_by_const3(num, s, rho);
        if (!div3) {
            DXH_UNIFORM_BRANCH();
            f = div_refine(f, num, s, rho);
        }
        fac[j] = f;
        // x in [2^-60, 2^60) as one unsigned comparison of the bit patterns (x >= 0 or NaN: a sum of squares), or x == 0
        const unsigned ub = __float_as_uint(xx);
        bool ok = (ub - __float_as_uint(kIsoLo)) < (__float_as_uint(kIsoHi) - __float_as_uint(kIsoLo)) || ub == 0u;
        if (!lean) {
            DXH_UNIFORM_BRANCH();
            ok = ok && (__float_as_uint(s) & 0x7FFFFFu) != 0x7FFFFFu;
        }
        okv[j] = ok;
        bad = bad || !ok;
    }
    if (__builtin_expect(__any(bad), 0)) {
        const bool kbad = !(k >= 0.0f);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (kbad || !okv[j]) {
                float xx = 0.0f;
                xx = xx + q0[j] * q0[j];
                xx = xx + q1[j] * q1[j];
                const float s = t_sqrt(xx);
                fac[j] = t_max0(s - k) / (s + 1e-18f);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        d0[j] = q0[j] * fac[j];
        b0[j] = q0[j] * (1.0f - fac[j]);
        d1[j] = q1[j] * fac[j];
        b1[j] = q1[j] * (1.0f - fac[j]);
    }
}

template <typename T, int DIM, bool ISO, bool FASTSIGN = false>
__device__ __forceinline__ void split_vars(const T (&q)[DIM], T k, T (&d)[DIM], T (&b)[DIM]) {
    if constexpr (ISO) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < DIM; ++j) s = s + q[j] * q[j];
        T fac;
        iso_s_fac<T>(s, k, s, fac);
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            d[j] = q[j] * fac;
            b[j] = q[j] * (T(1) - fac);
        }
    } else {
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            if constexpr (FASTSIGN)
                d[j] = q[j] - t_clamp_sym(q[j], k);
            else
                d[j] = t_max0(t_abs(q[j]) - k) * sgn(q[j]);
            b[j] = q[j] - d[j];
        }
    }
}

}  // namespace dxh
