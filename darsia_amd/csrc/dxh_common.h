// Shared helpers of libdarsia_hip: error reporting, device-side vector access,
// cross-lane shifts.  gfx950 only (wave64, DPP wave shifts).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

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
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }  // correctly rounded (no fast-math)
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }

// (d, b) of one pixel from its stored state q = b_old + D(u) ("dub" in the reference):
//   anisotropic  d = shrink(q, k), b = q - d                    split_bregman_tvd.py:80-81,143-146
//   isotropic    s = |q|_2, f = max(s-k,0)/(s+1e-18), d = q f, b = q (1-f)       :107-126
// FASTSIGN (callers guarantee k >= 0): max(|q| - k, 0) * sign(q) only transfers q's sign -- for q == 0
// the magnitude is max(-k, 0) = 0 as well -- so copysign gives the same value with 3 instructions
// instead of 8.  For k < 0 or NaN the forms differ at q == 0, hence the flag.
template <typename T, int DIM, bool ISO, bool FASTSIGN = false>
__device__ __forceinline__ void split_vars(const T (&q)[DIM], T k, T (&d)[DIM], T (&b)[DIM]) {
    if constexpr (ISO) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < DIM; ++j) s = s + q[j] * q[j];
        s = t_sqrt(s);
        T fac = t_max0(s - k) / (s + T(1e-18));
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            d[j] = q[j] * fac;
            b[j] = q[j] * (T(1) - fac);
        }
    } else {
#pragma unroll
        for (int j = 0; j < DIM; ++j) {
            if constexpr (FASTSIGN)
                d[j] = t_copysign(t_max0(t_abs(q[j]) - k), q[j]);
            else
                d[j] = t_max0(t_abs(q[j]) - k) * sgn(q[j]);
            b[j] = q[j] - d[j];
        }
    }
}

}  // namespace dxh
