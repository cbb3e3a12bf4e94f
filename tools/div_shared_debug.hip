// find the operands on which div_shared differs from the IEEE division (tools only)
#include "../darsia_amd/csrc/dxh_common.h"
using namespace dxh;
namespace dxh { void set_error(const char*, ...) {} }
__device__ __forceinline__ unsigned mix32(unsigned long long v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
    return (unsigned)v;
}
__global__ void k(unsigned* out, unsigned* cnt) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long it = 0; it < 4096; ++it) {
        const unsigned long long base = (tid * 4096 + it) * 8;
        float x[3], d;
        const unsigned mode = mix32(base + 7) % 3;
        auto gen = [&](unsigned long long k, bool positive) -> float {
            const unsigned r = mix32(base + k);
            if (mode == 0) return __uint_as_float(positive ? (r & 0x7FFFFFFFu) : r);
            if (mode == 1) {
                const float m = 1.0f + (float)(r & 0xFFFFFF) * 0x1p-24f;
                const int e = (int)((r >> 24) & 31) - 20;
                const float v = ldexpf(m, e);
                return (!positive && (r >> 31)) ? -v : ((r >> 29) == 0 ? 0.0f * v : v);
            }
            const int e = ((r >> 24) & 1) ? 62 : -62;
            const float v = ldexpf(1.0f + (float)(r & 0xFFFFFF) * 0x1p-24f, e + (int)((r >> 25) & 3) - 2);
            return (!positive && (r >> 31)) ? -v : v;
        };
        for (int i = 0; i < 3; ++i) x[i] = gen(i, false);
        d = gen(3, true);
        float q[3];
        div_shared<float, 3>(x, d, q);
        for (int i = 0; i < 3; ++i) {
            const float want = x[i] / d;
            const bool same = (want != want) ? (q[i] != q[i]) : (__float_as_uint(want) == __float_as_uint(q[i]));
            if (!same) {
                const unsigned k2 = atomicAdd(cnt, 1u);
                if (k2 < 8) { out[8 * k2] = __float_as_uint(x[0]); out[8 * k2 + 1] = __float_as_uint(x[1]); out[8 * k2 + 2] = __float_as_uint(x[2]);
                    out[8 * k2 + 3] = __float_as_uint(d); out[8 * k2 + 4] = i; out[8 * k2 + 5] = __float_as_uint(want); out[8 * k2 + 6] = __float_as_uint(q[i]); out[8 * k2 + 7] = mode; }
            }
        }
    }
}
int main() {
    unsigned *o, *c; hipMalloc(&o, 256); hipMalloc(&c, 4); hipMemset(c, 0, 4); hipMemset(o, 0, 256);
    hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, o, c);
    unsigned h[64], n; hipMemcpy(h, o, 256, hipMemcpyDeviceToHost); hipMemcpy(&n, c, 4, hipMemcpyDeviceToHost);
    printf("%u mismatches\n", n);
    for (unsigned i = 0; i < n && i < 8; ++i) {
        float f[7]; memcpy(f, h + 8 * i, 28);
        printf("x = %a %a %a  d = %a  which %u  want %a got %a  mode %u\n", f[0], f[1], f[2], f[3], h[8*i+4], f[5], f[6], h[8*i+7]);
        printf("   bits x %08x %08x %08x d %08x want %08x got %08x\n", h[8*i], h[8*i+1], h[8*i+2], h[8*i+3], h[8*i+5], h[8*i+6]);
    }
    return 0;
}
