// developer probe: which x in [2^-60, 2^60) make the rsq-seeded reciprocal differ from RN(1 / sqrt(x))?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(unsigned* out, unsigned* cnt) {
    const unsigned long long lo = 0x21800000ull, hi = 0x5D800000ull;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        const float s = fmaf(fmaf(-g, g, x), h, g);
        const float rho = fmaf(fmaf(-s, y, 1.0f), y, y);
        const float ref = 1.0f / s;
        if (__float_as_uint(rho) != __float_as_uint(ref)) {
            unsigned i = atomicAdd(cnt, 1u);
            if (i < 64) { out[4 * i] = (unsigned)b; out[4 * i + 1] = __float_as_uint(s); out[4 * i + 2] = __float_as_uint(rho); out[4 * i + 3] = __float_as_uint(ref); }
        }
    }
}
int main() {
    unsigned *d, *c, h[256], n = 0;
    hipMalloc(&d, sizeof h); hipMalloc(&c, 4); hipMemset(c, 0, 4);
    hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, d, c);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&n, c, 4, hipMemcpyDeviceToHost);
    printf("mismatches %u\n", n);
    for (unsigned i = 0; i < (n < 16 ? n : 16); ++i) {
        float x, s, r, f; memcpy(&x, &h[4*i], 4); memcpy(&s, &h[4*i+1], 4); memcpy(&r, &h[4*i+2], 4); memcpy(&f, &h[4*i+3], 4);
        printf("x=%08x (%a) s=%08x (%a) rho=%08x ref=%08x\n", h[4*i], x, h[4*i+1], s, h[4*i+2], h[4*i+3]);
    }
    return 0;
}
