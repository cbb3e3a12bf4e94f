// Is  y = rsq(x); g = x y; h = y / 2; r = fma(-g, g, x); s = fma(r, h, g)  the correctly rounded sqrtf(x)?
// Exhaustive over every positive normal float >= 2^-100 (tools only).  Also variants with a second correction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float sqrt_a(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = fmaf(-g, g, x);
    return fmaf(r, h, g);
}
__device__ __forceinline__ float sqrt_b(float x) {  // via v_sqrt_f32 + v_rcp
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(g);
    const float r = fmaf(-g, g, x);
    return fmaf(r, h, g);
}
__global__ void probe(unsigned long long* bad_a, unsigned long long* bad_b, unsigned* first_a, unsigned* first_b) {
    const uint64_t lo = 0x00800000ull /* 2^-126: every positive normal */, hi = 0x7F800000ull /* inf */;
    uint64_t ca = 0, cb = 0;
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)b);
        const float ref = sqrtf(x);  // correctly rounded (no fast-math)
        if (sqrt_a(x) != ref) { if (!ca) atomicMin(first_a, (unsigned)b); ++ca; }
        if (sqrt_b(x) != ref) { if (!cb) atomicMin(first_b, (unsigned)b); ++cb; }
    }
    if (ca) atomicAdd(bad_a, ca);
    if (cb) atomicAdd(bad_b, cb);
}
int main() {
    unsigned long long *ba, *bb; unsigned *fa, *fb;
    hipMalloc(&ba, 8); hipMalloc(&bb, 8); hipMalloc(&fa, 4); hipMalloc(&fb, 4);
    hipMemset(ba, 0, 8); hipMemset(bb, 0, 8); hipMemset(fa, 0xff, 4); hipMemset(fb, 0xff, 4);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, ba, bb, fa, fb);
    unsigned long long a = 0, b = 0; unsigned xa = 0, xb = 0;
    hipMemcpy(&a, ba, 8, hipMemcpyDeviceToHost); hipMemcpy(&b, bb, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&xa, fa, 4, hipMemcpyDeviceToHost); hipMemcpy(&xb, fb, 4, hipMemcpyDeviceToHost);
    printf("rsq form: %llu mismatches (first bits 0x%08x); sqrt+rcp form: %llu mismatches (first 0x%08x)\n", a, xa, b, xb);
    return 0;
}
