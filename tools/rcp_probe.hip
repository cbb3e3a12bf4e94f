// Which floats n have  chain(n) != RN(1/n)  for  r0 = rcp(n); e = fma(-n, r0, 1); r1 = fma(e, r0, r0); e2 = fma(-n, r1, 1); r = fma(e2, r1, r1) ?
// Exhaustive over positive floats in [2^-100, 2^100] (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float chain2(float n) {
    const float r0 = __builtin_amdgcn_rcpf(n);
    const float e = fmaf(-n, r0, 1.0f);
    const float r1 = fmaf(e, r0, r0);
    const float e2 = fmaf(-n, r1, 1.0f);
    return fmaf(e2, r1, r1);
}
__device__ __forceinline__ float chain1(float n) {
    const float r0 = __builtin_amdgcn_rcpf(n);
    const float e = fmaf(-n, r0, 1.0f);
    return fmaf(e, r0, r0);
}
__global__ void probe(unsigned long long* cnt, unsigned* samples) {
    const uint64_t lo = 0x0D800000ull, hi = 0x71800000ull;  // 2^-100 .. 2^100
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float n = __uint_as_float((unsigned)b);
        const float ref = 1.0f / n;
        if (chain2(n) != ref) {
            const unsigned long long k = atomicAdd(&cnt[0], 1ull);
            if (k < 16) samples[k] = (unsigned)b;
            if (((unsigned)b & 0x7FFFFFu) != 0x7FFFFFu) atomicAdd(&cnt[2], 1ull);  // not an all-ones significand
        }
        if (chain1(n) != ref) atomicAdd(&cnt[1], 1ull);
    }
}
int main() {
    unsigned long long* c; unsigned* s;
    hipMalloc(&c, 24); hipMalloc(&s, 64); hipMemset(c, 0, 24); hipMemset(s, 0, 64);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, c, s);
    unsigned long long h[3]; unsigned hs[16];
    hipMemcpy(h, c, 24, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 64, hipMemcpyDeviceToHost);
    printf("two corrections: %llu mismatches (%llu of them NOT all-ones significands); one correction: %llu mismatches\n", h[0], h[2], h[1]);
    for (int i = 0; i < 16 && i < (int)h[0]; ++i) printf("  0x%08x\n", hs[i]);
    return 0;
}
