// Does hipStreamWaitValue32 work on this stack, and on which kind of memory?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
__global__ void bump(unsigned* p, int spin) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin) __builtin_amdgcn_s_sleep(10);
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void mark(unsigned* flag, const unsigned* counter) { *flag = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main() {
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    for (int kind = 0; kind < 2; ++kind) {
        unsigned* ctr = nullptr; unsigned* flag = nullptr;
        if (kind == 0) CK(hipMalloc((void**)&ctr, 64));
        else CK(hipExtMallocWithFlags((void**)&ctr, 64, hipMallocSignalMemory));
        CK(hipMalloc((void**)&flag, 4));
        CK(hipMemset(ctr, 0, 64)); CK(hipMemset(flag, 0xff, 4));
        hipStream_t a, b;
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        hipError_t e = hipStreamWaitValue32(a, ctr, 3, hipStreamWaitValueGte, 0xffffffffu);
        printf("kind %d: hipStreamWaitValue32 -> %s\n", kind, hipGetErrorString(e));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, a, flag, ctr);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, b, ctr, 20000000 /* 0.2 s at 100 MHz */);
            CK(hipStreamSynchronize(a));
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            unsigned f = 0; CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
            printf("kind %d: waiter released after %.1f ms (expected ~600), saw counter = %u (expected 3)\n", kind, ms, f);
            CK(hipStreamSynchronize(b));
        }
        CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b)); CK(hipFree(ctr)); CK(hipFree(flag));
    }
    return 0;
}
