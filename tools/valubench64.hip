// Developer microbenchmark: issue cost (cycles per wave-instruction per SIMD) of the float64 / conversion / DPP
// instructions the numpy2 (mixed float32 / float64) pipelines are made of, next to v_add_f32, at 1 - 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valubench64.hip -o tools/valubench64 && tools/valubench64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

enum { ADD32, ADD64, FMA64, MUL64, CVT_F32_F64, CVT_F64_F32, MAX64, MIN64, MOVDPP, BFI, MED3, ADD32DPP, FMA32, MOV64, NMODES };
static const char* NAMES[NMODES] = {"v_add_f32", "v_add_f64", "v_fma_f64", "v_mul_f64", "v_cvt_f32_f64", "v_cvt_f64_f32",
                                    "v_max_f64", "v_min_f64", "v_mov_b32_dpp", "v_bfi_b32", "v_med3_f32", "v_add_f32_dpp",
                                    "v_fma_f32", "v_mov_b64"};

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
    double x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3 + i; f[i] = (float)x[i]; }
    const float af = (float)a, bf = (float)b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MODE == ADD32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(af));
            else if constexpr (MODE == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            else if constexpr (MODE == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            else if constexpr (MODE == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            else if constexpr (MODE == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(x[i]));
            else if constexpr (MODE == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[i]) : "v"(f[i]));
            else if constexpr (MODE == MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            else if constexpr (MODE == MIN64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            else if constexpr (MODE == MOVDPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f[i]) : "v"(f[(i + 1) & 7]));
            else if constexpr (MODE == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(f[i]) : "v"(af), "v"(bf));
            else if constexpr (MODE == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(af), "v"(bf));
            else if constexpr (MODE == ADD32DPP) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            else if constexpr (MODE == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(af), "v"(bf));
            else if constexpr (MODE == MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run_mode(double* out, hipEvent_t e0, hipEvent_t e1, int iters) {
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps;   // one 256-thread block = one wave per SIMD of a CU
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.5);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 0.5);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions per SIMD = iters * 8 * wps; cycles at a nominal 2.4 GHz (the clock under load is lower: compare rows)
        printf("%-16s waves/SIMD=%d  %8.3f ms  %6.2f cycles per wave-instruction per SIMD (2.4 GHz nominal)\n", NAMES[MODE], wps, ms,
               ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps));
    }
}

template <int M>
void all(double* out, hipEvent_t e0, hipEvent_t e1, int iters) {
    if constexpr (M < NMODES) {
        run_mode<M>(out, e0, e1, iters);
        all<M + 1>(out, e0, e1, iters);
    }
}

int main() {
    double* out;
    CK(hipMalloc(&out, 4096 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    all<0>(out, e0, e1, 20000);
    return 0;
}
