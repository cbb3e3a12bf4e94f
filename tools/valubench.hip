// Developer microbenchmark: issue rate of plain vs packed f32 VALU ops on gfx950 at a given occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b){
  float x0=threadIdx.x*1e-3f, x1=x0+1.f, x2=x0+2.f, x3=x0+3.f, x4=x0+4.f, x5=x0+5.f, x6=x0+6.f, x7=x0+7.f;
  f2 p0={x0,x1},p1={x2,x3},p2={x4,x5},p3={x6,x7}; f2 av={a,a}, bv={b,b};
  for(int i=0;i<iters;++i){
    if(MODE==0){ // 8 independent scalar fma
      x0=__builtin_fmaf(x0,a,b); x1=__builtin_fmaf(x1,a,b); x2=__builtin_fmaf(x2,a,b); x3=__builtin_fmaf(x3,a,b);
      x4=__builtin_fmaf(x4,a,b); x5=__builtin_fmaf(x5,a,b); x6=__builtin_fmaf(x6,a,b); x7=__builtin_fmaf(x7,a,b);
    } else if(MODE==1){ // 4 packed fma (same 8 results)
      p0=__builtin_elementwise_fma(p0,av,bv); p1=__builtin_elementwise_fma(p1,av,bv); p2=__builtin_elementwise_fma(p2,av,bv); p3=__builtin_elementwise_fma(p3,av,bv);
    } else if(MODE==2){ // 8 scalar adds
      x0+=a; x1+=a; x2+=a; x3+=a; x4+=a; x5+=a; x6+=a; x7+=a;
    } else { // 4 packed adds
      p0+=av; p1+=av; p2+=av; p3+=av;
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7+p0.x+p0.y+p1.x+p1.y+p2.x+p2.y+p3.x+p3.y;
}
int main(){
  float* out; CK(hipMalloc(&out, 4096*256*4)); hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int iters=20000;
  for(int wpc: {4,8,16}){
    int blocks=256*wpc/4;
    for(int mode=0;mode<4;++mode){
      auto run=[&]{ switch(mode){case 0: hipLaunchKernelGGL(k<0>,dim3(blocks),dim3(256),0,0,out,iters,1.0001f,0.5f);break; case 1: hipLaunchKernelGGL(k<1>,dim3(blocks),dim3(256),0,0,out,iters,1.0001f,0.5f);break; case 2: hipLaunchKernelGGL(k<2>,dim3(blocks),dim3(256),0,0,out,iters,1.0001f,0.5f);break; default: hipLaunchKernelGGL(k<3>,dim3(blocks),dim3(256),0,0,out,iters,1.0001f,0.5f);} };
      run(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
      double elem_ops=(double)blocks*256*iters*8; const char* nm[4]={"scalar fma x8","packed fma x4","scalar add x8","packed add x4"};
      printf("waves/CU=%2d %-14s %8.3f ms  %7.2f Gelem-op/s  (%.2f cycles per wave-instr per SIMD at 2.4GHz)\n",wpc,nm[mode],ms,elem_ops/ms*1e-6, ms*1e-3*2.4e9/((double)iters*(mode%2?4:8)*wpc/4.0));
    }
  }
  return 0;
}
