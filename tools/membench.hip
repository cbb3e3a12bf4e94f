// Developer microbenchmark (not product): streaming ceilings on this box for the access shapes k_fused uses.
//   copy      : float4 grid-stride copy (1 read + 1 write stream)
//   rw43      : 4 read + 3 write streams, linear grid-stride
//   march     : 4 read + 3 write streams, column-strip row-marching like k_fused (strip stride S floats, V store lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(float4 v, float* p){ v4f t={v.x,v.y,v.z,v.w}; __builtin_nontemporal_store(t,(v4f*)p); }
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n){
  for(size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=(size_t)gridDim.x*blockDim.x) b[i]=a[i];
}
__global__ void __launch_bounds__(256) k_rw43(const float4* __restrict__ a0,const float4* __restrict__ a1,const float4* __restrict__ a2,const float4* __restrict__ a3,
    float4* __restrict__ b0,float4* __restrict__ b1,float4* __restrict__ b2,size_t n){
  for(size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=(size_t)gridDim.x*blockDim.x){
    float4 x=a0[i],y=a1[i],z=a2[i],w=a3[i];
    float4 r; r.x=x.x+y.x; r.y=x.y+z.y; r.z=x.z+w.z; r.w=x.w+y.w;
    b0[i]=r; float4 s; s.x=z.x+w.x; s.y=y.y; s.z=x.z; s.w=w.w; b1[i]=s; float4 t; t.x=y.x; t.y=w.y+x.y; t.z=z.z; t.w=y.w; b2[i]=t;
  }
}
// marching: wave owns a window of 64 lanes*4 floats starting at strip*S - 4, stores lanes [1,1+V)
template<bool NT>
__global__ void __launch_bounds__(256) k_march(const float* __restrict__ a0,const float* __restrict__ a1,const float* __restrict__ a2,const float* __restrict__ a3,
    float* __restrict__ b0,float* __restrict__ b1,float* __restrict__ b2,long n0,long L,long pitch,int S,int V,int nstrips,long rpc,int nchunks){
  int lane=threadIdx.x&63, wave=threadIdx.x>>6; int nbx=(nstrips+3)/4; int bx=blockIdx.x%nbx, chunk=blockIdx.x/nbx; int strip=bx*4+wave;
  if(strip>=nstrips||chunk>=nchunks) return;
  long R0=chunk*rpc, R1=R0+rpc<n0?R0+rpc:n0;
  long x0=(long)strip*S+(lane-1)*4; long xl=x0<0?0:(x0>pitch-4?pitch-4:x0);
  bool st=lane>=1&&lane<1+V&&x0<L;
  float4 p0=*(const float4*)(a0+R0*pitch+xl),p1=*(const float4*)(a1+R0*pitch+xl),p2=*(const float4*)(a2+R0*pitch+xl),p3=*(const float4*)(a3+R0*pitch+xl);
  for(long r=R0;r<R1;++r){
    long rn=r+1<n0?r+1:n0-1;
    float4 q0=*(const float4*)(a0+rn*pitch+xl),q1=*(const float4*)(a1+rn*pitch+xl),q2=*(const float4*)(a2+rn*pitch+xl),q3=*(const float4*)(a3+rn*pitch+xl);
    float4 r0; r0.x=p0.x+p1.x; r0.y=p0.y+p2.y; r0.z=p0.z+p3.z; r0.w=p0.w+p1.w;
    float4 r1; r1.x=p2.x+p3.x; r1.y=p1.y; r1.z=p0.z; r1.w=p3.w;
    float4 r2; r2.x=p1.x; r2.y=p3.y+p0.y; r2.z=p2.z; r2.w=p1.w;
    if(st){
      if(NT){ nt_store(r0,b0+r*pitch+xl); nt_store(r1,b1+r*pitch+xl); nt_store(r2,b2+r*pitch+xl); }
      else { *(float4*)(b0+r*pitch+xl)=r0; *(float4*)(b1+r*pitch+xl)=r1; *(float4*)(b2+r*pitch+xl)=r2; }
    }
    p0=q0;p1=q1;p2=q2;p3=q3;
  }
}
int main(int argc,char**argv){
  long n0=4000,L=24000,pitch=24000; size_t n=(size_t)n0*pitch; size_t bytes=n*4;
  float* buf[7]; for(int i=0;i<7;++i){ CK(hipMalloc(&buf[i],bytes)); CK(hipMemset(buf[i],0,bytes)); }
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit=[&](const char* name,double gb,auto f){ f(); CK(hipDeviceSynchronize()); float best=1e9; for(int rep=0;rep<5;++rep){ CK(hipEventRecord(e0)); for(int i=0;i<20;++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=20; if(ms<best)best=ms;} printf("%-44s %.4f ms  %.0f GB/s\n",name,best,gb/best*1e-6); fflush(stdout); };
  for(int g: {2048,4096,8192,16384}){ char nm[64]; sprintf(nm,"copy grid=%d",g); timeit(nm,2.0*bytes,[&]{ hipLaunchKernelGGL(k_copy,dim3(g),dim3(256),0,0,(const float4*)buf[0],(float4*)buf[4],n/4); }); }
  for(int g: {2048,4096,8192}){ char nm[64]; sprintf(nm,"rw43 linear grid=%d",g); timeit(nm,7.0*bytes,[&]{ hipLaunchKernelGGL(k_rw43,dim3(g),dim3(256),0,0,(const float4*)buf[0],(const float4*)buf[1],(const float4*)buf[2],(const float4*)buf[3],(float4*)buf[4],(float4*)buf[5],(float4*)buf[6],n/4); }); }
  struct Cfg{int S,V,bpc;};
  for(Cfg c: {Cfg{244,61,4},Cfg{224,56,4},Cfg{244,61,2},Cfg{244,61,8},Cfg{224,56,8},Cfg{224,56,2},Cfg{244,61,6},Cfg{224,56,6}}){
    int nstrips=(L+c.S-1)/c.S; int nbx=(nstrips+3)/4; long target=256L*c.bpc; long nchunks=target/nbx; if(nchunks<1)nchunks=1; long rpc=(n0+nchunks-1)/nchunks; nchunks=(n0+rpc-1)/rpc;
    for(int nt=0;nt<2;++nt){ char nm[96]; sprintf(nm,"march S=%d V=%d blk/CU=%d rpc=%ld nt=%d",c.S,c.V,c.bpc,rpc,nt);
      timeit(nm,7.0*bytes,[&]{ if(nt) hipLaunchKernelGGL(k_march<true>,dim3(nbx*nchunks),dim3(256),0,0,buf[0],buf[1],buf[2],buf[3],buf[4],buf[5],buf[6],n0,L,pitch,c.S,c.V,nstrips,rpc,(int)nchunks);
        else hipLaunchKernelGGL(k_march<false>,dim3(nbx*nchunks),dim3(256),0,0,buf[0],buf[1],buf[2],buf[3],buf[4],buf[5],buf[6],n0,L,pitch,c.S,c.V,nstrips,rpc,(int)nchunks); }); }
  }
  return 0;
}
