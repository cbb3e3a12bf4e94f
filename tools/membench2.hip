// Developer microbenchmark #2: shape exploration of the row-marching 4R+3W stream (see membench.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int NTS> __device__ inline void st4(v4f v, float* p){ if(NTS) __builtin_nontemporal_store(v,(v4f*)p); else *(v4f*)p=v; }
template<int NTL> __device__ inline v4f ld4(const float* p){ if(NTL) return __builtin_nontemporal_load((const v4f*)p); else return *(const v4f*)p; }

struct A{ const float* a[4]; float* b[3]; long n0,L,pitch; int S,V,nstrips,nchunks,remap,wpb; long rpc; };

// VPL vectors per lane: lane holds VPL float4 at x0 + v*256 (v-th 1KB segment of the window): window = VPL*256 floats
template<int NTL,int NTS,int VPL,int PF>
__global__ void __launch_bounds__(256) k_march(A g){
  int lane=threadIdx.x&63, wave=threadIdx.x>>6; int wpb=g.wpb; int nbx=(g.nstrips+wpb-1)/wpb;
  int bid=blockIdx.x; int bx,chunk; if(g.remap){ chunk=bid%g.nchunks; bx=bid/g.nchunks; } else { bx=bid%nbx; chunk=bid/nbx; }
  int strip=bx*wpb+wave; if(strip>=g.nstrips||chunk>=g.nchunks) return;
  long R0=chunk*g.rpc, R1=R0+g.rpc<g.n0?R0+g.rpc:g.n0; long P=g.pitch;
  long xl[VPL]; bool st[VPL];
  #pragma unroll
  for(int v=0;v<VPL;++v){ long x0=(long)strip*g.S+(lane-1)*4+v*256; xl[v]=x0<0?0:(x0>P-4?P-4:x0); int gl=lane+v*64; st[v]=gl>=1&&gl<1+g.V&&x0<g.L; }
  v4f p[PF][VPL][4];
  #pragma unroll
  for(int k=0;k<PF;++k){ long rr=R0+k<g.n0?R0+k:g.n0-1;
    #pragma unroll
    for(int v=0;v<VPL;++v)
      #pragma unroll
      for(int i=0;i<4;++i) p[k][v][i]=ld4<NTL>(g.a[i]+rr*P+xl[v]); }
  for(long r=R0;r<R1;++r){
    long rn=r+PF<g.n0?r+PF:g.n0-1;
    v4f q[VPL][4];
    #pragma unroll
    for(int v=0;v<VPL;++v)
      #pragma unroll
      for(int i=0;i<4;++i) q[v][i]=ld4<NTL>(g.a[i]+rn*P+xl[v]);
    #pragma unroll
    for(int v=0;v<VPL;++v){
      v4f r0=p[0][v][0]+p[0][v][1]*p[0][v][2]; v4f r1=p[0][v][2]+p[0][v][3]; v4f r2=p[0][v][1]-p[0][v][3]*p[0][v][0];
      if(st[v]){ st4<NTS>(r0,g.b[0]+r*P+xl[v]); st4<NTS>(r1,g.b[1]+r*P+xl[v]); st4<NTS>(r2,g.b[2]+r*P+xl[v]); }
    }
    #pragma unroll
    for(int k=0;k<PF-1;++k)
      #pragma unroll
      for(int v=0;v<VPL;++v)
        #pragma unroll
        for(int i=0;i<4;++i) p[k][v][i]=p[k+1][v][i];
    #pragma unroll
    for(int v=0;v<VPL;++v)
      #pragma unroll
      for(int i=0;i<4;++i) p[PF-1][v][i]=q[v][i];
  }
}
int main(int argc,char**argv){
  long n0=argc>1?atol(argv[1]):4000, L=argc>2?atol(argv[2]):24000; long pitch=(L+31)/32*32; size_t n=(size_t)n0*pitch; size_t bytes=n*4;
  float* buf[7]; for(int i=0;i<7;++i){ CK(hipMalloc(&buf[i],bytes)); CK(hipMemset(buf[i],0,bytes)); }
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit=[&](const char* name,double gb,auto f){ f(); CK(hipDeviceSynchronize()); float best=1e9; for(int rep=0;rep<5;++rep){ CK(hipEventRecord(e0)); for(int i=0;i<20;++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=20; if(ms<best)best=ms;} printf("%-64s %.4f ms  %.0f GB/s\n",name,best,gb/best*1e-6); fflush(stdout); };
  double gb=7.0*(double)n0*L*4;
  auto run=[&](int ntl,int nts,int vpl,int pf,int S,int V,int wpc,int remap,int wpb){
    A g; for(int i=0;i<4;++i) g.a[i]=buf[i]; for(int i=0;i<3;++i) g.b[i]=buf[4+i]; g.n0=n0; g.L=L; g.pitch=pitch; g.S=S; g.V=V; g.remap=remap; g.wpb=wpb;
    g.nstrips=(L+S-1)/S; int nbx=(g.nstrips+wpb-1)/wpb; long target_blocks=256L*wpc/wpb; long nch=target_blocks/nbx; if(nch<1)nch=1; if(remap&&nch>=8) nch=nch/8*8;
    g.rpc=(n0+nch-1)/nch; nch=(n0+g.rpc-1)/g.rpc; g.nchunks=(int)nch;
    char nm[160]; sprintf(nm,"ntl%d nts%d vpl%d pf%d S=%d V=%d waves/CU=%d remap=%d wpb=%d rpc=%ld nch=%ld",ntl,nts,vpl,pf,S,V,wpc,remap,wpb,g.rpc,nch);
    dim3 grid(nbx*nch), blk(64*wpb);
#define L_(NTL,NTS,VPL,PF) if(ntl==NTL&&nts==NTS&&vpl==VPL&&pf==PF) timeit(nm,gb,[&]{ hipLaunchKernelGGL((k_march<NTL,NTS,VPL,PF>),grid,blk,0,0,g); });
    L_(0,0,1,1) L_(0,1,1,1) L_(1,1,1,1) L_(1,0,1,1) L_(0,1,1,2) L_(0,1,2,1) L_(0,0,2,1) L_(1,1,2,1) L_(0,1,2,2) L_(0,1,1,3)
  };
  // baseline from membench1
  run(0,1,1,1,224,56,8,0,4);
  run(0,1,1,1,224,56,8,1,4);
  run(0,1,1,1,224,56,16,1,4);
  run(0,1,1,1,224,56,8,1,1);
  run(0,1,1,1,224,56,8,1,2);
  run(1,1,1,1,224,56,8,1,4);
  run(1,0,1,1,224,56,8,1,4);
  run(0,1,1,2,224,56,8,1,4);
  run(0,1,1,3,224,56,8,1,4);
  run(0,1,1,2,224,56,4,1,4);
  run(0,1,1,1,224,56,4,1,4);
  run(0,1,1,1,224,56,12,1,4);
  // two vectors per lane: window 512 floats, stores 120 lanes = 480 floats (15 lines)
  run(0,1,2,1,480,120,8,1,4);
  run(0,1,2,1,480,120,4,1,4);
  run(0,1,2,1,480,120,8,0,4);
  run(0,0,2,1,480,120,8,1,4);
  run(1,1,2,1,480,120,8,1,4);
  run(0,1,2,2,480,120,4,1,4);
  run(0,1,2,1,480,120,6,1,2);
  return 0;
}
