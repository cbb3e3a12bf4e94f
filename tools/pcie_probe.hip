// Developer probe: pageable vs registered (pinned in place) host buffers for the 384 MB image upload / download.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
static double now(){ return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(){
  const size_t n = (size_t)4000*8000*3*4;
  char* h = (char*)malloc(n); memset(h, 1, n);
  char* h2 = (char*)malloc(n); memset(h2, 2, n);
  void* d; CK(hipMalloc(&d, n));
  CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double t1 = now();
    CK(hipMemcpy(h2, d, n, hipMemcpyDeviceToHost)); double t2 = now();
    printf("pageable   H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", t1-t0, n/(t1-t0)*1e-6, t2-t1, n/(t2-t1)*1e-6);
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); CK(hipHostRegister(h, n, hipHostRegisterDefault)); double t1 = now();
    CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double t2 = now();
    CK(hipHostUnregister(h)); double t3 = now();
    printf("registered register %.2f ms  H2D %.2f ms (%.1f GB/s)  unregister %.2f ms  total %.2f\n", t1-t0, t2-t1, n/(t2-t1)*1e-6, t3-t2, t3-t0);
    t0 = now(); CK(hipHostRegister(h2, n, hipHostRegisterDefault)); t1 = now();
    CK(hipMemcpy(h2, d, n, hipMemcpyDeviceToHost)); t2 = now();
    CK(hipHostUnregister(h2)); t3 = now();
    printf("registered register %.2f ms  D2H %.2f ms (%.1f GB/s)  unregister %.2f ms  total %.2f\n", t1-t0, t2-t1, n/(t2-t1)*1e-6, t3-t2, t3-t0);
  }
  // chunked: register + copy 32 MB chunks, pipelined on two streams
  return 0;
}
