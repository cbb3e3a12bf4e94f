// Developer probe: per-call cost of a few HIP runtime calls.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now(){ return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(){
  void* w; hipMalloc(&w, 1<<20);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    double t1 = now(); int v; hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0);
    double t2 = now(); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double t3 = now(); hipEventDestroy(e0); hipEventDestroy(e1);
    double t4 = now(); hipSetDevice(0);
    double t5 = now(); int n; hipGetDeviceCount(&n);
    double t6 = now(); hipDeviceSynchronize();
    double t7 = now();
    printf("GetDeviceProperties %.3f  GetAttribute %.3f  2x EventCreate %.3f  2x EventDestroy %.3f  SetDevice %.3f  GetDeviceCount %.3f  DeviceSynchronize %.3f ms\n",
           t1-t0, t2-t1, t3-t2, t4-t3, t5-t4, t6-t5, t7-t6);
  }
  return 0;
}
