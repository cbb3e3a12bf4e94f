// Developer probe: cost of hipMalloc / hipMemsetAsync / hipFree / stream create for the ~11 planes of a one-shot call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now(){ return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(){
  const size_t n = (size_t)4000*8000*3*4;
  void* w; hipMalloc(&w, 1<<20); hipFree(w);
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double t1 = now();
    std::vector<void*> p(11);
    for (auto& q : p) hipMalloc(&q, n);
    double t2 = now();
    for (auto& q : p) hipMemsetAsync(q, 0, n, st);
    hipStreamSynchronize(st);
    double t3 = now();
    for (auto& q : p) hipFree(q);
    double t4 = now();
    hipStreamDestroy(st);
    double t5 = now();
    printf("stream create %.2f  malloc x11 %.2f  memset x11 %.2f  free x11 %.2f  stream destroy %.2f ms\n", t1-t0, t2-t1, t3-t2, t4-t3, t5-t4);
  }
  return 0;
}
