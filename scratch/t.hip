#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* p){ p[threadIdx.x] = threadIdx.x; }
extern "C" int probe() {
  int n = -1; hipError_t e = hipGetDeviceCount(&n);
  printf("hipGetDeviceCount -> %d (%s), n=%d\n", (int)e, hipGetErrorString(e), n);
  int rv=0, dv=0; hipRuntimeGetVersion(&rv); hipDriverGetVersion(&dv); printf("runtime %d driver %d\n", rv, dv);
  if (e != hipSuccess) return 1;
  int* d; e = hipMalloc(&d, 256); printf("malloc %d\n",(int)e);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  e = hipGetLastError(); printf("launch %d (%s)\n",(int)e, hipGetErrorString(e));
  e = hipDeviceSynchronize(); printf("sync %d (%s)\n",(int)e, hipGetErrorString(e));
  int h[64]; hipMemcpy(h,d,256,hipMemcpyDeviceToHost); printf("h[5]=%d\n",h[5]);
  return 0;
}
