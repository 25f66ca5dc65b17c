// How fast does the GPU start workgroups?  Empty kernels of N workgroups of T threads with
// L bytes of LDS; reports the kernel time per launch (events around 200 back-to-back launches).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel(int* sink) {
  extern __shared__ char lds[];
  if (sink != nullptr && threadIdx.x == 0) {
    lds[0] = 1;
    sink[blockIdx.x] = lds[0];
  }
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int ns[] = {1, 256, 1024, 2048, 4096, 8192, 16384};
  const int ts[] = {64, 256};
  const int ls[] = {0, 18432, 65536};
  for (int t : ts)
    for (int l : ls)
      for (int n : ns) {
        if (l > 0) hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, l);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(n), dim3(t), l, st, (int*)nullptr);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(n), dim3(t), l, st, (int*)nullptr);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("threads %3d lds %5d wgs %5d: %7.2f us/launch  (%6.1f ns per WG)\n", t, l, n, ms * 1000 / 200, ms * 1e6 / 200 / n);
      }
  return 0;
}
