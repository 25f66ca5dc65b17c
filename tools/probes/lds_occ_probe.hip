// How many 256-thread workgroups with S bytes of static LDS does a CU hold?  (hipOccupancy query)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S> __global__ __launch_bounds__(256) void k(int* o) {
  __shared__ char s[S];
  s[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  o[threadIdx.x] = s[(threadIdx.x * 7) % S];
}
template <int S> void ask() {
  int n = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<S>, 256, 0);
  printf("static LDS %6d B: %d workgroups per CU (%s)\n", S, n, hipGetErrorString(e));
}
int main() {
  ask<65536 + 4096>(); ask<65536 + 8192>(); ask<65536 + 12288>(); ask<65536 + 14336>(); ask<65536 + 16384>();
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlock %zu\n", (size_t)p.sharedMemPerMultiprocessor, (size_t)p.maxSharedMemoryPerMultiProcessor, (size_t)p.sharedMemPerBlock);
  return 0;
}
